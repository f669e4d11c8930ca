#!/bin/bash
# r06 run H: the whole GPU suite on the tree as committed
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -80 > $O/r06_h_tests.txt
tail -5 $O/r06_h_tests.txt
