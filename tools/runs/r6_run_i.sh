#!/bin/bash
# r06 run I: persistent mel front end: parity (front-end tests) and time against the one-tile-per-workgroup form
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -x -k "frontend or mel" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 > $O/r06_i_tests.txt
tail -3 $O/r06_i_tests.txt
for i in 1 2; do
  echo "one-tile  $(PA_MEL_PERSIST=0 python tools/bench_mel.py 2>/dev/null | tail -1)"
  echo "persist8  $(PA_MEL_PERSIST=1 python tools/bench_mel.py 2>/dev/null | tail -1)"
  echo "persist16 $(PASST_AMD_LIB=$R/passt_amd/libpasst_amd_mel_p16.so python tools/bench_mel.py 2>/dev/null | tail -1)"
  echo "persist4  $(PASST_AMD_LIB=$R/passt_amd/libpasst_amd_mel_p4.so python tools/bench_mel.py 2>/dev/null | tail -1)"
done
