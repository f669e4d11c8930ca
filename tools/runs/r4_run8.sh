#!/bin/bash
R=$PWD
O=$R/gpurun_out/r04h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" 2>&1 | tail -8 > $O/pytest_kern.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -q 2>&1 | tail -8 > $O/pytest_model.txt
for i in 1 2; do
for cfg in c2 c5; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config $cfg 2>/dev/null | tail -1 | cut -c1-140 >> $O/ab.txt
PA_NT_SKINNY=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --config $cfg 2>/dev/null | tail -1 | cut -c1-140 >> $O/ab.txt
done; done
cat $O/pytest_kern.txt $O/pytest_model.txt $O/ab.txt
