#!/bin/bash
# round 4, first GPU call: the new data-parallel tests (self-launching bench, autograd attach / torch DDP), the parity
# additions, the default bench line and the autograd-path bench line
O=gpurun_out/r04a
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ddp.py -q -x > $O/pytest_ddp.txt 2>&1; echo "ddp rc=$?" >> $O/pytest_ddp.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "train_step or per_bucket or frontend or contract" > $O/pytest_model_sel.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "bit_deterministic or one_item" > $O/pytest_kern_sel.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline > $O/bench_c2.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --path autograd > $O/bench_c2_autograd.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --config c5 > $O/bench_c5.log 2>&1
timeout 400 python bench.py --no-cpu-baseline --config c5 --path autograd > $O/bench_c5_autograd.log 2>&1
tail -3 $O/pytest_ddp.txt; tail -3 $O/pytest_model_sel.txt; tail -3 $O/pytest_kern_sel.txt
for f in bench_c2 bench_c2_autograd bench_c5 bench_c5_autograd; do tail -1 $O/$f.log | cut -c1-330; done
