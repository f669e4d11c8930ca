#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
bash tools/collect_profiles.sh gpurun_out/r05final3 > gpurun_out/r05final3_collect.log 2>&1
timeout 600 python bench_kernels.py --out gpurun_out/r05final3/isolated_kernels.json > gpurun_out/r05final3/bench_kernels.log 2>&1
tail -3 gpurun_out/r05final3/bench_kernels.log
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05final3/bench_c2_b.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05final3/bench_c5.json.log
cut -c1-200 gpurun_out/r05final3/bench_c2_b.json.log gpurun_out/r05final3/bench_c5.json.log
tail -1 gpurun_out/r05final3/bench_c2.log | cut -c1-200
