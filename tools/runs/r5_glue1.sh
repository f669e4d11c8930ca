#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05g
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "patch or head_linear" > gpurun_out/r05g/pytest_glue.txt 2>&1
tail -5 gpurun_out/r05g/pytest_glue.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/r05g/pytest_model.txt 2>&1
tail -3 gpurun_out/r05g/pytest_model.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r05g/kt.log 2>&1
python /root/repo/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | head -1)" --steps 6 --top 44 > /root/repo/gpurun_out/r05g/kernel_stats.txt 2>&1
grep -E "batch_sum|patch_param|linear_fwd|mel_front|# " /root/repo/gpurun_out/r05g/kernel_stats.txt | cut -c1-150
