#!/bin/bash
R=$PWD
O=$R/gpurun_out/r04i
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "two_steps and bf16 and ce" 2>&1 | grep -E "^E|assert" | head -12 > $O/ce_fail.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ka
timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --config c5 > $O/kt.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/ka -name '*.db' | head -1)" --steps 7 --top 45 > $O/kernel_stats_c5.txt 2>&1
cd $R
cat $O/ce_fail.txt; head -48 $O/kernel_stats_c5.txt | cut -c1-150
