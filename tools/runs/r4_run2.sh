#!/bin/bash
# where the autograd (drop-in) path loses against TrainStep: phase times, host enqueue time, kernel stats; the cycle probe
O=gpurun_out/r04b
mkdir -p $O
R=$PWD
PASST_AMD_BENCH_PHASES=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd > $O/phases_c2.log 2>&1
PASST_AMD_BENCH_PHASES=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd --config c5 > $O/phases_c5.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd > $O/noroof_c2.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd --config c5 > $O/noroof_c5.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config c5 > $O/noroof_ts_c5.log 2>&1
timeout 300 python tools/cpu_enqueue_time.py > $O/cpu_enqueue.txt 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ka /tmp/kb
timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o k -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --path autograd > $O/kt_c2.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/ka -name '*.db' | head -1)" --steps 6 --top 45 > $O/kernel_stats_autograd_c2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kb -o k -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --path autograd --config c5 > $O/kt_c5.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kb -name '*.db' | head -1)" --steps 6 --top 45 > $O/kernel_stats_autograd_c5.txt 2>&1
cd $R
(cd tests/probes && ./probe_mfma_valu_overlap) > $O/mfma_valu_overlap.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_model.py -q -k "frontend_vs_oracle" 2>&1 | tail -3
grep -h "phases" $O/phases_c2.log $O/phases_c5.log
for f in noroof_c2 noroof_c5 noroof_ts_c5; do tail -1 $O/$f.log | cut -c1-200; done
cat $O/cpu_enqueue.txt | tail -3
head -30 $O/kernel_stats_autograd_c2.txt | cut -c1-150
tail -12 $O/mfma_valu_overlap.txt
