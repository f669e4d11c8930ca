#!/bin/bash
# same box, same call: TrainStep vs the autograd path -- bench lines without per-launch events, kernel stats + GPU idle gaps
R=$PWD
O=$R/gpurun_out/r04e
mkdir -p $O
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-160 >> $O/ab.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd 2>/dev/null | tail -1 | cut -c1-160 >> $O/ab.txt
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ka /tmp/kb
timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/kt_ts.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/ka -name '*.db' | head -1)" --steps 7 --top 30 --gaps 25 > $O/kernel_stats_trainstep_c2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kb -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --path autograd > $O/kt_ag.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kb -name '*.db' | head -1)" --steps 7 --top 30 --gaps 25 > $O/kernel_stats_autograd_c2.txt 2>&1
cd $R
cat $O/ab.txt
grep -A 28 "GPU timeline" $O/kernel_stats_trainstep_c2.txt | cut -c1-140
grep -A 28 "GPU timeline" $O/kernel_stats_autograd_c2.txt | cut -c1-140
