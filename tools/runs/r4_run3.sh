#!/bin/bash
# autograd-path kernel stats (what the GPU runs beyond TrainStep's kernels), timed-steps-only phases, and the arbitration probe
R=$PWD
O=$R/gpurun_out/r04c
mkdir -p $O
PASST_AMD_BENCH_PHASES=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd > $O/phases_c2.log 2>&1
PASST_AMD_BENCH_PHASES=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --path autograd --config c5 > $O/phases_c5.log 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ka /tmp/kb
timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o k -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --path autograd > $O/kt_c2.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/ka -name '*.db' | head -1)" --steps 6 --top 60 > $O/kernel_stats_autograd_c2.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kb -o k -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --path autograd --config c5 > $O/kt_c5.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kb -name '*.db' | head -1)" --steps 6 --top 60 > $O/kernel_stats_autograd_c5.txt 2>&1
cd $R
(cd tests/probes && ./probe_mfma_valu_overlap) > $O/mfma_valu_overlap.txt 2>&1
grep -h "phases" $O/phases_c2.log $O/phases_c5.log
grep -v "pa[0-9]*\|_ZN2pa" $O/kernel_stats_autograd_c2.txt | head -40 | cut -c1-170
tail -10 $O/mfma_valu_overlap.txt
