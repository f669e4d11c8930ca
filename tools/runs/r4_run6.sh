#!/bin/bash
# after the host-side diet of the autograd path (cached parameter list, one-call gradient views, no materialised dfeat) and
# with passt_amd.optim.AdamW: same box, same call A/B against TrainStep, c2 and c5; c5 traces for idle-gap analysis
R=$PWD
O=$R/gpurun_out/r04f
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "optim_adamw or contract or accumulation or frozen" 2>&1 | tail -3 > $O/pytest_sel.txt
for cfg in c2 c5; do
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config $cfg 2>/dev/null | tail -1 | cut -c1-150 >> $O/ab_$cfg.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config $cfg --path autograd 2>/dev/null | tail -1 | cut -c1-150 >> $O/ab_$cfg.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config $cfg --path autograd --optimizer pa_adamw 2>/dev/null | tail -1 | cut -c1-150 >> $O/ab_$cfg.txt
done; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ka /tmp/kb /tmp/kc
timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --config c5 > $O/kt_ts.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kb -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --config c5 --path autograd --optimizer pa_adamw > $O/kt_ag.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kc -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --path autograd --optimizer pa_adamw > $O/kt_ag2.log 2>&1
cp "$(find /tmp/ka -name '*.db' | head -1)" $O/trainstep_c5.db
cp "$(find /tmp/kb -name '*.db' | head -1)" $O/autograd_pa_c5.db
cp "$(find /tmp/kc -name '*.db' | head -1)" $O/autograd_pa_c2.db
cd $R
python tools/host_time_backward.py c5 2>&1 | grep -v Warning > $O/host_time.txt
cat $O/pytest_sel.txt $O/ab_c2.txt $O/ab_c5.txt $O/host_time.txt
