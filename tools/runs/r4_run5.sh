#!/bin/bash
R=$PWD
O=$R/gpurun_out/r04e
mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ka /tmp/kb
timeout 300 rocprofv3 --kernel-trace -d /tmp/ka -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/kt_ts.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kb -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --path autograd > $O/kt_ag.log 2>&1
cp "$(find /tmp/ka -name '*.db' | head -1)" $O/trainstep.db
cp "$(find /tmp/kb -name '*.db' | head -1)" $O/autograd.db
ls -la $O
