#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05final4
timeout 400 python bench.py > gpurun_out/r05final4/bench_c2.log 2>&1
tail -1 gpurun_out/r05final4/bench_c2.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['attention']['frac'], d['frontend'])"
timeout 300 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05final4/bench_c5.json.log
python -c "
import json; d=json.loads(open('gpurun_out/r05final4/bench_c5.json.log').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['attention']['frac'], d['frontend']['avg_us'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
