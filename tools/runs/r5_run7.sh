# Round 5, call 8: full GPU suite + smoke + headline line on the state with the single-pass attention backward as the default
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05h
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -4 > $O/smoke.txt; cat $O/smoke.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json.log
python bench.py --speedtest 2>/dev/null | grep '^{' > $O/bench_speedtest.json.log
for f in $O/bench_c*.json.log; do python -c "
import json,sys
d=json.loads(open('$f').read())
print('$f', d['value'], d['ms_per_step'], d.get('attention',{}).get('fwd_avg_us'), d.get('attention',{}).get('bwd_avg_us'), d.get('attention',{}).get('frac'), d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))"; done
cut -c1-330 $O/bench_speedtest.json.log
