# Round 5, call 11: sustained run, isolated kernels, two more default lines (box-to-box / run-to-run spread)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05j
mkdir -p $O
python bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_sustained_2000.json.log
python bench_kernels.py --out $O/isolated_kernels.json > $O/bench_kernels.log 2>&1
python bench.py 2>/dev/null | tail -1 > $O/bench_c2_default_a.json.log
python bench.py 2>/dev/null | tail -1 > $O/bench_c2_default_b.json.log
for f in $O/bench_c2*.json.log; do python -c "
import json
d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d.get('loss'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('attention',{}).get('frac'))"; done
tail -5 $O/bench_kernels.log
