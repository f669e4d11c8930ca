# Round 5, call 2: single-pass attention backward -- parity (fused vs two-pass, then the fp64 tests), A/B timing; c5 regression probe
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05b
mkdir -p $O
timeout 120 python tools/debug_attn_fused.py > $O/debug_fused.txt 2>&1; cat $O/debug_fused.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -25 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
python tools/bench_attn.py --tag fused --shapes 64x12x474,12x12x353 > $O/ab_attn.txt 2>&1
PASST_AMD_ATTN_BWD=two_pass python tools/bench_attn.py --tag two_pass --shapes 64x12x474,12x12x353 >> $O/ab_attn.txt 2>&1
cat $O/ab_attn.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_fused.json.log
PASST_AMD_ATTN_BWD=two_pass python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_twopass.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_20.json.log
python bench.py --config c5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c5_20_noroofline.json.log
python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_c5_60_noroofline.json.log
for f in $O/bench_*.json.log; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read())
print(d['value'], d['ms_per_step'], d.get('attention',{}).get('fwd_avg_us'), d.get('attention',{}).get('bwd_avg_us'), d.get('attention',{}).get('frac'), d.get('roofline',{}).get('frac'))"; done
