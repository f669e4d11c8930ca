# Round 5, call 3: single-pass attention backward v2 (absolute addresses, asm fragment reads, pipelined phase 2): parity + A/B;
# where do the per-launch HIP events cost so much at c5?
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05d
mkdir -p $O
timeout 120 python tools/debug_attn_fused.py > $O/debug_fused.txt 2>&1; cat $O/debug_fused.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -8 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
for rep in 1 2; do
PASST_AMD_ATTN_BWD=single_pass python tools/bench_attn.py --tag single_pass --shapes 64x12x474,12x12x353 >> $O/ab_attn.txt 2>&1
PASST_AMD_ATTN_BWD=two_pass python tools/bench_attn.py --tag two_pass --shapes 64x12x474,12x12x353 >> $O/ab_attn.txt 2>&1
done
grep '^{' $O/ab_attn.txt
for rep in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_single_$rep.json.log
PASST_AMD_ATTN_BWD=two_pass python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_twopass_$rep.json.log
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_single_roofline.json.log
python bench.py --config c5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c5_noroofline.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_roofline.json.log
for f in $O/bench_*.json.log; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read())
print(d['value'], d['ms_per_step'], d.get('attention',{}).get('fwd_avg_us'), d.get('attention',{}).get('bwd_avg_us'), d.get('attention',{}).get('frac'), d.get('roofline',{}).get('frac'))"; done
