#!/bin/bash
# r06 run G: suite after the AdamW arithmetic pin / bind fix; pick-table A/B (PA_NT_RATES=r01 vs the round-6 refit) over every bench
# configuration; epilogue-v2 two-workgroup tiles for the plain-store / residual GEMMs in the step
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -150 > $O/r06_g_tests.txt
tail -12 $O/r06_g_tests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; pe=r.get('per_epilogue',{}); print('$1', d['value'], d['ms_per_step'], r.get('frac'), {k: v['avg_us'] for k, v in pe.items()})"; }
run() { tag=$1; shift; cfg=$1; shift; env "$@" python bench.py --config $cfg --no-cpu-baseline --steps 60 > $O/r06_g_$tag.log 2>&1; tail -1 $O/r06_g_$tag.log | line $tag; }
for i in 1 2; do
  for cfg in c2 c4 c5; do
    run ${cfg}_r01_a$i $cfg PA_NT_RATES=r01 PA_NT_MLP_2WG=0
    run ${cfg}_r06_a$i $cfg PA_NT_MLP_2WG=0
    run ${cfg}_r06mlp_a$i $cfg A=1
    run ${cfg}_r06mlp_b$i $cfg A=1
    run ${cfg}_r06_b$i $cfg PA_NT_MLP_2WG=0
    run ${cfg}_r01_b$i $cfg PA_NT_RATES=r01 PA_NT_MLP_2WG=0
  done
done
for t in r01 r06; do
  for b in 64 2; do
    echo "eval $t batch $b: $(PA_NT_RATES=$t python tools/bench_eval.py --batch $b 2>/dev/null | tail -1 | cut -c1-200)"
  done
  echo "c5 batch 96 $t: $(PA_NT_RATES=$t python bench.py --config c5 --batch 96 --steps 40 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-160)"
done
for i in 1 2; do
  run c2_base_s$i c2 A=1
  run c2_store13_s$i c2 PASST_AMD_TUNE_STORE=13
  run c2_resid13_s$i c2 PASST_AMD_TUNE_RESID=13
  run c2_store19_s$i c2 PASST_AMD_TUNE_STORE=19
  run c2_resid19_s$i c2 PASST_AMD_TUNE_RESID=19
  run c2_both13_s$i c2 PASST_AMD_TUNE_RESID=13 PASST_AMD_TUNE_STORE=13
done
for i in 1 2; do
  for cfg in c5 c2; do
    python bench.py --config $cfg --path autograd --optimizer pa_adamw --mixup pa --steps 100 --no-cpu-baseline --no-roofline > $O/r06_g_autograd_${cfg}_bound$i.log 2>&1
    tail -1 $O/r06_g_autograd_${cfg}_bound$i.log | line autograd_${cfg}_bound$i
  done
done
