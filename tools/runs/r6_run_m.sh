#!/bin/bash
# r06 run M: tile choices re-checked under the adopted cache policy (environment knobs on the default library), config #2, ABBA
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
OUT=$O/r06_tiles_under_policy.txt
: > $OUT
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 > $O/r06_m_$tag.log 2>&1; tail -1 $O/r06_m_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; pe=r.get('per_epilogue',{}); print('$tag', d['value'], d['ms_per_step'], r['frac'], ' '.join(f'{k} {v[\"avg_us\"]}' for k,v in pe.items()))" | tee -a $OUT; }
V=("base A=1" "mlp2wg PA_NT_MLP_2WG=1" "rates PA_NT_RATES=r06" "resid6 PASST_AMD_TUNE_RESID=6" "resid7 PASST_AMD_TUNE_RESID=7" "store6 PASST_AMD_TUNE_STORE=6" "store7 PASST_AMD_TUNE_STORE=7" "gelu17 PASST_AMD_TUNE_GELU=17" "dgelu17 PASST_AMD_TUNE_DGELU=17")
for i in 1 2; do
  for v in "${V[@]}"; do set -- $v; t=$1; shift; run ${t}_a$i "$@"; done
  for ((k=${#V[@]}-1; k>=0; k--)); do set -- ${V[$k]}; t=$1; shift; run ${t}_b$i "$@"; done
done
