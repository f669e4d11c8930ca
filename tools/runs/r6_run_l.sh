#!/bin/bash
# r06 run L: the adopted cache policy (library default) against -DPA_NO_CACHE_POLICY on configs #2, #4, #5, the eval forward and the
# drop-in path, ABBA
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
OUT=$O/r06_cache_policy_configs.txt
: > $OUT
run() { tag=$1; lib=$2; shift 2; PASST_AMD_LIB=$R/passt_amd/$lib python bench.py --no-cpu-baseline "$@" > $O/r06_l_$tag.log 2>&1; tail -1 $O/r06_l_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; pe=r.get('per_epilogue',{}); a=d.get('attention') or {}; print('$tag', d['value'], d['ms_per_step'], r.get('frac'), ' '.join(f'{k} {v[\"avg_us\"]}' for k,v in pe.items()), 'attn', a.get('fwd_avg_us'), a.get('bwd_avg_us'))" | tee -a $OUT; }
for i in 1 2; do
  for c in c2 c4 c5; do
    run ${c}_policy_a$i libpasst_amd.so --config $c --steps 60
    run ${c}_nopolicy_a$i libpasst_amd_var_nopolicy.so --config $c --steps 60
    run ${c}_nopolicy_b$i libpasst_amd_var_nopolicy.so --config $c --steps 60
    run ${c}_policy_b$i libpasst_amd.so --config $c --steps 60
  done
done
run c5b96_policy libpasst_amd.so --config c5 --batch 96 --steps 60
run c5b96_nopolicy libpasst_amd_var_nopolicy.so --config c5 --batch 96 --steps 60
run c2fp32_policy libpasst_amd.so --precision fp32 --steps 10
run c2fp32_nopolicy libpasst_amd_var_nopolicy.so --precision fp32 --steps 10
run c2auto_policy libpasst_amd.so --path autograd --optimizer pa_adamw --mixup pa --steps 60
run c2auto_nopolicy libpasst_amd_var_nopolicy.so --path autograd --optimizer pa_adamw --mixup pa --steps 60
for l in libpasst_amd.so libpasst_amd_var_nopolicy.so; do PASST_AMD_LIB=$R/passt_amd/$l python tools/bench_eval.py 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT; done
