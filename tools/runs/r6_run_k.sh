#!/bin/bash
# r06 run K: cache-policy sweep (nt on the GEMM memory instructions by role, csrc/gemm.hip PA_AUX_*), config #2 training step, ABBA
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
OUT=$O/r06_cache_policy_raw.txt
: > $OUT
run() { tag=$1; lib=$2; PASST_AMD_LIB=$R/passt_amd/$lib python bench.py --no-cpu-baseline --steps ${STEPS:-60} > $O/r06_k_$tag.log 2>&1; tail -1 $O/r06_k_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; pe=r.get('per_epilogue',{}); print('$tag', d['value'], d['ms_per_step'], r['frac'], ' '.join(f'{k} {v[\"avg_us\"]}' for k,v in pe.items()))" | tee -a $OUT; }
VARS="${VARS:-pre_nt ldaux_nt dmaA_nt dmaB_nt dmaTN_nt out_nt res_nt allst_nt}"
for i in 1 2; do
  run base_a$i libpasst_amd.so
  for v in $VARS; do run ${v}_a$i libpasst_amd_var_$v.so; done
  for v in $(echo $VARS | tr ' ' '\n' | tac); do run ${v}_b$i libpasst_amd_var_$v.so; done
  run base_b$i libpasst_amd.so
done
