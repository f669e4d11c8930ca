#!/bin/bash
# r06 run P: per-role cache policy at ESC-50's batch (M = 4 236: every activation fits the Infinity Cache), ABBA, 300-step lines
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
OUT=$O/r06_cache_policy_c5.txt
: > $OUT
run() { tag=$1; lib=$2; PASST_AMD_LIB=$R/passt_amd/$lib python bench.py --no-cpu-baseline --no-roofline --config c5 --steps 300 > $O/r06_p_$tag.log 2>&1; tail -1 $O/r06_p_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])" | tee -a $OUT; }
for i in 1 2; do
  run base_a$i libpasst_amd.so
  for v in $VARS; do run ${v}_a$i libpasst_amd_var_$v.so; done
  for v in $(echo $VARS | tr ' ' '\n' | tac); do run ${v}_b$i libpasst_amd_var_$v.so; done
  run base_b$i libpasst_amd.so
done
