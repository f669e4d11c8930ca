#!/bin/bash
# r06 run O: runtime environment knobs on the launch path (kernel arguments in device memory), ESC-50 (B = 12: ~370 short launches per step) and config #2
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
OUT=$O/r06_runtime_env.txt
: > $OUT
run() { tag=$1; cfg=$2; steps=$3; shift 3; env "$@" python bench.py --no-cpu-baseline --no-roofline --config $cfg --steps $steps $EXTRA > $O/r06_o_$tag.log 2>&1; tail -1 $O/r06_o_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])" | tee -a $OUT; }
for i in 1 2; do
  run c5_base_a$i c5 300 A=1
  run c5_devkernarg_a$i c5 300 HIP_FORCE_DEV_KERNARG=1
  run c5_devkernarg_b$i c5 300 HIP_FORCE_DEV_KERNARG=1
  run c5_base_b$i c5 300 A=1
  run c5graph_base_$i c5 300 A=1 PASST_BENCH_GRAPH=1
done
run c2_base c2 60 A=1
run c2_devkernarg c2 60 HIP_FORCE_DEV_KERNARG=1
run c5auto_base c5 300 A=1 BENCH_EXTRA=1
