#!/bin/bash
# r06 run F: the whole GPU suite on the current tree; every NT variant at the seven block shapes (for pick_nt_variant's table);
# MLP-epilogue tile 3 at configs #4 / #5; weight-gradient slice counts; the drop-in path with the bound optimizer
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -60 > $O/r06_f_tests.txt
tail -4 $O/r06_f_tests.txt
python bench_kernels.py --variants 0,1,2,3,9,6,7,8,17,18 --iters 30 --out $O/r06_f_kernels.json 2>&1 | grep "gemm " > $O/r06_nt_variants.jsonl
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; pe=r.get('per_epilogue',{}); print('$1', d['value'], d['ms_per_step'], r.get('frac'), {k: v['avg_us'] for k, v in pe.items()})"; }
run() { tag=$1; shift; cfg=$1; shift; env "$@" python bench.py --config $cfg --no-cpu-baseline --steps 60 > $O/r06_f_$tag.log 2>&1; tail -1 $O/r06_f_$tag.log | line $tag; }
for i in 1 2; do
  run c4_base_$i c4 A=1
  run c4_t3_$i c4 PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=3
  run c5_base_$i c5 A=1
  run c5_t3_$i c5 PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=3
  run c5_t3_b$i c5 PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=3
  run c5_base_b$i c5 A=1
  run c4_t3_b$i c4 PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=3
  run c4_base_b$i c4 A=1
done
for s in 7 3 4 5 6 8 9 10 7; do
  run c2_slices$s c2 PASST_AMD_WGRAD_SLICES=$s
done
run c2_slices_default c2 A=1
for i in 1 2; do
  for cfg in c5 c2; do
    PASST_AMD_NO_FLAT_GRADS=1 python bench.py --config $cfg --path autograd --optimizer pa_adamw --mixup pa --steps 100 --no-cpu-baseline --no-roofline > $O/r06_f_autograd_${cfg}_unbound$i.log 2>&1
    python bench.py --config $cfg --path autograd --optimizer pa_adamw --mixup pa --steps 100 --no-cpu-baseline --no-roofline > $O/r06_f_autograd_${cfg}_bound$i.log 2>&1
    python bench.py --config $cfg --steps 100 --no-cpu-baseline --no-roofline > $O/r06_f_trainstep_${cfg}_$i.log 2>&1
    for k in unbound bound; do tail -1 $O/r06_f_autograd_${cfg}_$k$i.log | line autograd_${cfg}_$k$i; done
    tail -1 $O/r06_f_trainstep_${cfg}_$i.log | line trainstep_${cfg}_$i
  done
done
cat $O/r06_nt_variants.jsonl | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['kernel'], d['us'], d['tflops'])"
