#!/bin/bash
# r06 run D: W16 attention backward (parity + timing vs the eight-wave single pass, ABBA), in-step A/B of the MLP-epilogue GEMM tiles,
# fused optimizer + staging after the load hoist
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -6 > $O/r06_d_attn_tests.txt
tail -3 $O/r06_d_attn_tests.txt
for m in single_pass single_pass_w16 single_pass_w16 single_pass two_pass; do
  PASST_AMD_ATTN_BWD=$m python tools/bench_attn.py --tag $m --shapes 64x12x474,32x16x790 >> $O/r06_d_attn_w16.jsonl 2>&1
done
cat $O/r06_d_attn_w16.jsonl
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline > $O/r06_d_step_$tag.log 2>&1; tail -1 $O/r06_d_step_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], r['frac'], {k: round(v['us'],1) if isinstance(v, dict) and 'us' in v else v for k,v in r.get('per_epilogue',{}).items()}, d.get('attention'))"; }
run base_1 A=1
run gelu6_1 PASST_AMD_TUNE_GELU=6
run dgelu6_1 PASST_AMD_TUNE_DGELU=6
run both6_1 PASST_AMD_TUNE_GELU=6 PASST_AMD_TUNE_DGELU=6
run gelu3_1 PASST_AMD_TUNE_GELU=3
run gelu3d6_1 PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=6
run gelu9_1 PASST_AMD_TUNE_GELU=9
run w16_1 PASST_AMD_ATTN_BWD=single_pass_w16
run w16_2 PASST_AMD_ATTN_BWD=single_pass_w16
run gelu9_2 PASST_AMD_TUNE_GELU=9
run gelu3d6_2 PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=6
run gelu3_2 PASST_AMD_TUNE_GELU=3
run both6_2 PASST_AMD_TUNE_GELU=6 PASST_AMD_TUNE_DGELU=6
run dgelu6_2 PASST_AMD_TUNE_DGELU=6
run gelu6_2 PASST_AMD_TUNE_GELU=6
run base_2 A=1
run sepstage PASST_AMD_NO_FUSED_STAGE=1
run base_3 A=1
