#!/bin/bash
# r06 run E: counters of the sixteen-wave single-pass attention backward against the eight-wave one; ablation builds (no phase 2 / no
# phase 2 and no dV, dK products) of both; repeated in-step A/B of the fc1 + GELU tile (ABBA)
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/r06_attention_w16_pmc.txt
for m in single_pass single_pass_w16; do
  rm -rf /tmp/pa1 /tmp/pa2
  PASST_AMD_ATTN_BWD=$m timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d /tmp/pa1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
  PASST_AMD_ATTN_BWD=$m timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pa2 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
  echo "== $m" >> $O/r06_attention_w16_pmc.txt
  for d in pa1 pa2; do python $R/tools/rocpd_stats.py "$(find /tmp/$d -name '*.db' | head -1)" --top 2 >> $O/r06_attention_w16_pmc.txt 2>&1; done
done
cd $R
: > $O/r06_attention_w16_abl.jsonl
for lib in libpasst_amd.so libpasst_amd_attn_abl1.so libpasst_amd_attn_abl5.so; do
  for m in single_pass single_pass_w16; do
    PASST_AMD_LIB=$R/passt_amd/$lib PASST_AMD_ATTN_BWD=$m python tools/bench_attn.py --tag "$lib:$m" --shapes 64x12x474 >> $O/r06_attention_w16_abl.jsonl 2>/dev/null
  done
done
cat $O/r06_attention_w16_abl.jsonl
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 60 > $O/r06_e_step_$tag.log 2>&1; tail -1 $O/r06_e_step_$tag.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; pe=r.get('per_epilogue',{}); print('$tag', d['value'], d['ms_per_step'], r['frac'], 'gelu', pe['gelu']['avg_us'], 'dgelu', pe['dgelu']['avg_us'])"; }
for i in 1 2 3; do
  run base_a$i A=1
  run gelu3_a$i PASST_AMD_TUNE_GELU=3
  run gelu3d6_a$i PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=6
  run gelu3d3_a$i PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=3
  run gelu3d3_b$i PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=3
  run gelu3d6_b$i PASST_AMD_TUNE_GELU=3 PASST_AMD_TUNE_DGELU=6
  run gelu3_b$i PASST_AMD_TUNE_GELU=3
  run base_b$i A=1
done
