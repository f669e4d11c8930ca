# Round 5, call 5: ablation timing of the single-pass attention backward (which part costs what)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05e
mkdir -p $O
export PASST_AMD_ATTN_BWD=single_pass
python tools/bench_attn.py --tag full --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ablation.txt
for v in 1 2 4 8 16 6 14 15; do
  PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_abl$v.so python tools/bench_attn.py --tag abl$v --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ablation.txt
done
python tools/bench_attn.py --tag full --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ablation.txt
python -c "
import json
for l in open('$O/ablation.txt'):
    d=json.loads(l); print(d['lib'], d['bwd_us'])"
