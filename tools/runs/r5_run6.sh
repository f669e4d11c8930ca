# Round 5, call 7: prologue variants of the single-pass attention backward
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05g
mkdir -p $O
timeout 120 python tools/debug_attn_fused.py 474 67 20 > $O/debug_fused.txt 2>&1; cat $O/debug_fused.txt
export PASST_AMD_ATTN_BWD=single_pass
for rep in 1 2; do
python tools/bench_attn.py --tag pro0 --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
for v in pro1 pro2 pro3 abl15; do
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_$v.so python tools/bench_attn.py --tag $v --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
done
done
python -c "
import json
for l in open('$O/ab.txt'):
    d=json.loads(l); print(d['lib'], d['bwd_us'])"
