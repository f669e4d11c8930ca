# Round 5, call 16: staggered phase order of the two waves of a SIMD in the single-pass attention backward
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05n
mkdir -p $O
timeout 120 python tools/debug_attn_fused.py > $O/debug_fused.txt 2>&1; cat $O/debug_fused.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -4
export PASST_AMD_ATTN_BWD=single_pass
for rep in 1 2 3; do
python tools/bench_attn.py --tag stagger --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_nostagger.so python tools/bench_attn.py --tag nostagger --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
done
python -c "
import json
for l in open('$O/ab.txt'):
    d=json.loads(l); print(d['lib'], d['bwd_us'])"
