# Round 5, call 6: single-pass attention backward v3 (coalesced prologue, V via LDS, early column fragments; interleave variant)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05f
mkdir -p $O
timeout 120 python tools/debug_attn_fused.py > $O/debug_fused.txt 2>&1; cat $O/debug_fused.txt
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_il.so timeout 120 python tools/debug_attn_fused.py 474 67 > $O/debug_fused_il.txt 2>&1; cat $O/debug_fused_il.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -5 > $O/pytest_attn.txt; cat $O/pytest_attn.txt
export PASST_AMD_ATTN_BWD=single_pass
for rep in 1 2; do
python tools/bench_attn.py --tag v3 --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_il.so python tools/bench_attn.py --tag v3_interleave --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_abl15.so python tools/bench_attn.py --tag abl15 --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_abl1.so python tools/bench_attn.py --tag abl1 --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
PASST_AMD_ATTN_BWD=two_pass python tools/bench_attn.py --tag two_pass --shapes 64x12x474 2>/dev/null | grep '^{' >> $O/ab.txt
done
python -c "
import json
for l in open('$O/ab.txt'):
    d=json.loads(l); print(d['lib'], d['bwd_us'])"
