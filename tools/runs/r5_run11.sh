# Round 5, call 15: PMC counters of the single-pass attention backward (and the two-kernel form) at the bench shape
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05m
mkdir -p $O
cd /tmp; rm -rf /tmp/pa1 /tmp/pa2 /tmp/pb1 /tmp/pb2
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
PASST_AMD_ATTN_BWD=single_pass timeout 200 rocprofv3 --kernel-trace --pmc $C1 -d /tmp/pa1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
PASST_AMD_ATTN_BWD=single_pass timeout 200 rocprofv3 --kernel-trace --pmc $C2 -d /tmp/pa2 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
PASST_AMD_ATTN_BWD=two_pass timeout 200 rocprofv3 --kernel-trace --pmc $C1 -d /tmp/pb1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
PASST_AMD_ATTN_BWD=two_pass timeout 200 rocprofv3 --kernel-trace --pmc $C2 -d /tmp/pb2 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
cd $R
for d in pa1 pa2 pb1 pb2; do echo "== $d" >> $O/attention_pmc.txt; python tools/rocpd_stats.py "$(find /tmp/$d -name '*.db' | head -1)" --top 4 >> $O/attention_pmc.txt 2>&1; done
cat $O/attention_pmc.txt | cut -c1-260
