# forward-kernel ablations (PA_ATTN_ABLATE builds) + SQ counters of the new kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
for a in 0 1 2 3 4 8 16 19 20 23 31; do
  lib=libpasst_amd_attn_abl$a.so; [ $a = 0 ] && lib=libpasst_amd.so
  PASST_AMD_LIB=$R/passt_amd/$lib timeout 120 python tools/bench_attn.py --shapes 64x12x474 --tag abl$a 2>&1 | grep lib >> gpurun_out/r2_ablate.txt
done
cd /tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d /tmp/p1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD -d /tmp/p2 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT -d /tmp/p3 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
cd $R
for d in p1 p2 p3; do python tools/rocpd_stats.py "$(find /tmp/$d -name '*.db' | head -1)" --top 4 >> gpurun_out/r2_attn_pmc.txt 2>&1; done
cat gpurun_out/r2_ablate.txt; cut -c1-1200 gpurun_out/r2_attn_pmc.txt
