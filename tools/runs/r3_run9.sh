export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "attention" 2>&1 | grep -E "Error|assert|passed|failed" | head -8 >> gpurun_out/r9.txt
for lib in libpasst_amd_attn_nopipe.so libpasst_amd.so libpasst_amd_attn_ones1.so libpasst_amd_attn_w3.so libpasst_amd_attn_w3ones.so; do
  cd /tmp; rm -rf /tmp/k1
  PASST_AMD_LIB=$R/passt_amd/$lib timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 4 2>&1 | grep '"lib"' >> $R/gpurun_out/r9.txt
  cd $R
  python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 3 | grep attn_fwd | cut -c1-140 >> gpurun_out/r9.txt 2>&1
done
cat gpurun_out/r9.txt
