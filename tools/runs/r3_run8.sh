export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | grep -E "Error|assert|passed|failed" | head -5 >> gpurun_out/r8.txt
for lib in libpasst_amd_attn_nopipe.so libpasst_amd.so; do
  cd /tmp; rm -rf /tmp/k1
  PASST_AMD_LIB=$R/passt_amd/$lib timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474,12x12x353 --iters 4 2>&1 | grep '"lib"' >> $R/gpurun_out/r8.txt
  cd $R
  python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 3 | grep attn_fwd | cut -c1-140 >> gpurun_out/r8.txt 2>&1
done
cat gpurun_out/r8.txt
