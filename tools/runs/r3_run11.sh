export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "attention or colscale" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | head -12 >> gpurun_out/r11.txt
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -6 >> gpurun_out/r11.txt
for lib in libpasst_amd_r2.so libpasst_amd.so; do
  cd /tmp; rm -rf /tmp/k1
  PASST_AMD_LIB=$R/passt_amd/$lib timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 4 2>&1 | grep '"lib"' >> $R/gpurun_out/r11.txt
  cd $R
  python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 4 | grep attn_ | cut -c1-140 >> gpurun_out/r11.txt 2>&1
done
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600 >> gpurun_out/r11.txt
cat gpurun_out/r11.txt
