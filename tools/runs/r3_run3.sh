# static wave priorities: A/B on the attention kernels (events + rocprof kernel times)
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
for lib in libpasst_amd_attn_prio0.so libpasst_amd.so libpasst_amd_attn_prio2.so; do
  PASST_AMD_LIB=$R/passt_amd/$lib timeout 120 python tools/bench_attn.py --shapes 64x12x474,12x12x353 2>&1 | grep lib >> gpurun_out/r3_prio.txt
  cd /tmp; rm -rf /tmp/k1
  PASST_AMD_LIB=$R/passt_amd/$lib timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 4 > /dev/null 2>&1
  cd $R
  echo "== $lib" >> gpurun_out/r3_prio.txt
  python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 3 | cut -c1-140 >> gpurun_out/r3_prio.txt 2>&1
done
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -5 >> gpurun_out/r3_prio.txt
cat gpurun_out/r3_prio.txt
