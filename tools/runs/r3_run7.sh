export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
for a in 0 15 7 14 16 19 31; do
  cd /tmp; rm -rf /tmp/k1
  PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_abl$a.so timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 4 > /dev/null 2>&1
  cd $R
  echo "abl$a $(python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 3 | grep attn_fwd | cut -c60-140)" >> gpurun_out/r7.txt 2>&1
done
cat gpurun_out/r7.txt
