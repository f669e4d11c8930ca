# A/B of two library builds on the attention kernels (rocprof kernel times) + the stress / determinism check
# usage: bash tools/runs/ab_attn.sh libA.so libB.so ...
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
OUT=gpurun_out/ab_attn.txt; rm -f $OUT
for lib in "$@"; do
  for rep in 1 2; do
    cd /tmp; rm -rf /tmp/k1
    PASST_AMD_LIB=$R/passt_amd/$lib timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 4 > /dev/null 2>&1
    cd $R
    echo "== $lib rep $rep: $(python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 4 | grep attn_ | awk '{printf "%s %s | ", substr($1,9,14), $4}')" >> $OUT
  done
done
PASST_AMD_LIB=$R/passt_amd/${@: -1} python tools/debug_attn.py 64 2>&1 | tail -1 >> $OUT
cat $OUT
