set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -15 > gpurun_out/r1_pytest_attn.txt
for lib in libpasst_amd_r2.so libpasst_amd.so libpasst_amd_attn_noones.so libpasst_amd_attn_w2.so; do
  PASST_AMD_LIB=$PWD/passt_amd/$lib timeout 300 python tools/bench_attn.py >> gpurun_out/r1_bench_attn.txt 2>&1
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o attn -- python $GRAFT_REPO_ROOT/tools/bench_attn.py --shapes 64x12x474 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof1 -name "*kernel_stats*" | head -3
f=$(find /tmp/prof1 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r1_attn_kernel_stats.csv
cat gpurun_out/r1_pytest_attn.txt gpurun_out/r1_bench_attn.txt
head -8 gpurun_out/r1_attn_kernel_stats.csv | cut -c1-200
