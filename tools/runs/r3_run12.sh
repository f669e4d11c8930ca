export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 >> gpurun_out/r12.txt
cd /tmp; rm -rf /tmp/k1
timeout 200 rocprofv3 --kernel-trace -d /tmp/k1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 4 2>&1 | grep '"lib"' >> $R/gpurun_out/r12.txt
cd $R
python tools/rocpd_stats.py "$(find /tmp/k1 -name '*.db' | head -1)" --top 4 | grep attn_ | cut -c1-140 >> gpurun_out/r12.txt 2>&1
timeout 600 python bench.py 2>&1 | tail -1 >> gpurun_out/r12_bench.json
cut -c1-700 gpurun_out/r12_bench.json >> gpurun_out/r12.txt
cat gpurun_out/r12.txt
