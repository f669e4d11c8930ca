export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
python bench.py --config c5 2>/dev/null | tail -1 > gpurun_out/c5_bench.json
PASST_AMD_PROFILE_BY_SHAPE=1 python bench.py --config c5 2>/dev/null | tail -1 > gpurun_out/c5_bench_shapes.json
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- python $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | head -1)" --steps 6 --top 22 | cut -c1-150 > gpurun_out/c5_kernel_stats.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c5_bench.json').read())
print(d['value'], d['ms_per_step'], d.get('attention',{}).get('frac'), d['roofline']['frac'])
d=json.loads(open('gpurun_out/c5_bench_shapes.json').read())
for k,v in sorted(d['roofline']['per_epilogue'].items()): print(k, v)
PY
cat gpurun_out/c5_kernel_stats.txt
