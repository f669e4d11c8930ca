export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 >> gpurun_out/r13.txt
timeout 900 python bench.py 2>&1 | tail -1 >> gpurun_out/r13_bench.json
python - <<'PY' >> gpurun_out/r13.txt
import json
d=json.loads(open('gpurun_out/r13_bench.json').read().strip().split('\n')[-1])
for k in ('value','ms_per_step','parity_mode_clips_s','cpu_baseline','cpu_baseline_forward','attention'):
    print(k, d.get(k))
print('roofline', {k:v for k,v in d['roofline'].items() if k in ('achieved','frac','traffic','traffic_source','per_epilogue')})
PY
cat gpurun_out/r13.txt
