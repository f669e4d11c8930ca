#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05m
timeout 400 python -m pytest tests/test_gpu_model.py -x -q -k "frontend or train_step or speed" > gpurun_out/r05m/pytest_mel5.txt 2>&1
tail -3 gpurun_out/r05m/pytest_mel5.txt
timeout 120 python tools/bench_mel.py > gpurun_out/r05m/mel_isolated_g.json 2>/dev/null
timeout 120 python tools/bench_mel.py > gpurun_out/r05m/mel_isolated_h.json 2>/dev/null
cat gpurun_out/r05m/mel_isolated_g.json gpurun_out/r05m/mel_isolated_h.json
PASST_AMD_LIB=passt_amd/libpasst_amd_mel_probe.so timeout 200 python tools/probe_mel.py > gpurun_out/r05m/mel_probe5.json 2> gpurun_out/r05m/mel_probe5.err
python - <<'PY'
import json
t=open('gpurun_out/r05m/mel_probe5.json').read()
d=json.loads(t[t.index('{'):])
print({k:v['median'] for k,v in d['phases'].items()}, d['wave_lifetime'], d['launch_span_us'], d['workgroup_starts_per_10us'])
PY
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05m/bench_c2.log 2>&1; tail -1 gpurun_out/r05m/bench_c2.log | cut -c1-300
timeout 300 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r05m/bench_c5.log 2>&1; tail -1 gpurun_out/r05m/bench_c5.log | cut -c1-200
