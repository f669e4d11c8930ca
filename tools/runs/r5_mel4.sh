#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05m
timeout 400 python -m pytest tests/test_gpu_model.py -x -q -k "frontend" > gpurun_out/r05m/pytest_mel4.txt 2>&1
tail -3 gpurun_out/r05m/pytest_mel4.txt
timeout 120 python tools/bench_mel.py > gpurun_out/r05m/mel_isolated_e.json 2>/dev/null
timeout 120 python tools/bench_mel.py > gpurun_out/r05m/mel_isolated_f.json 2>/dev/null
cat gpurun_out/r05m/mel_isolated_e.json gpurun_out/r05m/mel_isolated_f.json
PASST_AMD_LIB=passt_amd/libpasst_amd_mel_probe0.so timeout 200 python tools/probe_mel.py > gpurun_out/r05m/mel_probe4.json 2> gpurun_out/r05m/mel_probe4.err
python - <<'PY'
import json
t=open('gpurun_out/r05m/mel_probe4.json').read()
d=json.loads(t[t.index('{'):])
print({k:v['median'] for k,v in d['phases'].items()}, d['wave_lifetime'], d['launch_span_us'], d['workgroup_starts_per_10us'])
PY
