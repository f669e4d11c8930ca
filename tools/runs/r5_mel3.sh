#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05m
for a in 0 1 2; do
  PASST_AMD_LIB=passt_amd/libpasst_amd_mel_probe$a.so timeout 200 python tools/probe_mel.py > gpurun_out/r05m/abl_probe$a.json 2> gpurun_out/r05m/abl_probe$a.err
  PASST_AMD_LIB=passt_amd/libpasst_amd_mel_probe$a.so timeout 120 python tools/bench_mel.py > gpurun_out/r05m/abl_iso$a.json 2>/dev/null
done
python - <<'PY'
import json
for a in (0,1,2):
    t=open(f'gpurun_out/r05m/abl_probe{a}.json').read()
    d=json.loads(t[t.index('{'):])
    i=json.loads(open(f'gpurun_out/r05m/abl_iso{a}.json').read().strip().split('\n')[-1])
    print(a, {k:v['median'] for k,v in d['phases'].items()}, d['wave_lifetime']['median'], d['launch_span_us'], d['workgroup_starts_per_10us'], i['B64_L320000']['us'])
PY
