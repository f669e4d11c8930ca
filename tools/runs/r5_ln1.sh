#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05n
: > gpurun_out/r05n/ln_ab.jsonl
for rep in 1 2; do
for lib in default v1 v2 v2f; do
  for blk in 1024 768 1536; do
    if [ $lib = default ]; then L=""; else L="PASST_AMD_LIB=passt_amd/libpasst_amd_ln_$lib.so"; fi
    env $L PA_LN_BWD_BLOCKS=$blk timeout 100 python tools/bench_ln.py >> gpurun_out/r05n/ln_ab.jsonl 2>/dev/null
  done
done
done
cat gpurun_out/r05n/ln_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['lib'][-12:], d['blocks'], d['M30336'], d['M4236']['bwd_us'])
"
