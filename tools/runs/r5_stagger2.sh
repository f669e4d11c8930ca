#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05s
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r05s/b_default_$rep.json
  PASST_AMD_LIB=passt_amd/libpasst_amd_gemm_st1k.so python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r05s/b_st1k_$rep.json
done
python - <<'PY'
import json
for lib in ("default","st1k"):
    for rep in (1,2,3):
        d=json.loads(open(f"gpurun_out/r05s/b_{lib}_{rep}.json").read())
        pe=d["roofline"]["per_epilogue"]
        print(lib, rep, d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["avg_us"] for k,v in pe.items()})
PY
