#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05s
for rep in 1 2; do
for lib in default st1 st2 st4; do
  if [ $lib = default ]; then L=""; else L="PASST_AMD_LIB=passt_amd/libpasst_amd_gemm_$lib.so"; fi
  env $L timeout 300 python bench_kernels.py --out gpurun_out/r05s/k_${lib}_$rep.json > gpurun_out/r05s/k_${lib}_$rep.log 2>&1
done
done
python - <<'PY'
import json
rows={}
for lib in ("default","st1","st2","st4"):
    for rep in (1,2):
        try: d=json.load(open(f"gpurun_out/r05s/k_{lib}_{rep}.json"))
        except Exception as e: print(lib,rep,e); continue
        for r in d if isinstance(d,list) else d.get("kernels",d):
            k=r["kernel"]
            if k.startswith("gemm") and "variant 0" in k: rows.setdefault(k[:40],{}).setdefault(lib,[]).append(r["us"])
for k,v in rows.items(): print(k.ljust(42), {l:[round(x,1) for x in u] for l,u in v.items()})
PY
