#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05s
for rep in 1 2 3; do
  for lib in default gemm_nostagger gemm_tnst attn_st; do
    if [ $lib = default ]; then L=""; else L="PASST_AMD_LIB=passt_amd/libpasst_amd_$lib.so"; fi
    env $L python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r05s/e_${lib}_$rep.json
  done
done
python - <<'PY'
import json
for lib in ("default","gemm_nostagger","gemm_tnst","attn_st"):
    ms=[]
    for rep in (1,2,3):
        d=json.loads(open(f"gpurun_out/r05s/e_{lib}_{rep}.json").read()); ms.append(d["ms_per_step"])
    pe=d["roofline"]["per_epilogue"]
    print(lib, ms, {k:v["avg_us"] for k,v in pe.items()}, d["attention"]["fwd_avg_us"], d["attention"]["bwd_avg_us"])
PY
