#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05s
i=0
for lib in gemm_nostagger default default gemm_nostagger gemm_nostagger default default gemm_nostagger default gemm_nostagger gemm_nostagger default; do
  i=$((i+1))
  if [ $lib = default ]; then L=""; else L="PASST_AMD_LIB=passt_amd/libpasst_amd_$lib.so"; fi
  env $L python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r05s/f_${i}_${lib}.json
done
python - <<'PY'
import json,glob,re
rows=[]
for f in sorted(glob.glob("gpurun_out/r05s/f_*_*.json"), key=lambda x:int(re.search(r"f_(\d+)_",x).group(1))):
    d=json.loads(open(f).read()); rows.append((re.search(r"f_\d+_(.*)\.json",f).group(1), d["ms_per_step"]))
print(rows)
import statistics
for k in ("default","gemm_nostagger"):
    v=[m for n,m in rows if n==k]; print(k, round(statistics.mean(v),3), v)
PY
