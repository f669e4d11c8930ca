#!/bin/bash
# LDS-free (v3) epilogues: equality test against v2, the model parity tests, A/B of the step against the round-3 arrangement
R=$PWD
O=$R/gpurun_out/r04g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "epilogue_v3 or gemm" 2>&1 | tail -5 > $O/pytest_kern.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -5 > $O/pytest_model.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_v3_$i.json
PA_EPILOGUE_V3=0 PASST_AMD_DGELU_COLSUM=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_v2_$i.json
done
PASST_AMD_DGELU_COLSUM=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_v3_dgelucolsum.json
cat $O/pytest_kern.txt $O/pytest_model.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04g/bench_*.json")):
    try:
        d=json.loads(open(f).read())
        pe=d["roofline"]["per_epilogue"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["avg_us"],v["tflops"]) for k,v in pe.items()})
    except Exception as e: print(f, "ERR", e)
PY
