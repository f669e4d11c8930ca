export TMPDIR=/tmp
mkdir -p gpurun_out/r29
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 > gpurun_out/r29/rev_$i.json.log
PASST_AMD_REVERSE_CONSUMERS=0 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 > gpurun_out/r29/fwd_$i.json.log
done
PASST_AMD_PROFILE_BY_SHAPE=1 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > gpurun_out/r29/rev_shapes.json.log
PASST_AMD_PROFILE_BY_SHAPE=1 PASST_AMD_REVERSE_CONSUMERS=0 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > gpurun_out/r29/fwd_shapes.json.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r29/*_[12].json.log")):
    d = json.loads(open(f).read())
    pe = d["roofline"]["per_epilogue"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["loss"], " ".join(f"{k}={v['avg_us']:.1f}" for k, v in sorted(pe.items())))
for n in ("rev", "fwd"):
    d = json.loads(open(f"gpurun_out/r29/{n}_shapes.json.log").read())
    pe = d["roofline"]["per_epilogue"]
    print(n, " ".join(f"{k}={v['avg_us']:.1f}" for k, v in sorted(pe.items()) if "M30336" in k and ("K3072" in k or "K2304" in k)))
PY
