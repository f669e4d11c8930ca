export TMPDIR=/tmp
mkdir -p gpurun_out/r23
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 > gpurun_out/r23/persist_$i.json.log
PA_NT_PERSISTENT=0 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 > gpurun_out/r23/nopersist_$i.json.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r23/*.json.log")):
    d = json.loads(open(f).read())
    pe = d["roofline"]["per_epilogue"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], " ".join(f"{k}={v['avg_us']:.1f}" for k, v in sorted(pe.items())))
PY
