export TMPDIR=/tmp
mkdir -p gpurun_out/r20
python bench.py --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | tail -1 > gpurun_out/r20/base.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 60 --overlap-wgrad 2>/dev/null | tail -1 > gpurun_out/r20/overlap.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | tail -1 > gpurun_out/r20/base2.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 60 --overlap-wgrad 2>/dev/null | tail -1 > gpurun_out/r20/overlap2.json.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r20/*.json.log")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("loss"))
PY
