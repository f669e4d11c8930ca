# per-bucket optimizer launches from the backward (gradients still in the caches) vs one launch after it; one GPU
export TMPDIR=/tmp
mkdir -p gpurun_out/r18
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > gpurun_out/r18/block_$i.json.log
PASST_AMD_BLOCK_OPT=0 python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > gpurun_out/r18/single_$i.json.log
done
python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > gpurun_out/r18/c5_block.json.log
PASST_AMD_BLOCK_OPT=0 python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > gpurun_out/r18/c5_single.json.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r18/*.json.log")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("loss"))
PY
