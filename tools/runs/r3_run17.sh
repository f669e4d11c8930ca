# finishing stream (split-K reductions + per-block optimizer next to the remaining backward): model / ddp tests, A/B in the step
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py -q -m gpu -x 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fin.json.log
PASST_AMD_NO_FINISH_STREAM=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_nofin.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > $O/bench_fin_100.json.log
PASST_AMD_NO_FINISH_STREAM=1 python bench.py --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > $O/bench_nofin_100.json.log
python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > $O/bench_c5_fin.json.log
PASST_AMD_NO_FINISH_STREAM=1 python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 100 2>/dev/null | tail -1 > $O/bench_c5_nofin.json.log
python - <<'PY'
import json
for n in ("bench_fin", "bench_nofin", "bench_fin_100", "bench_nofin_100", "bench_c5_fin", "bench_c5_nofin"):
    try:
        d = json.loads(open(f"gpurun_out/r17/{n}.json.log").read())
    except Exception as e:
        print(n, "unreadable", e); continue
    print(n, d["value"], d["ms_per_step"], d.get("loss"))
PY
