# split-K NT (ESC-50 shapes + prefix-only tail): parity tests, bench c2 / c5 with and without it, one step's launch sequence
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r14
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split_k or colscale or gemm_store_bias or epilogues" 2>&1 | tail -4 > $O/pytest.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_c2.json.log
PA_NT_SPLITK=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_nosplit.json.log
PASST_AMD_PROFILE_BY_SHAPE=1 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json.log
PA_NT_SPLITK=0 PASST_AMD_PROFILE_BY_SHAPE=1 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_nosplit.json.log
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | head -1)" --steps 3 --top 45 --sequence 420 | cut -c1-140 > $O/c2_sequence.txt
python - <<'PY'
import json
for n in ("bench_c2", "bench_c2_nosplit", "bench_c5", "bench_c5_nosplit"):
    try:
        d = json.loads(open(f"gpurun_out/r14/{n}.json.log").read())
    except Exception as e:
        print(n, "unreadable", e); continue
    print(n, d["value"], d["ms_per_step"], "gemm", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "attn", d.get("attention", {}).get("frac"))
    if "c5" in n:
        for k, v in sorted(d["roofline"]["per_epilogue"].items()):
            print("   ", k, v["avg_us"], v["tflops"])
PY
cat $O/pytest.txt
