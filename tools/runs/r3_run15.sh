# stream-K weight gradients: parity tests, A/B against the split-K form in the step (c2, c5), kernel durations under rocprof
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r15
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad or split_k" 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2.json.log
PASST_AMD_TN_STREAMK=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_split.json.log
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_b.json.log
PASST_AMD_TN_STREAMK=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_split_b.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json.log
PASST_AMD_TN_STREAMK=0 python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_split.json.log
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | head -1)" --steps 4 --top 16 | cut -c1-140 > $O/c2_kernel_stats.txt
python - <<'PY'
import json
for n in ("bench_c2", "bench_c2_split", "bench_c2_b", "bench_c2_split_b", "bench_c5", "bench_c5_split"):
    try:
        d = json.loads(open(f"gpurun_out/r15/{n}.json.log").read())
    except Exception as e:
        print(n, "unreadable", e); continue
    w = d["roofline"]["per_epilogue"].get("wgrad_tn", {})
    print(n, d["value"], d["ms_per_step"], "gemm", d["roofline"]["frac"], "wgrad", w.get("avg_us"), w.get("tflops"))
PY
head -14 $O/c2_kernel_stats.txt
