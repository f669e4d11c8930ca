# A/B of GEMM translation-unit variants inside the step (same box, same call): 2 k-substeps per barrier pair, L2 patch shapes
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r16
mkdir -p $O
for v in base sub2 gm8 gm2 base2; do
  L=$R/passt_amd/libpasst_amd_gemm_$v.so
  if [ $v = base ] || [ $v = base2 ]; then L=$R/passt_amd/libpasst_amd.so; fi
  PASST_AMD_LIB=$L python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$v.json.log
done
python - <<'PY'
import json
for n in ("base", "sub2", "gm8", "gm2", "base2"):
    try:
        d = json.loads(open(f"gpurun_out/r16/bench_{n}.json.log").read())
    except Exception as e:
        print(n, "unreadable", e); continue
    pe = d["roofline"]["per_epilogue"]
    print(n, d["value"], d["ms_per_step"], "gemm", d["roofline"]["frac"], " ".join(f"{k}={v['avg_us']:.1f}" for k, v in sorted(pe.items())))
PY
