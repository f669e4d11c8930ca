#!/bin/bash
# First-node runbook: ONE command for the first lease of a multi-GPU MI355X node.
#
#     bash tools/run_on_node.sh [out_dir]          (default out_dir: profiles/node_<date>)
#
# Produces, under out_dir (copy / commit it as it is):
#   scale_<transport>_<persist>_<wire>.jsonl   bench.py --sweep-gpus 1,2,4,8: one JSON line per N (the 1 / 2 / 4 / 8 curve north_star asks
#                                              for: clips/s, allreduce_bus_GBps_idle, allreduce_exposed_wait_ms_per_step, rccl_nranks) for
#                                              {torch.distributed nccl (= RCCL), the C ABI's own RCCL transport} x
#                                              {GEMMs one work item per workgroup next to the all-reduce (default for N > 1),
#                                               persistent GEMMs (PASST_AMD_DDP_PERSISTENT=1)} x {fp32, bf16 gradient wire}
#   scale_c5_*.jsonl                           the same sweep at N = 1,2,4 for BASELINE config #5 (ESC-50, 4-GPU DDP)
#   ddp_tests.txt                              pytest -m gpu tests/test_gpu_ddp.py: the 8 RCCL N > 1 parity cases that skip on a 1-GPU box
#   topology.txt                               rocm-smi --showtopo, device count
# Reference flow this stands in for: `DDP=N python ex_audioset.py` (ex_audioset.py:499-524: one process per GPU, NCCL all-reduce).
# Every sweep is bounded by `timeout`; a failing N prints an error line and the sweep goes on (bench.py --sweep-gpus).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-$R/profiles/node_$(date +%Y%m%d_%H%M)}
STEPS=${STEPS:-100}          # timed steps per bench line (STEPS=20 for a quick dry run)
mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
{ echo "devices: $NG"; rocm-smi --showtopo 2>&1 | head -80; } > "$O/topology.txt"
GPUS=$(python - "$NG" <<'EOF'
import sys
n = int(sys.argv[1])
print(",".join(str(g) for g in (1, 2, 4, 8) if g <= n))
EOF
)
echo "sweeping N = $GPUS on $NG devices -> $O"

python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1 || { echo "build failed, see $O/build.log"; exit 1; }

for transport in torch rccl_abi; do
  for persist in item persistent; do
    for wire in fp32 bf16; do
      tag=${transport}_${persist}_${wire}
      env_persist=""
      [ "$persist" = persistent ] && env_persist="PASST_AMD_DDP_PERSISTENT=1"
      echo "== $tag"
      env $env_persist timeout 1500 python bench.py --sweep-gpus "$GPUS" --steps "$STEPS" --warmup 10 --no-cpu-baseline \
          --transport $transport --comm-dtype $wire > "$O/scale_$tag.jsonl" 2> "$O/scale_$tag.err"
      grep -c '"metric"' "$O/scale_$tag.jsonl" | sed "s/^/   lines: /"
    done
  done
done

# BASELINE config #5: ESC-50 fine-tune on 4 GPUs (per-GPU batch 12), front-end GB/s in every line
C5=$(python - "$NG" <<'EOF'
import sys
n = int(sys.argv[1])
print(",".join(str(g) for g in (1, 2, 4) if g <= n))
EOF
)
for transport in torch rccl_abi; do
  timeout 900 python bench.py --config c5 --sweep-gpus "$C5" --steps $((2 * STEPS)) --warmup 10 --no-cpu-baseline --transport $transport \
      > "$O/scale_c5_$transport.jsonl" 2> "$O/scale_c5_$transport.err"
done

# the multi-rank parity cases (2 / 4 / 8 ranks over nccl and over the C ABI's communicator == one process on the concatenated batch)
timeout 1800 python -m pytest -m gpu tests/test_gpu_ddp.py -q -rs > "$O/ddp_tests.txt" 2>&1
tail -3 "$O/ddp_tests.txt"

python - "$O" <<'EOF'
import glob, json, os, sys
o = sys.argv[1]
print("\nfile, N, clips/s, ms/step, efficiency vs N=1 of the same file, idle bus GB/s, exposed all-reduce ms")
for f in sorted(glob.glob(os.path.join(o, "scale_*.jsonl"))):
    base = None
    for line in open(f):
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if "value" not in d:
            print(os.path.basename(f), d.get("n_gpus"), "ERROR", str(d.get("error"))[:100])
            continue
        if d["n_gpus"] == 1:
            base = d["value"]
        eff = d["value"] / (base * d["n_gpus"]) if base else float("nan")
        print(os.path.basename(f), d["n_gpus"], round(d["value"], 1), d["ms_per_step"], round(eff, 3),
              d.get("allreduce_bus_GBps_idle"), d.get("allreduce_exposed_wait_ms_per_step"))
EOF
