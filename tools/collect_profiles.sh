#!/bin/bash
# The measurements behind bench.py's roofline object and DESIGN.md's tables, in one go (run on the GPU box):
#   bash tools/collect_profiles.sh gpurun_out/rNN
# Writes into the given directory:
#   bench_c2.log        the default bench line (with cpu_baseline)
#   kernel_stats.txt    per-kernel time of 6 steps (5 timed + 1 warm-up)  (rocprofv3 --kernel-trace, summarised by rocpd_stats.py)
#   pmc_fetch.txt / pmc_write.txt / gemm_traffic.json / mel_traffic.json
#                       HBM bytes per launch: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md:
#                       FETCH_SIZE counts 32-byte units twice on gfx950; tools/gemm_traffic.py applies the corrections)
#   pmc_mfma.txt        SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel
# Copy what should be judged into profiles/ (tracked) afterwards.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/${1:-gpurun_out/prof}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"

timeout 400 $BENCH > "$O/bench_c2.log" 2>&1

rm -rf /tmp/kt /tmp/pf /tmp/pw /tmp/pm
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o k -- $BENCH --steps 5 --warmup 1 --no-cpu-baseline > "$O/kt.log" 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kt -name '*.db' | head -1)" --steps 6 --top 40 > "$O/kernel_stats.txt" 2>&1

timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline > "$O/pf.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline > "$O/pw.log" 2>&1
F=$(find /tmp/pf -name '*.db' | head -1); W=$(find /tmp/pw -name '*.db' | head -1)
python $R/tools/gemm_traffic.py "$F" "$W" "$O/mel_traffic.json" > "$O/gemm_traffic.json" 2> "$O/traffic.err"
python $R/tools/rocpd_stats.py "$F" --top 25 > "$O/pmc_fetch.txt" 2>&1
python $R/tools/rocpd_stats.py "$W" --top 25 > "$O/pmc_write.txt" 2>&1

timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace -d /tmp/pm -o m -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > "$O/pm.log" 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/pm -name '*.db' | head -1)" --top 16 > "$O/pmc_mfma.txt" 2>&1

cd "$R"
tail -1 "$O/bench_c2.log" | cut -c1-400
head -34 "$O/kernel_stats.txt" | cut -c1-160
head -14 "$O/gemm_traffic.json"; cat "$O/mel_traffic.json"; tail -3 "$O/traffic.err"
