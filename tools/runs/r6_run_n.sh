#!/bin/bash
# r06 run N: memory-side counters of the training step with the adopted cache policy and without it (-DPA_NO_CACHE_POLICY): requests and
# occupancy (sum of outstanding requests per cycle) at the L2's fabric interface -> mean read / write latency by Little's law, credit stalls
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$O/r06_cache_policy_pmc.txt
: > $OUT
for lib in libpasst_amd.so libpasst_amd_var_nopolicy.so; do
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
    rm -rf /tmp/pn
    PASST_AMD_LIB=$R/passt_amd/$lib timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pn -o n -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/r06_n.log 2>&1
    echo "== $lib :: $set" >> $OUT
    python $R/tools/rocpd_stats.py "$(find /tmp/pn -name '*.db' | head -1)" --top 12 >> $OUT 2>&1
  done
done
tail -5 $O/r06_n.log
