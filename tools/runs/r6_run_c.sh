#!/bin/bash
# r06 run C: kernel stats of the fused optimizer + staging against the separate pair (c2 and c5), the EPI 1 / 3 variant table, the
# rest of the GPU suite after the attention test fix
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c5; do
  for mode in sep fused; do
    rm -rf /tmp/kt_$cfg$mode
    env $( [ $mode = sep ] && echo PASST_AMD_NO_FUSED_STAGE=1 ) timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$cfg$mode -o k -- python $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-roofline > $O/r06_c_kt_$cfg$mode.log 2>&1
    python $R/tools/rocpd_stats.py "$(find /tmp/kt_$cfg$mode -name '*.db' | head -1)" --steps 6 --top 30 > $O/r06_c_kernel_stats_$cfg$mode.txt 2>&1
  done
done
cd $R
python tools/bench_epi13.py > $O/r06_gemm_variants_epi13.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/r06_c_tests.txt
grep -h "adamw\|stage_w" $O/r06_c_kernel_stats_*.txt
cat $O/r06_gemm_variants_epi13.txt
tail -3 $O/r06_c_tests.txt
