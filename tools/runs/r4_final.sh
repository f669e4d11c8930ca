#!/bin/bash
# Round-4 evidence in one GPU call: profiles (kernel stats, HBM traffic, MFMA busy), bench lines of every configuration and of
# the drop-in (autograd) path, the self-launching N = 2 dry run, the whole GPU test suite.  Output: gpurun_out/r04final/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04final
mkdir -p $O
bash tools/collect_profiles.sh gpurun_out/r04final > $O/collect.log 2>&1
python bench.py --no-cpu-baseline --path autograd 2>/dev/null | tail -1 > $O/bench_c2_autograd.json.log
python bench.py --no-cpu-baseline --path autograd --optimizer pa_adamw 2>/dev/null | tail -1 > $O/bench_c2_autograd_pa_adamw.json.log
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4.json.log
python bench.py --config c4_ref --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_ref.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json.log
python bench.py --config c5 --no-cpu-baseline --path autograd --optimizer pa_adamw 2>/dev/null | tail -1 > $O/bench_c5_autograd_pa_adamw.json.log
python bench.py --config c5 --batch 96 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_b96.json.log
python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_fp32.json.log
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_sustained_400.json.log
python tools/bench_eval.py 2>/dev/null | tail -1 > $O/bench_eval.json.log
env -u RANK -u WORLD_SIZE PASST_AMD_BENCH_DRY_GLOO=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_dry_gloo_2ranks_selflaunch.json.log
env -u RANK -u WORLD_SIZE PASST_AMD_BENCH_DRY_GLOO=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --path autograd 2>/dev/null | tail -1 > $O/bench_dry_gloo_2ranks_selflaunch_autograd.json.log
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|skipped|^FAILED|^ERROR" | tail -8 > $O/pytest_gpu.txt
cp gpurun_out/kernel_parity_metrics.json $O/ 2>/dev/null; cp gpurun_out/model_parity_metrics.json $O/ 2>/dev/null
ls $O; cat $O/pytest_gpu.txt; tail -1 $O/bench_c2.log | cut -c1-200; for f in $O/bench_*.json.log; do echo $f; cut -c1-170 $f; done
