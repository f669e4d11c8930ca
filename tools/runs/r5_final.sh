#!/bin/bash
# Round-5 evidence in one GPU call: profiles (kernel stats, HBM traffic, MFMA busy), bench lines of every configuration, of the
# drop-in (autograd) path with its two one-word replacements, of the reference's own speed-test flow, the self-launching sweep
# (dry: two ranks on one device over gloo), the whole GPU test suite.  Output: gpurun_out/r05final/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r05final}
mkdir -p $O
bash tools/collect_profiles.sh gpurun_out/${1:-r05final} > $O/collect.log 2>&1
python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_c2_trainstep_noroofline.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 --path autograd 2>/dev/null | tail -1 > $O/bench_c2_autograd.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 --path autograd --optimizer pa_adamw 2>/dev/null | tail -1 > $O/bench_c2_autograd_pa_adamw.json.log
python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 --path autograd --optimizer pa_adamw --mixup pa 2>/dev/null | tail -1 > $O/bench_c2_autograd_pa_adamw_pa_mixup.json.log
PASST_AMD_ATTN_BWD=two_pass python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2_attn_two_pass.json.log
python bench.py --speedtest 2>/dev/null | grep '^{' > $O/bench_speedtest.json.log
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4.json.log
python bench.py --config c4_ref --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_ref.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json.log
python bench.py --config c5 --no-cpu-baseline --no-roofline --steps 80 --warmup 10 --path autograd --optimizer pa_adamw --mixup pa 2>/dev/null | tail -1 > $O/bench_c5_autograd_pa_adamw_pa_mixup.json.log
python bench.py --config c5 --batch 96 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_b96.json.log
python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_fp32.json.log
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_sustained_400.json.log
python tools/bench_eval.py 2>/dev/null | tail -1 > $O/bench_eval.json.log
env -u RANK -u WORLD_SIZE PASST_AMD_BENCH_DRY_GLOO=1 python bench.py --sweep-gpus 1,2 --steps 5 --warmup 2 --no-cpu-baseline --batch 8 2>/dev/null | grep '^{' > $O/bench_dry_gloo_sweep_1_2.json.log
cd /tmp; rm -rf /tmp/kt5; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt5 -o k -- python $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kt5 -name '*.db' | head -1)" --steps 6 --top 30 > "$O/c5_kernel_stats.txt" 2>&1
cd $R
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|skipped|^FAILED|^ERROR" | tail -8 > $O/pytest_gpu.txt
cp gpurun_out/kernel_parity_metrics.json $O/ 2>/dev/null; cp gpurun_out/model_parity_metrics.json $O/ 2>/dev/null
ls $O; cat $O/pytest_gpu.txt; tail -1 $O/bench_c2.log | cut -c1-200; for f in $O/bench_*.json.log; do echo $f; cut -c1-170 $f; done
