#!/bin/bash
# r06 final evidence on the committed sources: profiles (kernel stats, PMC traffic, MFMA busy), every bench line, the GPU suite, a dry run
# of the first-node runbook.  Copy what should be judged from gpurun_out/r06/ into profiles/r06_*.
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
bash tools/collect_profiles.sh gpurun_out/r06 > $O/collect.log 2>&1
tail -1 $O/bench_c2.log | cut -c1-300
B="python bench.py --no-cpu-baseline"
$B --steps 400 --no-roofline > $O/bench_c2_sustained_400.json.log 2>&1
$B --config c4 > $O/bench_c4.json.log 2>&1
$B --config c5 > $O/bench_c5.json.log 2>&1
$B --config c5 --steps 400 --no-roofline > $O/bench_c5_sustained_400.json.log 2>&1
$B --config c5 --batch 96 > $O/bench_c5_b96.json.log 2>&1
$B --precision fp32 > $O/bench_c2_fp32.json.log 2>&1
$B --graph --no-roofline > $O/bench_c2_graph.json.log 2>&1
$B --path autograd --no-roofline > $O/bench_c2_autograd.json.log 2>&1
$B --path autograd --optimizer pa_adamw --mixup pa --no-roofline > $O/bench_c2_autograd_pa_adamw_pa_mixup.json.log 2>&1
$B --config c5 --path autograd --no-roofline --steps 100 > $O/bench_c5_autograd.json.log 2>&1
$B --config c5 --path autograd --optimizer pa_adamw --mixup pa --no-roofline --steps 100 > $O/bench_c5_autograd_pa_adamw_pa_mixup.json.log 2>&1
$B --config c5 --no-roofline --steps 100 > $O/bench_c5_trainstep_noroofline.json.log 2>&1
python bench.py --speedtest > $O/bench_speedtest.json.log 2>&1
python tools/bench_eval.py > $O/bench_eval.json.log 2>&1
python tools/bench_mel.py > $O/mel_isolated.json 2>&1
python bench_kernels.py --iters 30 --out $O/isolated_kernels.json > $O/isolated_kernels.log 2>&1
# kernel stats of ESC-50
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt5; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt5 -o k -- python $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-roofline > $O/kt5.log 2>&1
python $R/tools/rocpd_stats.py "$(find /tmp/kt5 -name '*.db' | head -1)" --steps 6 --top 30 > $O/c5_kernel_stats.txt 2>&1
cd $R
python -m pytest tests -m gpu -q -rf 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
cp gpurun_out/model_parity_metrics.json $O/model_parity_metrics.json; cp gpurun_out/kernel_parity_metrics.json $O/kernel_parity_metrics.json
STEPS=10 bash tools/run_on_node.sh $O/node_dry_run > $O/node_dry_run.log 2>&1; tail -15 $O/node_dry_run.log
for f in $O/bench_*.json.log; do echo "$(basename $f): $(tail -1 $f | cut -c1-140)"; done
