#!/bin/bash
# after the last code changes of round 4: the whole GPU suite again + the bench lines that changed meaning (drop-in path without
# per-launch events, next to TrainStep in the same call; graph mode)
R=$PWD
O=$R/gpurun_out/r04final2
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|skipped|^FAILED|^ERROR" | tail -8 > $O/pytest_gpu.txt
cp gpurun_out/kernel_parity_metrics.json $O/ 2>/dev/null; cp gpurun_out/model_parity_metrics.json $O/ 2>/dev/null
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_noroofline.json.log
python bench.py --no-cpu-baseline --no-roofline --path autograd 2>/dev/null | tail -1 > $O/bench_c2_autograd.json.log
python bench.py --no-cpu-baseline --no-roofline --path autograd --optimizer pa_adamw 2>/dev/null | tail -1 > $O/bench_c2_autograd_pa_adamw.json.log
python bench.py --no-cpu-baseline --no-roofline --config c5 2>/dev/null | tail -1 > $O/bench_c5_noroofline.json.log
python bench.py --no-cpu-baseline --no-roofline --config c5 --path autograd --optimizer pa_adamw 2>/dev/null | tail -1 > $O/bench_c5_autograd_pa_adamw.json.log
python bench.py --no-cpu-baseline --graph --config c5 2>/dev/null | tail -1 > $O/bench_c5_graph.json.log
python bench.py --no-cpu-baseline --graph --config c5 --batch 4 2>/dev/null | tail -1 > $O/bench_c5_b4_graph.json.log
python bench.py --no-cpu-baseline --no-roofline --config c5 --batch 4 2>/dev/null | tail -1 > $O/bench_c5_b4_eager.json.log
python __graft_entry__.py smoke 2>&1 | tail -3 > $O/smoke.txt
cat $O/pytest_gpu.txt $O/smoke.txt; for f in $O/bench_*.json.log; do echo $(basename $f) $(cut -c60-150 $f); done
