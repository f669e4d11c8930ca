#!/bin/bash
R=$PWD
O=$R/gpurun_out/r04k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "graph_equals_eager or optim_adamw or two_steps" 2>&1 | tail -12 > $O/pytest_sel.txt
for i in 1 2; do
for cfg in c5 c2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config $cfg 2>$O/err_eager_$cfg.txt | tail -1 | cut -c1-150 >> $O/ab.txt
timeout 300 python bench.py --no-cpu-baseline --graph --config $cfg 2>$O/err_graph_$cfg.txt | tail -1 | cut -c1-150 >> $O/ab.txt
done; done
timeout 300 python bench.py --no-cpu-baseline --graph --config c5 --batch 4 2>/dev/null | tail -1 | cut -c1-150 >> $O/ab_b4.txt
timeout 300 python bench.py --no-cpu-baseline --no-roofline --config c5 --batch 4 2>/dev/null | tail -1 | cut -c1-150 >> $O/ab_b4.txt
cat $O/pytest_sel.txt $O/ab.txt $O/ab_b4.txt; tail -5 $O/err_graph_c5.txt
