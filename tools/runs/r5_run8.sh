# Round 5, call 9: merged block-finishing launch (A/B) + LayerNorm backward geometry at small M
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -x -k "finishing or layernorm or golden or train_step" 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 10 $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; }
for rep in 1 2; do
EXTRA="" run c2_merged X=1
EXTRA="" run c2_separate PASST_AMD_NO_DEFER_ROWS=1
done
for rep in 1 2; do
EXTRA="--config c5 --steps 80" run c5_merged X=1
EXTRA="--config c5 --steps 80" run c5_separate PASST_AMD_NO_DEFER_ROWS=1
EXTRA="--config c5 --steps 80" run c5_merged_rpw2 PA_LN_BWD_ROWS_PER_WAVE=2
EXTRA="--config c5 --steps 80" run c5_merged_rpw4 PA_LN_BWD_ROWS_PER_WAVE=4
done
