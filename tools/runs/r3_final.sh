# Round-3 evidence in one GPU call: profiles (kernel stats, HBM traffic, MFMA busy), bench lines of every configuration,
# sustained run, attention counters, the GPU test log.  Output: gpurun_out/r03/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03f
mkdir -p $O
bash tools/collect_profiles.sh gpurun_out/r03f > $O/collect.log 2>&1
python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_c2_sustained_400.json.log
python bench.py --config c4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4.json.log
python bench.py --config c4_ref --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c4_ref.json.log
python bench.py --config c5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5.json.log
python bench.py --config c5 --batch 96 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c5_b96.json.log
python bench.py --precision fp32 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_c2_fp32.json.log
python tools/bench_eval.py 2>/dev/null | tail -1 > $O/bench_eval.json.log
python bench_kernels.py --out $O/isolated_kernels.json > /dev/null 2>&1
python tools/bench_mel.py 2>/dev/null | tail -1 > $O/mel_isolated.json
cd /tmp; rm -rf /tmp/pa1 /tmp/pa2
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d /tmp/pa1 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pa2 -o a -- python $R/tools/bench_attn.py --shapes 64x12x474 --iters 2 > /dev/null 2>&1
cd $R
for d in pa1 pa2; do python tools/rocpd_stats.py "$(find /tmp/$d -name '*.db' | head -1)" --top 3 >> $O/attention_pmc.txt 2>&1; done
(cd tests/probes && ./probe_mfma_valu_overlap) > $O/mfma_valu_overlap.txt 2>&1
python tools/debug_attn.py 64 > $O/attention_stress.txt 2>&1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|skipped|^FAILED|^ERROR" | tail -5 > $O/pytest_gpu.txt
cp gpurun_out/kernel_parity_metrics.json $O/ 2>/dev/null; cp gpurun_out/model_parity_metrics.json $O/ 2>/dev/null
python tools/bench_vs_vendor.py > $O/vs_vendor.json 2>/dev/null
ls -la $O; cat $O/pytest_gpu.txt; tail -1 $O/bench_c2.log | cut -c1-200; for f in $O/bench_*.json.log; do echo $f; cut -c1-160 $f; done
