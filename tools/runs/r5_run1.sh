# Round 5, call 1: the new parity / caller-flow tests, this box's baseline line, the speed-test flow, the drop-in path with and
# without the my_mixup replacement.  Output: gpurun_out/r05a/
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "batch64 or speed_test or graph or my_mixup or optim_adamw" 2>&1 | tail -15 > $O/pytest_new.txt
timeout 600 python -m pytest tests/test_gpu_ddp.py -q -m gpu -x -k "sweep or self_launch" 2>&1 | tail -8 >> $O/pytest_new.txt
cat $O/pytest_new.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c2.json.log
python bench.py --speedtest 2>$O/speedtest.err | grep '^{' > $O/bench_speedtest.json.log
python bench.py --speedtest --no-compile --batch 64 2>/dev/null | grep '^{' > $O/bench_speedtest_nocompile.json.log
for mx in ref pa; do
  python bench.py --path autograd --optimizer pa_adamw --mixup $mx --no-roofline --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_c2_autograd_pa_adamw_mixup_$mx.json.log
  python bench.py --config c5 --path autograd --optimizer pa_adamw --mixup $mx --no-roofline --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_c5_autograd_pa_adamw_mixup_$mx.json.log
done
python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 > $O/bench_c2_trainstep_noroofline.json.log
python bench.py --config c5 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_c5.json.log
for f in $O/bench_*.json.log; do echo $f; cut -c1-260 $f; done
cat gpurun_out/model_parity_metrics.json | python -c "import json,sys; d=json.load(sys.stdin); [print(k, v) for k,v in d.items() if 'b64' in k or 'speed' in k]"
