export TMPDIR=/tmp
mkdir -p gpurun_out/r24
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ddp.py -q -m gpu -k "one_item or split_k or ddp or bench or rccl or ranks" 2>&1 | tail -4
bash tools/collect_profiles.sh gpurun_out/r24 > gpurun_out/r24/collect.log 2>&1
tail -1 gpurun_out/r24/bench_c2.log | cut -c1-300
