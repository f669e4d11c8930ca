# the N > 1 code path of bench.py on a one-GPU box: two ranks on device 0 over gloo (DRY RUN, labelled so in the line)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PASST_AMD_BENCH_DRY_GLOO=1
mkdir -p gpurun_out/r30
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 2>gpurun_out/r30/err.txt | tail -1 > gpurun_out/r30/bench_dry_gloo_2ranks.json.log
cut -c1-400 gpurun_out/r30/bench_dry_gloo_2ranks.json.log; tail -3 gpurun_out/r30/err.txt
