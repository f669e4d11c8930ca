#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05m
for rep in 1 2; do
timeout 120 python tools/bench_mel.py 2>/dev/null | tail -1
PASST_AMD_LIB=passt_amd/libpasst_amd_mel_f8.so timeout 120 python tools/bench_mel.py 2>/dev/null | tail -1
done > gpurun_out/r05m/mel_f8_ab.jsonl
cat gpurun_out/r05m/mel_f8_ab.jsonl
PASST_AMD_LIB=passt_amd/libpasst_amd_mel_f8.so timeout 300 python -m pytest tests/test_gpu_model.py -x -q -k frontend 2>&1 | tail -2
