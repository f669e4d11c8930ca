#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05m
PASST_AMD_LIB=passt_amd/libpasst_amd_mel_probe.so timeout 200 python tools/probe_mel.py > gpurun_out/r05m/mel_probe.json 2> gpurun_out/r05m/mel_probe.err
tail -5 gpurun_out/r05m/mel_probe.err
cat gpurun_out/r05m/mel_probe.json
