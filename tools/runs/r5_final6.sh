#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
bash tools/collect_profiles.sh gpurun_out/r05final6 > gpurun_out/r05final6_collect.log 2>&1
ls gpurun_out/r05final6; cat gpurun_out/r05final6/mel_traffic.json | head -20
