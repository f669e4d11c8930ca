#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
for i in 1 2; do
  echo "one-tile       $(PA_MEL_PERSIST=0 python tools/bench_mel.py 2>/dev/null | tail -1 | cut -c1-110)"
  echo "persist8       $(python tools/bench_mel.py 2>/dev/null | tail -1 | cut -c1-110)"
  for v in nostore nodma nostore_nodma; do
    echo "persist8 $v  $(PASST_AMD_LIB=$R/passt_amd/libpasst_amd_mel_$v.so python tools/bench_mel.py 2>/dev/null | tail -1 | cut -c1-160)"
    echo "one-tile $v  $(PA_MEL_PERSIST=0 PASST_AMD_LIB=$R/passt_amd/libpasst_amd_mel_$v.so python tools/bench_mel.py 2>/dev/null | tail -1 | cut -c1-160)"
  done
done
