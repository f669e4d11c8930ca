# Round 5, call 12: front end on 2-vectors (packed f32) + 8 frames per workgroup for small grids: parity + timing (A/B vs the old build)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05k
mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -x -k "frontend or mel or wave" 2>&1 | tail -4 > $O/pytest_mel.txt; cat $O/pytest_mel.txt
for rep in 1 2; do
python tools/bench_mel.py 2>/dev/null | tail -1 | tee -a $O/mel_isolated.json
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_mel_old.so python tools/bench_mel.py 2>/dev/null | tail -1 | tee -a $O/mel_isolated.json
done
