export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
PASST_AMD_LIB=$R/passt_amd/libpasst_amd_attn_probe.so timeout 120 python tools/probe_attn.py 2>&1 | tail -3 >> gpurun_out/r5.txt
cat gpurun_out/r5.txt
