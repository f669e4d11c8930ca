#!/bin/bash
# A/B build of the whole library with per-translation-unit defines:
#   tools/build_variant.sh NAME "GEMM_DEFS" ["ATTN_DEFS" ["LN_DEFS" ["OPTIM_DEFS"]]]   ->  passt_amd/libpasst_amd_var_NAME.so
# (select it with PASST_AMD_LIB=...; the default objects under csrc/build must exist: run make first)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/passt_amd/csrc
N=$1; GD=$2; AD=$3; LD=$4; OD=$5
B=build/var_$N
mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result"
objs=""
for o in api patch head_loss comm mel; do objs="$objs build/$o.o"; done
/opt/rocm/bin/hipcc $F -fno-slp-vectorize $GD -c gemm.hip -o $B/gemm.o &
if [ -n "$AD" ]; then /opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize $AD -c attention.hip -o $B/attention.o & a=$B/attention.o; else a=build/attention.o; fi
if [ -n "$LD" ]; then /opt/rocm/bin/hipcc $F $LD -c layernorm.hip -o $B/layernorm.o & l=$B/layernorm.o; else l=build/layernorm.o; fi
if [ -n "$OD" ]; then /opt/rocm/bin/hipcc $F $OD -c optim.hip -o $B/optim.o & op=$B/optim.o; else op=build/optim.o; fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpasst_amd_var_$N.so $objs $B/gemm.o $a $l $op -ldl
echo built libpasst_amd_var_$N.so
