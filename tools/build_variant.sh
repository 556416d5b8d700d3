#!/bin/bash
# Builds a variant of liblade_hip.so for A/B experiments on the GPU box: tools/build_variant.sh NAME "EXTRA HIPCC FLAGS"
# -> lookaheaddecoding_amd/liblade_hip_NAME.so (select it with LADE_HIP_LIB=<path>); objects under csrc/build_NAME/.
set -e
cd "$(dirname "$0")/../lookaheaddecoding_amd/csrc"
NAME=$1; shift
EXTRA="$*"
mkdir -p build_$NAME
OBJS=""
for f in attn.hip kv.hip intops.hip glue.hip sampling.hip gemm.hip gemm_bf16.hip gemm_f16.hip cabi.cpp lpcomm.cpp; do
    o=build_$NAME/${f%.*}.o
    x=""; [ "${f##*.}" = "cpp" ] && x="-x hip"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable $EXTRA $x -c $f -o $o &
    OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblade_hip_$NAME.so $OBJS -ldl
echo built ../liblade_hip_$NAME.so
