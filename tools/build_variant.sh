#!/bin/bash
# Builds a variant of liblade_hip.so for A/B experiments on the GPU box: tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" [EXPERIMENTAL=1]
# -> lookaheaddecoding_amd/liblade_hip_NAME.so (select it with LADE_HIP_LIB=<path>); objects under csrc/build_NAME/.
set -e
cd "$(dirname "$0")/../lookaheaddecoding_amd/csrc"
NAME=$1; EXTRA=${2:-}; shift; shift || true
make -j8 BUILD=build_$NAME LIB=../liblade_hip_$NAME.so EXTRA="$EXTRA" "$@"
echo built ../liblade_hip_$NAME.so
