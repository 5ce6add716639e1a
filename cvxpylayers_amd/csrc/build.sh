#!/bin/bash
# Builds the C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value \
    -I../../include cone_engine.hip -o libcone_engine.so "$@"
