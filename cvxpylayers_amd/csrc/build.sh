#!/bin/bash
# Builds the C-ABI shared library for gfx950 (cross-compiles without a GPU): csrc/Makefile, kernel families in parallel.
# Extra compiler flags: EXTRA="-DCE_TIMING" ./build.sh   (forces a full rebuild when EXTRA changes: run `make clean` first)
set -e
cd "$(dirname "$0")"
make -j"${CE_BUILD_JOBS:-8}" "$@"
