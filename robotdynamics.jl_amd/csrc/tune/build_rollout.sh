#!/bin/bash
# usage: tune/build_rollout.sh <tag> [extra hipcc flags...]   builds tune/tune_rollout_<tag> (developer tool; binaries are not part of the library)
cd "$(dirname "$0")"
tag=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize "$@" tune_rollout.hip -o tune_rollout_$tag 2>&1 | grep -E "error" -A3 | head -8
