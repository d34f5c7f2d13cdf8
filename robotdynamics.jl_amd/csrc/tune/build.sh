#!/bin/bash
# usage: tune/build.sh '<variant list>'   e.g.  '{"ng4_kw1", launch_unrolled<4,1>}, {"ng8_kw1", launch_unrolled<8,1>}'
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -ffinite-math-only -DTUNE_VARIANTS="$1" $2 -o ${3:-tune_jac} -Rpass-analysis=kernel-resource-usage tune_jac.hip 2> remarks.txt || { grep -E "error" -A3 remarks.txt | head -40; exit 1; }
python kstat.py 'jac_' < remarks.txt
