#!/bin/bash
# tools/gpu_profile.sh <tag> -- run on the GPU box (via gpurun): GPU parity tests, bench line, rocprofv3 kernel-trace stats and the
# HBM PMC passes (each counter in its own run; never combined with trace domains).  Everything lands in gpurun_out/<tag>/.
tag=${1:-r01}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
tail -3 $out/pytest_gpu.log
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json
# kernel traces of the bench command: (a) as the driver runs it (launch overlap on: consecutive launches of the flagship kernel overlap in time, so a launch's
# DURATION is longer than the job's time per launch), (b) --no-overlap (one stream: the per-kernel average duration is the time per launch)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python $root/bench.py --no-cpu-baseline --streams 1 > $out/trace.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_single -o $tag -- python $root/bench.py --no-cpu-baseline --streams 1 --no-overlap --no-extras > $out/trace_single.log 2>&1)
# HBM byte counters (their own passes; counter collection serialises the kernels, so the single-stream form of the command is the honest one)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o $tag -- python $root/bench.py --no-cpu-baseline --streams 1 --no-overlap --no-extras > $out/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o $tag -- python $root/bench.py --no-cpu-baseline --streams 1 --no-overlap --no-extras > $out/pmc_write.log 2>&1)
# calibration of the two counters on known byte counts (1 GiB streaming fill / read, 16 B per lane)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -o cal -- $root/tools/fp64_ubench calib > $out/cal_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -o cal -- $root/tools/fp64_ubench calib > $out/cal_write.log 2>&1)
find $out -name "*.db" -delete
find $out -type f | head -50
