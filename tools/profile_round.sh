tag=${1:-r03p}
mkdir -p gpurun_out; bash tools/gpu_profile.sh $tag > gpurun_out/$tag.log 2>&1
out=gpurun_out/$tag
python tools/bench_configs.py --sweep > $out/configs.jsonl 2> $out/configs.err
python tools/bench_configs.py --only sweep --overlap --reps 100 > $out/sweep_overlap.jsonl 2>> $out/configs.err
python tools/bench_configs.py --only c5 > $out/c5.jsonl 2>> $out/configs.err
python tools/bench_configs.py --only c5full >> $out/c5.jsonl 2>> $out/configs.err
python tools/bench_configs.py --only body > $out/body.jsonl 2>> $out/configs.err
bash tools/c5_timeline.sh 4096 > $out/c5_whole_timeline.txt 2>&1; bash tools/c5_timeline.sh 512 > $out/c5_share_timeline.txt 2>&1
(python tools/c5_overlap.py 4096 40 plain; python tools/c5_overlap.py 4096 40; python tools/c5_overlap.py 512 40 plain; python tools/c5_overlap.py 512 40) 2>/dev/null | grep trajectories > $out/c5_overlap.txt
python tools/bench_configs.py --only implicit --overlap > $out/implicit_overlap.jsonl 2>> $out/configs.err
python tools/host_overhead.py > $out/host_overhead.txt 2>&1
python tools/two_stream_bench.py 65536 1000 > $out/multi_stream.jsonl 2>/dev/null
tools/abi_bench robotdynamics.jl_amd/lib/librdhip.so > $out/abi_bench.txt 2>&1
python tools/ift_probe.py 1,4 300 2>/dev/null | grep ift > $out/ift_probe.jsonl
python tools/small_probe.py 2>/dev/null > $out/small_batches.jsonl; RDHIP_TUNE=small=tile python tools/small_probe.py 2>/dev/null >> $out/small_batches.jsonl
tail -5 gpurun_out/$tag.log | cut -c1-300
wc -l $out/configs.jsonl; cat $out/abi_bench.txt
# round 6 additions (each writes one file under gpurun_out/<tag>/; copy what is to be judged into profiles/<tag>/)
python tools/bench_configs.py --only split > $out/split_rollout.jsonl 2>> $out/configs.err
python tools/body_probe.py > $out/body_probe.jsonl 2>> $out/configs.err
python tools/multi_vs_single.py > $out/multi_vs_single.txt 2>&1
python tools/load_time.py > $out/load_time.txt 2>&1
(python tools/c5_one_call.py 4096 40; python tools/c5_one_call.py 4096 40 tj; python tools/c5_one_call.py 512 40; python tools/c5_one_call.py 512 40 tj) 2>/dev/null | grep trajectories > $out/c5_layouts.txt
timeout 1500 python tools/fused_stress.py 100000 > $out/fused_stress.log 2>&1
timeout 1500 python tools/mixed_stress.py 60000 > $out/mixed_stress.log 2>&1
# separately (their own rocprofv3 passes): bash tools/pmc_kernels.sh <tag>pmc; bash tools/pmc_valu.sh <tag>valu; bash tools/kernel_reach.sh <tag>/reach
