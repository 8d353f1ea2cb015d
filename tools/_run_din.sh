#!/bin/bash
# scratch: DIN first light on the GPU
mkdir -p gpurun_out/din
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "din" > gpurun_out/din/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/din/tests.log
timeout 600 python bench.py --workload din --steps 6 --warmup 2 --queries_per_step 2048 --cpu_seconds 6 > gpurun_out/din/bench.json 2> gpurun_out/din/bench.err
echo "bench rc=$?" >> gpurun_out/din/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/din/prof -o din -- python /root/repo/bench.py --workload din --steps 4 --warmup 2 --queries_per_step 2048 --timed_only > /root/repo/gpurun_out/din/prof.log 2>&1
cd /root/repo
f=$(ls gpurun_out/din/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" > gpurun_out/din/kernel_stats_head.csv
tail -5 gpurun_out/din/tests.log; cat gpurun_out/din/bench.json; tail -3 gpurun_out/din/bench.err; cat gpurun_out/din/kernel_stats_head.csv
