#!/bin/bash
# rocprofv3 kernel summary of the bench's timed region on the final build + the driver's torchrun line at N = 1
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
d=$OUT/trace_tmp; rm -rf "$d"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- python bench.py --gpus 1 --steps 20 --warmup 5 --timed_only > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
S=$(find "$d" -name "*kernel_stats.csv" | head -1); T=$(find "$d" -name "*kernel_trace.csv" | head -1)
[ -n "$S" ] && cp "$S" "$OUT/rocprofv3_kernel_stats.csv"
[ -n "$T" ] && python tools/trace_overlap.py "$T" > "$OUT/kernel_overlap.txt"
rm -rf "$d"
head -4 "$OUT/rocprofv3_kernel_stats.csv"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no_cpu_baseline > "$OUT/bench_torchrun_n1.json" 2> "$OUT/torchrun.err"; echo "torchrun rc=$?"
python -c "import json; d=json.load(open('$OUT/bench_torchrun_n1.json')); print(d['value'], d['n_gpus'], d['config']['collective'])"
