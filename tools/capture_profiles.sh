#!/bin/bash
# One gpurun session -> everything under profiles/ (run from the repo root ON THE GPU BOX):
#   gpurun --timeout 1500 -- 'bash tools/capture_profiles.sh r01'
# writes gpurun_out/<tag>/..., which tools/publish_profiles.py copies into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
run() { timeout "$@"; }

# 1. the bench line (N=1, defaults), with the CPU baseline leg
run 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# 2. rocprofv3's own per-kernel summary of the same command (+ the trace it is made from)
#    (--timed_only: warm-up + timed region, no extra legs, so the summary's sls_kernel row is the
#    benchmark's own 8-query gather launches and nothing else)
run 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python bench.py --timed_only > "$OUT/bench_traced.json" 2> "$OUT/trace.err"
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
S=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp "$S" "$OUT/rocprofv3_kernel_stats.csv"
[ -n "$T" ] && run 120 python tools/trace_stats.py "$T" > "$OUT/kernel_trace_by_grid.txt"
# 3. PMC passes, each in its own run (never combined with a trace domain other than kernel-trace)
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  run 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- \
      python bench.py --no_cpu_baseline --steps 1600 --warmup 160 > /dev/null 2> "$OUT/pmc_$name.err"
  C=$(find "$OUT/pmc_$name" -name "*counter_collection.csv" | head -1)
  if [ -n "$C" ]; then
    echo "== rocprofv3 --kernel-trace --pmc $pass -- python bench.py --no_cpu_baseline --steps 1600 --warmup 160" >> "$OUT/pmc_summary.txt"
    run 120 python tools/pmc_stats.py "$C" | grep -v rocclr >> "$OUT/pmc_summary.txt"
  fi
done
# 4. other operating points (one line each)
run 300 python bench.py --no_cpu_baseline --steps 8000 --warmup 800 --set shared_stream=1 > "$OUT/bench_single_stream.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --steps 4000 --warmup 400 --coalesce 1 > "$OUT/bench_coalesce1.json" 2>/dev/null
for w in rmc1_ref rmc2_ref rmc3_ref rmc1_dot wnd ncf; do
  run 300 python bench.py --workload $w --no_cpu_baseline --steps 4000 --warmup 400 > "$OUT/bench_$w.json" 2>/dev/null
done
run 400 python bench.py --workload rmc3 --batch 512 --no_cpu_baseline --steps 2000 --warmup 200 > "$OUT/bench_rmc3.json" 2>/dev/null
# 4b. reference-format characterisation tables (accelerator/predict_execution.py "***" files)
for m in rm1 rm2 rm3; do
  run 300 python tools/characterize.py --model $m --out "$OUT/accelerator_mi355x/" > "$OUT/characterize_$m.txt" 2>&1
done
# 4c. the queue harness end to end: one accel engine, RMC1 and the W&D + NCF mixed stream
run 300 python tools/serve.py --avg_arrival_rate 0.01 2>/dev/null | tail -1 > "$OUT/serve_rmc1.json"
run 300 python tools/serve.py --mix --avg_arrival_rate 0.01 2>/dev/null | tail -1 > "$OUT/serve_mix_wnd_ncf.json"
# 5. the driver's multi-GPU launch line, on the one GPU of this box
run 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 8000 --warmup 800 --no_cpu_baseline > "$OUT/bench_torchrun_n1.json" 2> "$OUT/bench_torchrun_n1.err"
# keep the merge small: the raw traces stay on the box except the one kernel trace
find "$OUT" -name "*.csv" -size +20M -delete
ls -la "$OUT"
