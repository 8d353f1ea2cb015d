#!/bin/bash
# One gpurun session -> everything under profiles/ (run from the repo root ON THE GPU BOX):
#   gpurun --timeout 3000 -- 'bash tools/capture_profiles.sh r04'
# writes gpurun_out/<tag>/..., which tools/publish_profiles.py copies into profiles/.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
run() { timeout "$@"; }
# per-kernel counter averages of one `rocprofv3 --pmc` pass over a command, appended to a summary
pmc() { # summary-file, counters, command...
  local sum=$1 ctr=$2; shift 2
  local d=$OUT/pmc_tmp; rm -rf "$d"
  run 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$d" -- "$@" > /dev/null 2> "$OUT/pmc_last.err"
  local C=$(find "$d" -name "*counter_collection.csv" | head -1)
  echo "== rocprofv3 --kernel-trace --pmc $ctr -- $*" >> "$sum"
  [ -n "$C" ] && run 120 python tools/pmc_stats.py "$C" | grep -v rocclr >> "$sum"
  rm -rf "$d"
}
# rocprofv3's own per-kernel summary of a command: <name>_kernel_stats.csv + by-grid text
trace() { # name, command...
  local n=$1; shift
  local d=$OUT/trace_tmp; rm -rf "$d"
  run 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -- "$@" > "$OUT/${n}_traced.json" 2> "$OUT/${n}_trace.err"
  local S=$(find "$d" -name "*kernel_stats.csv" | head -1); local T=$(find "$d" -name "*kernel_trace.csv" | head -1)
  [ -n "$S" ] && cp "$S" "$OUT/${n}_rocprofv3_kernel_stats.csv"
  [ -n "$T" ] && run 120 python tools/trace_stats.py "$T" > "$OUT/${n}_kernel_trace_by_grid.txt"
  [ -n "$T" ] && run 120 python tools/trace_overlap.py "$T" > "$OUT/${n}_kernel_overlap.txt"
  [ -n "$T" ] && run 120 python tools/trace_timeline.py "$T" 30 > "$OUT/${n}_kernel_timeline.txt"
  [ -n "$T" ] && run 120 python tools/trace_ramp.py "$T" > "$OUT/${n}_duration_by_position.txt"
  rm -rf "$d"
}

# 1. the bench line exactly as the driver runs it (N=1), with the CPU baseline legs
mkdir -p "$OUT/cpu_epyc9575f/raw_data"
run 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu_table "$OUT/cpu_epyc9575f/raw_data/results_rm1.txt" > "$OUT/bench.json" 2> "$OUT/bench.err"
# 2. rocprofv3's per-kernel summary of the same warm-up + timed region (--timed_only: none of the
#    extra legs, so every gather launch in the trace is a launch of the benchmark itself)
trace bench python bench.py --gpus 1 --steps 20 --warmup 5 --timed_only
mv "$OUT/bench_rocprofv3_kernel_stats.csv" "$OUT/rocprofv3_kernel_stats.csv" 2>/dev/null
mv "$OUT/bench_kernel_trace_by_grid.txt" "$OUT/kernel_trace_by_grid.txt" 2>/dev/null
mv "$OUT/bench_kernel_overlap.txt" "$OUT/kernel_overlap.txt" 2>/dev/null
mv "$OUT/bench_kernel_timeline.txt" "$OUT/kernel_timeline.txt" 2>/dev/null
mv "$OUT/bench_traced.json" "$OUT/bench_traced.json" 2>/dev/null
# 3. PMC passes on RMC1 (the headline workload), each in its own run
B1="python bench.py --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 2048"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  pmc "$OUT/pmc_summary.txt" "$pass" $B1
done
# 4. BASELINE config 3 (RMC3, 12 x 10M x 32, batch 512): the gather beside its GEMM launches
#    (pipelined, default) and with the chip to itself (shared_stream=1); traffic counters; and the
#    MFMA-busy counters of gemm_kernel / stream_kernel on the MLP-bound model
R3="python bench.py --workload rmc3 --batch 512 --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 2048"
run 600 $R3 > "$OUT/rmc3_bench.json" 2>/dev/null
run 600 $R3 --set shared_stream=1 > "$OUT/rmc3_bench_single_stream.json" 2>/dev/null
trace rmc3 $R3
trace rmc3_single_stream $R3 --set shared_stream=1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  pmc "$OUT/rmc3_pmc_summary.txt" "$pass" $R3 --set shared_stream=1
done
pmc "$OUT/wnd_pmc_summary.txt" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
    python bench.py --workload wnd --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 2048 --set shared_stream=1
# the wide-layer GEMM alone (tools/gemm_bench.py): durations + MFMA-busy; at 2 048 rows (every shape) and, for
# RM3 config 3's real launch, 8 192 rows (gemm32_kernel's 128 x 128 form) against gemm_kernel on the same box
trace gemm python tools/gemm_bench.py --iters 2000
pmc "$OUT/gemm_pmc_summary.txt" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python tools/gemm_bench.py --iters 20
# (sustained runs: a GPU that was idle needs some hundred launches to reach its sustained shader clock -- a
#  30-launch trace of an MFMA-bound kernel reads 10-15 % slow; tools/trace_ramp.py prints duration by position)
trace gemm_8192 python tools/gemm_bench.py --rows 8192 --iters 3000 --shapes 2560x1024,1024x256
# (gemm_kernel on the same launch: "mlp_gemm32" 0 is a lab option since round 6 -- profiles/r05_gemm_sustained/ holds the pair)
run 200 python tools/clock_trace.py --out "$OUT/gemm_8192_clock_trace.txt" --hz 20 -- python tools/gemm_bench.py --rows 8192 --iters 15000 --shapes 2560x1024 > /dev/null 2>&1
pmc "$OUT/gemm_pmc_summary.txt" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" python tools/gemm_bench.py --rows 8192 --iters 10 --shapes 2560x1024
# per-workgroup phase stamps, shader clock and bit check of the same launch (tools/ubench/gemm_lab.hip, built by `make lab`)
[ -x tools/ubench/gemm_lab ] && run 200 tools/ubench/gemm_lab 8192 2560 1024 > "$OUT/gemm_lab.txt" 2>&1
[ -x tools/ubench/gemm_lab ] && run 200 tools/ubench/gemm_lab 4096 2560 1024 >> "$OUT/gemm_lab.txt" 2>&1
# (round 6: "mlp_layout" 1 and "mlp_gemm32" 0 are lab options; their RM3 lines are profiles/r05_rmc3_bench_layout1.json / _gemm32_0.json)
# 5. other operating points and shapes (one line each)
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --set shared_stream=1 > "$OUT/bench_single_stream.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --coalesce 1 > "$OUT/bench_coalesce1.json" 2>/dev/null
# round 6: small launch sets with and without the column-split MLP launch ("mlp_nsplit" 4, the default for DLRM), the arms alternating
for rep in 1 2; do
  for c in 1 2 4; do
    for ns in 0 4; do
      run 300 python bench.py --no_cpu_baseline --timed_only --steps 4 --warmup 2 --queries_per_step 20480 --coalesce $c --set mlp_nsplit=$ns > "$OUT/bench_coalesce${c}_nsplit${ns}_$rep.json" 2>/dev/null
    done
  done
done
run 300 python bench.py --no_cpu_baseline --timed_only --steps 4 --warmup 2 --queries_per_step 20480 --coalesce 1 --slots 4 > "$OUT/bench_coalesce1_slots4.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --timed_only --steps 4 --warmup 2 --queries_per_step 20480 --coalesce 1 --slots 4 --set mlp_nsplit=0 > "$OUT/bench_coalesce1_slots4_nsplit0.json" 2>/dev/null
trace coalesce1 python bench.py --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 8192 --coalesce 1
[ -f deeprecsys_amd/libdrs_hip_tl.so ] && TL_ROWS=80 run 200 python tools/mlp_timeline.py --coalesce 1 > "$OUT/mlp_timeline_one_query_nsplit4.txt" 2>&1
[ -f deeprecsys_amd/libdrs_hip_tl.so ] && TL_ROWS=80 run 200 python tools/mlp_timeline.py --coalesce 1 --set mlp_nsplit=0 > "$OUT/mlp_timeline_one_query_plain.txt" 2>&1
# the 8-wave packed MLP launch in place of stream4_kernel (what each costs
# the gather beside it), and launch sets of 8 instead of 12 queries
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --set mlp_stream=2 > "$OUT/bench_mlp_stream2.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --set mlp_stream_2cu=1 > "$OUT/bench_mlp_stream4_2cu.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --coalesce 8 > "$OUT/bench_coalesce8.json" 2>/dev/null
# round 4: what the non-temporal row loads and the 32-row MLP form are worth on this box (same line, one option off)
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --set sls_nt=0 > "$OUT/bench_sls_nt0.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 --set mlp_rows32=0 > "$OUT/bench_mlp_rows32_0.json" 2>/dev/null
run 300 python bench.py --no_cpu_baseline --steps 5 --warmup 2 > "$OUT/bench_again.json" 2>/dev/null
# the MLP launch ALONE (one stream, 8-query sets = 128 workgroups) in its three forms: durations from
# rocprofv3, MFMA counters, and the in-kernel timeline of stream4_kernel (needs libdrs_hip_tl.so: make timeline)
MA="python bench.py --no_cpu_baseline --timed_only --steps 2 --warmup 1 --queries_per_step 2048 --coalesce 8 --set shared_stream=1"
trace mlp_alone_stream4 $MA --set mlp_stream=4 --set mlp_stream_2cu=0
trace mlp_alone_stream2 $MA --set mlp_stream=2 --set mlp_stream_2cu=0
# (2 048 rows take stream4_kernel's 32-row form by default since round 4: the two lines above both ran it; the
#  16-row forms alone, for the record)
trace mlp_alone_stream4_16rows $MA --set mlp_stream=4 --set mlp_stream_2cu=0 --set mlp_rows32=0
pmc "$OUT/mlp_alone_pmc_summary.txt" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $MA --set mlp_stream=4
pmc "$OUT/mlp_alone_pmc_summary.txt" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" $MA --set mlp_stream=4
[ -f deeprecsys_amd/libdrs_hip_tl.so ] && TL_ROWS=40 run 200 python tools/mlp_timeline.py --coalesce 8 --set mlp_stream=4 --set shared_stream=1 > "$OUT/mlp_timeline_stream4.txt" 2>&1
for w in rmc1_ref rmc2_ref rmc3_ref rmc1_dot wnd ncf mtwnd din dien; do
  run 400 python bench.py --workload $w --no_cpu_baseline --steps 5 --warmup 2 --queries_per_step 4096 > "$OUT/bench_$w.json" 2>/dev/null
  run 400 python bench.py --workload $w --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 4096 --set shared_stream=1 > "$OUT/bench_${w}_single_stream.json" 2>/dev/null
done
run 400 python bench.py --workload rmc3 --batch 512 --no_cpu_baseline --steps 5 --warmup 2 --queries_per_step 2048 > "$OUT/bench_rmc3.json" 2>/dev/null
run 600 python bench.py --workload rmc3 --batch 512 --steps 3 --warmup 1 --queries_per_step 2048 --timed_only > /dev/null 2>&1
# round 6: the one-lookup gather as a row copy (sls_one_kernel) against the lane-group-per-bag walk ("sls_one" 0), arms alternating;
# kernel traces of W&D and MT-WnD (is some kernel running all the time?  MT-WnD's output transfer has its own stream now)
for rep in 1 2; do
  for w in wnd mtwnd dien ncf; do
    for one in 0 1; do
      run 300 python bench.py --workload $w --no_cpu_baseline --steps 5 --warmup 2 --set sls_one=$one > "$OUT/bench_${w}_sls_one${one}_$rep.json" 2>/dev/null
    done
  done
done
trace wnd python bench.py --workload wnd --no_cpu_baseline --timed_only --steps 3 --warmup 1
trace mtwnd python bench.py --workload mtwnd --no_cpu_baseline --timed_only --steps 3 --warmup 1
# CPU baseline legs on the MLP-bound shapes as well (port + torch)
run 600 python bench.py --workload wnd --steps 3 --warmup 1 --queries_per_step 4096 > "$OUT/bench_wnd_cpu.json" 2>/dev/null
# 5b. DIN (fused gather + attention launch) and DIEN (recurrence on the matrix cores): per-kernel
#     summaries, HBM traffic of the fused launch, MFMA-busy of the recurrent launch, CPU legs
DN="python bench.py --workload din --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 2048"
DE="python bench.py --workload dien --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 2048"
trace din $DN
trace dien $DE
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  pmc "$OUT/din_pmc_summary.txt" "$pass" $DN
done
pmc "$OUT/dien_pmc_summary.txt" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $DE
run 400 python bench.py --workload din --steps 5 --warmup 2 --queries_per_step 4096 --cpu_seconds 6 > "$OUT/bench_din_cpu.json" 2>/dev/null
run 400 python bench.py --workload dien --steps 5 --warmup 2 --queries_per_step 4096 --cpu_seconds 6 > "$OUT/bench_dien_cpu.json" 2>/dev/null
run 300 python bench.py --workload din --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 2048 --set din_fused=0 > "$OUT/bench_din_two_launch.json" 2>/dev/null
run 300 python bench.py --workload dien --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 2048 --set dien_mfma=0 > "$OUT/bench_dien_valu.json" 2>/dev/null
# (round 4: the top MLP in a launch of its own behind the recurrence, as before; NCF with three launch sets in flight)
run 300 python bench.py --workload dien --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 4096 --set dien_fuse_top=0 > "$OUT/bench_dien_two_launch.json" 2>/dev/null
run 300 python bench.py --workload ncf --slots 3 --no_cpu_baseline --timed_only --steps 3 --warmup 1 --queries_per_step 4096 > "$OUT/bench_ncf_slots3.json" 2>/dev/null
# 6. reference-format characterisation tables (accelerator/predict_execution.py "***" files)
for m in rm1 rm2 rm3 wnd ncf mtwnd din dien; do
  run 300 python tools/characterize.py --model $m --out "$OUT/accelerator_mi355x/" > "$OUT/characterize_$m.txt" 2>&1
done
# 7. the queue harness end to end: one accel engine, RMC1 and the W&D + NCF mixed stream
run 300 python tools/serve.py --avg_arrival_rate 0.01 --nepochs 512 2>/dev/null | tail -1 > "$OUT/serve_rmc1.json"
# (at 0.01 ms the load generator's integer-millisecond Poisson gaps -- 1 % of the queries wait 1 ms -- offer ~70-100 k
#  queries/s; at 0.001 ms the engine process is what saturates)
run 300 python tools/serve.py --avg_arrival_rate 0.001 --nepochs 512 2>/dev/null | tail -1 > "$OUT/serve_rmc1_rate0001.json"
run 300 python tools/serve.py --mix --avg_arrival_rate 0.001 --nepochs 512 2>/dev/null | tail -1 > "$OUT/serve_mix_wnd_ncf_rate0001.json"
run 300 python tools/serve.py --mix --avg_arrival_rate 0.01 --nepochs 512 2>/dev/null | tail -1 > "$OUT/serve_mix_wnd_ncf.json"
# 8. the driver's multi-GPU launch line, on the one GPU of this box (RCCL communicator of size 1 is
#    not created: world == 1), and the self-spawn path
run 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no_cpu_baseline > "$OUT/bench_torchrun_n1.json" 2> "$OUT/bench_torchrun_n1.err"
# 9. locality-aware traces: the gather on uniform / shipped-profile / reuse-heavy index streams,
#    through the same line as everything else (bench.py --trace = --data_generation synthetic)
TR="python bench.py --no_cpu_baseline --steps 3 --warmup 1 --queries_per_step 8192 --num_batches 16"
run 400 $TR > "$OUT/bench_trace_uniform.json" 2>/dev/null
for p in shipped hot; do
  run 600 $TR --trace $p > "$OUT/bench_trace_$p.json" 2>/dev/null
  pmc "$OUT/traces_tcc_summary.txt" "TCC_HIT_sum TCC_MISS_sum" $TR --timed_only --steps 1 --trace $p
done
run 600 $TR --trace hot --trace_unique > "$OUT/bench_trace_hot_unique.json" 2>/dev/null
# 10. where the host time of a query goes; raw PCIe rate of the box
run 300 python tools/host_probe.py > "$OUT/host_probe.txt" 2>&1
run 120 python tools/pcie_probe.py > "$OUT/pcie_probe.txt" 2>&1
find "$OUT" -name "*.csv" -size +20M -delete
ls -la "$OUT"
