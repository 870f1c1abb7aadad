#!/bin/bash
# rocprofv3 kernel trace of the default bench workload, summarised over its last 5 steps.  Run on the GPU box from the repo root:
#     bash tools/profile_bench.sh gpurun_out/prof_bench
set -e
OUT=$(realpath -m "${1:-gpurun_out/prof_bench}")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --preheat-seconds 1 --no-cpu-baseline --no-inference-leg --no-dataloader-leg --no-step-roofline --no-offsets-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
python3 "$ROOT/tools/summarize_kernel_stats.py" "$(ls "$OUT"/trace/*/*kernel_trace.csv | head -1)" 5 > "$OUT/kernel_summary.txt"
python3 "$ROOT/tools/summarize_kernel_stats.py" "$(ls "$OUT"/trace/*/*kernel_trace.csv | head -1)" 5 120 > "$OUT/kernel_top120.txt"
cp "$(ls "$OUT"/trace/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
