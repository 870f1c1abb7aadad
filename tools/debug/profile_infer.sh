#!/bin/bash
# rocprofv3 kernel trace of the inference bench (eval forward, 50 queries, B = 16), summarised over its last 10 steps
OUT=$(realpath -m "${1:-gpurun_out/prof_infer}")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --mode infer --steps 10 --warmup 2 --preheat-seconds 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
python3 "$ROOT/tools/summarize_kernel_stats.py" "$(ls "$OUT"/trace/*/*kernel_trace.csv | head -1)" 10 > "$OUT/kernel_summary.txt"
rm -rf "$OUT/trace"
head -60 "$OUT/kernel_summary.txt"
