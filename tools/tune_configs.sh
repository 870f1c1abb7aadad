#!/bin/bash
# MIOpen find + PyTorch TunableOp search for the OTHER BASELINE configurations (config 4: ResNet-101 at 1408x376; config 5:
# 1920x1280, B = 4, mixed cameras), appended to copies of the shipped databases.  Run on the GPU box from the repo root:
#     bash tools/tune_configs.sh gpurun_out/tune45
# then copy <out>/db/*.txt and <out>/db/tunableop_gfx950.csv over monosowa_amd/miopen_db/.
OUT=$(realpath -m "${1:-gpurun_out/tune45}")
ROOT=$(pwd)
mkdir -p "$OUT/db"
cp "$ROOT"/monosowa_amd/miopen_db/*.txt "$OUT/db/"
for ord in 0; do cp "$ROOT/monosowa_amd/miopen_db/tunableop_gfx950.csv" "$OUT/db/tunableop$ord.csv"; done
export MIOPEN_FIND_MODE=1 MIOPEN_USER_DB_PATH="$OUT/db" MIOPEN_CUSTOM_CACHE_DIR="$OUT/db"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$OUT/db/tunableop.csv"
C4="--backbone resnet101 --resolution 1408x376"
C5="--resolution 1920x1280 --batch 4 --mixed-cameras"
echo "config 4 search" >> "$OUT/progress.log"
timeout -k 10 500 python bench.py --miopen-find $C4 --steps 3 --warmup 1 --no-cpu-baseline --no-inference-leg > "$OUT/find_c4.json" 2> "$OUT/find_c4.err"
echo "config 5 search" >> "$OUT/progress.log"
timeout -k 10 500 python bench.py --miopen-find $C5 --steps 3 --warmup 1 --no-cpu-baseline --no-inference-leg > "$OUT/find_c5.json" 2> "$OUT/find_c5.err"
cp "$OUT/db/tunableop0.csv" "$OUT/db/tunableop_gfx950.csv"
ls -la "$OUT/db" >> "$OUT/progress.log"
