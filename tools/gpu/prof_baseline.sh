#!/bin/bash
# Round-2 starting-point profiles (run through gpurun from the repo root):
#   rocprofv3 --kernel-trace --stats of the VIGOR training step (B=64), the bf16 C1-model forward (B=64) and bf16 C2 (B=32)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_train_vigor -o tv -- python3 $R/bench.py --train --model vigor20 --steps 3 --warmup 1 > $OUT/train_vigor.json 2> $OUT/train_vigor.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_bf16_c1 -o c1 -- python3 $R/bench.py --precision bf16 --model prior0 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events > $OUT/bf16_c1.json 2> $OUT/bf16_c1.err
rocprofv3 --kernel-trace --stats -d $OUT/prof_bf16_c2 -o c2 -- python3 $R/bench.py --precision bf16 --model vigor20 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events > $OUT/bf16_c2.json 2> $OUT/bf16_c2.err
# drop the per-launch traces (large); keep the stats
find $OUT/prof_train_vigor $OUT/prof_bf16_c1 $OUT/prof_bf16_c2 -name '*kernel_trace.csv' -size +20M -delete
python3 $R/bench.py --precision bf16 --model prior0 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/bf16_c1_layers.json 2> $OUT/bf16_c1_layers.err
python3 $R/tools/train_layer_times.py vigor 64 > $OUT/train_layers.log 2>&1
ls -la $OUT/prof_train_vigor $OUT/prof_bf16_c1
