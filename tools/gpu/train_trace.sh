#!/bin/bash
# Single-stream kernel table of the training step (the three-stream step's per-kernel durations are inflated by the kernels
# that run beside them): bash tools/gpu/train_trace.sh [tag]   ->  gpurun_out/train_trace_<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${1:-single}
OUT=$R/gpurun_out/train_trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export CCVPE_TRAIN_TWO_STREAMS=0 CCVPE_TRAIN_DEFER_WGRAD=0 CCVPE_TRAIN_DECODER_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $R/bench.py --train --model vigor20 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/train.json 2> $OUT/train.err
tail -1 $OUT/train.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("single-stream step:", d["ms_per_step"], "ms")'
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
head -45 $OUT/kernel_stats.csv | cut -c1-160
