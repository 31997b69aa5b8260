#!/bin/bash
# rocprofv3 kernel stats (CSV) of the three measured workloads -> gpurun_out/prof_r02/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f32 -o f32 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events > $OUT/f32.json 2> $OUT/f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16 -o bf16 -- python3 $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events > $OUT/bf16.json 2> $OUT/bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- python3 $R/bench.py --train --model vigor20 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/train.json 2> $OUT/train.err
find $OUT -name '*kernel_trace.csv' -size +30M -delete
ls -la $OUT/*
