#!/bin/bash
# Per-kernel times with NO two kernels sharing the chip: every stream switch off, then rocprofv3 kernel stats of the fp32
# forward, the bf16 forward and the training step.  usage (through gpurun): bash tools/gpu/solo_profiles.sh <tag> [f32,bf16_c1,train]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/solo_${1:-r03}
mkdir -p $OUT
export TMPDIR=/tmp
export CCVPE_EVAL_TWO_STREAMS=0 CCVPE_OVERLAP_DECODERS=0 CCVPE_TRAIN_TWO_STREAMS=0 CCVPE_TRAIN_DEFER_WGRAD=0 CCVPE_TRAIN_DECODER_STREAMS=0
cd /tmp
F32="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events"
BF1="python3 $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events"
TRN="python3 $R/bench.py --train --model vigor20 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events"
WHAT=${2:-f32,bf16_c1,train}
case ,$WHAT, in *,f32,*) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f32 -o f32 -- $F32 > $OUT/f32.json 2> $OUT/f32.err;; esac
case ,$WHAT, in *,bf16_c1,*) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16_c1 -o bf16_c1 -- $BF1 > $OUT/bf16_c1.json 2> $OUT/bf16_c1.err;; esac
case ,$WHAT, in *,train,*) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- $TRN > $OUT/train.json 2> $OUT/train.err;; esac
find $OUT -name '*kernel_trace.csv' -size +8M -delete
find $OUT -name '*kernel_trace.csv' -delete
ls $OUT
