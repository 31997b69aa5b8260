#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py tests/test_bf16_gpu.py tests/test_backward_gpu.py -m gpu -q -x 2>&1 | tail -3
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/f32_j.json 2> $OUT/f32_j.err
python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bf16_j.json 2> $OUT/bf16_j.err
python3 bench.py --train --model vigor20 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/train_j.json 2> $OUT/train_j.err
python3 -c "
import json
for f in ('f32','bf16','train'):
    try:
        d=json.load(open('$OUT/%s_j.json'%f)); print(f,d['value'],d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'), d['roofline']['whole_step'].get('frac'))
    except Exception as e: print(f,'failed',e)
"
