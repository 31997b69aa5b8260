#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py -m gpu -q -x -k "mbconv or front or forward or se_gate or dwconv" 2>&1 | tail -3
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/f32_b.json 2> $OUT/f32_b.err
python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/bf16_b.json 2> $OUT/bf16_b.err
python3 -c "
import json
for f in ('f32','bf16'):
    d=json.load(open('$OUT/%s_b.json'%f)); print(f,d['value'],d['ms_per_step'])
"
grep mbconv_front $OUT/f32_b.err
grep mbconv_front $OUT/bf16_b.err
