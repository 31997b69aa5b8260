#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_forward_gpu.py -m gpu -q -x 2>&1 | tail -2
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/f32_c.json 2> $OUT/f32_c.err
python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/bf16_c.json 2> $OUT/bf16_c.err
python3 -c "
import json
for f in ('f32','bf16'):
    d=json.load(open('$OUT/%s_c.json'%f)); print(f,d['value'],d['ms_per_step'])
"
grep mbconv_front $OUT/f32_c.err | head -12
grep mbconv_front $OUT/bf16_c.err | head -12
