#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_forward_gpu.py tests/test_backward_gpu.py -m gpu -q -x > $OUT/t3.log 2>&1
tail -15 $OUT/t3.log
for pw in 1 0; do
  CCVPE_PW_GEMM=$pw python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/f32_pw$pw.json 2> $OUT/f32_pw$pw.err
  CCVPE_PW_GEMM=$pw python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --per-layer > $OUT/bf16_pw$pw.json 2> $OUT/bf16_pw$pw.err
  python3 -c "
import json
for f in ('f32','bf16'):
    d=json.load(open('$OUT/%s_pw$pw.json'%f)); print('pw=$pw',f,d['value'],d['ms_per_step'])
"
done
CCVPE_PW_GEMM=1 python3 bench.py --train --model vigor20 --steps 3 --warmup 1 --no-cpu-baseline --per-layer > $OUT/train_pw1.json 2> $OUT/train_pw1.err
python3 -c "
import json
d=json.load(open('$OUT/train_pw1.json')); print('train pw=1',d['value'],d['ms_per_step'])
"
grep -E "pw_gemm|igemm" $OUT/f32_pw1.err | head -40
