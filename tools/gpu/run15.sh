#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py tests/test_backward_gpu.py -m gpu -q -x 2>&1 | tail -2
for v in a b; do
  if [ $v = b ]; then export CCVPE_LIB=$R/gpurun_ab/libccvpe_noslp.so; fi
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/f32_$v.json 2> $OUT/f32_$v.err
  python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bf16_$v.json 2> $OUT/bf16_$v.err
  python3 bench.py --train --model vigor20 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/train_$v.json 2> $OUT/train_$v.err
done
python3 -c "
import json
for v in 'ab':
  for f in ('f32','bf16','train'):
    try:
        d=json.load(open('$OUT/%s_%s.json'%(f,v))); print(v,f,d['value'],d['ms_per_step'])
    except Exception as e: print(v,f,'failed',e)
"
