#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
export CCVPE_CONV3_WREG=1
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py tests/test_backward_gpu.py -m gpu -q -x 2>&1 | tail -3
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/f32_w1.json 2> $OUT/f32_w1.err
python3 bench.py --train --model vigor20 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/train_w1.json 2> $OUT/train_w1.err
python3 -c "
import json
for f in ('f32','train'):
    try:
        d=json.load(open('$OUT/%s_w1.json'%f)); print(f,d['value'],d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('frac'))
        ak=d['roofline']['all_kernels']
        for k in sorted(ak):
            if 'conv3x3' in k: print('   ',k, ak[k]['ms_per_step'], ak[k]['tflops'])
    except Exception as e: print(f,'failed',e)
"
