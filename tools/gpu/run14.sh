#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1200 python3 -m pytest tests/test_backward_gpu.py tests/test_train_forward_gpu.py tests/test_train_backward_gpu.py tests/test_train_glue_gpu.py tests/test_train_trajectory_gpu.py tests/test_abi.py -q -x 2>&1 | tail -5
python3 bench.py --train --model vigor20 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/train_d.json 2> $OUT/train_d.err
CCVPE_PACK_GRAPH=0 python3 bench.py --train --model vigor20 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/train_d0.json 2> $OUT/train_d0.err
python3 -c "
import json
for f in ('train_d','train_d0'):
    try:
        d=json.load(open('$OUT/%s.json'%f)); print(f,d['value'],d['ms_per_step'])
    except Exception as e: print(f,'failed',e)
"
tail -3 $OUT/train_d.err
