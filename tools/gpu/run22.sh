#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests/test_backward_gpu.py tests/test_train_glue_gpu.py tests/test_train_backward_gpu.py tests/test_train_trajectory_gpu.py tests/test_rccl_single_rank_gpu.py tests/test_abi.py -q -x 2>&1 | tail -3
python3 bench.py --train --model vigor20 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/train_k.json 2> $OUT/train_k.err
python3 -c "
import json
d=json.load(open('$OUT/train_k.json')); print('train',d['value'],d['ms_per_step'])
"
bash tools/gpu/prof_train.sh 2>&1 | tail -46
