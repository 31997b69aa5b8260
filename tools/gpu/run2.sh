#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_train_glue_gpu.py tests/test_drop_connect_gpu.py tests/test_rccl_single_rank_gpu.py tests/test_train_trajectory_gpu.py tests/test_fullsize_gpu.py tests/test_bf16_gpu.py tests/test_train_backward_gpu.py -m gpu -q -s > $OUT/t_new.log 2>&1
tail -40 $OUT/t_new.log
timeout 600 python3 tools/bf16_argmax_rate.py 64 vigor20 > $OUT/argmax_rate.log 2>&1
cat $OUT/argmax_rate.log
timeout 900 python3 bench.py --steps 10 --warmup 3 > $OUT/bench_r02a.json 2> $OUT/bench_r02a.err
tail -c 1500 $OUT/bench_r02a.err
python3 -c "
import json; d=json.load(open('$OUT/bench_r02a.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])
for k,v in d['extra'].items(): print(k, {a:b for a,b in v.items() if a not in ('roofline','workload','parity')}, (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('whole_step'))
"
