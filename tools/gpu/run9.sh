#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_backward_gpu.py -m gpu -q -x -k "wgrad or conv" 2>&1 | tail -3
python3 tools/wgrad_probe.py 10 2>/dev/null
python3 bench.py --train --model vigor20 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/train9.json 2> $OUT/train9.err
python3 -c "
import json
d=json.load(open('$OUT/train9.json')); print('train',d['value'],d['ms_per_step']); r=d['roofline']; print({k:(v['ms_per_step'],v['tflops']) for k,v in r['all_kernels'].items() if 'wgrad' in k})
"
