#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -x > $OUT/t7.log 2>&1
tail -4 $OUT/t7.log
for pw in 1 0; do
  echo "== pw=$pw"; CCVPE_PW_GEMM=$pw python3 tools/pw_probe.py fp32 20 2>/dev/null
  CCVPE_PW_GEMM=$pw python3 tools/pw_probe.py bf16 20 2>/dev/null
done
for pw in 1 0; do
  CCVPE_PW_GEMM=$pw python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/f32_pw$pw.json 2> $OUT/f32_pw$pw.err
  CCVPE_PW_GEMM=$pw python3 bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bf16_pw$pw.json 2> $OUT/bf16_pw$pw.err
  python3 -c "
import json
for f in ('f32','bf16'):
    d=json.load(open('$OUT/%s_pw$pw.json'%f)); print('pw=$pw',f,d['value'],d['ms_per_step'])
"
done
python3 bench.py --train --model vigor20 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/train7.json 2> $OUT/train7.err
python3 -c "
import json
d=json.load(open('$OUT/train7.json')); print('train',d['value'],d['ms_per_step'])
"
