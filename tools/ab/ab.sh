for v in ld20 ld24; do
CCVPE_LIB=tools/ab/libccvpe_hip_$v.so python bench.py --legs c2,c1bf16,train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['frac'], {k:v.get('ms_per_step') for k,v in d['config'].items() if isinstance(v,dict)})"
done
