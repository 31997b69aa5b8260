#!/bin/bash
# A/B of two builds of the library in ONE GPU call (same box, same clocks): the default bench line's headline + the bf16 and
# training legs, once per library.  Build each variant in-tree, copy ccvpe_amd/libccvpe_hip.so to tools/ab/libccvpe_hip_<tag>.so
# (git-ignored, but it travels with gpurun), then:   gpurun -- 'bash tools/ab/ab.sh <tagA> <tagB>'
# (round 3: ld20 / ld24 = LDS row pitch 20 vs 24 floats in the GEMM kernels, DESIGN section 4 "LDS bank conflicts").
for v in "$@"; do
CCVPE_LIB=tools/ab/libccvpe_hip_$v.so python bench.py --legs c2,c1bf16,train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d['config'].items() if k.endswith('_ms')})"
done
