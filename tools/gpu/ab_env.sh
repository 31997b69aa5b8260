#!/bin/bash
# Generic same-box A/B of one environment switch on the C1 / C2 forwards (bf16 unless the extra bench arguments say
# `--precision fp32`; the printed dtype is the one that ran): bash tools/gpu/ab_env.sh VAR=a VAR=b [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
A=$1; B=$2; shift; shift
run() {  # label, env, args
  label=$1; e=$2; shift; shift
  line=$(env $e python3 bench.py --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1)
  echo "$label [$e]: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["dtype"], d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
for rep in 1 2; do
for e in $A $B; do
run "C1" $e --precision bf16 "$@"
run "C2" $e --precision bf16 --model vigor20 --batch 32 "$@"
done
done
