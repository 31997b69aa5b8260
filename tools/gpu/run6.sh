#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for ab in 8 15; do
  echo "== ablate=$ab"
  for sh in 65536,112,672 65536,672,112; do
  CCVPE_PW_ABLATE=$ab python3 tools/pw_probe.py bf16 20 $sh 2>&1 | grep -E "pwprof|bf16"
  CCVPE_PW_ABLATE=$ab python3 tools/pw_probe.py fp32 20 $sh 2>&1 | grep -E "pwprof|fp32"
  done
done
