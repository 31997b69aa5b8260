#!/bin/bash
# Round-end run: the whole GPU test suite, then the evidence script (bench line, rocprofv3 stats, PMC passes).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/pytest_gpu_final.log
cat gpurun_out/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 tools/gemm_ceiling.py gpurun_out/gemm_ceiling.json 2>&1 | grep -v amdgpu | tail -10
bash tools/gpu/evidence_r02.sh $1 2>&1 | tail -25
