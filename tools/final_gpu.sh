set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/final/pytest_gpu.txt
python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/final/bench_prof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-kernel-events --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-kernel-events --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/final/*/*/ | head -30
cat gpurun_out/final/pytest_gpu.txt
cut -c1-400 gpurun_out/final/bench_n1.json
