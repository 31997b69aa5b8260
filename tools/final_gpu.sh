# Round-end evidence run on the GPU box: full -m gpu suite, smoke(), the default bench line, rocprofv3 kernel stats of the
# same command, the two PMC passes (FETCH_SIZE / WRITE_SIZE) for profiles/pmc_traffic.json, and the C3 training line.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err
python bench.py --train --model kitti --batch 64 --steps 5 --warmup 2 > gpurun_out/final/bench_train_kitti_b64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/final/bench_prof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-kernel-events --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-kernel-events --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/stats_train -- python3 $GRAFT_REPO_ROOT/bench.py --train --model kitti --batch 64 --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/final/pytest_gpu.txt gpurun_out/final/smoke.txt
cut -c1-300 gpurun_out/final/bench_n1.json
cut -c1-200 gpurun_out/final/bench_train_kitti_b64.json
