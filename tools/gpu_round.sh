set -e
mkdir -p gpurun_out/r01c
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r01c/pytest.log 2>&1
tail -3 gpurun_out/r01c/pytest.log
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01c/bench.json 2> gpurun_out/r01c/bench.err
cat gpurun_out/r01c/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r01c/pmc_w -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01c/pmc_w.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r01c/pmc_f -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r01c/pmc_f.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/stage_timing.py > gpurun_out/r01c/stage_timing.txt 2>&1 || true
tail -25 gpurun_out/r01c/stage_timing.txt
