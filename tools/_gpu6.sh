cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r3m/gputests.txt
python bench.py > gpurun_out/r3m/bench_default.txt 2>&1
python bench.py --workload big > gpurun_out/r3m/bench_big.txt 2>&1
