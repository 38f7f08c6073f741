cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_train.py tests/test_gpu_train_ops.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3n/t.txt
python bench.py --no-cpu-baseline --no-strong > gpurun_out/r3n/bench.txt 2>&1
python bench.py --workload train > gpurun_out/r3n/train.txt 2>&1
python bench.py --workload train --graph 1 > gpurun_out/r3n/train_graph.txt 2>&1
