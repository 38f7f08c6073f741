cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pairs and (forward or plumbing or trunc)" 2>&1 | tail -8 > gpurun_out/r3d/parity.txt
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --sustained 0 > gpurun_out/r3d/bench_nofuse.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --fuse-pairs 1 --sustained 0 > gpurun_out/r3d/bench_fuse.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --sustained 0 > gpurun_out/r3d/bench_nofuse2.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --fuse-pairs 1 --sustained 0 > gpurun_out/r3d/bench_fuse2.txt 2>&1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/prof -- python3 bench.py --steps 10 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 --conv-mode f16x2 --fuse-pairs 1 > gpurun_out/r3d/prof_run.txt 2>&1
find gpurun_out/r3d/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3d/kernel_stats_fuse.csv
rm -rf gpurun_out/r3d/prof
