cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
for i in 1 2; do
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --fuse-pairs 1 --sustained 0 > gpurun_out/r3e/bench_fuse_$i.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --fuse-pairs 1 --sustained 0 --streams 2 > gpurun_out/r3e/bench_fuse_s2_$i.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e/prof -- python3 bench.py --steps 10 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 --conv-mode f16x2 --fuse-pairs 1 > gpurun_out/r3e/prof_run.txt 2>&1
find gpurun_out/r3e/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3e/kernel_stats.csv
rm -rf gpurun_out/r3e/prof
