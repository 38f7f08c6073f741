cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --conv-mode f16x2 > gpurun_out/r3b/bench_f16x2.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode f16x2 --streams 2 > gpurun_out/r3b/bench_f16x2_s2.txt 2>&1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-strong --no-other-mode --conv-mode bf16x3 --streams 2 > gpurun_out/r3b/bench_bf16x3_s2.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3b/prof -- python3 bench.py --steps 10 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 --conv-mode f16x2 > gpurun_out/r3b/prof_run.txt 2>&1
find gpurun_out/r3b/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3b/kernel_stats_f16x2.csv
rm -rf gpurun_out/r3b/prof
