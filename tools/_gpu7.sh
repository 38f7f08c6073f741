cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3o/smoke.txt 2>&1
python bench.py --full-schedule --no-cpu-baseline --no-strong --no-other-mode > gpurun_out/r3o/bench_full.txt 2>&1
( time python -m score_based_channels_amd.test_score --synthetic --synthetic_weights 2024 --seed 1 --no_plot ) > gpurun_out/r3o/cli.txt 2>&1
SBC_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --sustained 0 --no-other-mode > gpurun_out/r3o/selflaunch2.txt 2>&1
python bench.py --force-dist --steps 10 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 > gpurun_out/r3o/rccl1.txt 2>&1
