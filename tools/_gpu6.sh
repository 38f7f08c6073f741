cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_pair" 2>&1 | tail -6 > gpurun_out/r3k/ops.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capi.py -x -q -m gpu -k "f16w or big_array" 2>&1 | tail -8 > gpurun_out/r3k/par.txt
python bench.py --workload big --steps 8 --warmup 2 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 > gpurun_out/r3k/big_fuse.txt 2>&1
python bench.py --workload big --steps 8 --warmup 2 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 --fuse-pairs 0 > gpurun_out/r3k/big_nofuse.txt 2>&1
python bench.py --workload big --steps 8 --warmup 2 --no-cpu-baseline --no-strong --no-other-mode --sustained 0 --streams 1 > gpurun_out/r3k/big_fuse_s1.txt 2>&1
