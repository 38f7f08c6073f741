#!/bin/bash
# Runs on the GPU box (gpurun): the four rocprofv3 passes behind profiles/rNN_*_<workload>.  Counters are collected in their
# own runs with --kernel-trace only (no --stats / sys-trace next to --pmc); FETCH_SIZE and WRITE_SIZE need separate passes.
# usage: tools/profile_passes.sh [cdlc|big]
W=${1:-cdlc}
R=$PWD; OUT=$R/gpurun_out/profile_passes_$W; mkdir -p $OUT
# one stream: a kernel then has the chip to itself, which is what the per-kernel numbers (and bench.py's roofline segment) mean
ARGS="--workload $W --streams 1 --no-cpu-baseline --no-strong --no-other-mode --no-exact-mode --sustained 0 $EXTRA_ARGS"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 10 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o p -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -o p -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -o p -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/pmc3.log 2>&1
tail -c 600 $OUT/stats.log; ls $OUT/*
