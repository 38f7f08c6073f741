#!/bin/bash
# PMC counters of one convolution launch (GPU box): tools/pmc_conv.sh <out-name> "<prof_conv.py arguments>" [counter ...]
# PROF=prof_dp.py (or another tools/prof_*.py taking --iters) selects the launcher.  Default counters: matrix-pipe busy, issue / wait split, vector and LDS instruction counts, LDS bank conflicts.
name=$1; args=$2; shift 2
R=$PWD; OUT=$R/gpurun_out/pmc_$name; mkdir -p $OUT
C1="${@:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE}"
C2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/a -o p -- python3 $R/tools/${PROF:-prof_conv.py} $args --iters 3 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $OUT/b -o p -- python3 $R/tools/${PROF:-prof_conv.py} $args --iters 3 > $OUT/b.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv' not in k: continue
        agg[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print('   %-28s %14.0f  (per launch, %d launches)' % (n, sum(v) / len(v), len(v)))
    g = lambda n: sum(c[n]) / len(c[n]) if n in c else float('nan')
    print('   MFMA busy = %.3f   VALU active = %.3f   wait_inst = %.3f   wait_any = %.3f   LDS conflict / LDS active = %.3f   LDS active / busy = %.3f'
          % (g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES') / 4, g('SQ_ACTIVE_INST_VALU') * 4 / g('SQ_WAVE_CYCLES') if 0 else g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'),
             g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'), g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'),
             g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'), g('SQ_LDS_IDX_ACTIVE') / g('SQ_BUSY_CYCLES')))
PY
