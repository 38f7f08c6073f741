#!/bin/bash
# LDS counters of every kernel of the one-stream Langevin step (GPU box): conflict share and LDS-array occupancy per kernel.
R=$PWD; OUT=$R/gpurun_out/pmc_lds; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-exact-mode --no-strong --no-other-mode --no-cpu-baseline --sustained 0 > $OUT/run.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + '/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:64]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE'])
print('%-64s %7s %9s %9s %9s %9s' % ('kernel', 'calls', 'cycles/8', 'LDS busy', 'conflict', 'MFMA busy'))
for k, c in rows[:16]:
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    print('%-64s %7d %9.0f %9.3f %9.3f %9.3f' % (k, n[k], cyc / max(n[k], 1), c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc if cyc else 0,
          c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'] if c['SQ_LDS_IDX_ACTIVE'] else 0, c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc if cyc else 0))
PY
