#!/bin/bash
# Vector-issue accounting of every kernel of the one-stream Langevin step (GPU box): instructions per launch and the share of the
# SIMD cycles they occupy under the serial-issue model measured by tools/experiments/issue_overlap.hip (4 cycles per vector
# instruction, 16 per v_mfma_f32_16x16x32_f16 and per transcendental).
R=$PWD; OUT=$R/gpurun_out/pmc_issue; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --streams 1 --no-exact-mode --no-strong --no-other-mode --no-cpu-baseline --sustained 0"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT -o p -- python3 $R/bench.py $ARGS > $OUT/run.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + '/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:62]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
tot = sum(c['GRBM_GUI_ACTIVE'] for c in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE'])
print('%-62s %5s %6s %9s %9s %8s %8s %7s' % ('kernel', 'calls', 'time%', 'valu/call', 'mfma/call', 'trans', 'lds', 'issue'))
for k, c in rows[:18]:
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    valu = c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']
    trans = c.get('SQ_INSTS_VALU_TRANS', 0.0)
    issue = (4 * (valu - trans) + 16 * trans + 16 * c['SQ_INSTS_MFMA']) / 1024 / cyc if cyc else 0
    print('%-62s %5d %6.1f %9.0f %9.0f %8.0f %8.0f %7.3f' % (k, n[k], 100 * c['GRBM_GUI_ACTIVE'] / tot, valu / n[k], c['SQ_INSTS_MFMA'] / n[k], trans / n[k], c['SQ_INSTS_LDS'] / n[k], issue))
PY
