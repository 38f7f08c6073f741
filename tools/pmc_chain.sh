#!/bin/bash
# PMC view of the SBC_OP_CHAIN kernels in isolation (tools/prof_chain.py): where the SIMD cycles go.  GPU box.
R=$PWD; OUT=$R/gpurun_out/pmc_chain; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- python3 $R/tools/prof_chain.py 1700 10 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- python3 $R/tools/prof_chain.py 1700 10 > $OUT/b.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ('a', 'b'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(out + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void sbc::', '') + ' grid ' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
    for k, c in sorted(agg.items()):
        if 'chain' not in k: continue
        cyc = c['GRBM_GUI_ACTIVE'] / 8 / n[k]
        if sub == 'a':
            wc = c['SQ_WAVE_CYCLES']
            print('%-44s n=%3d cycles/launch %8.0f  mfma_busy %.3f  wave-cycles: wait_any %.3f wait_inst %.3f (lds %.3f) active %.3f' % (
                k, n[k], cyc, c['SQ_VALU_MFMA_BUSY_CYCLES'] / n[k] / 1024 / cyc, c['SQ_WAIT_ANY'] / wc, c['SQ_WAIT_INST_ANY'] / wc, c['SQ_WAIT_INST_LDS'] / wc, c['SQ_ACTIVE_INST_ANY'] / wc))
        else:
            print('%-44s n=%3d per launch: valu %9.0f mfma %9.0f lds %9.0f vmem %8.0f | lds idx_active/cycle %.3f conflict share %.3f' % (
                k, n[k], (c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / n[k], c['SQ_INSTS_MFMA'] / n[k], c['SQ_INSTS_LDS'] / n[k], c['SQ_INSTS_VMEM'] / n[k],
                c['SQ_LDS_IDX_ACTIVE'] / n[k] / 256 / cyc, c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1)))
PY
