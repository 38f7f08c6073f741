"""Turn gpurun_out/profile_passes_<workload>/ (tools/profile_passes.sh) into the committed summaries under profiles/.
usage: digest_profiles.py [cdlc|big] [round tag, default r03]"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = sys.argv[1] if len(sys.argv) > 1 else 'cdlc'
TAG = sys.argv[2] if len(sys.argv) > 2 else 'r06'
R = os.path.join(ROOT, 'gpurun_out', 'profile_passes_' + W) + '/'
P = os.path.join(ROOT, 'profiles') + '/'
BIG = W == 'big'
# the kernel classes bench.py tags; the dominant one = largest total time in the --stats pass
CANDIDATES = (['conv_x3_kernel<32, 32, 3, 2, 1, 4, 1, true, 1>', 'conv_wx3_kernel<32, 32, 1, true, 3, true, 1, 1, 1>', 'conv_pair_kernel<64, 4, 1, 8, 32>',
               'conv_x3_kernel<64, 64, 3, 2, 2, 4, 1, true, 1>', 'conv_wx3_kernel<64, 64, 1, true, 2, false, 2, 1, 1>'] if BIG else
              ['conv_wx3_kernel<32, 32, 1, true, 4, true, 1, 1, 2>', 'conv_pair_roll_kernel<2>', 'conv_pair_p3_kernel<16, 8, 2, 4, 32>', 'conv_down_kernel<32, 64, 16>', 'conv_down_kernel<64, 64, 8>',
               'conv_pool_kernel<16, 8, 2, 4, 32>', 'conv_wx3_kernel<64, 64, 1, true, 2, false, 2, 1, 2>',
               'conv_dp_kernel<64, 8, 8, 1, false, 4, false>', 'conv_dp_kernel<64, 8, 8, 1, false, 4, true>', 'conv_res_kernel', 'conv_chain_kernel<128, 2, 8, 1>', 'conv_chain_kernel<64, 2, 4, 1>',
               'conv_chain_kernel<64, 4, 4, 1>', 'conv_chain_kernel<64, 8, 8, 1>', 'conv_chain_kernel<32, 8, 4, 1>'])
PX = (256 * 64) if BIG else (64 * 16)
T = 1024 if BIG else 1700
CMD = '--workload %s --streams 1 --no-cpu-baseline --no-strong --no-other-mode --no-exact-mode --sustained 0' % W
DESC = ('conv_mode f16w + fused RCU pairs + folded statistics, one stream, T=1024 (256x64 arrays)' if BIG else
        'conv_mode f16x2 (calibrated activation scales) + fused RCU pairs, CRP stages, ResidualBlocks and low-resolution chains + folded statistics, one stream, T=1700')


def find(d, suffix):
    hits = glob.glob(R + d + '/**/*' + suffix, recursive=True)
    assert hits, (d, suffix)
    return hits[0]

rows = list(csv.reader(open(find('stats', 'kernel_stats.csv'))))
with open(P + '%s_kernel_stats_%s_steps10.csv' % (TAG, W), 'w', newline='') as f:
    n_steps = max([int(r[1]) for r in rows[1:] if 'langevin' in r[0]] + [0])
    f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 %s '
            '(%d Langevin steps in all: 3 warm-up + 10 timed + the one-stream roofline segments of bench.py), 1x MI355X, %s\n'
            % (CMD, n_steps, DESC))
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows:
        w.writerow(r)


def load(d):
    kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(find(d, 'kernel_trace.csv')))}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(d, 'counter_collection.csv'))):
        k = kt[r['Dispatch_Id']]
        name = r['Kernel_Name'].split('(')[0].replace('void sbc::', '').replace('sbc::', '')
        if 'at::native' in name or 'rocclr' in name:
            continue
        dur = int(k['End_Timestamp']) - int(k['Start_Timestamp'])
        agg[(name, int(k['Grid_Size_X']))][r['Counter_Name']].append((float(r['Counter_Value']), dur))
    return agg


a1 = load('pmc1')
with open(P + '%s_pmc_mfma_util_%s.csv' % (TAG, W), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY\n'
            '#   -- python3 bench.py --steps 2 --warmup 1 %s; per (kernel, grid) mean over dispatches.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration\n' % CMD +
            '#   (over-estimates for kernels < 60 us: the counter runs a little before/after the dispatch); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * cycles).\n'
            'kernel,grid_threads,dispatches,avg_dur_us,clock_ghz,mfma_busy_frac,wait_inst_any_frac,active_inst_frac\n')
    for key, d in sorted(a1.items(), key=lambda kv: -sum(t for _, t in kv[1]['GRBM_GUI_ACTIVE'])):
        n = len(d['GRBM_GUI_ACTIVE'])
        dur = sum(t for _, t in d['GRBM_GUI_ACTIVE']) / n
        m = lambda c: sum(v for v, _ in d[c]) / n                                  # noqa: E731
        clk = m('GRBM_GUI_ACTIVE') / 8 / dur
        f.write('"%s",%d,%d,%.1f,%.2f,%.3f,%.3f,%.3f\n' % (key[0], key[1], n, dur / 1e3, clk,
                m('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * dur * clk), m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'),
                m('SQ_ACTIVE_INST_ANY') / m('SQ_WAVE_CYCLES')))
a2, a3 = load('pmc2'), load('pmc3')
tr = {}
with open(P + '%s_pmc_hbm_traffic_%s.csv' % (TAG, W), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) of bench.py --steps 2 --warmup 1 %s; %s.\n' % (CMD, DESC) +
            '# Raw counter values are KB per dispatch.  Per MI355X_MICROARCH.md FETCH_SIZE reports half of the bytes of wide coalesced (16 B/lane) reads on gfx950,\n'
            '# so hbm_read_MB = 2 * FETCH_SIZE / 1024; WRITE_SIZE is uncalibrated (it matches the algorithmic output bytes here).  MALL hits are counted as traffic.\n'
            'kernel,grid_threads,dispatches,avg_dur_us,FETCH_SIZE_KB_raw,WRITE_SIZE_KB_raw,hbm_read_MB_corrected,hbm_write_MB,GBps_corrected\n')
    for key, d in sorted(a2.items(), key=lambda kv: -sum(t for _, t in kv[1]['FETCH_SIZE'])):
        n = len(d['FETCH_SIZE'])
        dur = sum(t for _, t in d['FETCH_SIZE']) / n
        fs = sum(v for v, _ in d['FETCH_SIZE']) / n
        ws = sum(v for v, _ in a3[key]['WRITE_SIZE']) / max(1, len(a3[key]['WRITE_SIZE']))
        rd, wr = 2 * fs / 1024, ws / 1024
        if key[0] not in tr or key[1] > tr[key[0]][3]:
            tr[key[0]] = (fs, ws, dur, key[1])
        f.write('"%s",%d,%d,%.1f,%.0f,%.0f,%.1f,%.1f,%.0f\n' % (key[0], key[1], n, dur / 1e3, fs, ws, rd, wr, (rd + wr) * 1e6 / dur))
kern = {}
for name in CANDIDATES:
    if name not in tr:
        continue
    fs, ws, dur, grid = tr[name]
    kern[name] = {'grid_threads': grid, 'fetch_size_kb_raw': round(fs), 'write_size_kb': round(ws),
                  'hbm_bytes_per_launch': int(round((2 * fs + ws) * 1024)), 'avg_dur_us_in_the_pmc_pass': round(dur / 1e3, 1)}
json.dump({
    'trajectories_per_launch': T, 'conv_mode': 'f16w' if BIG else 'f16x2',
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/%s_pmc_hbm_traffic_%s.csv' % (TAG, W),
    'correction': 'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for wide coalesced reads, '
                  'MI355X_MICROARCH.md); WRITE_SIZE as reported',
    'note': 'per kernel: its largest-grid dispatches (the 64 -> 64 symbol also runs the next level down with a smaller grid)',
    'kernels': kern}, open(P + '%s_traffic_%s.json' % (TAG, W), 'w'), indent=1)
for r in rows[1:6]:
    print(r[0][:70], r[1], '%.1f us' % (float(r[3]) / 1e3))
print(open(P + '%s_traffic_%s.json' % (TAG, W)).read())
