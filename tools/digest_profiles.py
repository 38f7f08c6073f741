"""Turn gpurun_out/profile_passes/ (tools/profile_passes.sh) into the committed summaries under profiles/."""
import collections, csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, 'gpurun_out', 'profile_passes') + '/'
P = os.path.join(ROOT, 'profiles') + '/'
DOMINANT = 'conv_wx3_kernel<32, 32, 1, true, 3, true, 1, 1>'

rows = list(csv.reader(open(R + 'stats/s_kernel_stats.csv')))
with open(P + 'r01_kernel_stats_bench_steps10.csv', 'w', newline='') as f:
    f.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --no-cpu-baseline '
            '(13 steps incl. 3 warm-up), 1x MI355X, defaults: conv_mode bf16x3, one stream, T=1700\n')
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows:
        w.writerow(r)


def load(d):
    kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(R + d + '/p_kernel_trace.csv'))}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(R + d + '/p_counter_collection.csv')):
        k = kt[r['Dispatch_Id']]
        name = r['Kernel_Name'].split('(')[0].replace('void sbc::', '').replace('sbc::', '')
        if 'at::native' in name or 'rocclr' in name:
            continue
        dur = int(k['End_Timestamp']) - int(k['Start_Timestamp'])
        agg[(name, int(k['Grid_Size_X']))][r['Counter_Name']].append((float(r['Counter_Value']), dur))
    return agg


a1 = load('pmc1')
with open(P + 'r01_pmc_mfma_util.csv', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY\n'
            '#   -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; per (kernel, grid) mean over dispatches.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration\n'
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
with open(P + 'r01_pmc_hbm_traffic.csv', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) of bench.py --steps 2 --warmup 1 --no-cpu-baseline, T=1700.\n'
            '# Raw counter values are KB per dispatch.  Per MI355X_MICROARCH.md FETCH_SIZE reports half of the bytes of wide coalesced (16 B/lane) reads on gfx950,\n'
            '# so hbm_read_MB = 2 * FETCH_SIZE / 1024; WRITE_SIZE is uncalibrated (it matches the algorithmic output bytes here).  MALL hits are counted as traffic.\n'
            'kernel,grid_threads,dispatches,avg_dur_us,FETCH_SIZE_KB_raw,WRITE_SIZE_KB_raw,hbm_read_MB_corrected,hbm_write_MB,GBps_corrected\n')
    for key, d in sorted(a2.items(), key=lambda kv: -sum(t for _, t in kv[1]['FETCH_SIZE'])):
        n = len(d['FETCH_SIZE'])
        dur = sum(t for _, t in d['FETCH_SIZE']) / n
        fs = sum(v for v, _ in d['FETCH_SIZE']) / n
        ws = sum(v for v, _ in a3[key]['WRITE_SIZE']) / max(1, len(a3[key]['WRITE_SIZE']))
        rd, wr = 2 * fs / 1024, ws / 1024
        tr[key[0]] = (fs, ws, dur, key[1])
        f.write('"%s",%d,%d,%.1f,%.0f,%.0f,%.1f,%.1f,%.0f\n' % (key[0], key[1], n, dur / 1e3, fs, ws, rd, wr, (rd + wr) * 1e6 / dur))
fs, ws, dur, grid = tr[DOMINANT]
json.dump({
    'kernel': '%s: 3x3 32->32 at 64x16 (grid %d threads = %d trajectories per launch)' % (DOMINANT, grid, grid // 256 * 128 // 1024),
    'trajectories_per_launch': grid // 256 * 128 // 1024,
    'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/r01_pmc_hbm_traffic.csv',
    'fetch_size_kb_raw': round(fs), 'write_size_kb': round(ws),
    'correction': 'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for wide coalesced reads, '
                  'MI355X_MICROARCH.md); WRITE_SIZE as reported',
    'hbm_bytes_per_launch': int(round((2 * fs + ws) * 1024)),
    'algorithmic_bytes_per_launch': 'input + residual (15 of 18 launches) + output, 131072 B per trajectory each'},
    open(P + 'r01_traffic.json', 'w'), indent=1)
for r in rows[1:6]:
    print(r[0][:70], r[1], '%.1f us' % (float(r[3]) / 1e3))
print(open(P + 'r01_traffic.json').read())
