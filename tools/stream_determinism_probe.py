"""Tuning / debugging aid: 12 multi-stream (2 or 3 sub-batch streams, eager and hipGraph) runs of a small schedule must be
bit-identical to the single-stream run.  usage: stream_determinism_probe.py [f16x2|bf16x3|f32|f16w] [reps] [channels]; SBC_LIB_PATH selects a library
variant (tools/build_variant.sh).  With packed-fp32 instructions in the build 8 of 12 runs differed."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from score_based_channels_amd import synth
from score_based_channels_amd.ald import snr_to_noise
from score_based_channels_amd.config import default_config
from score_based_channels_amd.driver import run_trajectories
from score_based_channels_amd.scorenet import ScoreNet
from score_based_channels_amd.weights import seeded_state_dict
mode = sys.argv[1] if len(sys.argv) > 1 else 'f16x2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nch_arg = int(sys.argv[3]) if len(sys.argv) > 3 else 24
cfg = default_config(); sd = seeded_state_dict(cfg, 2024)
net = ScoreNet(cfg, conv_mode=mode).cuda().load_state_dict(sd)
nch, nt, nr, npil = nch_arg, 64, 16, 38
raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=41)
H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(42), nch, nt, npil), (0, 2, 1)))
snr = np.array([-10.0, 0.0, 12.5, 30.0])
idx = np.tile(np.arange(nch), len(snr)); ln = np.repeat(snr_to_noise(snr, nt), nch)
init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(7))
def run(n, g):
    return run_trajectories(net, H, Pm, idx, idx, ln, 3e-11, 0.01, [0, 1155, 2310], 3, 11, init, n_streams=n, use_graph=g, return_final=True)
ref = run(1, False)
bad = 0
for rep in range(reps):
    log, est = run(2 + rep % 2, bool(rep & 2))
    bad += not (np.array_equal(log, ref[0]) and np.array_equal(est, ref[1]))
print('RESULT mode', mode, 'env', {k: v for k, v in os.environ.items() if k.startswith('SBC_')}, 'mismatching runs', bad, 'of', reps, '(%d trajectories)' % (nch * len(snr)))
