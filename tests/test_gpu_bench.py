"""bench.py keeps the driver's contract: one JSON line with the headline keys, the roofline and (N = 1) cpu_baseline objects,
and it starts its own ranks for --gpus N.  Short runs on the GPU box (``pytest -m gpu``)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline'}


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_single_gpu():
    d = _run('--steps', '4', '--warmup', '1', '--channels', '8', '--snr-points', '4', '--sustained', '6', '--no-cpu-baseline')
    assert KEYS <= set(d) and d['n_gpus'] == 1 and d['steps'] == 4 and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['value'] > 0 and abs(d['value'] - 32 / (6933 * d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert 'workload' in d['config'] and 'model' not in d['config'] and d['config']['nmse_finite']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    # five full-/half-resolution classes (the direct 64 -> 64 class of round 4 became chain records), the five chain kernels, the two conv_down kernels
    assert len(r['kernels']) == 12 and sum(k.startswith('conv_chain_kernel') for k in r['kernels']) == 5 and 'conv_res_kernel' in r['kernels']
    assert sum(k.startswith('conv_down_kernel') for k in r['kernels']) == 2
    assert d['config']['streams_is_cli_default'] and d['config']['f16x2_range_flag'] == 0
    # frac is the ALGORITHMIC fraction of the dense fp16 MFMA peak; the matrix pipe's busy fraction is a separate field
    assert r['peak'] == 2516.6 and 0 < r['frac'] < r['mfma_busy'] < 1 and 0 < r['frac_step'] < 1
    assert d['dtype'].startswith('f16x2')
    ss = d['strong_small']['by_world_size']
    assert [ss[w]['trajectories_per_gpu'] for w in ('2', '4', '8')] == [16, 8, 4] and all(ss[w]['ms_per_step'] > 0 for w in ss)
    assert d['exact_mode']['conv_mode'] == 'bf16x3' and d['exact_mode']['value'] > 0 and len(d['per_rank']['ms_per_step_by_rank']) == 1
    assert d['sustained_steps'] == 6 and d['sustained_ms_per_step'] > 0 and 'other_launch_mode' in d
    assert d['strong']['scaling'] == 'strong' and d['strong']['trajectories_total'] == 20400
    # round 6: the small-batch numbers and the exact-mode rate as plain fields of `config`; blocks this small run as one batch with launch lanes
    assert all(ss[w]['launch_lanes'] and ss[w]['streams'] == 1 for w in ss)
    assert d['config']['rank_share_ms_per_step_T16'] > 0 and d['config']['exact_mode_bf16x3_channels_per_s'] > 0 and d['config']['sustained_channels_per_s'] > 0


def test_bench_self_launches_its_ranks():
    """``python bench.py --gpus 2`` with no torchrun environment: the parent starts the ranks as children and rank 0 prints the
    line with the aggregate over both.  On this one-GPU box two ranks must share the device, which bench.py only accepts when it
    is asked for explicitly (SBC_DIST_BACKEND=gloo; RCCL refuses two ranks on one device) -- without it, a local rank beyond the
    visible devices is a launch error, not something to fold silently."""
    import torch
    env_clean = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SBC_DIST_BACKEND')}
    if torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                              '--channels', '4', '--snr-points', '1', '--sustained', '0', '--no-strong', '--no-other-mode'],
                             capture_output=True, text=True, timeout=900, env=env_clean)
        assert bad.returncode != 0 and 'one rank per GPU' in (bad.stderr + bad.stdout)
        env_clean['SBC_DIST_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--channels', '8', '--snr-points', '4', '--sustained', '0', '--no-other-mode'],
                       capture_output=True, text=True, timeout=900, env=env_clean)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert d['n_gpus'] == 2 and d['config']['trajectories_per_gpu'] == 32 and d['config']['world_size_seen_by_backend'] == 2
    assert abs(d['value'] - 2 * 32 / (6933 * d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    # round 6: with N > 1 the line also MEASURES BASELINE configs[1] strong-sharded over the ranks of the job (here: this run's 32
    # trajectories, 16 per rank), and repeats the numbers as plain fields of `config` (what the driver's record keeps)
    sc = d['strong_cfg2']
    assert sc['scaling'] == 'strong' and sc['trajectories_total'] == 32 and sc['trajectories_per_gpu'] == 16 and sc['value'] > 0
    assert abs(sc['value'] - 32 / (6933 * sc['ms_per_step'] * 1e-3)) < 1e-6 * sc['value']
    assert abs(d['config']['strong_cfg2_channels_per_s'] - sc['value']) < 1e-2 and d['config']['strong_cfg2_ms_per_step'] > 0
    assert d['strong']['trajectories_total'] == 20400 and d['strong']['trajectories_per_gpu'] == 10200
