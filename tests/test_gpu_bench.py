"""GPU: bench.py end to end, including the self-launching N > 1 path (all ranks on device 0 over gloo: a plumbing
dry run of what the driver's SCALE run does on an 8-GPU node; its numbers are not an N-GPU measurement)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*argv, **env):
    e = dict(os.environ, **env)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_is_self_consistent():
    line = _bench('--steps', '400', '--warmup', '50', '--search-batch', '4')
    assert line['n_gpus'] == 1 and line['unit'] == 'MPix/s' and line['vs_baseline'] is None
    roof, extra = line['roofline'], line['extra']
    assert extra['kernel_ms'] <= line['ms_per_step'] * 1.001                 # a kernel cannot outlast its step
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3 and 0 < roof['frac'] < 1
    pix = 64 * 256 * 256
    assert abs(roof['achieved'] - 64 * pix / (extra['kernel_ms'] * 1e-3) / 1e9) < 5.0      # kernel_ms is rounded to 0.01 us
    assert abs(line['value'] - pix / (line['ms_per_step'] * 1e-3) / 1e6) < 0.01 * line['value']
    # the matrix pipes' busy share (f32 and f16 instructions against their own nameplates) is a utilisation; the effective figure
    # prices direct-convolution FLOPs against the f32 peak and passes 1 where Winograd / the f16 pipe do the work
    assert 0 < extra['cnn_mfma_issued_frac'] < 1 and 'cnn_effective_frac' not in extra
    assert extra['cnn_arith'].startswith('2 x f16 split') and extra['cnn_mfma_f16_issued_TFLOPs'] > 0
    assert 0 < extra['cnn_f32_mfma_issued_frac'] < 1 and extra['cnn_f32_MPix_s'] < extra['cnn_MPix_s'] * 1.02
    assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['value'] > 0
    s = extra['search_step']
    assert s['n_gpus'] == 1 and s['per_rank_batch'] == 4 and s['s_per_step'] > 0
    # every quoted configuration is in the driver-run line: configs 3 and 5 and the reference's shipped search geometry
    c3, c5, sh = extra['config3'], extra['config5'], extra['search_step_shipped']
    assert 'error' not in c3 and 'error' not in c5 and 'error' not in sh, (c3, c5, sh)
    assert 0.05 < c3['s_per_step'] < 2 and 0 < c3['mfma_f16_issued_of_nameplate'] < c3['mfma_f16_issued_of_sustained'] < 1
    shares = c3['kernel_time_share']
    assert abs(sum(shares.values()) - 1) < 0.01 and sum(v for k, v in shares.items() if 'wide' in k) > 0.4 and any('few-channel' in k for k in shares)
    assert 0.5 * c3['s_per_step'] * 1e3 < c3['kernel_ms_per_step_between_events'] < 1.5 * c3['s_per_step'] * 1e3
    assert 2 < c5['ms_per_frame'] < 200 and c5['finite'] and c5['first_frame_ms'] >= c5['ms_per_frame'] * 0.9
    assert sh['per_rank_batch'] == 4 and sh['iters_per_s'] > 1 and sh['c_abi_calls_per_iter'] > 100
    assert not any(k.endswith('frac') and isinstance(v, float) and v > 1 for k, v in extra.items())


def test_bench_gpus2_self_launch_dry_run():
    line = _bench('--gpus', '2', '--steps', '100', '--warmup', '10', '--no-cnn', '--no-configs', '--search-batch', '4',
                  RISP_BENCH_ONE_DEVICE='1')
    assert line['n_gpus'] == 2 and line['dry_run_all_ranks_on_one_device'] is True
    assert line['config']['global_batch'] == 128 and 'cpu_baseline' not in line
    s = line['extra']['search_step']
    assert s['n_gpus'] == 2 and s['per_rank_batch'] == 2 and s['allreduce_calls_per_step'] == 4
    assert s['allreduce_s_per_step'] > 0 and s['one_gpu']['s_per_step'] > 0
    sh = line['extra']['search_step_shipped']                    # the reference's shipped geometry, its batch of 4 over the ranks
    assert sh['n_gpus'] == 2 and sh['per_rank_batch'] == 2 and sh['allreduce_s_per_iter'] > 0
    # two ranks with half of the batch each compute the same averaged gradients as one rank with the whole batch
    assert abs(s['loss_rank0'] - s['one_gpu']['loss']) < 0.2 * abs(s['one_gpu']['loss']) + 1e-6


def test_bench_gpus8_self_launch_dry_run():
    """what the driver's SCALE run starts on an 8-GPU node, with all eight ranks on device 0 over gloo: launcher,
    rendezvous, batch / tile sharding, the four all-reduces per search iteration, one line"""
    line = _bench('--gpus', '8', '--steps', '50', '--warmup', '5', '--batch', '8', '--search-batch', '8', '--no-cnn', '--no-configs',
                  RISP_BENCH_ONE_DEVICE='1')
    assert line['n_gpus'] == 8 and line['dry_run_all_ranks_on_one_device'] is True
    assert line['config']['global_batch'] == 64 and 'cpu_baseline' not in line
    s = line['extra']['search_step']
    assert 'error' not in s, s
    assert s['n_gpus'] == 8 and s['per_rank_batch'] == 1 and s['allreduce_calls_per_step'] == 4
    assert s['one_gpu']['s_per_step'] > 0 and s['speedup_vs_one_gpu'] > 0
