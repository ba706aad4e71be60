"""`python bench.py --gpus N` must start its own rank processes (the driver's SCALE run does exactly that) and pass ONE
JSON line through.  RISP_BENCH_LAUNCH_ONLY=1 makes the ranks stop after rendezvous + one gloo collective, so the
launcher, the torch.distributed.run plumbing on 127.0.0.1 and the line forwarding are exercised without a GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(n):
    env = dict(os.environ, RISP_BENCH_LAUNCH_ONLY='1', OMP_NUM_THREADS='1')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1'],
                          env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus2_launches_its_own_ranks():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['launch_only'] is True
    assert line['rank_sum'] == 3.0                       # ranks 0 and 1 both joined the collective


def test_bench_refuses_more_ranks_than_gpus_without_the_dry_run_switch():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'RISP_BENCH_LAUNCH_ONLY', 'RISP_BENCH_ONE_DEVICE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64'], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and 'GPU(s) visible' in r.stderr and not r.stdout.strip()
