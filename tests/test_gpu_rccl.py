"""GPU: the RCCL code path executed on the one GPU this pool has - a process group of one rank on backend 'nccl'.
DartsModel(dist=True) sends its four gradient sets per iteration through ncclAllReduce (DartsModel._allreduce_mean;
models/darts_model.py:31,173 is DDP's bucket all-reduce in the reference) and run_frame its tiles through ncclAllGather;
with one rank both must leave every bit unchanged.  (A scaling curve needs an 8-GPU node, which this pool does not
have: none is claimed anywhere.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_one_rank_process_group_on_rccl_changes_no_bit(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rccl.pt')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_world1_job.py'), out, str(port)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    res = torch.load(out)
    assert res['backend'] == 'nccl'
    plain, ranked = res['plain'], res['ranked']
    assert ranked['darts'].pop('comm_seconds').item() > 0          # the four all-reduces of the iteration really ran
    plain['darts'].pop('comm_seconds')
    calls = ranked['darts'].pop('allreduce_ops')                    # four gradient sets per iteration, each: copy in, collective, copy out
    # (the first use of a gradient set also allocates its flat buffer and cuts the views: metadata, no launches)
    calls = [[o for o in ops if not any(m in o for m in ('aten.empty', 'aten.split_with_sizes', 'aten.view'))] for ops in calls]
    assert len(calls) == 4 and all(len(ops) <= 3 for ops in calls), calls
    assert all(sum('allreduce' in o for o in ops) == 1 and sum('_foreach_copy_' in o for o in ops) == 2 for ops in calls), calls
    assert set(plain['darts']) == set(ranked['darts'])
    for k in plain['darts']:
        assert torch.equal(plain['darts'][k], ranked['darts'][k]), k
    assert torch.equal(plain['frame'], ranked['frame'])
