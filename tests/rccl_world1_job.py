"""Child process of tests/test_gpu_rccl.py: a process group of ONE rank on backend 'nccl' (= RCCL on ROCm) on cuda:0.
Runs one DARTS iteration with dist=True - every gradient set goes through DartsModel._allreduce_mean, i.e. through
ncclAllReduce - and one tiled frame through run_frame's all_gather, and writes the results next to the same jobs without
torch.distributed.  Reference: train.py:20-55 (init_dist), models/darts_model.py:31,173, test_split.py:82-106."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)


def darts_job(distributed):
    from conftest import load_golden
    from test_host_logic import darts_opt, seed_darts
    from reconfigisp_amd.codes.models import create_model
    g = load_golden('darts_step')
    opt = darts_opt(torch.device('cuda'))
    opt['dist'] = distributed
    torch.manual_seed(0)
    model = create_model(opt)
    seed_darts(model)
    ops_per_call = []
    if distributed:
        model.comm_seconds = 0.0                   # bracket the collectives: proves they ran
        # the framework operations one _allreduce_mean issues (one list-wide copy in, the collective, one list-wide copy out)
        from torch.utils._python_dispatch import TorchDispatchMode

        class Count(TorchDispatchMode):
            def __init__(self):
                super().__init__()
                self.names = []

            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                self.names.append(str(func))
                return func(*args, **(kwargs or {}))

        inner = model._allreduce_mean

        def counted(tensors):
            with Count() as c:
                r = inner(tensors)
            ops_per_call.append(c.names)
            return r
        model._allreduce_mean = counted
    data = tuple(torch.from_numpy(np.asarray(g[k])) for k in ('img', 'gt', 'val_img', 'val_gt'))
    model.feed_data(data)
    model.update_learning_rate(0, warmup_iter=-1)
    model.optimize_alphas()
    out = {'val_loss': model.val_loss.detach().cpu()}
    out.update({'alpha_grad%d' % k: a.grad.cpu() for k, a in enumerate(model.netG.alphas)})
    model.optimize_parameters()
    out.update({k: v.detach().cpu() for k, v in model.netG.state_dict().items()})
    out['comm_seconds'] = torch.tensor(model.comm_seconds or 0.0)
    if distributed:
        out['allreduce_ops'] = ops_per_call
    return out


def frame_job(collective):
    from collections import OrderedDict
    from reconfigisp_amd.codes.models import create_model
    from reconfigisp_amd.codes.test_split import run_frame
    opt = OrderedDict(model='isp', gpu_ids=[0], dist=collective, is_train=False,
                      network_G=dict(which_model_G='IspUniversal', architecture='Bayer_02_Demosaic_01_sRGB_11_01_14',
                                     individual_module_paths=[None] * 8, module_path=None),
                      path=dict(pretrain_model_G=None, strict_load=True))
    model = create_model(opt)
    g = np.random.Generator(np.random.PCG64(5))
    frame = torch.from_numpy(g.random((1, 1, 88, 124)).astype(np.float32))
    return run_frame(model, frame, (32, 32), (24, 24), tile_batch=4, rank=0, world=1, collective=collective).cpu()


def main():
    out_path, port = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    plain = {'darts': darts_job(False), 'frame': frame_job(False)}
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % port, rank=0, world_size=1)
    try:
        t = torch.ones(4, device='cuda')
        dist.all_reduce(t)                          # RCCL is up
        assert t.tolist() == [1.0] * 4
        ranked = {'darts': darts_job(True), 'frame': frame_job(True)}
        backend = dist.get_backend()
    finally:
        dist.destroy_process_group()
    torch.save({'plain': plain, 'ranked': ranked, 'backend': backend}, out_path)


if __name__ == '__main__':
    main()
