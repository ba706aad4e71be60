#!/usr/bin/env python3
"""GPU box: BASELINE.json config 5 - test_split.py-style tiled inference of one 3000x4000 frame
(patch 512 / stride 480 -> 63 tiles) through Bayer_01_Demosaic_02_sRGB_13 (Path-Restore-Bayer -> proxy
bilinear demosaic -> WbQuadratic); MPix/s is quoted on the 12 MPix frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get('RISP_BENCH_NO_NARROW3') == '1':         # A/B on one box: Path-Restore's 3x3 tails back on the vector kernel
    import reconfigisp_amd.convnets as CN
    CN.small_has_narrow3 = lambda *a: False
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.test_split import run_frame

tile_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 21
H, W = 3000, 4000
opt = OrderedDict(model='isp', gpu_ids=[0], dist=False, is_train=False,
                  network_G=dict(which_model_G='IspUniversal', architecture='Bayer_01_Demosaic_02_sRGB_13',
                                 individual_module_paths=[None] * 3, module_path=None),
                  path=dict(pretrain_model_G=None, strict_load=True))
torch.manual_seed(10)
model = create_model(opt)
g = torch.Generator().manual_seed(1)
frame = (torch.randint(0, 1024, (1, 1, H, W), generator=g).float() / 1023.).cuda()
out = run_frame(model, frame, (512, 512), (480, 480), tile_batch)
torch.cuda.synchronize()
t = time.perf_counter()
reps = 3
for _ in range(reps):
    out = run_frame(model, frame, (512, 512), (480, 480), tile_batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
flop = 239680.0 * 63 * 512 * 512
print('full frame %dx%d, 63 tiles (batch %d): %.1f ms/frame, %.1f MPix/s on the 12 MPix frame, %.1f TFLOP/s (%.2f of fp32 MFMA peak), out %s finite=%s'
      % (W, H, tile_batch, dt * 1e3, H * W / dt / 1e6, flop / dt / 1e12, flop / dt / 1e12 / 157.3, tuple(out.shape),
         bool(torch.isfinite(out).all())))
