import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from reconfigisp_amd.codes.models import networks
net = networks.define_G({'network_G': {'which_model_G': 'IspUniversal', 'architecture': 'Bayer_01_Demosaic_02_sRGB_13_12',
                                       'module_path': None, 'individual_module_paths': [None] * 8}}).cuda().eval()
g = torch.Generator().manual_seed(1)
x = (torch.randint(0, 1024, (1, 1, 3000, 4000), generator=g).float() / 1023.).cuda()
with torch.no_grad():
    y = net(x); torch.cuda.synchronize()
    t = time.perf_counter(); y = net(x); torch.cuda.synchronize(); dt = time.perf_counter() - t
    # crop consistency: the same pipeline on a crop must agree in the crop's interior (receptive field < 64)
    yc = net(x[:, :, 1000:1512, 2000:2768].contiguous())
d = (y[:, :, 1000 + 64:1512 - 64, 2000 + 64:2768 - 64] - yc[:, :, 64:-64, 64:-64]).abs().max().item()
print('full frame untiled: %.1f ms, finite=%s, max |full - crop| in the crop interior %.2e, peak mem %.1f GB'
      % (dt * 1e3, bool(torch.isfinite(y).all()), d, torch.cuda.max_memory_allocated() / 2**30))
