#!/usr/bin/env python3
"""GPU box diagnostic: host timeline of the short (K = 20) timed window of bench.py - where its fixed cost goes."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
dev = torch.device('cuda')
net = bench.build_pipeline('Demosaic_01_sRGB_07_11_01_14', dev, which='OriginUniversal')
bays = [make_batch(64, 256, 256, seed=10 + k)[0].to(dev) for k in range(4)]
state = {'k': 0}
def fn():
    with torch.no_grad():
        net(bays[state['k'] % 4])
    state['k'] += 1
for _ in range(1500):
    fn()
torch.cuda.synchronize()
P = time.perf_counter
MODE = sys.argv[1] if len(sys.argv) > 1 else 'plain'
tiny = torch.zeros(1, device=dev)
rows = []
for rep in range(12):
    gc.collect(); gc.disable()
    for _ in range(1200): fn()
    for _ in range(5): fn()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if MODE in ('event', 'kernel'):          # wake the submission path after the blocking wait, outside the window
        evd = torch.cuda.Event()
        if MODE == 'kernel':
            tiny.add_(1.0)
        evd.record()
        while not evd.query():
            pass
    elif MODE == 'spin':                     # spin-wait instead of a blocking synchronize: the host never sleeps
        pass
    t0 = P(); ev0.record(); t1 = P()
    fn(); t2 = P()
    for _ in range(19): fn()
    t3 = P(); ev1.record(); t4 = P()
    torch.cuda.synchronize(); t5 = P()
    torch.cuda.synchronize(); t6 = P()
    gc.enable()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t0) * 1e6, (t6 - t5) * 1e6,
                 ev0.elapsed_time(ev1) * 1e3, (t6 - t0) * 1e6))
rows.sort(key=lambda r: r[-1])
r = rows[len(rows) // 2]
print(MODE, 'median window (us): ev0.record %.1f | first forward %.1f | 19 more forwards %.1f | ev1.record %.1f | t0 -> sync returns %.1f | '
      'second sync %.1f | GPU ev0->ev1 %.1f | wall %.1f  => wall - GPU window = %.1f, per step %.2f' %
      (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[7] - r[6], r[7] / 20))
