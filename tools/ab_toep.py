"""A/B of the Toeplitz-band split-precision kernel (risp_conv2d_toep) against the vector-FMA kernel (risp_conv2d_small) on the
layers it takes over: error against float64 and time per launch.  Run on the GPU box:
    python tools/ab_toep.py            # RISP_AB_N=32 RISP_AB_HW=256 RISP_AB_G=1 (members of a grouped launch)"""
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reconfigisp_amd import convnets as CN  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    n, hw, G = int(os.environ.get('RISP_AB_N', 32)), int(os.environ.get('RISP_AB_HW', 256)), int(os.environ.get('RISP_AB_G', 1))
    gen = torch.Generator('cuda').manual_seed(1)
    cases = [('9x9 64->3 backward-data + add', 9, 64, 3, True, CN.EPI_ADD), ('5x5 32->3 forward + add', 5, 32, 3, False, CN.EPI_ADD),
             ('9x9 64->4 backward-data, PixelShuffle', 9, 64, 4, True, CN.EPI_SHUFFLE2),
             ('5x5 32->12 forward, PixelShuffle', 5, 32, 12, False, CN.EPI_SHUFFLE2)]
    for name, k, cin, cout, tr, epi in cases:
        if tr:
            ws = [torch.randn(cin, cout + 9, k, k, device='cuda', generator=gen) * 0.05 for _ in range(G)]
            scs = [CN.SmallConv(w, None, transpose=True, keep=cout) for w in ws]
            refw = [w[:, :cout].double() for w in ws]
        else:
            ws = [torch.randn(cout, cin, k, k, device='cuda', generator=gen) * 0.05 for _ in range(G)]
            bs = [torch.randn(cout, device='cuda', generator=gen) * 0.1 for _ in range(G)]
            scs = [CN.SmallConv(w, b) for w, b in zip(ws, bs)]
        x = torch.randn(G * n, cin, hw, hw, device='cuda', generator=gen) * (1e-4 if tr else 1.0)
        add = torch.randn(G * n, 3, hw, hw, device='cuda', generator=gen) * (1e-4 if tr else 1.0) if epi & CN.EPI_ADD else None
        sc = CN.stack_small(scs) if G > 1 else scs[0]
        group = (G, 0) if G > 1 else None

        def run(arith):
            CN.CONV_ARITH = arith
            return CN.conv_small(x, sc, n, hw, hw, epi=epi, add=add, add_c=3 if add is not None else 0, group=group)
        if os.environ.get('RISP_AB_STAMPS'):       # library built with EXTRA=-DRISP_TP_STAMPS: where a wave's life goes
            import ctypes as C
            from reconfigisp_amd import lib as L
            st = torch.zeros(512 * 4 * 6, dtype=torch.int64, device='cuda')
            y = torch.empty((G * n, cout, hw, hw), device='cuda')
            d = L.ConvDesc(N=n, H=hw, W=hw, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=16 | (epi & 8), add_c=0,
                           x=x.data_ptr(), wpack=sc.toep.data_ptr(), bias=None, cvals=st.data_ptr(), add=None, mask=None, y=y.data_ptr())
            CN._group_fields(d, n, group, sc.toep, None)
            L.call('risp_conv2d_toep', C.byref(d), None)
            torch.cuda.synchronize()
            t = st.view(-1, 6).double()
            t = t[t[:, 5] > 0]
            names = ('max+barrier A', 'split+write+barrier B', 'feed', 'matrix', 'epilogue', 'life')
            print('   stamps (%d waves): ' % t.shape[0] + ', '.join('%s %.1f%%' % (nm, 100 * t[:, i].sum().item() / t[:, 5].sum().item()) for i, nm in enumerate(names[:5]))
                  + ', life %.0f cycles' % t[:, 5].mean().item(), flush=True)
        ref = []
        for g in range(min(G, 2)):                 # float64 reference of the first members, image 0 and the last
            for i in (g * n, g * n + n - 1):
                xi = x[i:i + 1].double()
                r = TF.conv_transpose2d(xi, refw[g], padding=k // 2) if tr else TF.conv2d(xi, ws[g].double(), bs[g].double(), padding=k // 2)
                if add is not None:
                    r = r + add[i:i + 1].double()
                if epi & CN.EPI_SHUFFLE2:
                    r = TF.pixel_shuffle(r, 2)
                ref.append((i, r))
        line = '%-40s N=%d x %d members %dx%d:' % (name, n, G, hw, hw)
        for arith in ('f32', 'f16x2'):
            y = run(arith)
            m = max(r.abs().max().item() for _, r in ref)
            e = torch.cat([(y[i:i + 1].double() - r).flatten() for i, r in ref])
            us = timed(lambda: run(arith))
            line += '  %s %8.1f us rms %.2e max %.2e' % (arith, us, e.pow(2).mean().sqrt().item() / m, e.abs().max().item() / m)
        print(line, flush=True)


def border_case(v, L, P=4):
    return v if v < P else (2 * P - (L - 1 - v) if v >= L - P else P)


def first_layers():
    """the 9x9 first layers: risp_conv2d_toep_first against risp_conv2d_k3 (fp32 matrix pipe)"""
    n, hw, G = int(os.environ.get('RISP_AB_N', 32)), int(os.environ.get('RISP_AB_HW', 256)), int(os.environ.get('RISP_AB_G', 1))
    gen = torch.Generator('cuda').manual_seed(2)
    for name, cin, load in (('9x9 3->64 + case table + ReLU', 3, CN.LOAD_PLAIN), ('9x9 4->64 on the mosaic + ReLU', 4, CN.LOAD_UNSHUFFLE2)):
        ws = [torch.randn(64, cin, 9, 9, device='cuda', generator=gen) * 0.05 for _ in range(G)]
        bs = [torch.randn(64, device='cuda', generator=gen) * 0.1 for _ in range(G)]

        class M:
            pass
        pcs = []
        for w, b in zip(ws, bs):
            pcs.append(CN.PackedConv(w, b))
        pc = CN.stack_packed(pcs) if G > 1 else pcs[0]
        group = (G, CN.L.GROUP_SHARED_X) if G > 1 else None
        x = torch.rand((n, 1, 2 * hw, 2 * hw) if cin == 4 else (n, 3, hw, hw), device='cuda', generator=gen)
        table = torch.randn(G * n, 64 * 81, device='cuda', generator=gen) * 0.1 if cin == 3 else None
        epi = CN.EPI_RELU | (CN.EPI_CASEBIAS if cin == 3 else 0)

        def run(arith):
            CN.CONV_ARITH = arith
            return CN.conv(x, pc, n, hw, hw, load=load, epi=epi, cvals=table, group=group, infer=True)
        ref = []
        idx = torch.tensor([border_case(v, hw) for v in range(hw)], device='cuda')
        for g, i in ((0, 0), (G - 1, n - 1)):
            xi = x[i:i + 1].double()
            if cin == 4:
                xi = TF.pixel_unshuffle(xi, 2)
            r = TF.conv2d(xi, ws[g].double(), bs[g].double(), padding=4)
            if table is not None:
                t = table[g * n + i].view(64, 9, 9).double()
                r = r + t[:, idx][:, :, idx].unsqueeze(0)
            ref.append((g * n + i, torch.relu(r)))
        line = '%-40s N=%d x %d members %dx%d:' % (name, n, G, hw, hw)
        for arith in ('f32', 'f16x2'):
            y = run(arith)
            m = max(r.abs().max().item() for _, r in ref)
            e = torch.cat([(y[i:i + 1].double() - r).flatten() for i, r in ref])
            us = timed(lambda: run(arith))
            line += '  %s %8.1f us rms %.2e max %.2e' % (arith, us, e.pow(2).mean().sqrt().item() / m, e.abs().max().item() / m)
        print(line, flush=True)


if __name__ == '__main__':
    if os.environ.get('RISP_AB_FIRST', '1') != '0':
        first_layers()
    if os.environ.get('RISP_AB_SMALL', '1') != '0':
        main()
