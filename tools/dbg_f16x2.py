import ctypes as C, os, subprocess, sys, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from reconfigisp_amd import lib as L, convnets as CN
so='/tmp/f16x2_dbg.so'
subprocess.check_call(['/opt/rocm/bin/hipcc','-O3','-std=c++17','-fPIC','--offload-arch=gfx950','-ffp-contract=off', '-fno-slp-vectorize','-I'+ROOT+'/include','-I'+ROOT+'/reconfigisp_amd/csrc','-x','hip','-shared','-o',so,ROOT+'/reconfigisp_amd/csrc/risp_conv_f16x2.hip',ROOT+'/reconfigisp_amd/csrc/risp_core.cpp']+sys.argv[1:])
l=C.CDLL(so); l.risp_conv2d_f16x2.restype=C.c_int; l.risp_conv2d_f16x2.argtypes=[C.c_void_p,C.c_void_p]
torch.manual_seed(0)
n,h,w,cin,cout=2,16,64,64,64
x=torch.rand(n,cin,h,w,device='cuda'); wt=torch.randn(cout,cin,3,3,device='cuda')*0.05; b=torch.randn(cout,device='cuda')*0.01
y=torch.full((n,cout,h,w),float('nan'),device='cuda')
ph=CN.f16x2_weights(wt,False)
d=L.ConvDesc(N=n,H=h,W=w,cin=cin,cout=cout,ksize=3,load_mode=0,cin_img=0,epilogue=0,add_c=0,x=x.data_ptr(),wpack=ph.data_ptr(),bias=b.data_ptr(),cvals=None,add=None,mask=None,y=y.data_ptr())
print('status', l.risp_conv2d_f16x2(C.byref(d),None)); torch.cuda.synchronize()
ref=torch.nn.functional.conv2d(x.double(),wt.double(),b.double(),padding=1)
err=(y.double()-ref).abs()
print('max err', err.max().item(), 'nan', torch.isnan(y).sum().item())
bad=(err>1e-3)
print('bad fraction', bad.float().mean().item())
for dim,name in enumerate(['n','cout','row','col']):
    other=[i for i in range(4) if i!=dim]
    print(name, [round(v,2) for v in bad.float().mean(dim=other).tolist()])
r=(y.double()/ref)[bad]
print('ratio y/ref on bad: median', r.median().item() if r.numel() else None)
# which (chunk, tap) terms are present in the bad outputs?  least squares of y - bias on the 36 partial convolutions
terms=[]
for ch in range(4):
    for ky in range(3):
        for kx in range(3):
            wz=torch.zeros_like(wt); wz[:,16*ch:16*ch+16,ky,kx]=wt[:,16*ch:16*ch+16,ky,kx]
            terms.append(torch.nn.functional.conv2d(x.double(),wz.double(),None,padding=1))
T=torch.stack(terms,-1)                      # (n,cout,h,w,36)
A=T[bad]; yy=(y.double()-b.double()[None,:,None,None])[bad]
# per-element: cannot solve 36 unknowns from 1 equation; instead assume alpha shared over all bad elements
sol=torch.linalg.lstsq(A,yy[:,None]).solution[:,0]
print('alpha per (chunk, ky, kx) over bad outputs:'); print(sol.view(4,3,3))
good=~bad
sol2=torch.linalg.lstsq(T[good][:20000],(y.double()-b.double()[None,:,None,None])[good][:20000,None]).solution[:,0]
print('alpha over good outputs:'); print(sol2.view(4,3,3))
idx=bad.nonzero()[:8]; print(idx.tolist()); print([ (y[tuple(i)].item(), ref[tuple(i)].item()) for i in idx])
