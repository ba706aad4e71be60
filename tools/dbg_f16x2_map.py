import ctypes as C, os, subprocess, sys, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from reconfigisp_amd import lib as L, convnets as CN
so='/tmp/f16x2_dbg.so'
subprocess.check_call(['/opt/rocm/bin/hipcc','-O3','-std=c++17','-fPIC','--offload-arch=gfx950','-ffp-contract=off', '-fno-slp-vectorize','-I'+ROOT+'/include','-I'+ROOT+'/reconfigisp_amd/csrc','-x','hip','-shared','-o',so,ROOT+'/reconfigisp_amd/csrc/risp_conv_f16x2.hip',ROOT+'/reconfigisp_amd/csrc/risp_core.cpp','-DRISP_H2_DBG=4'])
l=C.CDLL(so); l.risp_conv2d_f16x2.restype=C.c_int; l.risp_conv2d_f16x2.argtypes=[C.c_void_p,C.c_void_p]
n,h,w,cin,cout=1,8,64,64,64
x=torch.rand(n,cin,h,w,device='cuda'); wt=torch.randn(cout,cin,3,3,device='cuda')*0.05
y=torch.full((n,cout,h,w),float('nan'),device='cuda')
ph=CN.f16x2_weights(wt,False)
inv=ph[:2].view(torch.float32).item()
d=L.ConvDesc(N=n,H=h,W=w,cin=cin,cout=cout,ksize=3,load_mode=0,cin_img=0,epilogue=16,add_c=0,x=x.data_ptr(),wpack=ph.data_ptr(),bias=None,cvals=None,add=None,mask=None,y=y.data_ptr())
print('status', l.risp_conv2d_f16x2(C.byref(d),None)); torch.cuda.synchronize()
code=(y/inv).round().long().cpu()
exp=torch.zeros_like(code)
for co in range(64):
    b,r=co//32,co%32; j,hl,i=r//8,(r%8)//4,r%4; e=4*j+i
    for row in range(8):
        wave,rb=row//2,row%2
        for col in range(64):
            Q,t=col//4,col%4
            lane=hl*32+rb*16+Q
            exp[0,co,row,col]=((t*2+b)*16+e)*64+lane
bad=(code!=exp)
print('mismatches', int(bad.sum()))
for idx in bad.nonzero()[:12].tolist():
    g=code[tuple(idx)].item(); ex=exp[tuple(idx)].item()
    dec=lambda v:(v//64//16//2, (v//64//16)%2, (v//64)%16, v%64)
    print(idx, 'got (t,b,e,lane)', dec(g), 'expected', dec(ex))
