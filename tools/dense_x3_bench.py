import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from sst_amd import dense as D
dev='cuda:0'
m=90107
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for k,n in ((128,128),(128,256),(256,128)):
    x=torch.randn(m,k,device=dev); w=torch.randn(n,k,device=dev)*0.1; b=torch.randn(n,device=dev); aux=torch.randn(m,n,device=dev)
    res={}
    for mode in ('f32','f32x3'):
        D.set_matmul_mode(mode)
        res[mode]=[timed(lambda: D.lds_linear(x,w,b,D.EPI_BIAS)), timed(lambda: D.lds_linear(x,w,b,D.EPI_GELU,want_pre=True)), timed(lambda: D.lds_linear(x,w,b,D.EPI_MUL_GELU_GRAD,aux_in=aux))]
        if n==128:
            lw=torch.randn(128,device=dev); r=torch.randn(m,128,device=dev)
            res[mode].append(timed(lambda: D.lds_linear_add_ln(x,w,b,r,lw,lw,1e-5)))
    by=(k+n)*4*m
    print(f'K={k} N={n}: bytes {by/1e6:.0f} MB -> {by/8e12*1e6:.1f} us at 8 TB/s | f32 {[round(v,1) for v in res["f32"]]} | f32x3 {[round(v,1) for v in res["f32x3"]]}')
D.set_matmul_mode('f32')
