"""What the in-GEMM row statistics cost a few-row launch (round 5): plain / + residual / + residual + statistics / + residual followed by the
row_stats kernel, cold operands.  Result (us): o_proj M655 28.4 / 30.6 / 36.1 / 32.6; down M655 52.3 / 52.2 / 57.7 / 54.4; o_proj M207 16.6 / 18.5 / 23.9 / 27.1;
SigLIP out M576 12.7 / 13.5 / 17.5 / 25.7; fc2 M576 22.2 / 22.3 / 25.8 / 29.2 - the ticket + fold tail is 4-5.5 us per launch, a separate kernel is
only cheaper at M = 655 (by 3.5 us) and dearer everywhere else: kept in the GEMM.    python tools/attic/stats_cost_probe.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from aki_amd import ops
dev="cuda"
g=torch.Generator(device=dev).manual_seed(0)
rnd=lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g)*sc).to(torch.bfloat16)
def timed(fn, NB=12, reps=4):
    for i in range(NB): fn(i)
    torch.cuda.synchronize()
    best=1e9
    for _ in range(3):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps*NB): fn(k%NB)
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/(reps*NB)*1e3)
    return best
for (M,N,K,ln,name) in ((655,3072,3072,False,"o_proj M655"),(655,3072,8192,False,"down M655"),(207,3072,3072,False,"o_proj M207"),(576,1152,1152,True,"siglip out M576"),(576,1152,4352,True,"siglip fc2 M576")):
    x=[rnd(M,K) for _ in range(12)]; w=[rnd(N,K,sc=0.02) for _ in range(12)]; r=[rnd(M,N) for _ in range(12)]
    y=torch.empty(M,N,device=dev,dtype=torch.bfloat16); st=ops.new_stats(M,dev,ln=ln)
    a=timed(lambda i: ops.linear(x[i],w[i],out=y))
    b=timed(lambda i: ops.linear(x[i],w[i],residual=r[i],out=y))
    c=timed(lambda i: ops.linear(x[i],w[i],residual=r[i],stats_out=st,stats_eps=1e-5,out=y))
    def sep(i):
        ops.linear(x[i],w[i],residual=r[i],out=y); ops.row_stats(y,1e-5,ln=ln)
    d=timed(sep)
    print(f"{name:18s} plain {a:6.1f}  +res {b:6.1f}  +res+stats {c:6.1f}  +res then row_stats kernel {d:6.1f} us")
