#!/usr/bin/env python3
"""HBM ceilings of the box with torch ops: copy (read + write) and read-only reduction, the practical limits the
roofline fractions in DESIGN.md are read against."""
import torch, time
dev=torch.device('cuda')
n=(1<<32)//8
a=torch.empty(n,dtype=torch.float64,device=dev).normal_()
b=torch.empty_like(a)
def t(f,reps=5):
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e-3
tc=t(lambda: b.copy_(a)); print('copy  GB/s', 2*n*8/tc/1e9)
ts=t(lambda: a.sum()); print('sum   GB/s', n*8/ts/1e9)
tf=t(lambda: b.fill_(1.0)); print('fill  GB/s', n*8/tf/1e9)
ta=t(lambda: torch.add(a,b,out=b)); print('add   GB/s', 3*n*8/ta/1e9)
af=a.view(torch.float32)
ts=t(lambda: af.sum()); print('sum32 GB/s', n*8/ts/1e9)
