#!/usr/bin/env python3
"""Where the interpolation calls of the real shape mix (tests/golden/trace_*.npz) spend their time: vvcgpu_if_batch on sub-sets of the mix
(call count of every sub-set as in the full 8.4 M-sample batch).  usage: python tools/if_mix_parts.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops, shape_mix as sm
hist,_=sm.load_trace()
sig=sm.signatures(hist,'interp',lambda w,h,a,b,c: 2<=w<=256 and h<=256)
def run(sel,label):
    rng=np.random.default_rng(3)
    s=sig[sel]
    if not len(s): print(label,'none'); return
    # keep the call COUNT proportional to the full mix: scale so that the full mix would have 8.4M samples
    tot=float((sig[:,0]*sig[:,1]*sig[:,5]).sum()); part=float((s[:,0]*s[:,1]*s[:,5]).sum())
    calls=sm.draw(s,int(8388608*part/tot),rng)
    fn,n,_=sm.build_interp(calls,rng)
    ms=sm.gpu_ms(fn,5)
    print('%-34s %7d calls %9d samples %.4f ms'%(label,len(calls),n,ms))
w,h=sig[:,0],sig[:,1]
run(np.ones(len(sig),bool),'all')
run(w%4!=0,'w % 4 != 0')
run((w%4!=0)&(w<4),'  of these: w < 4 (partial units)')
run((w%4!=0)&(w>=4)&(w*h<=512),'  w >= 5 odd, <= 512 samples')
run((w%4!=0)&(w>=4)&(w*h>512),'  w >= 5 odd, > 512 samples (heavy)')
run((w%4==0)&(w*h<=256),'w%4==0, <= 256 samples')
run((w%4==0)&(w*h>256)&(w*h<=512),'257..512 samples')
run((w%4==0)&(w*h>512),'> 512 samples (heavy)')
run(sig[:,2]==8,'taps 8'); run(sig[:,2]==4,'taps 4'); run(sig[:,2]==2,'taps 2')
run((sig[:,3]&1)==1,'vertical'); run((sig[:,3]&1)==0,'horizontal')
