#!/usr/bin/env python3
"""vvcgpu_if_batch on homogeneous lists (121 k calls of one shape / direction) and on two-shape lists, shuffled and sorted: the per-call floor of the
batch kernel without the real mix's heavy calls.  usage: python tools/if_shape_time.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops, shape_mix as sm
rng=np.random.default_rng(1)
def run(calls,label):
    fn,n,_=sm.build_interp(calls,rng)
    ms=sm.gpu_ms(fn,5)
    print('%-40s %7d calls %8d samples %.4f ms  %.2f ns/call'%(label,len(calls),n,ms,ms*1e6/len(calls)))
n=121000
def mk(w,h,taps,flags): 
    c=np.zeros((n,5),np.int64); c[:,0]=w;c[:,1]=h;c[:,2]=taps;c[:,3]=flags; return c
run(mk(4,4,8,2|4),'4x4 hor 8tap first+last')
run(mk(4,4,8,1|4),'4x4 ver 8tap last')
run(mk(4,11,8,2),'4x11 hor 8tap first')
run(mk(8,8,8,2),'8x8 hor')
run(mk(16,16,8,2)[:n//16],'16x16 hor (n/16 calls)')
a=mk(4,11,8,2); b=mk(4,4,8,1|4); c=np.concatenate([a[:n//2],b[:n//2]]); rng.shuffle(c,axis=0)
run(c,'4x11 hor / 4x4 ver shuffled')
c2=np.concatenate([a[:n//2],b[:n//2]])
run(c2,'4x11 hor then 4x4 ver (sorted)')
