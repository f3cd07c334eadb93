#!/usr/bin/env python3
"""vvcgpu_mc_batch (bi-prediction, luma) on a 4K picture tiled with 16x16 / 32x32 / 64x64 PUs: the packed fast path against the tile walker of the generic
kernel.  usage: python tools/mc_shape_time.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vvcsoftware_vtm_amd import ops
rng=np.random.default_rng(1)
W,H=3840+64,2160+64
ref=torch.from_numpy(rng.integers(0,1024,(H,W),dtype=np.int16)).cuda()
dst=torch.zeros((H,W),dtype=torch.int16,device='cuda')
for B in (16,32,64):
    ys,xs=np.meshgrid(np.arange(16,2160-B,B),np.arange(16,3840-B,B),indexing='ij')
    n=ys.size
    d=np.zeros(n,ops.MC_DESC)
    d['ref0_off']=d['ref1_off']=(ys*W+xs).ravel(); d['dst_off']=(ys*W+xs).ravel()
    d['ref0_stride']=d['ref1_stride']=d['dst_stride']=W; d['w']=d['h']=B
    d['frac_x0']=rng.integers(1,16,n); d['frac_y0']=rng.integers(1,16,n); d['frac_x1']=rng.integers(0,16,n); d['frac_y1']=rng.integers(0,16,n)
    d['is_luma']=1; d['bi']=1
    dd=ops.struct_to_device(d)
    fn=lambda: ops.mc_batch(ref,ref,dst,dd,n,10,(0,1023))
    fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): fn()
    b.record(); torch.cuda.synchronize()
    print('mc_batch bi luma %dx%d: %d PUs %.3f ms'%(B,B,n,a.elapsed_time(b)/5))
