#!/usr/bin/env python3
"""How often does a FAR value move?  (build container or GPU box: CPU only, on the oracle.)

A pass of the packed table form is flagged while it holds an entry beyond the 8-level codes (8 or more stamps behind its
subject); the coded merge cannot carry such values, so flagged passes used to run the 64-step chain.  This counts, on
BASELINE configs[4] (128 UE / 64 res, mobility_vary, velocities redrawn every 25 slots), over slots 150 .. T: the share of
8-column passes that hold a far entry when the slot begins, and of those the share in which some entry TAKES a value that is
itself beyond the codes (Vehicle.received_update copying a stale stamp, vehicle.py:35-47) - the only thing the coded
merge would get wrong.  csrc/step_wide.hpp `wide_far_guard` proves the absence of that per slot and pass.

  python profiles/far_propagation.py [envs] [slots]
"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle.oracle import Oracle, SQ_IEEE
from diral_amd.config import bench_config
N,A,L=128,64,4000.0
B=int(sys.argv[1]) if len(sys.argv)>1 else 48
T=int(sys.argv[2]) if len(sys.argv)>2 else 500
cfg=bench_config(N,A,L,mobility_vary=True)
rng=np.random.default_rng(1234)
o=Oracle(cfg,batch=B,sq_mode=SQ_IEEE,threads=16)
o.reset(rng.integers(0,int(L),size=(B,N)).astype(float),np.zeros((B,N)),np.full((B,N),1.7))
flag_tot=0; prop_tot=0; pass_tot=0
hist=[]
for t in range(T):
    a=rng.integers(0,A,size=(B,N)).astype(np.int32)
    e0=o.export()
    o.step(0,a,t)
    if t%25==24: o.update_velocity(rng.integers(1,4,size=(B,N)).astype(np.uint8))
    if t<150: continue
    e1=o.export()
    seq0=e0['seq']; seq1=e1['seq']      # [B, viewer, subject]
    own1=np.einsum('bkk->bk',seq1)      # subject's own seq after the step (stamped)
    lag1=own1[:,None,:]-seq1            # lag after merge
    own0=np.einsum('bkk->bk',seq0)
    lag0s=(own0[:,None,:]+1)-seq0       # lag after stamp, before merge
    far_before=(lag0s>=8)               # code 0 after the stamp (incl never heard)
    changed_far=(seq1!=seq0)&(lag1>=8)  # took a value that is itself beyond the codes
    flagged=(lag1>=7)                   # keeps the quad flagged for the next slot (approx)
    # passes of 8 subject columns
    fp=far_before.reshape(B,N,N//8,8).any(axis=(1,3))     # pass holds a far entry at start
    cp=changed_far.reshape(B,N,N//8,8).any(axis=(1,3))
    flag_tot+=fp.sum(); prop_tot+=(cp&fp).sum(); pass_tot+=fp.size
    hist.append((fp.sum(axis=1)>0).mean())
print("passes with a far entry: %.3f of all; of those, far value propagated: %.4f"%(flag_tot/pass_tot, prop_tot/max(flag_tot,1)))
print("share of envs with any flagged pass (mean over slots): %.3f"%np.mean(hist))
