#!/usr/bin/env python
"""Per-kernel durations of one evaluation on non-uniform batches: events emitted by moving dots (few dots -> sharp IWE with hot pixels at the
evaluated motion, border pile-up from clipping) next to the uniform batches bench.py uses.  cfg3 / cfg4 / cfg5 shapes; HIP-event brackets of
cmax_set_profiling (4 launches per bracket)."""
import sys, numpy as np, torch
sys.path.insert(0,'.')
import event_based_optical_flow_amd as E
def run(tag, H, W, ev, model, cost, sigma, T, motion):
    h=E.CMaxHandle((H,W)).set_events(torch.from_numpy(ev).cuda(), time_bin=T)
    desc=E.make_descriptor(cost, model, sigma=sigma, time_bin=T)
    m=torch.as_tensor(np.ascontiguousarray(motion),device="cuda",dtype=torch.float32)
    for _ in range(10): h.evaluate(desc,m,True)
    torch.cuda.synchronize()
    import time
    t0=time.perf_counter()
    for _ in range(50): h.evaluate(desc,m,True)
    torch.cuda.synchronize(); wall=(time.perf_counter()-t0)/50*1e6
    h.set_profiling(True, 4)
    for _ in range(20): h.evaluate(desc,m,True)
    torch.cuda.synchronize()
    p=h.read_profile(); h.set_profiling(False)
    print(tag, "eval %.1f us"%wall, {k:round(v[0]/max(v[1],1)*1e3,2) for k,v in p.items() if v[1]}, flush=True)
vel=(12.3,-7.7)
for name,H,W,n,model,cost,sigma,T in (("cfg3",480,640,5_000_000,"dense-flow","gradient_magnitude",0.0,0),
                                      ("cfg4",260,346,2_000_000,"dense-flow-voxel","image_variance",1.0,10),
                                      ("cfg5",720,1280,2_500_000,"dense-flow","image_variance",0.0,0)):
    f0=-(E.utils.generate_smooth_flow((H,W),3.0,seed=3)+np.array(vel)[:,None,None])
    motion=f0 if T==0 else np.stack([f0]*T)
    evu=E.utils.generate_events(n,H,W,tmin=0,tmax=0.05,seed=46)
    run(name+" uniform   ",H,W,evu,model,cost,sigma,T,motion)
    for nd in (n//400, n//20):
        evs=E.utils.generate_structured_events(n,H,W,vel,n_dots=nd,tmin=0,tmax=0.05,seed=46)
        run(name+f" dots={nd:7d}",H,W,evs,model,cost,sigma,T,motion)
