"""Host cost of one cfg2 evaluation: how long the host needs to ENQUEUE it (CMaxHandle.evaluate, the raw ctypes call with fixed
pointers) against how long the GPU needs to drain it -- whether a loop of evaluations is host- or GPU-bound (profiles/r02_ablation.txt).
   python tools/probe_host_cost.py        (GPU box)"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
import event_based_optical_flow_amd as E
size, n = (260, 346), 1_000_000
ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
h = E.CMaxHandle(size).set_events(ev)
desc = E.make_descriptor("image_variance", "2d-translation")
theta = torch.tensor([12.3, -7.7], device="cuda", dtype=torch.float32)
for _ in range(50): h.evaluate(desc, theta)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(400): h.evaluate(desc, theta)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {(t1-t0)/400*1e6:.2f} us per evaluation, until drained {(t2-t0)/400*1e6:.2f} us per evaluation")
# a tiny batch: the GPU side is ~3 launch floors, what is left is the host
ev2 = ev[:2000]
h2 = E.CMaxHandle(size).set_events(ev2)
for _ in range(50): h2.evaluate(desc, theta)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): h2.evaluate(desc, theta)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"2000-event batch: enqueue {(t1-t0)/2000*1e6:.2f} us, drained {(t2-t0)/2000*1e6:.2f} us per evaluation")
import ctypes
from event_based_optical_flow_amd import functional as F
lib = h._lib
res = torch.empty(8, dtype=torch.float64, device="cuda"); grad = torch.empty(2, dtype=torch.float64, device="cuda")
mp, rp, gp, st = theta.data_ptr(), res.data_ptr(), grad.data_ptr(), F._stream()
dref = ctypes.byref(desc)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): lib.cmax_objective(h._h, dref, mp, rp, gp, st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"raw ctypes call, fixed pointers: enqueue {(t1-t0)/400*1e6:.2f} us, drained {(t2-t0)/400*1e6:.2f} us per evaluation")
t0 = time.perf_counter()
for _ in range(2000): F._stream()
print(f"F._stream(): {(time.perf_counter()-t0)/2000*1e6:.2f} us")
t0 = time.perf_counter()
for _ in range(2000): torch.empty(8, dtype=torch.float64, device="cuda")
print(f"torch.empty: {(time.perf_counter()-t0)/2000*1e6:.2f} us")
t0 = time.perf_counter()
for _ in range(2000): h._motion32(theta)
print(f"_motion32: {(time.perf_counter()-t0)/2000*1e6:.2f} us")
