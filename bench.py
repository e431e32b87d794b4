#!/usr/bin/env python
"""Headline benchmark: events/s through ONE objective evaluation = warp + IWE accumulate + cost +
analytic gradient (BASELINE.json metric), on synthetic events already resident in HBM.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                            torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE configs[1] = cfg2 = 1M synthetic events, 346x260 (H=260, W=346), 2-DoF
translational flow, image-variance cost + analytic gradient.  `value` times the COMPLETE evaluation: cmax_objective = K1
(warp + vote) + K3 (gather + gradient) + the one-wave finishing kernel, loss and gradient left on the device
(config.result_form; round 3 timed the raw form -- VERDICT r3 / ADVICE r3).  `also` carries the other forms (cfg2_raw =
cmax_objective_raw: K1 + K3, 32 x 6 partial sums folded by the consumer; cfg2_host_result = cmax_objective_host, every
evaluation delivered to the host before the next starts -- their fractions are repeated in roofline.frac_raw_form /
frac_host_result), the headline with blur, on a sharp image and at a motion of 80 px (cfg2_theta80), the other single-GPU
configurations (cfg3, cfg4, a cfg5 shard) -- cfg3 and the cfg5 shard also with the per-pixel random flow of
src/utils/flow_utils.py:20-30 (cfg3_rough, cfg5_rough: the unfriendly case for coalesced gathers and LDS windows) -- cfg5 AS
BASELINE STATES IT on one GPU (cfg5_strong: 20M events, the N = 1 point of its strong-scaling curve) and `hbm`: 64M events,
a packed stream larger than the 256 MiB Infinity Cache.  roofline.launch_floor_us = two dependent EMPTY launches with the
evaluation's grids, measured in the same run.  For N > 1 every rank owns one 1M-event time
slice of an N x 1M-event batch (weak scaling); both all-reduces of an evaluation (IWE, gradient) are enqueued by
libcmax_hip.so itself (RCCL, cmax_objective_dist), and `also.cfg5_strong` is cfg5 split into N time slices.

Timing: W warm-up steps, then `windows` (default 25) windows of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides, MAX over ranks per window; ms_per_step / value are the MEDIAN window.

One JSON line on stdout (rank 0).  Extra objects:
  roofline      bound hbm; achieved = SURVEY 8(d) algorithmic bytes of ONE EVALUATION / median step time
                (frac = achieved / 8 TB/s); `kernels` = the same per kernel class (bytes per launch / mean
                launch duration, HIP events on the launch stream in an instrumented pass of the same steps,
                cmax_set_profiling), `dominant` names the longest one
  cpu_baseline  the CPU oracle (oracle/cmax_oracle.c, scalar C, 1 core) timed on this host on the
                same workload -- a reported baseline, not the target; cpu_baseline_torch: the same evaluation
                written the way the reference is (torch tensor ops + autograd, oracle/torch_cpu.py), each row of a
                small thread sweep in a FRESH process on every CPU the launcher allowed; CPU model and core
                counts are printed with both
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (H, W, n_events per GPU, model, cost, sigma)
    "cfg2": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0,
                 desc="cfg2: 1M synthetic events, 346x260, 2-DoF translational flow, image_variance + analytic grad"),
    "cfg3": dict(H=480, W=640, n=5_000_000, model="dense-flow", cost="gradient_magnitude", sigma=0.0,
                 desc="cfg3: 5M synthetic events, 640x480, dense per-pixel flow, gradient_magnitude + analytic grad"),
    "cfg4": dict(H=260, W=346, n=2_000_000, model="dense-flow-voxel", cost="image_variance", sigma=1.0,
                 desc="cfg4: 2M synthetic events, 346x260, Burgers voxel T=10 (t0 middle) + voxel warp, image_variance"),
    "cfg5": dict(H=720, W=1280, n=2_500_000, model="dense-flow", cost="image_variance", sigma=0.0,
                 desc="cfg5 shard: 20M/8 = 2.5M events per GPU, 1280x720, dense flow, image_variance"),
    # cfg5 as BASELINE states it: ONE 20M-event batch, time-sliced over the N ranks (strong scaling)
    "cfg5_strong": dict(H=720, W=1280, n=20_000_000, model="dense-flow", cost="image_variance", sigma=0.0, strong=True,
                        desc="cfg5: 20M synthetic events, 1280x720, dense flow, image_variance, time-sliced over the ranks"),
    # out of the 256 MiB Infinity Cache: the packed stream alone is 512 MB and is read twice per evaluation (K1, K3), so HBM
    # -- not the cache -- feeds the kernels.  Events are drawn on the device (torch.Generator): 2 GB of fp64 [n, 4] never cross PCIe.
    "hbm": dict(H=720, W=1280, n=64_000_000, model="dense-flow", cost="image_variance", sigma=0.0, device_gen=True,
                desc="hbm: 64M synthetic events, 1280x720, dense flow, image_variance (512 MB packed stream: larger than the Infinity Cache)"),
    # SURVEY 8(d) second rows of the headline: the YAMLs' blur, and a sharp image (2500 dots warped with their true motion --
    # what an optimiser converges to; the friendliest case for LDS / L2 atomics is the uniform stream above)
    # the headline's other two forms (see run_workload): results left ON THE DEVICE by a finishing kernel (cmax_objective), and
    # results delivered TO THE HOST after every evaluation (cmax_objective_host: what a sequential optimiser sees)
    "cfg2_raw": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, form="raw",
                     desc="cfg2 through cmax_objective_raw: K1 + K3, 32 x 6 partial sums left on the device and folded by the consumer"),
    "cfg2_batch8": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, form="batch", batch=8,
                        desc="cfg2 through cmax_objective_batch: 8 candidate thetas per call, one launch of each kernel (blockIdx.z = candidate); "
                             "ms_per_step and events/s are PER EVALUATION (8 per call)"),
    "cfg2_batch32": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, form="batch", batch=32,
                         desc="cfg2 through cmax_objective_batch: 32 candidate thetas per call -- the chunk the solver's grid initialisers use "
                              "(solver/translation_search.py); ms_per_step and events/s are PER EVALUATION (32 per call)"),
    "cfg2_theta80": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, theta=(80.0, -50.0), slabs=4,
                         desc="cfg2 at a large motion: theta = (80, -50) px over the batch, events in 4 time slabs (cmax_set_time_slabs: "
                              "windows of the source tiles' size + 20 x 13 px)"),
    "cfg2_theta150": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, theta=(150.0, -100.0), slabs=4,
                          desc="cfg2 at the search range of configs/*.yaml: theta = (150, -100) px over the batch, events in 4 time slabs"),
    # "dense F ~ U(-5, 5) per pixel *and* a smooth field" -- the per-pixel random flow of src/utils/flow_utils.py:20-30
    "cfg3_rough": dict(H=480, W=640, n=5_000_000, model="dense-flow", cost="gradient_magnitude", sigma=0.0, rough=5.0,
                       desc="cfg3 with a per-pixel random flow F ~ U(-5, 5): 5M events, 640x480, dense flow, gradient_magnitude"),
    "cfg5_rough": dict(H=720, W=1280, n=2_500_000, model="dense-flow", cost="image_variance", sigma=0.0, rough=5.0,
                       desc="cfg5 shard with a per-pixel random flow F ~ U(-5, 5): 2.5M events, 1280x720, dense flow, image_variance"),
    "cfg2_host_result": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, form="host",
                             desc="cfg2 through cmax_objective_host: every evaluation returns loss + gradient to the host before the next starts"),
    "cfg2_sigma1": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=1.0,
                        desc="cfg2 with the YAMLs' blur: 1M uniform events, 346x260, 2-DoF, image_variance, sigma 1"),
    "cfg2_structured": dict(H=260, W=346, n=1_000_000, model="2d-translation", cost="image_variance", sigma=0.0, structured=True,
                            desc="cfg2 on a sharp image: 1M events of 2500 dots moving with theta, 346x260, 2-DoF, image_variance"),
}
KERNEL_NAMES = {"vote": "k_vote (K1 warp + bilinear vote)", "grad": "k_grad (K3 gather + gradient)", "stats": "k_stats* (K2 image statistics)",
                "gimage": "k_gimage* (K2b dL/dIWE)", "finish": "k_finish* (final reduction)", "comm": "RCCL all-reduce"}


def algorithmic_bytes(kernel: str, n: int, H: int, W: int, model: str, T: int) -> float:
    """per evaluation B = 24 N + 16 HW + B_model, split per kernel as
    K1 vote : 12 N (x, y, t fp32) + 4 HW (IWE write) + 8 HW T' (flow read, dense / voxel)
    K3 grad : 12 N               + 4 HW (G read)     + 8 HW T' (flow-gradient write)
    K2 stats: 4 HW (IWE read);  K2b gimage: 4 HW (G write)                                  """
    tp = 0 if model == "2d-translation" else (T if model == "dense-flow-voxel" else 1)
    if kernel in ("vote", "grad"):
        return 12.0 * n + 4.0 * H * W + 8.0 * H * W * tp
    if kernel in ("stats", "gimage"):
        return 4.0 * H * W
    return 0.0


def evaluation_bytes(n: int, H: int, W: int, model: str, T: int) -> float:
    return sum(algorithmic_bytes(k, n, H, W, model, T) for k in ("vote", "stats", "gimage", "grad"))


def measured_traffic(workload: str, kernel: str):
    """HBM bytes per launch from the PMC counters (profiles/r*_pmc_<workload>.json, collected with
    tools/prof_pmc.sh and corrected as MI355X_MICROARCH.md prescribes); None if not collected."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)), reverse=True):
        try:
            return json.load(open(path))["traffic_bytes_per_launch"][workload][kernel]
        except (KeyError, ValueError, OSError):
            continue
    return None


def traffic_source(workload: str):
    """Where `traffic` comes from: the committed counter summary of an EARLIER run of the same workload, not this run."""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)), reverse=True)
    if not paths:
        return None
    return ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this workload, tools/prof_pmc.sh; "
            "replayed, not measured in this run)" % os.path.relpath(paths[0], ROOT))


PMC_CLASS = {"k_vote": "vote", "k_stats": "stats", "k_gimage": "gimage", "k_grad": "grad", "k_finish": "finish", "k_finish_deferred": "finish",
             "k_finish_raw": "finish", "k_finish_lines": "finish", "k_stats_gimage_gm": "stats", "k_blur_stats_gimage_gm": "stats",
             "k_blur_stats_var": "stats", "k_gimage_blur_adj_var": "gimage", "k_blur_stats_adj_var": "stats"}


def pmc_traffic_this_run(workload: str, timeout_s: float = 150.0):
    """HBM-side bytes per launch of the workload's kernels measured ON THIS BOX, NOW: this script re-executed twice
    under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` -- separate passes, no tracing option beside --pmc, as
    MI355X_MICROARCH.md's HBM section prescribes -- with a short run of the same workload; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB
    (gfx950 tallies the 128-byte requests of a wide coalesced read at 64: FETCH_SIZE doubled; WRITE_SIZE as reported).  Returns
    ({class: bytes per launch}, description) or (None, reason).  Never raises: the replayed summary under profiles/ is the fall-back."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    raw = {}
    tmp = tempfile.mkdtemp(prefix="cmax_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", out_dir, "-o", "c", "--", sys.executable, os.path.abspath(__file__),
                   "--no-cpu-baseline", "--no-also", "--no-pmc", "--windows", "2", "--steps", "20", "--warmup", "3", "--ramp", "0",
                   "--workload", workload, "--verbose-out", os.path.join(tmp, "line.json")]
            try:
                p = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s timed out after %.0f s" % (ctr, timeout_s)
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s: rc %d, %d counter file(s): %s" % (ctr, p.returncode, len(files), (p.stderr or "")[-160:].replace("\n", " "))
            per = collections.defaultdict(list)
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") == ctr:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
            for kname, vals in per.items():
                m = re.search(r"cmax::(?:[tbm]\d+::)?(k_\w+)", kname)
                if m and len(vals) >= 10:
                    raw.setdefault(m.group(1), {})[ctr] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = {}
    for kname, v in raw.items():
        cls = PMC_CLASS.get(kname)
        if cls and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic[cls] = traffic.get(cls, 0) + int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
    if not traffic:
        return None, "no counter rows for the workload's kernels"
    return traffic, ("this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes of `bench.py --workload %s --steps 20 "
                     "--windows 2` on this box, no tracing); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, mean over launches" % workload)


def make_inputs(cfg, rank, world, seed=46, structured=False):
    """This rank's time slice [rank, rank + 1) * period / world of the batch (host fp64 [n, 4]) and the motion."""
    import event_based_optical_flow_amd as E

    H, W = cfg["H"], cfg["W"]
    n = cfg["n"] // world if cfg.get("strong") else cfg["n"]
    period = 0.05
    t0, t1 = rank * period / world, (rank + 1) * period / world
    if (structured or cfg.get("structured")) and cfg["model"] == "2d-translation":
        ev = E.utils.generate_structured_events(n, H, W, (12.3, -7.7), n_dots=2500, tmin=t0, tmax=t1, seed=seed + rank)
    elif cfg.get("device_gen"):
        import torch

        g = torch.Generator(device="cuda")
        g.manual_seed(seed + rank)
        ev = torch.empty((n, 4), dtype=torch.float64, device="cuda")
        ev[:, 0] = torch.randint(0, H, (n,), generator=g, device="cuda")
        ev[:, 1] = torch.randint(0, W, (n,), generator=g, device="cuda")
        ev[:, 2] = torch.sort(torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * (t1 - t0) + t0).values
        ev[:, 3] = torch.randint(0, 2, (n,), generator=g, device="cuda")
    else:
        ev = E.utils.generate_events(n, H, W, tmin=t0, tmax=t1, seed=seed + rank)
    T = 0
    if cfg["model"] == "2d-translation":
        motion = np.array(cfg.get("theta", (12.3, -7.7)), dtype=np.float64)
    elif cfg["model"] == "dense-flow" and cfg.get("rough"):
        motion = E.utils.generate_dense_optical_flow((H, W), cfg["rough"], seed=seed + 1000)  # per pixel, src/utils/flow_utils.py:20-30
    elif cfg["model"] == "dense-flow":
        motion = E.utils.generate_smooth_flow((H, W), 20, seed=seed + 1000)
    else:
        T = 10
        motion = None  # built on the GPU from the t0 flow
    return ev, motion, T


def host_cpu_info():
    """CPU model, sockets, physical cores and logical CPUs of this host (/proc/cpuinfo), for the CPU rows of the line."""
    info = {"model": None, "sockets": None, "physical_cores": None, "logical_cpus": os.cpu_count()}
    try:
        cores, sockets, phys, core = set(), set(), None, None
        for ln in open("/proc/cpuinfo"):
            key, _, val = ln.partition(":")
            key, val = key.strip(), val.strip()
            if key == "model name" and info["model"] is None:
                info["model"] = val
            elif key == "physical id":
                phys = val
                sockets.add(val)
            elif key == "core id":
                core = val
            elif not key and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        info["sockets"] = len(sockets) or None
        info["physical_cores"] = len(cores) or None
    except OSError:
        pass
    return info


def cpu_baseline(cfg, ev, motion, budget_s=12.0):
    """Oracle (fp64 scalar C port of the reference path) on this host, same inputs."""
    from oracle import oracle as orc

    size = (cfg["H"], cfg["W"])
    orc.objective(ev[:1000], motion, cfg["model"], size, cost=cfg["cost"], sigma=cfg["sigma"])  # warm the .so
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.objective(ev, motion, cfg["model"], size, cost=cfg["cost"], sigma=cfg["sigma"])
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 5000:
            break
    cpu = host_cpu_info()
    return {"value": ev.shape[0] * reps / el, "unit": "events/s", "cores": 1, "kind": "port",
            "sample": f"{reps} full evaluations (value+gradient) of the same {ev.shape[0]}-event workload, "
                      f"oracle/cmax_oracle.c fp64 scalar C, {el:.1f} s",
            "host_cpus": os.cpu_count(), "cpu_model": cpu["model"], "host": cpu}


def _cpu_torch_worker(workload: str, threads: int, budget_s: float):
    """Child process of cpu_baseline_torch: a FRESH interpreter (its own OpenMP pool, sized by OMP_NUM_THREADS before torch is
    imported) on every CPU the launcher allowed -- not the GPU process, which is pinned to one NUMA node and whose thread
    pool was sized before the pinning (round 2's driver line: 128 threads squeezed after the fact, 7 s per evaluation)."""
    import torch

    from oracle import torch_cpu

    torch.set_num_threads(threads)
    cfg = WORKLOADS[workload]
    ev, motion, _ = make_inputs(cfg, 0, 1)
    size = (cfg["H"], cfg["W"])
    torch_cpu.value_and_grad(ev, motion, cfg["model"], size, cfg["cost"])  # warm-up: thread pool, allocator
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        torch_cpu.value_and_grad(ev, motion, cfg["model"], size, cfg["cost"])
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s or len(times) >= 200:
            break
    print(json.dumps({"n": int(ev.shape[0]), "reps": len(times), "total_s": float(sum(times)), "median_s": float(np.median(times)),
                      "threads": torch.get_num_threads(), "affinity": len(os.sched_getaffinity(0))}), flush=True)


def cpu_baseline_torch(workload, cfg, affinity, budget_s=4.0):
    """The reference's own kind of CPU code (tensor ops + torch.autograd, oracle/torch_cpu.py) on this host's cores: one row
    with one thread per physical core the launcher allowed, one row with ONE thread, two in between.  Each row
    runs in its own fresh interpreter (see _cpu_torch_worker).  `affinity`: the CPU set this process had BEFORE it pinned itself."""
    if cfg["sigma"] > 0 or cfg["model"] not in ("2d-translation", "dense-flow") or cfg["cost"] not in ("image_variance", "gradient_magnitude"):
        return None
    cpu = host_cpu_info()
    n_aff = len(affinity)
    smt = max(1, (cpu["logical_cpus"] or n_aff) // (cpu["physical_cores"] or n_aff))
    n_threads = max(1, n_aff // smt)  # one thread per physical core: SMT siblings only contend for these memory-bound ops
    # the scatter-add / gather ops of this path do not scale with threads (on the 2-socket GPU hosts one thread beats 128), so a
    # small sweep is reported and the BEST row is the headline of this object
    sweep = sorted({1, min(8, n_threads), min(32, n_threads), n_threads})
    rows = {}
    for threads in sweep:
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-torch-worker", workload, str(threads), str(budget_s)]
        try:
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=budget_s * 6 + 120,
                                 preexec_fn=lambda: os.sched_setaffinity(0, affinity))
            r = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # never fail the bench line over a baseline
            rows[str(threads)] = {"error": f"{type(e).__name__}: {e}"[:300]}
            continue
        rows[str(threads)] = {"value": r["n"] / r["median_s"], "unit": "events/s", "cores": r["threads"], "kind": "port",
                              "sample": f"{r['reps']} full evaluations (value + autograd gradient) of the same {r['n']}-event workload, torch-CPU "
                                        f"fp64 restatement of the reference's tensor path (oracle/torch_cpu.py), median of the evaluations, "
                                        f"{r['total_s']:.1f} s, fresh process on {r['affinity']} CPUs"}
    ok = {k: v for k, v in rows.items() if "value" in v}
    if not ok:
        return {"error": next(iter(rows.values())).get("error", "no result"), "host": cpu}
    best = max(ok, key=lambda k: ok[k]["value"])
    out = dict(ok[best])
    out["threads_sweep"] = {k: (v.get("value") or v.get("error")) for k, v in rows.items()}
    out["one_thread"] = rows.get("1")
    out["all_cores"] = rows.get(str(n_threads))
    out["host_cpus"] = os.cpu_count()
    out["cpu_model"] = cpu["model"]
    out["host"] = cpu
    return out


def run_workload(name, args, rank, world, dev, steps, warmup, windows, profile=True, keep_inputs=False):
    """Sets the workload up on this rank, times `windows` windows of `steps` evaluations, optionally profiles the kernel
    classes.  Returns a dict (rank 0 uses it; every rank must call this: the timing holds collectives)."""
    import torch
    import torch.distributed as dist

    import event_based_optical_flow_amd as E
    from event_based_optical_flow_amd.distributed import TimeSlicedObjective

    cfg = WORKLOADS[name]
    H, W = cfg["H"], cfg["W"]
    ev, motion, T = make_inputs(cfg, rank, world, structured=args.events == "structured")
    n_local = ev.shape[0]
    handle = E.CMaxHandle((H, W))
    if args.deterministic:
        handle.set_deterministic(True)
    sliced = TimeSlicedObjective(handle, in_library=not args.torch_collectives)
    ev_dev = ev.to(dev) if isinstance(ev, torch.Tensor) else torch.from_numpy(ev).to(dev)  # fp64 [n,4] resident in HBM before anything is timed
    sliced.set_local_events(ev_dev, time_bin=T, device=dev)  # first call: workspace allocation, code-object load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sliced.set_local_events(ev_dev, time_bin=T, device=dev)
    if cfg.get("slabs"):
        handle.set_time_slabs(cfg["slabs"])  # large motions: slab-major order (a second pass of the counting sort)
    torch.cuda.synchronize()
    prepare_ms = (time.perf_counter() - t0) * 1e3  # once per batch: pack + counting sort + work list (not in `value`)
    del ev_dev

    if cfg["model"] == "dense-flow-voxel":
        f0 = torch.from_numpy(E.utils.generate_smooth_flow((H, W), 20, seed=1046)).to(dev)
        motion_dev = E.utils.construct_dense_flow_voxel_torch(f0 / 20.0, T, "burgers", "middle").float() * 20.0
        motion = motion_dev.double().cpu().numpy()
    else:
        motion_dev = torch.from_numpy(np.asarray(motion)).to(dev).float().contiguous()
    desc = E.make_descriptor(cfg["cost"], cfg["model"], sigma=cfg["sigma"], time_bin=T)

    # one evaluation = one prepared library call (outputs allocated once, pointers resolved once: what a solver loop in C
    # would do; `evaluate` spends ~6 us per call in Python, which on a busy host is the difference between a GPU-bound and a
    # host-bound 17 us evaluation -- profiles/r02_ablation.txt)
    # Form of the evaluation.  "device" (default: the COMPLETE evaluation): cmax_objective, loss and gradient finished on the device.
    # "raw" (2-DoF objectives on one GPU): K1 + K3 leave 32 x 6 partial sums on the device and the CONSUMER folds them on the host
    # when it reads the result (cmax_objective_raw + cmax_finalize_raw_host; here: once per run, for `loss`) -- no finishing launch.
    # "host": cmax_objective_host, every step waits for its numbers on the host (sequential: what one optimiser iteration costs).
    form = cfg.get("form") or args.form
    finalize = None
    per_call = 1  # evaluations per call (cfg2_batch8: 8)
    if form == "batch" and world == 1:
        per_call = int(cfg["batch"])
        thetas = np.asarray(motion, dtype=np.float64)[None, :] * np.linspace(0.6, 1.3, per_call)[:, None]  # a line search's steps along theta
        call, res_b, grad_b = handle.prepare_batch(desc, torch.from_numpy(thetas).to(dev).float().contiguous())
        res, grad = res_b[per_call // 2], grad_b
    elif form == "raw" and world == 1 and handle.has_raw(desc):
        call, res, finalize = handle.prepare_raw(desc, motion_dev)
        grad, form = None, "raw"
    elif form == "host" and world == 1:
        call, res, grad = handle.prepare_host(desc, motion_dev)
    else:
        call, res, grad = sliced.prepare(desc, motion_dev, want_grad=True)
        form = "device"

    def step():
        call()
        return res, grad

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        res, grad = step()
    # clock ramp: a process that starts on an idle GPU measures its first ~second in a lower power state (the first bench process
    # on a fresh box: kernels 6-7 % slower through all 25 windows -- 0.1 s of GPU work in total -- than the same command run
    # again; profiles/r02_ablation.txt).  Untimed evaluations until `--ramp` seconds have passed; the K timed steps follow.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp:
        for _ in range(200):
            step()
        torch.cuda.synchronize()
    # Every window of EXACTLY `steps` evaluations sits between barrier + synchronize on both sides and is timed twice: with HIP events
    # recorded on the launch stream (torch's current stream IS the stream the library launches on, functional._stream) --
    # `value` -- and with the host clock around the closing synchronize (round 1-5's figure, kept as host_clock_*: it contains the
    # synchronize's return latency, a few us per 340-us window that moved with the box).
    times, host_times = [], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    settle = []  # headline row only: untimed windows until the clocks have settled (two consecutive windows within 1 % of the best), <= 6 s
    t_settle = time.perf_counter()
    while keep_inputs and args.ramp > 0 and world == 1 and time.perf_counter() - t_settle < 6.0:
        ev0.record()
        for _ in range(max(steps, 100)):
            step()
        ev1.record()
        torch.cuda.synchronize()
        settle.append(ev0.elapsed_time(ev1))
        if len(settle) >= 4 and max(settle[-2:]) <= 1.01 * min(settle):
            break
    for _ in range(windows):
        sync_all()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            res, grad = step()
        ev1.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        host_times.append(time.perf_counter() - t0)
        times.append(ev0.elapsed_time(ev1) * 1e-3)
    both = torch.tensor([times, host_times], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(both, op=dist.ReduceOp.MAX)  # a window lasts as long as its slowest rank
    both = both.cpu().numpy()
    times, host_times = np.sort(both[0]), np.sort(both[1])
    elapsed = float(np.median(times))
    loss = float(finalize()[0][0]) if finalize else float(res[0].item() if hasattr(res[0], "item") else res[0])
    n_total = n_local * world
    evals = steps * per_call  # evaluations per window (a batch call evaluates the objective per_call times): every figure below is per EVALUATION
    out = {"workload": cfg["desc"], "events_per_gpu": n_local, "events_total": n_total, "image": [H, W], "motion_model": cfg["model"],
           "cost": cfg["cost"], "blur_sigma": cfg["sigma"], "time_bins": T,
           "ms_per_step": elapsed / evals * 1e3, "value": n_total * evals / elapsed,
           "window_ms_per_step": {"min": float(times[0]) / evals * 1e3, "median": elapsed / evals * 1e3, "max": float(times[-1]) / evals * 1e3,
                                  "windows": windows, "steps_per_window": evals, "clock": "HIP events on the launch stream",
                                  "host_clock_median": float(np.median(host_times)) / evals * 1e3},
           "loss": loss, "prepare_ms_once_per_batch": prepare_ms, "collectives": sliced.collectives, "deterministic": bool(args.deterministic),
           "result_form": {"raw": "raw sums on the device, folded by the consumer on the host (cmax_objective_raw + cmax_finalize_raw_host)",
                           "device": "loss + gradient on the device (cmax_objective)",
                           "batch": "loss + gradient of every candidate on the device (cmax_objective_batch)",
                           "host": "loss + gradient on the host after every evaluation (cmax_objective_host)"}[form]}
    if world > 1:
        out["rccl"] = dict(zip(("nranks", "rank", "version"), handle.comm_info()))
    # B / t_eval.  Per GPU: N/g events + the full images (every rank evaluates the image-space part)
    eval_bytes = evaluation_bytes(n_local, H, W, cfg["model"], T)
    out["evaluation_bytes_per_gpu"] = eval_bytes
    out["evaluation_GBps_per_gpu"] = eval_bytes / (out["ms_per_step"] * 1e-3) / 1e9
    out["evaluation_frac"] = out["evaluation_GBps_per_gpu"] / HBM_PEAK_GBS

    if profile and form != "batch":
        # instrumented passes (after the timed region, same inputs): HIP events recorded on the launch stream
        # around every launch of the kernel classes.
        #  (a) one launch per bracket: what an evaluation actually runs, but each bracket adds ~2.5 us of
        #      marker / dispatch latency to kernels that only run ~8 us;
        #  (b) REPEAT launches per bracket: amortises that latency -> the per-launch duration used for the
        #      per-kernel roofline (agrees with the rocprofv3 kernel durations under profiles/).
        REPEAT = 8
        psteps = min(steps, 100)
        handle.set_profiling(True)
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        prof_single = handle.read_profile()
        chunks = []  # five short passes, median per class: one disturbed bracket must not move a 7 us figure
        for _ in range(5):
            handle.set_profiling(True, repeat=REPEAT)
            for _ in range(max(psteps // 10, 8)):
                step()
            torch.cuda.synchronize()
            chunks.append({k: (ms / cnt * 1e3 if cnt else 0.0) for k, (ms, cnt) in handle.read_profile().items()})
        handle.set_profiling(False)
        hot = ("vote", "stats", "gimage", "grad")
        per_kernel = {k: float(np.median([c[k] for c in chunks])) for k in chunks[0]}  # us per launch
        single = {k: (ms / cnt * 1e3 if cnt else 0.0) for k, (ms, cnt) in prof_single.items()}
        kernels = {}
        for k in hot:
            if per_kernel[k] <= 0:
                continue
            ab = algorithmic_bytes(k, n_local, H, W, cfg["model"], T)
            gbps = ab / (per_kernel[k] * 1e-6) / 1e9
            kernels[k] = {"kernel": KERNEL_NAMES[k], "launch_us": per_kernel[k], "single_launch_bracket_us": single[k],
                          "algorithmic_bytes_per_launch": ab, "GBps": gbps, "frac": gbps / HBM_PEAK_GBS,
                          "traffic": measured_traffic(name, k)}
        for k in ("finish", "comm"):  # never repeated inside a bracket: the single-launch figure is the only one
            if single.get(k, 0.0) > 0:
                kernels[k] = {"kernel": KERNEL_NAMES[k], "single_launch_bracket_us": single[k],
                              "launches_per_evaluation": prof_single[k][1] / psteps}
        out["kernels"] = kernels
        out["dominant"] = max((k for k in hot if k in kernels), key=lambda k: kernels[k]["launch_us"])
        out["profile_method"] = ("HIP events on the launch stream; each bracket of a hot class holds %d back-to-back launches of the "
                                 "kernel (instrumented passes after the timed region, same inputs; median of 5 passes)" % REPEAT)
    # counter bytes of one evaluation (profiles/r*_pmc_<workload>.json: FETCH_SIZE x 2 + WRITE_SIZE per launch, summed over the kernel
    # classes) / step time = the PHYSICAL rate through the memory-side counters, beside the algorithmic fraction: the packed event is
    # 8 or 4.5 bytes where SURVEY 8(d) charges 12, so the two differ by up to 2.3x on the large rows
    phys = [measured_traffic(name, k) for k in ("vote", "stats", "gimage", "grad", "finish")]
    if any(phys):
        out["physical_bytes_per_evaluation"] = int(sum(v for v in phys if v))
        out["physical_GBps"] = out["physical_bytes_per_evaluation"] / (out["ms_per_step"] * 1e-3) / 1e9
    if keep_inputs:
        out["_inputs"] = (cfg, ev, motion)
        if world == 1:
            try:
                out["launch_floor_us"] = handle.launch_floor_us()
            except Exception as e:  # never fail the bench line over a side figure
                out["launch_floor_us"] = None
                out["launch_floor_error"] = f"{type(e).__name__}: {e}"[:200]
    handle.close()
    return out


# DESIGN.md section 5's cost model of one all-reduce of S bytes over N ranks (fully connected xGMI mesh): alpha(N) + S / beta at N = 2
# (one link), alpha(N) + 2 S / (N beta) from N = 4 on (reduce-scatter + all-gather, every link carries S / N each way)
COMM_MODEL = {"alpha_us": {2: 12.0, 4: 18.0, 8: 25.0}, "beta_GBps": 100.0}


def comm_model_us(nbytes, world):
    alpha = COMM_MODEL["alpha_us"].get(world, 25.0 if world > 8 else 12.0)
    wire = nbytes / (COMM_MODEL["beta_GBps"] * 1e3)  # us
    return alpha + (wire if world <= 2 else 2.0 * wire / world)


# The partition DESIGN section 5 names beside BASELINE's time slices: rank r owns a BAND OF SOURCE TILE ROWS (every event whose source pixel
# lies in it, whatever its time).  Its events vote into its own rows plus a halo of max |flow dt| rows on either side, so C1 shrinks from an
# all-reduce of the image to two neighbour exchanges of halo rows (partial sums to the owner, finished rows back), the image statistics become
# an all-reduce of two doubles, and a dense flow gradient needs no C2 at all (every source pixel has one owner; the solver's patch gradient
# stays a 4 KB all-reduce).  Assumed like COMM_MODEL: alpha_p2p = 8 us per grouped send / receive pair, beta = 100 GB/s per link.
BAND_MODEL = {"alpha_p2p_us": 8.0, "halo_rows": 20}


def band_model_us(compute_us, width, world, halo_rows=None, patch_gradient=True):
    """One evaluation under the row-band partition: compute + 2 halo exchanges + the statistics' all-reduce (+ the patch gradient's)."""
    if world <= 1:
        return compute_us
    halo = BAND_MODEL["halo_rows"] if halo_rows is None else halo_rows
    wire = halo * width * 4 / (COMM_MODEL["beta_GBps"] * 1e3)  # us per direction and exchange (both neighbours in parallel on their own links)
    chain = 2.0 * (BAND_MODEL["alpha_p2p_us"] + wire) + comm_model_us(16, world)
    return compute_us + chain + (comm_model_us(4096, world) if patch_gradient else 0.0)


def comm_probe_run(dev, world, rank, args, reps=50):
    """N > 1: the collectives of an evaluation measured ON THEIR OWN, beside what DESIGN section 5's model predicts for them, so that one
    SCALE record confirms or refutes the model.  Sizes: the 16-byte 2-DoF gradient, the 4 KB gradient of the solver's
    patch objective (2 x 256 doubles), cfg2's single exchange (I, E0, F1 planes), cfg5's C1 (one 1280x720 image) and C2 (its flow
    gradient).  `reps` back-to-back calls on the stream between two synchronisations, median of 5 such windows, MAX over ranks."""
    import torch
    import torch.distributed as dist

    import event_based_optical_flow_amd as E
    from event_based_optical_flow_amd.distributed import TimeSlicedObjective

    # The probe is collective: every rank must take the same path.  Set-up (handle, communicator, the largest buffer) happens under a
    # per-rank try, then the ranks AGREE on its success (one MIN all-reduce) before any barrier / all-reduce of the timed loops -- a
    # rank that failed alone would otherwise leave the others blocked in dist.barrier.
    handle = sliced = None
    err = ""
    try:
        handle = E.CMaxHandle((64, 64))
        sliced = TimeSlicedObjective(handle, in_library=not args.torch_collectives)
        in_lib = sliced.collectives.startswith("in-library")
        probe = torch.zeros(2 * 720 * 1280, dtype=torch.float32, device=dev)
        if in_lib:
            handle.comm_allreduce  # (attribute exists: the library path is available on this rank)
        del probe
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"[:200]
    ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() < 1.0:
        if handle is not None:
            handle.close()
        return {"error": "set-up failed on at least one rank" + (f" (this rank: {err})" if err else ""), "n_ranks": world}
    flag = torch.tensor([1.0 if in_lib else 0.0], dtype=torch.float64, device=dev)  # ... and on WHICH collectives (all in-library or none)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    in_lib = in_lib and flag.item() >= 1.0
    sizes = {"grad_2dof_16B": (2, torch.float64), "grad_patch_4KB": (512, torch.float64),
             "cfg2_single_exchange": (3 * 260 * 346 + 260 + 346, torch.float32), "cfg5_C1_image": (720 * 1280, torch.float32),
             "cfg5_C2_flow_gradient": (2 * 720 * 1280, torch.float32)}
    rows = {}
    for name, (count, dtype) in sizes.items():
        buf = torch.zeros(count, dtype=dtype, device=dev)
        op = (lambda: handle.comm_allreduce(buf)) if in_lib else (lambda: dist.all_reduce(buf))
        for _ in range(5):
            op()
        wins = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                op()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / reps * 1e6)
        t = torch.tensor([float(np.median(wins))], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        nbytes = count * (8 if dtype == torch.float64 else 4)
        rows[name] = {"bytes": nbytes, "us": _r(float(t.item())), "model_us": _r(comm_model_us(nbytes, world))}
    # ... and what the ROW-BAND partition would exchange instead of C1: cfg5's 20 halo rows (102 KB) to the next rank and
    # from the previous one, as one grouped send / receive pair -- measures alpha_p2p, the one number that model assumes and no all-reduce shows
    halo = None
    try:
        count = BAND_MODEL["halo_rows"] * 1280
        send, recv = torch.zeros(count, dtype=torch.float32, device=dev), torch.empty(count, dtype=torch.float32, device=dev)
        nxt, prv = (rank + 1) % world, (rank - 1) % world

        def exchange():
            for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, nxt), dist.P2POp(dist.irecv, recv, prv)]):
                r.wait()

        for _ in range(5):
            exchange()
        wins = []
        for _ in range(5):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                exchange()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / reps * 1e6)
        t = torch.tensor([float(np.median(wins))], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        halo = {"bytes": count * 4, "us": _r(float(t.item())),
                "model_us": _r(BAND_MODEL["alpha_p2p_us"] + count * 4 / (COMM_MODEL["beta_GBps"] * 1e3)),
                "note": "torch.distributed batch_isend_irecv ring (next / previous rank), host-waited per exchange: an upper bound of a stream-ordered ncclSend / ncclRecv pair"}
    except Exception as e:  # noqa: BLE001 (a side figure; every rank takes the same path up to here)
        halo = {"error": f"{type(e).__name__}: {e}"[:200]}
    f1 = lambda m: 19.3 + 3.29 * m  # one-GPU us per evaluation at 720p from the measured 2.5M ... 64M-event rows
    band = {"cfg5_20M_fused": _r(band_model_us(f1(20.0 / world), 1280, world, patch_gradient=False)),
            "cfg5_20M_solver_objective": _r(band_model_us(f1(20.0 / world), 1280, world)),
            "hbm_64M_fused": _r(band_model_us(f1(64.0 / world), 1280, world, patch_gradient=False))}
    handle.close()
    return {"collectives": sliced.collectives, "n_ranks": world, "model": COMM_MODEL, "allreduce": rows,
            "band_partition": {"assumed": BAND_MODEL, "halo_exchange": halo, "model_us_per_evaluation": band},
            "note": "us = measured per call (back-to-back calls on one stream, max over ranks); model_us = DESIGN.md section 5's alpha-beta prediction"}


def graph_replay_rate(cfg, ev, motion, dev, steps, windows):
    """The same evaluations replayed from a hipGraph (torch.cuda.CUDAGraph over the library's launches): `steps` evaluations
    per replay, so the host does nothing between them.  Reported NEXT TO `value`, never as it: an optimiser reads loss and
    gradient after every evaluation and cannot run this way -- the figure shows what the GPU does when the host is out of
    the picture (on a shared node the eager loop's 10 us of enqueue work per 17 us evaluation is what neighbours disturb)."""
    import torch

    import event_based_optical_flow_amd as E

    try:
        handle = E.CMaxHandle((cfg["H"], cfg["W"]))
        handle.set_events(torch.from_numpy(ev).to(dev))
        desc = E.make_descriptor(cfg["cost"], cfg["model"], sigma=cfg["sigma"])
        m = torch.from_numpy(np.asarray(motion)).to(dev).float().contiguous()
        finalize = None
        call, res, grad = handle.prepare(desc, m)  # the headline's own form: cmax_objective
        k = steps + (steps & 1)  # even: the handle's double-buffered images end a replay where they began it
        for _ in range(50):
            call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(k):
                call()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        times = []
        for _ in range(windows):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / k)
        t = float(np.median(times))
        out = {"ms_per_step": t * 1e3, "value": cfg["n"] / t, "evaluations_per_graph": k, "windows": windows,
               "loss": float(finalize()[0][0]) if finalize else float(res[0].item()),
               "note": "hipGraph replay of the same launches; not the way a solver can call the path"}
        handle.close()
        return out
    except Exception as e:  # capture not available / refused: report, never fail the bench line
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def _r(x, digits=4):
    """Round to `digits` significant digits (compact line)."""
    if x is None or isinstance(x, (str, bool, int)):
        return x
    return float("%.*g" % (digits, x))


def compact_line(out, verbose_path):
    """The ONE line bench.py prints: the contract's keys, `roofline`, `cpu_baseline` and -- under `configs` -- one short entry per
    other configuration: {us: microseconds per evaluation, frac: evaluation roofline fraction (SURVEY 8d bytes / time / 8 TB/s),
    dom_us: the dominant kernel's launch, prep_ms: pack + sort + work list once per batch}.  Everything else (per-kernel
    dictionaries, window statistics, descriptions, the CPU thread sweep) is in the file named by `verbose`."""
    r = out["roofline"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(out["value"], 6), _r(out["ms_per_step"], 6)
    c = out["config"]
    line["config"] = {"workload": c["workload"], "events_per_gpu": c["events_per_gpu"], "image": c.get("image"), "motion_model": c["motion_model"],
                      "cost": c["cost"], "blur_sigma": c["blur_sigma"], "parallelism": c["parallelism"],
                      "result_form": c["result_form"].split(" (")[-1].rstrip(")"), "collectives": c["collectives"], "rccl": c["rccl"]}
    dom = r["dominant"]
    line["roofline"] = {"bound": "hbm", "achieved": _r(r["achieved"]), "peak": r["peak"], "unit": "GB/s", "frac": _r(r["frac"]),
                        "traffic": r["traffic"], "algorithmic_bytes": r["algorithmic_bytes_per_evaluation"],
                        "scope": "one evaluation: B = 24N + 16HW + B_model (SURVEY 8d) / median step time",
                        "dominant": {"kernel": dom["kernel"].split(" ")[0], "launch_us": _r(dom["launch_us"]), "frac": _r(dom["frac"]),
                                     "bytes": dom["algorithmic_bytes_per_launch"], "traffic": dom["traffic"]},
                        "kernels_us": {k: _r(v.get("launch_us", v.get("single_launch_bracket_us"))) for k, v in r["kernels"].items()},
                        "launch_floor_us": _r(r.get("launch_floor_us")), "frac_raw_form": _r(r.get("frac_raw_form")),
                        "frac_host_result": _r(r.get("frac_host_result")), "frac_batch8": _r(r.get("frac_batch8")),
                        "frac_batch32": _r(r.get("frac_batch32")), "frac_host_clock": _r(r.get("frac_host_clock")),
                        "physical_GBps": _r(r.get("physical_GBps")), "traffic_source": (r.get("traffic_source") or "")[:60] or None}
    line["timing_us"] = {k: _r(out["timing"][k] * 1e3) for k in ("min", "median", "max")}
    if "host_clock_median" in out["timing"]:  # (records of rounds 1-5 were timed with the host clock alone)
        line["timing_us"]["clock"] = "hip_events"
        line["timing_us"]["host_clock_median"] = _r(out["timing"]["host_clock_median"] * 1e3)
    line["prepare_ms_once_per_batch"] = _r(out["prepare_ms_once_per_batch"])
    line["loss"] = _r(out["loss"], 8)
    if out.get("also"):
        line["configs"] = {k: {"us": _r(a["ms_per_step"] * 1e3), "frac": _r(a["evaluation_frac"]), "dom_us": _r(a["dominant_kernel_us"]),
                               "prep_ms": _r(a["prepare_ms_once_per_batch"], 3)} for k, a in out["also"].items()}
        for k, a in out["also"].items():  # physical rate through the memory-side counters, where a counter summary of the row exists
            if a.get("physical_GBps"):
                line["configs"][k]["phys_GBps"] = _r(a["physical_GBps"], 3)
        if out["n_gpus"] > 1:
            for k, a in out["also"].items():
                line["configs"][k]["value"] = _r(a["value"], 5)
    if "graph_replay" in out:
        line["graph_replay_us"] = _r(out["graph_replay"].get("ms_per_step", 0.0) * 1e3) if "ms_per_step" in out["graph_replay"] else None
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
                                "cpu_model": cb.get("cpu_model"), "host_cpus": cb.get("host_cpus")}
    ct = out.get("cpu_baseline_torch")
    if ct and "value" in ct:
        line["cpu_baseline_torch"] = {"value": _r(ct["value"]), "unit": "events/s", "cores": ct["cores"], "kind": "port",
                                      "sample": "oracle/torch_cpu.py (reference-style tensor ops + autograd, fp64), best row of the thread sweep",
                                      "threads_sweep": {k: _r(v) if not isinstance(v, str) else v[:40] for k, v in ct.get("threads_sweep", {}).items()}}
    if out.get("comm_probe"):
        line["comm_probe"] = out["comm_probe"]
    line["verbose"] = os.path.relpath(verbose_path, ROOT) if os.path.isabs(verbose_path) else verbose_path
    return line


def pin_to_gpu_numa_node(dev_index):
    """Best effort: run this process on the CPUs of the NUMA node its GPU hangs off (sysfs).  The launch thread of a process
    that lands on the far socket reaches the GPU's doorbell and queue across the inter-socket link: of 18 unpinned / far-node
    runs of the default command 3 measured 19 us per cfg2 evaluation instead of 16.5-17.5, of 14 runs pinned to the GPU's node
    none (profiles/r02_ablation.txt).  Returns a description for the JSON line, or None when the topology is not exposed."""
    try:
        import torch

        p = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)  # never widen what the launcher allowed
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"NUMA node {node} of GPU {bdf} ({len(cpus)} CPUs)"
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the ranks ourselves."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not args.share_gpu:
        raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) are visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-torch-worker":
        _cpu_torch_worker(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=25, help="timed windows of --steps evaluations; the median window is reported")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the process to the CPUs of its GPU's NUMA node")
    ap.add_argument("--ramp", type=float, default=1.5, help="seconds of untimed evaluations after the warm-up steps, before the timed "
                    "windows (lets an idle GPU reach its clocks; 0 = none)")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other configurations (cfg3, cfg4, cfg5) reported under `also`")
    ap.add_argument("--events", default="uniform", choices=["uniform", "structured"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise the N > 1 path on a 1-GPU box)")
    ap.add_argument("--torch-collectives", action="store_true",
                    help="N > 1: all-reduce with torch.distributed around the phase-split calls instead of inside the library")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (1-GPU box testing with --backend gloo)")
    ap.add_argument("--form", default="device", choices=["auto", "raw", "device", "host"],
                    help="how an evaluation hands over its result: device (default; auto is the same) = cmax_objective, loss and gradient "
                         "finished on the device; raw = cmax_objective_raw where the objective has that form (2-DoF), the consumer folds "
                         "the partial sums; host = cmax_objective_host (every step waits for its numbers)")
    ap.add_argument("--deterministic", action="store_true", help="cmax_set_deterministic(1): integer accumulation, bit-repeatable results (slower)")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1: do not re-execute the headline workload under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                    "for roofline.traffic (two short passes, ~40 s); the committed counter summary under profiles/ is replayed instead")
    ap.add_argument("--verbose", action="store_true", help="print the full record (per-kernel dictionaries, every row's windows: ~30 KB) instead "
                    "of the compact line; the full record is written to --verbose-out either way")
    ap.add_argument("--verbose-out", default=None, help="file for the full record (default gpurun_out/bench_verbose_<workload>_n<N>.json)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    launch_affinity = set(os.sched_getaffinity(0))  # what the launcher allowed (the CPU baselines run on all of it)
    host_affinity = None if args.no_pin else pin_to_gpu_numa_node(dev.index)

    also = {}
    if not args.no_also:
        # the other configurations, fewer steps (their evaluations are 2-10x longer); same timing protocol
        # N = 1: the other single-GPU configurations, cfg5 as BASELINE states it on ONE GPU (the N = 1 point of its strong-scaling
        # curve), a working set larger than the Infinity Cache, and the headline's second rows (blur; sharp image)
        names = ([w for w in ("cfg3", "cfg4", "cfg5", "cfg5_strong", "hbm", "cfg2_raw", "cfg2_host_result", "cfg2_sigma1", "cfg2_structured",
                              "cfg2_theta80", "cfg2_theta150", "cfg2_batch8", "cfg2_batch32", "cfg3_rough", "cfg5_rough")
                  if w != args.workload]
                 if world == 1 else ["cfg5_strong"])
        for wname in names:
            # (the K-step contract binds the headline line only: these rows use windows of >= 50 evaluations, so that the barrier +
            # synchronize around a window -- ~20 us -- does not weigh on a 30 us evaluation; steps_per_window is reported per row)
            r = run_workload(wname, args, rank, world, dev, max(50, args.steps // 2), max(3, args.warmup // 4), max(5, args.windows // 2))
            dom = r.get("dominant")
            also[wname] = {"workload": r["workload"], "events_total": r["events_total"], "events_per_gpu": r["events_per_gpu"],
                           "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "events/s",
                           "scaling": "strong" if WORKLOADS[wname].get("strong") else "weak",
                           "evaluation_frac": r["evaluation_frac"], "evaluation_GBps_per_gpu": r["evaluation_GBps_per_gpu"],
                           "evaluation_bytes_per_gpu": r["evaluation_bytes_per_gpu"], "packed_event_stream_bytes": 8 * r["events_per_gpu"],
                           "window_ms_per_step": r["window_ms_per_step"], "loss": r["loss"], "result_form": r["result_form"],
                           "dominant_kernel": r["kernels"][dom]["kernel"] if dom else None,
                           "dominant_kernel_us": r["kernels"][dom]["launch_us"] if dom else None,
                           "dominant_kernel_frac": r["kernels"][dom]["frac"] if dom else None,
                           "kernels_us": {k: v.get("launch_us", v.get("single_launch_bracket_us")) for k, v in r.get("kernels", {}).items()},
                           "physical_GBps": r.get("physical_GBps"), "host_clock_ms_per_step": r["window_ms_per_step"]["host_clock_median"],
                           "collectives": r["collectives"], "prepare_ms_once_per_batch": r["prepare_ms_once_per_batch"]}

    # The headline row runs LAST, behind the other rows' ~40 s of GPU work (round 6): a process that starts on an idle box measures its
    # first seconds in a lower power state -- the default command as the FIRST GPU process on a fresh box gave K1 / K3 7.3 / 7.1 us and
    # 18.2 us per evaluation, the same command behind a test run 6.5 / 6.6 and 15.2 (gpurun_out/r06_bench_a / _b) -- and the 1.5 s ramp
    # does not cover it.  Its own W warm-up steps and the ramp still precede its K timed steps.
    main_res = run_workload(args.workload, args, rank, world, dev, args.steps, args.warmup, args.windows, keep_inputs=True)
    cfg, ev, motion = main_res.pop("_inputs")

    # roofline.traffic measured in THIS run (N = 1; the sub-runs pass --no-pmc): else the replayed summary of an earlier run
    pmc_traffic, pmc_note = None, None
    if world == 1 and not args.no_pmc and not args.deterministic:
        torch.cuda.synchronize()
        pmc_traffic, pmc_note = pmc_traffic_this_run(args.workload)
        if pmc_traffic:
            for k, v in main_res.get("kernels", {}).items():
                if k in pmc_traffic:
                    v["traffic"] = pmc_traffic[k]
            ptotal = sum(pmc_traffic.get(k, 0) for k in ("vote", "stats", "gimage", "grad", "finish"))
            main_res["physical_bytes_per_evaluation"] = ptotal
            main_res["physical_GBps"] = ptotal / (main_res["ms_per_step"] * 1e-3) / 1e9

    comm_probe = None
    if world > 1:
        try:
            comm_probe = comm_probe_run(dev, world, rank, args)
        except Exception as e:  # never fail the bench line over a side figure (every rank takes the same path: the probe is collective)
            comm_probe = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        H, W, n = cfg["H"], cfg["W"], main_res["events_per_gpu"]
        dom = main_res["dominant"]
        kd = main_res["kernels"][dom]
        out = {
            "metric": "events/sec through warp+IWE+cost+grad (one objective evaluation)",
            "value": main_res["value"],
            "unit": "events/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if cfg.get("strong") else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (%s events, seed 46)" % args.events,
            "config": {"workload": cfg["desc"], "events_per_gpu": n, "image": [H, W], "motion_model": cfg["model"],
                       "cost": cfg["cost"], "blur_sigma": cfg["sigma"], "parallelism": f"time-slice x{world}", "deterministic": bool(args.deterministic),
                       "result_form": main_res["result_form"],
                       "collectives": None if world == 1 else main_res["collectives"] + ": all-reduce(IWE) + all-reduce(grad) per evaluation",
                       "rccl": main_res.get("rccl")},
            "timing": dict(main_res["window_ms_per_step"], ramp_s=args.ramp, host_affinity=host_affinity,
                           statistic="median window; every window = `steps` evaluations between barrier + synchronize on both sides, max "
                                     "over ranks; `ramp_s` seconds of untimed evaluations precede the windows (GPU clocks); one evaluation = "
                                     "one prepared library call (CMaxHandle.prepare)"),
            "roofline": {"bound": "hbm",
                         "scope": "one evaluation (SURVEY 8d: B = 24 N + 16 HW + B_model algorithmic bytes / median step time), per GPU",
                         "achieved": main_res["evaluation_GBps_per_gpu"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": main_res["evaluation_frac"],
                         # the same evaluation in its other two forms (also.cfg2_raw / also.cfg2_host_result), when those rows ran
                         "frac_raw_form": also.get("cfg2_raw", {}).get("evaluation_frac") if args.workload == "cfg2" else None,
                         "frac_host_result": also.get("cfg2_host_result", {}).get("evaluation_frac") if args.workload == "cfg2" else None,
                         # ... and per evaluation of a K-candidate call (cmax_objective_batch): the only form of the 1M-event evaluation that
                         # clears 0.40 -- a single evaluation is a chain of launch latencies (launch_floor_us)
                         "frac_batch8": also.get("cfg2_batch8", {}).get("evaluation_frac") if args.workload == "cfg2" else None,
                         "frac_batch32": also.get("cfg2_batch32", {}).get("evaluation_frac") if args.workload == "cfg2" else None,
                         "frac_host_clock": main_res["evaluation_frac"] * main_res["ms_per_step"] / main_res["window_ms_per_step"]["host_clock_median"],
                         "physical_GBps": main_res.get("physical_GBps"),
                         # two dependent EMPTY launches with this evaluation's grids, measured in this run (cmax_debug_launch_floor)
                         "launch_floor_us": main_res.get("launch_floor_us"),
                         "algorithmic_bytes_per_evaluation": main_res["evaluation_bytes_per_gpu"],
                         "traffic": (sum(pmc_traffic.get(k, 0) for k in ("vote", "stats", "gimage", "grad", "finish")) if pmc_traffic else
                                     sum(v for v in (measured_traffic(args.workload, k) for k in ("vote", "stats", "gimage", "grad", "finish")) if v) or None),
                         "traffic_source": pmc_note if pmc_traffic else traffic_source(args.workload),
                         "traffic_this_run_error": None if (pmc_traffic or args.no_pmc or world > 1) else pmc_note,
                         "dominant": {"kernel": kd["kernel"], "achieved": kd["GBps"], "frac": kd["frac"], "launch_us": kd["launch_us"],
                                      "algorithmic_bytes_per_launch": kd["algorithmic_bytes_per_launch"], "traffic": kd["traffic"]},
                         "kernels": main_res["kernels"], "method": main_res["profile_method"]},
            "loss": main_res["loss"],
            "prepare_ms_once_per_batch": main_res["prepare_ms_once_per_batch"],
        }
        if also:
            out["also"] = also
        if world == 1 and not args.deterministic and cfg["model"] == "2d-translation":
            out["graph_replay"] = graph_replay_rate(cfg, ev, motion, dev, max(args.steps, 50), min(args.windows, 11))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, ev, motion)
            tc = cpu_baseline_torch(args.workload, cfg, launch_affinity)
            if tc is not None:
                out["cpu_baseline_torch"] = tc
        if comm_probe is not None:
            out["comm_probe"] = comm_probe
        # The full record goes to a file; stdout carries ONE compact line (the verbose line was ~30 KB and the driver
        # keeps the last 8 KB of stdout, so the rows for cfg3 / cfg4 / cfg5 / cfg5_strong / hbm never reached BENCH_r04.json).
        vpath = args.verbose_out or os.path.join(ROOT, "gpurun_out", "bench_verbose_%s_n%d.json" % (args.workload, world))
        try:
            if os.path.dirname(vpath):  # (a bare file name has no directory part: makedirs('') raises)
                os.makedirs(os.path.dirname(vpath), exist_ok=True)
            with open(vpath, "w") as f:
                json.dump(out, f)
        except OSError as e:
            vpath = "not written: %s" % e
        print(json.dumps(out if args.verbose else compact_line(out, vpath)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
