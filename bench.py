#!/usr/bin/env python3
"""bench.py -- EKF steps/s of the batched VI-EKF hot path on MI355X.

One "step" of one filter = one propagate (dt = 4 ms) + N_feat sequential active pixel updates
(SURVEY.md 8d).  Default workload = BASELINE.json headline: batch 1024 filters x N_feat 50 per
GPU.  Filters are independent, so N GPUs run N x batch filters with no data-path collective
(weak scaling); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (dominant kernel,
HIP-event timed on the launch stream) and `cpu_baseline` (the CPU oracle = a port of the
reference's dense algorithm, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def shard(total, world, rank):
    """contiguous partition of `total` filters: rank r owns [lo, hi)  """
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi


def reduce_times(local_seconds, world):
    """max over ranks of the timed-region seconds (RCCL/gloo all_reduce MAX of one double)"""
    if world == 1:
        return local_seconds
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([local_seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_record(steps, seconds, nbytes, max_rel_err, world):
    """the one collective of the run -- every rank's {steps, seconds, bytes, max_rel_err} reduced to
    {sum, max, sum, max} (two all_reduce calls of two doubles each over RCCL / gloo)."""
    if world == 1:
        return {"steps": float(steps), "seconds": float(seconds), "bytes": float(nbytes), "max_rel_err": float(max_rel_err)}
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    tot = torch.tensor([float(steps), float(nbytes)], dtype=torch.float64, device=dev)
    mx = torch.tensor([float(seconds), float(max_rel_err)], dtype=torch.float64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return {"steps": float(tot[0].item()), "seconds": float(mx[0].item()), "bytes": float(tot[1].item()),
            "max_rel_err": float(mx[1].item())}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """host threads this process may really use: the scheduler affinity, capped by a cgroup CPU quota if one is set (a GPU box
    hands a one-GPU job a share of the host, not all of it) -> (threads, how it was determined)"""
    n = os.cpu_count() or 1
    how = "os.cpu_count()"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, how = aff, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    q = int(int(txt[0]) / int(txt[1]))
                    if 0 < q < n:
                        n, how = q, "cgroup cpu.max"
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0 and 0 < quota // period < n:
                    n, how = quota // period, "cgroup cfs quota"
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n), how


def oracle_run(sc, N, params, which, steps_cmp, threads, structured=False):
    """The oracle on filters `which` x `steps_cmp` steps of the SAME inputs -> (x, P, seconds).
    Checker / baseline only -- never the product path."""
    from oracle import oracle as orc
    keys = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c",
            "p_b_c", "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")
    fs = []
    for b in which:
        f = orc.OracleFilter(N).init(**{k: params[k] for k in keys})
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    u = np.ascontiguousarray(sc["u"][:steps_cmp, which].transpose(1, 0, 2))
    z = np.ascontiguousarray(sc["z"][:steps_cmp, which].transpose(1, 0, 2, 3))
    t0 = time.perf_counter()
    orc.run_steps_mt(fs, threads, u, float(sc["dt"][0]), z, sc["slot"][which], sc["R"], structured=structured)
    secs = time.perf_counter() - t0
    return np.stack([f.x for f in fs]), np.stack([f.P for f in fs]), secs


def rel_err(gpu_x, gpu_P, xr, Pr):
    ex = float(np.abs(gpu_x - xr).max() / np.abs(xr).max())
    eP = float(np.abs(gpu_P - Pr).max() / np.abs(Pr).max())
    return max(ex, eP)


ORACLE_KEYS = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c",
               "p_b_c", "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")


def cadence_parity(g, sc, N, params, uniq, init_filters, step, d_u, d_dt, d_z, d_slot, d_R, d_res, torch, nsample=6):
    """One 25-sample cycle of the two cadence routes (propagate / step launches; viekf_batch_step_n with K = 8, 8, 9) from freshly
    initialised filters vs the oracle on a strided sample -> max rel err of each.  Checker leg, never timed."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    B = g.B
    which = [int(b) for b in np.unique(np.linspace(0, B - 1, nsample).round().astype(int))]
    dt = float(sc["dt"][0])

    def oracle_cycle(frames_after):
        """frames_after[k] = the frame index whose updates follow IMU sample k (or None)"""
        def fly(b):
            f = orc.OracleFilter(N).init(**{k: params[k] for k in ORACLE_KEYS})
            for i in range(N):
                f.init_feature(sc["pix"][b, i], i)
            for k, (ui, fr) in enumerate(frames_after):
                if fr is None:
                    f.propagate(sc["u"][ui, b], dt)
                else:
                    f.run_steps(sc["u"][ui, b][None], dt, sc["z"][fr, b][None], sc["slot"][b], sc["R"])
            return f.x.copy(), f.P.copy()
        with ThreadPoolExecutor(max_workers=min(len(which), os.cpu_count() or 1)) as ex:
            r = list(ex.map(fly, which))
        return np.stack([a for a, _ in r]), np.stack([p for _, p in r])

    # raw route: the launches of the timed loop below (k % 8 == 7: a full step with frame k % uniq's pixels, else a propagate)
    init_filters()
    for k in range(25):
        if k % 8 == 7:
            step(k)
        else:
            g.propagate(d_u[k % uniq], d_dt)
    torch.cuda.synchronize()
    xr, Pr = oracle_cycle([(k % uniq, (k % uniq) if k % 8 == 7 else None) for k in range(25)])
    e_raw = rel_err(g.get_state()[which], g.get_covariance()[which], xr, Pr)
    # fused route: three viekf_batch_step_n launches, K = 8, 8, 9 (the inputs the timed loop uses)
    ks = (8, 8, 9)
    init_filters()
    plan = []
    for q, kq in enumerate(ks):
        idx = [(j + 3 * q) % uniq for j in range(kq)]
        uu = torch.stack([d_u[i] for i in idx]).contiguous()
        dd = torch.stack([d_dt for _ in range(kq)]).contiguous()
        g.step_n(uu, dd, d_z[q % uniq], d_slot, d_R, result=d_res)
        plan += [(i, None) for i in idx[:-1]] + [(idx[-1], q % uniq)]
    torch.cuda.synchronize()
    xf, Pf = oracle_cycle(plan)
    e_fused = rel_err(g.get_state()[which], g.get_covariance()[which], xf, Pf)
    return {"raw": e_raw, "fused": e_fused,
            "what": "one 25-sample cycle of each route vs the oracle on filters %s (x and P)" % which}


def secondary_config(v, scene, torch, dev, dev_index, B, N, K, W, parity_steps, parity_filters):
    """ms per step, steps/s, roofline fraction (own B_alg) and an in-run parity figure of another BASELINE config"""
    import glob
    uniq = min(K + W, 8)
    sc = scene.make_scene(B, N, uniq, seed=0xC0F0 + N)
    g = v.BatchVIEKF(B, N, sc["params"], device=dev_index)
    g.use_torch_stream()
    d_u, d_z = torch.tensor(sc["u"], device=dev), torch.tensor(sc["z"], device=dev)
    d_dt, d_slot, d_R = torch.tensor(sc["dt"], device=dev), torch.tensor(sc["slot"], device=dev), torch.tensor(sc["R"], device=dev)
    d_res = torch.empty((B, N), dtype=torch.int32, device=dev)
    d_pix = torch.tensor(np.ascontiguousarray(sc["pix"].transpose(1, 0, 2)), device=dev)
    d_nan = torch.full((B,), float("nan"), dtype=torch.float64, device=dev)

    def init_filters():
        g.reset()
        for i in range(N):
            g.init_feature(d_pix[i], d_nan)

    def step(s):
        g.step(d_u[s % uniq], d_dt, d_z[s % uniq], d_slot, d_R, result=d_res)

    out = {"batch": B, "n_feat": N, "kernel": g.describe()}
    if parity_filters > 0:
        which = np.unique(np.linspace(0, B - 1, parity_filters).round().astype(int))
        init_filters()
        for s in range(parity_steps):
            step(s)
        torch.cuda.synchronize()
        gx, gP = g.get_state()[which], g.get_covariance()[which]
        threads = max(1, min(len(which), usable_cpus()[0]))
        xr, Pr, secs_o = oracle_run(sc, N, sc["params"], which, parity_steps, threads)
        out["parity_max_rel_err"] = rel_err(gx, gP, xr, Pr)
        out["parity"] = "%d filters (strided) x %d step(s) vs the dense reference-order oracle, %.1f s on %d threads" % (
            len(which), parity_steps, secs_o, threads)
        if out["parity_max_rel_err"] > 1e-6:
            raise SystemExit("PARITY FAILURE vs oracle (B=%d, N=%d): rel err %.3e" % (B, N, out["parity_max_rel_err"]))
    init_filters()
    for s in range(W):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        step(W + s)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    st = g.get_status()
    b_alg = scene.algorithmic_bytes_per_step(N)
    achieved = B * K / secs * b_alg / 1e9
    traffic, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc_traffic.json"))):
        try:
            j = json.load(open(f))
            if j["config"]["batch"] == B and j["config"]["n_feat"] == N:
                traffic, src = j.get("traffic_bytes_per_launch", j.get("traffic_bytes_per_step")), os.path.relpath(f, ROOT)
        except Exception:
            pass
    out.update({"steps": K, "warmup": W, "ms_per_step": secs / K * 1e3, "steps_per_s": B * K / secs,
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "alg_bytes_per_step": b_alg * B, "traffic": traffic, "traffic_source": src},
                "flops": flops_block(N, B * K / secs),
                "bad_filters": int((st & (1 | 2 | 8) != 0).sum())})
    g.close()
    return out


def flops_block(N, steps_per_s):
    """Structured algorithmic flops of one step: Phi P Phi^T + G Q G^T + N rank-2 Lambda-masked sweeps, priced against
    the part's fp64 peak (the secondary ceiling of SURVEY 8d: the fused step's intensity is above the fp64 ridge)."""
    n_ = 16 + 3 * N
    f_alg = 2 * 2 * n_ * (256 + 57 * N) + 12 * n_ * n_ + N * 6 * n_ * n_
    # what the kernels EXECUTE: P is symmetric and one block / tile of every pair is held, so the sweeps and the
    # propagate touch n (n + 1) / 2 elements -- roughly half the algorithmic count above, which prices the full matrix
    f_exec = 2 * n_ * (256 + 57 * N) + 6 * n_ * n_ + N * 3 * n_ * (n_ + 1)
    return {"alg_flop_per_step": f_alg, "achieved_tflops": f_alg * steps_per_s / 1e12, "peak_tflops": 78.6,
            "frac": f_alg * steps_per_s / 1e12 / 78.6,
            "executed_flop_per_step": f_exec, "executed_tflops": f_exec * steps_per_s / 1e12,
            "executed_frac": f_exec * steps_per_s / 1e12 / 78.6,
            "note": "frac prices the full n x n matrix (SURVEY 8d); the kernels hold and sweep the symmetric half: executed_frac is the pipe "
                    "utilisation. Measured on this part (tools/micro): v_fma_f64 issues every 4.4-5.8 clk per SIMD, "
                    "v_mfma_f64_16x16x4 every 64 clk = the same 16 FMA/clk/SIMD: the 78.6 TFLOP/s peak holds for both pipes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="filters per GPU")
    ap.add_argument("--feat", type=int, default=50, help="N_feat")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 streaming, 2 resident")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="viekf_batch_set_tuning(KEY, VALUE) on the benchmarked batch (integers, include/viekf.h VIEKF_TUNE_*): A/B runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cadence run (profiling: keeps the kernel statistics to the timed steps)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="milliseconds of the same step run untimed BEFORE the W warm-up steps (a GPU that was just handed over idles at "
                         "low clocks); 0 = none. The figure for the bare command-line contract is always reported too (no_prewarm)")
    ap.add_argument("--cpu-filters", type=int, default=0)
    ap.add_argument("--cpu-steps", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # (rehearsals of the N > 1 path on a one-GPU box: VIEKF_DIST_BACKEND=gloo shares the card between the ranks)
    backend = os.environ.get("VIEKF_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import vi_ekf_amd as v
    from vi_ekf_amd import scene

    B, N, K, W = args.batch, args.feat, args.steps, args.warmup
    total_steps = K + W
    lo, hi = shard(B * world, world, rank)
    assert hi - lo == B
    # synthetic inputs of this rank's filters (seeded per rank so every rank has different filters)
    uniq = min(total_steps, 32)  # distinct input frames, cycled: keeps staging small at N=150
    sc = scene.make_scene(B, N, uniq, seed=0x5EED0000 + rank)
    params = sc["params"]

    g = v.BatchVIEKF(B, N, params, device=dev_index)
    if args.kernel:
        g.set_kernel(args.kernel)
    for kv in args.tune:
        key, value = kv.split("=")
        g.set_tuning(int(key), int(value))
    g.use_torch_stream()
    desc = g.describe()
    d_u = torch.tensor(sc["u"], device=dev)
    d_z = torch.tensor(sc["z"], device=dev)
    d_dt = torch.tensor(sc["dt"], device=dev)
    d_slot = torch.tensor(sc["slot"], device=dev)
    d_R = torch.tensor(sc["R"], device=dev)
    d_res = torch.empty((B, N), dtype=torch.int32, device=dev)
    d_res_all = torch.zeros((K, B, N), dtype=torch.int32, device=dev)   # result codes of every timed update
    d_pix = torch.tensor(np.ascontiguousarray(sc["pix"].transpose(1, 0, 2)), device=dev)  # [N][B][2]
    d_nan = torch.full((B,), float("nan"), dtype=torch.float64, device=dev)

    def init_filters():
        g.reset()
        for i in range(N):
            g.init_feature(d_pix[i], d_nan)

    def step(s, res=None):
        g.step(d_u[s % uniq], d_dt, d_z[s % uniq], d_slot, d_R, result=d_res if res is None else res)

    # ---- phase A: parity vs the oracle on a STRIDED sample of this rank's filters (first / last / every dispatch round),
    #      and -- rank 0 of the N=1 run only -- the CPU baselines on a bounded sample of the same workload
    cpu = None
    parity = 0.0
    if not args.no_cpu_baseline:
        ncpu = os.cpu_count() or 1
        n = 16 + 3 * N
        flop = (8 + 4 * N) * float(n) ** 3      # dense cost per filter-step; ~6 GFLOP/s per thread for the plain-C loops
        if world == 1:
            # SURVEY 8(d) / BASELINE.md 3: one filter per thread over ALL host cores this job may use (r02 capped this at 16)
            threads, threads_how = usable_cpus()
            nf = args.cpu_filters or min(B, 4 * threads)
            # ~10-15 s of CPU work for the all-core dense figure, capped by the distinct frames
            sc_steps = args.cpu_steps or int(max(1, min(uniq, round(12.0 * 6.0e9 * threads / (flop * nf)))))
        else:   # N > 1: every rank checks 8 of its own filters, sharing the host cores with the other ranks
            threads = max(1, min(8, ncpu // world))
            nf = min(B, 8)
            sc_steps = int(max(1, min(uniq, 3, round(4.0 * 6.0e9 * threads / (flop * nf)))))
        which = np.unique(np.linspace(0, B - 1, nf).round().astype(int))   # strided: filters of every dispatch round
        init_filters()
        for s in range(sc_steps):
            step(s)
        torch.cuda.synchronize()
        gx, gP = g.get_state()[which], g.get_covariance()[which]
        xr, Pr, secs_d = oracle_run(sc, N, params, which, sc_steps, threads)
        parity = rel_err(gx, gP, xr, Pr)
        if parity > 1e-6:
            raise SystemExit("PARITY FAILURE vs oracle (rank %d): rel err %.3e" % (rank, parity))
        if rank == 0 and world == 1:
            cpu = {"value": len(which) * sc_steps / secs_d, "unit": "EKF steps/s", "cores": threads, "kind": "port",
                   "cpu_model": cpu_model(), "host_cores": ncpu, "cores_from": threads_how,
                   "sample": "%d filters (strided over the batch) x %d steps of the same inputs, dense reference-order oracle "
                             "(vi_ekf.cpp:301-304, vi_ekf_meas.cpp:232-257), %.1f s wall; extrapolates linearly to the batch"
                             % (len(which), sc_steps, secs_d)}
            # one-core figure of the same flavour: a few filters, ~3 s
            n1 = int(max(1, min(len(which), round(3.0 * 6.0e9 / (flop * sc_steps)))))
            _, _, secs_1 = oracle_run(sc, N, params, which[:n1], sc_steps, 1)
            cpu["one_core"] = {"value": n1 * sc_steps / secs_1, "sample": "%d filters x %d steps, 1 thread, %.1f s" % (n1, sc_steps, secs_1)}
            # structured flavour: the block-sparse propagate / rank-2 update the HIP kernels use, on the CPU
            xs, Ps, secs_s = oracle_run(sc, N, params, which, sc_steps, threads, structured=True)
            cpu["structured"] = {"value": len(which) * sc_steps / secs_s, "cores": threads,
                                 "max_rel_err_vs_dense": rel_err(xs, Ps, xr, Pr),
                                 "sample": "same filters and steps, %.2f s wall" % secs_s}
            _, _, secs_s1 = oracle_run(sc, N, params, which[:min(len(which), 4)], sc_steps, 1, structured=True)
            cpu["structured"]["one_core"] = min(len(which), 4) * sc_steps / secs_s1
            # the job may use `threads` of the host's `ncpu` hardware threads (cgroup quota): what the WHOLE host would give is
            # a projection from the one-core figures (filters are independent: linear in cores at best), labelled as such
            cpu["whole_host_projection"] = {"cores": ncpu, "dense": cpu["one_core"]["value"] * ncpu,
                                            "structured": cpu["structured"]["one_core"] * ncpu,
                                            "note": "one-core rate x host hardware threads, not measured: the job's CPU quota is %d threads" % threads}

    # ---- phase B: warmup + timed region
    # A GPU that has just been handed over idles at low clocks: the first ~10 ms of launches run 8 % slower (measured: 20 timed
    # steps after 5 warm-up steps 0.378 ms per step, after 150 more 0.359).  The region is therefore timed TWICE with exactly the
    # command line's W warm-up + K timed steps: first as the bare contract has it (reported as `no_prewarm`, comparable with the
    # r01 / r02 driver figures), then again after --prewarm-ms of the same step (reported as prewarm_steps); `value` is the second.
    def timed_region(with_events):
        init_filters()
        for s in range(W):
            step(s)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # HIP events on the launch stream bracket GROUPS of launches (an event between every two launches costs the queue a marker
        # packet and a completion signal per step: measured 4 % of the headline step): every 10th step, four groups for short runs
        ES = 10 if K >= 40 else max(1, K // 4)
        marks = [0]
        ev = [torch.cuda.Event(enable_timing=True)] if with_events else []
        t0 = time.perf_counter()
        if with_events:
            ev[0].record()
        for s in range(K):
            step(W + s, d_res_all[s])
            if with_events and ((s + 1) % ES == 0 or s == K - 1):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
                marks.append(s + 1)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev, marks

    secs_cold, _, _ = timed_region(False)
    secs_cold = reduce_times(secs_cold, world)
    prewarm = 0
    if args.prewarm_ms > 0:
        init_filters()
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3 or prewarm < 3:
            step(prewarm)
            prewarm += 1
            if prewarm % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    secs, ev, marks = timed_region(True)
    alg_bytes = scene.algorithmic_bytes_per_step(N) * B  # one launch processes B filter-steps
    rec = reduce_record(B * K, secs, alg_bytes * K, parity, world)
    secs = rec["seconds"]
    status = g.get_status()
    n_bad = int((status & 1).sum())
    # how much work the timed region skipped: a gated / NaN-guarded / invalid update does not run its sweeps
    codes = d_res_all.view(-1)
    counts = torch.bincount(codes.clamp(0, 7), minlength=8).cpu().numpy()
    n_upd = int(codes.numel())
    work = {"updates": n_upd, "gated_frac": float(counts[1]) / n_upd, "nan_frac": float(counts[2]) / n_upd,
            "invalid_frac": float(counts[3]) / n_upd,
            "flag_counts": {"nan": int((status & 1 != 0).sum()), "blowup": int((status & 2 != 0).sum()),
                            "negative_depth": int((status & 4 != 0).sum()), "internal": int((status & 8 != 0).sum())}}

    # ---- secondary numbers, outside the contract's timed region: the realistic cadence 25 IMU propagates :
    #      3 camera frames (250 Hz : 30 Hz, params/sim_params.yaml:149,160) -- 22 propagate-only launches + 3 full steps
    cadence = None
    if world == 1 and not args.no_secondary:
        cyc = 4
        cad_par = None
        if not args.no_cpu_baseline:
            # what this leg times, checked first: one 25-sample cycle of both routes from freshly initialised
            # filters against the oracle (K vo_propagate + the frame's updates) on a strided sample; outside every timed window
            cad_par = cadence_parity(g, sc, N, params, uniq, init_filters, step, d_u, d_dt, d_z, d_slot, d_R, d_res, torch)
            if max(cad_par["raw"], cad_par["fused"]) > 1e-6:
                raise SystemExit("PARITY FAILURE of the cadence routes vs oracle: %s" % cad_par)
            init_filters()
            for s in range(W):
                step(s)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for c in range(cyc):
            for k in range(25):
                if k % 8 == 7:
                    step(k)
                else:
                    g.propagate(d_u[k % uniq], d_dt)
        torch.cuda.synchronize()
        tc = time.perf_counter() - tc
        cadence = {"imu_steps_per_s": B * 25 * cyc / tc, "frames_per_s": B * 3 * cyc / tc,
                   "what": "25 propagates : 3 frames of %d feature updates, %d cycles" % (N, cyc)}
        if cad_par is not None:
            cadence["parity_max_rel_err"] = cad_par["raw"]
            cadence["fused_parity_max_rel_err"] = cad_par["fused"]
            cadence["parity"] = cad_par["what"]
        # the same work with the IMU samples between two frames fused into the frame's launch (viekf_batch_step_n, K = 8, 8, 9)
        try:
            ks = (8, 8, 9)
            uu = [torch.stack([d_u[(j + 3 * q) % uniq] for j in range(k)]).contiguous() for q, k in enumerate(ks)]
            dd = [torch.stack([d_dt for _ in range(k)]).contiguous() for k in ks]
            g.step_n(uu[0], dd[0], d_z[0], d_slot, d_R, result=d_res)
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for c in range(cyc):
                for q in range(3):
                    g.step_n(uu[q], dd[q], d_z[q % uniq], d_slot, d_R, result=d_res)
            torch.cuda.synchronize()
            tf = time.perf_counter() - tf
            cadence["fused_imu_steps_per_s"] = B * 25 * cyc / tf
            cadence["fused_what"] = "3 launches of K = 8, 8, 9 propagates + %d updates (viekf_batch_step_n)" % N
        except Exception as e:
            cadence["fused_error"] = str(e)[:200]

    # ... and the reference's own CPU-runnable case (BASELINE configs[0]): ONE filter, N_feat = 12, step latency
    single = None
    if world == 1 and not args.no_secondary:
        try:
            sc1 = scene.make_scene(1, 12, 8, seed=5)
            g1 = v.BatchVIEKF(1, 12, sc1["params"], device=dev_index)
            g1.use_torch_stream()
            for i in range(12):
                g1.init_feature(sc1["pix"][:, i, :].copy(), np.full(1, np.nan))
            du, dz = torch.tensor(sc1["u"], device=dev), torch.tensor(sc1["z"], device=dev)
            ddt, dsl, dR1 = torch.tensor(sc1["dt"], device=dev), torch.tensor(sc1["slot"], device=dev), torch.tensor(sc1["R"], device=dev)
            for k in range(20):
                g1.step(du[k % 8], ddt, dz[k % 8], dsl, dR1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(400):
                g1.step(du[k % 8], ddt, dz[k % 8], dsl, dR1)
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t1) / 400
            single = {"ms_per_step": t1 * 1e3, "steps_per_s": 1.0 / t1, "what": "batch=1, N_feat=12, one fused launch per step"}
        except Exception as e:   # (a secondary number must never cost the primary one)
            single = {"error": str(e)[:200]}

    # ... and the DROP-IN route: the same filters driven through viekf_seq_propagate / _add_frame /
    # _handle_measurements at 250 Hz : 30 Hz with the camera's 30 ms delay -- every frame rewinds and replays (tools/seq_bench.py)
    seq_cad = None
    if world == 1 and not args.no_secondary:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import seq_bench
            pf = 0 if args.no_cpu_baseline else 5      # frames flown against oracle/seq_oracle.SeqOracle before the timed run
            seq_cad = {"shared_clock": seq_bench.run(B, N, False, 40, parity_frames=pf),
                       "independent_clocks": seq_bench.run(B, N, True, 24, parity_frames=pf)}
            if cadence is not None:
                seq_cad["shared_vs_raw_cadence"] = seq_cad["shared_clock"]["imu_steps_per_s"] / cadence["imu_steps_per_s"]
        except SystemExit:
            raise                          # (a parity failure of a timed route fails the run, like the primary leg's)
        except Exception as e:
            seq_cad = {"error": str(e)[:300]}

    # ... and the other single-GPU BASELINE configs (VERDICT r03 missing #3): configs[1] B=256, N=25 and configs[4] B=1024, N=150
    # (5 warm-up + 20 timed steps), each with its own algorithmic bytes, kernel description and in-run parity figure
    configs = None
    if world == 1 and not args.no_secondary and (B, N) == (1024, 50):
        configs = {}
        for name, cb, cn, ck, cw, psteps, pfilters in (("b256_n25", 256, 25, 200, 20, 3, 8), ("b1024_n150", 1024, 150, 20, 5, 1, 2)):
            try:
                configs[name] = secondary_config(v, scene, torch, dev, dev_index, cb, cn, ck, cw, psteps,
                                                 0 if args.no_cpu_baseline else pfilters)
            except SystemExit:
                raise
            except Exception as e:
                configs[name] = {"error": str(e)[:300]}

    # per-launch duration of the step's kernels from HIP events on the launch stream
    launch_ms = np.array([ev[i].elapsed_time(ev[i + 1]) / (marks[i + 1] - marks[i]) for i in range(len(ev) - 1)])
    launch_s = float(np.median(launch_ms)) * 1e-3   # (median over the groups of the groups' average launch duration)
    kernel_achieved = alg_bytes / launch_s / 1e9            # from the HIP-event launch duration
    achieved = rec["steps"] / secs * scene.algorithmic_bytes_per_step(N) / world / 1e9   # steps/s x B_alg, per GPU

    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (bench.py cannot run the
    # profiler on itself): FETCH_SIZE (x2, gfx950 rule) + WRITE_SIZE, see profiles/<round>/*_pmc_traffic.json
    traffic, traffic_src = None, None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc_traffic.json"))):
            j = json.load(open(f))
            if j["config"]["batch"] == B and j["config"]["n_feat"] == N and args.kernel in (0, 2):
                traffic, traffic_src = j.get("traffic_bytes_per_launch", j.get("traffic_bytes_per_step")), os.path.relpath(f, ROOT)
    except Exception:
        traffic = None

    if rank == 0:
        out = {
            "metric": "EKF steps/sec (IMU-rate propagate + N_feat updates), batch=%d, N_feat=%d" % (B, N),
            "value": rec["steps"] / secs,
            "unit": "EKF steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "prewarm_steps": prewarm,
            "no_prewarm": {"ms_per_step": secs_cold / K * 1e3, "value": B * world * K / secs_cold,
                           "what": "the same W warm-up + K timed steps run first, straight after set-up (no --prewarm-ms)"},
            "ms_per_step": secs / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "batch=%d filters per GPU, N_feat=%d (n=%d), static-hover synthetic IMU+pixels, "
                                   "1 propagate + %d active FEAT updates per step" % (B, N, 16 + 3 * N, N),
                       "batch_per_gpu": B, "n_feat": N, "parallelism": "filters sharded %d-way, no collective" % world,
                       "kernel_family": args.kernel},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "achieved_from": "value x B_alg(N) per GPU (SURVEY 8d); kernel_* = the same bytes / the median HIP-event launch duration",
                         "kernel_achieved": kernel_achieved, "kernel_frac": kernel_achieved / HBM_PEAK_GBS,
                         "kernel": ("%s; one fused launch per step, median HIP-event duration per launch %.4f ms"
                                    if desc.startswith(("k_step_resident", "k_step_tiles")) else
                                    "%s; one propagate and one update launch per step, median HIP-event duration of the pair %.4f ms")
                                   % (desc, launch_s * 1e3),
                         "alg_bytes_per_launch": alg_bytes},
            "nan_filters": n_bad,
            "gated_frac": work["gated_frac"],
            "timed_work": work,
            "reduction": rec,
        }
        out["flops"] = flops_block(N, B / launch_s)
        if cadence is not None:
            out["cadence_250_30"] = cadence
        if single is not None:
            out["single_filter_n12"] = single
        if seq_cad is not None:
            out["seq_cadence_250_30"] = seq_cad
        if configs is not None:
            out["configs"] = configs
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if not args.no_cpu_baseline:
            out["parity_max_rel_err"] = rec["max_rel_err"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
