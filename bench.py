#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: robot-configs/sec (RNEA+ABA), 30-DoF humanoid, batch 4096 per GPU.

One "step" = one pass of the hot path over one batch of synthetic input: RNEA (q, qd, qdd -> tau) and ABA (q, qd, tau_in -> qdd)
of the same 4096 configurations of the 30-DoF humanoid (SixDoF pelvis + 24 revolute joints) through mh_rnea_aba_f64 -- one fused
launch at this batch size (--separate: mh_rnea_f64 then mh_aba_f64) -- with inputs resident in HBM before the timed region.  Multi-GPU: one process per GPU, every rank owns its own 4096
configurations (weak scaling), no collective on the data path; the model is broadcast once over RCCL before timing and
the outputs are all-gathered once after it (reported separately as gather_ms).

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  "roofline":     algorithmic bytes of the dominant kernel / its mean launch duration (HIP events on the launch stream)
  "cpu_baseline": the CPU oracle (a port, not Mecano/JVM) timed on the host cores (one thread each) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096                 # BASELINE.json metric: batch=4096 (per GPU; weak scaling)
MODEL_SEED, STATE_SEED = 43, 2342
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
BYTES_RNEA = 968             # (nq + 2 nv + nv) * 8 with nq = 31, nv = 30   
BYTES_ABA = 968


def cpu_baseline(desc, q, qd, qdd, tau, gravity, target_s=10.0):
    """Oracle (CPU port) on the host cores: RNEA+ABA pairs per second on a bounded sample of the same batch.  One thread per core,
    each walking its own contiguous slice (the C calls release the GIL; the oracle keeps its scratch thread-local); the
    single-thread rate is measured first and quoted in `sample`."""
    import threading
    from oracle.cpu_oracle import OracleModel
    om = OracleModel(desc)
    n = 256
    om.rnea(q[:n], qd[:n], qdd[:n], gravity)  # warm
    t0 = time.perf_counter()
    reps1 = 0
    while time.perf_counter() - t0 < 2.0:
        om.rnea(q[:n], qd[:n], qdd[:n], gravity)
        om.aba(q[:n], qd[:n], tau[:n], gravity)
        reps1 += 1
    single = reps1 * n / (time.perf_counter() - t0)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU quota (cgroup v2) is the number of cores it can really keep busy
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    # whole batches per thread, bounded: about target_s seconds of wall clock, at most 1024 batches in total
    reps = int(max(1, min(1024 // cores + 1, target_s * single / len(q))))
    done = [0] * cores

    def work(t):
        for _ in range(reps):
            om.rnea(q, qd, qdd, gravity)
            om.aba(q, qd, tau, gravity)
            done[t] += len(q)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    return {"value": total / dt, "unit": "configs/s", "cores": cores, "kind": "port",
            "sample": f"{total} RNEA+ABA pairs ({reps} passes over the same humanoid batch per thread), oracle/mecano_oracle.c (C restatement, not "
                      f"Mecano/JVM), {cores} threads, {dt:.1f} s; one thread alone: {single:.0f} configs/s"}


def measured_traffic(fused_launch, B):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE,
    separate passes, same command; FETCH_SIZE x 2 as calibrated on this access pattern, profiles/r01_pmc_calibration_8B_per_lane.txt);
    only quoted for the configuration they were collected on, else null."""
    path = os.path.join(ROOT, "profiles", "r01_fused_split_b4096_hbm_pmc.json")
    if not (fused_launch and B == BATCH and os.path.exists(path)):
        return None
    pmc = json.load(open(path))
    return (pmc.get("FETCH_SIZE_correction", 1.0) * pmc["FETCH_SIZE_KB_per_launch_mean"] + pmc["WRITE_SIZE_KB_per_launch_mean"]) * 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="configurations per GPU per step (default: the metric's 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate", action="store_true", help="two calls per step (mh_rnea_f64, mh_aba_f64) instead of mh_rnea_aba_f64")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mecano_amd import build, distributed as mdist, random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer

    rank, world, local_rank = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if not os.path.exists(build.LIB):
        build.build_lib()

    # ---- model: built on rank 0, broadcast over RCCL (north_star: "RCCL broadcast of the model")
    sys_ = rt.nextHumanoid(np.random.default_rng(MODEL_SEED))
    desc = mdist.broadcast_model_desc(sys_.toModelDesc() if rank == 0 else None, src=0)
    model = HipModel(desc)
    B = args.batch
    model.reserve(B)
    q, qd, qdd, tau_in = rt.nextState(np.random.default_rng(STATE_SEED + rank), sys_, B)
    gravity = (0.0, 0.0, -9.81)
    tq, tqd, tqdd, ttau = (torch.tensor(x, device="cuda") for x in (q, qd, qdd, tau_in))
    stream = torch.cuda.current_stream().cuda_stream

    # caller-owned output buffers, as the C-ABI prescribes (Mecano's calculators also write into pre-allocated matrices)
    tau, acc = torch.empty_like(tqd), torch.empty_like(tqd)
    fused_step = model.bind_rnea_aba(tq, tqd, tqdd, ttau, tau, acc, gravity)

    def step():
        if args.separate:
            return model.rnea(tq, tqd, tqdd, gravity), model.aba(tq, tqd, ttau, gravity)
        fused_step()
        return tau, acc

    for _ in range(args.warmup):
        tau, acc = step()
    # ---- timed region: exactly K steps between barrier + synchronize pairs; HIP events on the launch stream around every launch
    K = args.steps
    # fused: ONE event pair brackets the K launches of the timed region (per-launch event pairs put ~5 us of markers between
    # two ~30 us kernels); average launch duration = elapsed / K, gaps included.  --separate: an event pair per launch.
    t_all = HipTimer()
    t_a = [HipTimer() for _ in range(K)] if args.separate else []
    t_b = [HipTimer() for _ in range(K)] if args.separate else []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_all.start(stream)
    for k in range(K):
        if args.separate:
            t_a[k].start(stream)
            tau = model.rnea(tq, tqd, tqdd, gravity)
            t_a[k].stop(stream)
            t_b[k].start(stream)
            acc = model.aba(tq, tqd, ttau, gravity)
            t_b[k].stop(stream)
        else:
            fused_step()
    t_all.stop(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.separate:
        ms_a = float(np.mean([t.elapsed_ms() for t in t_a])) if K else 0.0
        ms_b = float(np.mean([t.elapsed_ms() for t in t_b])) if K else 0.0
    else:
        ms_a, ms_b = (t_all.elapsed_ms() / K if K else 0.0), 0.0

    # ---- after the timed region: one all-gather of the outputs over xGMI (north_star: "a final gather")
    gather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        full = mdist.all_gather_rows(acc, B * world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3
        assert full.shape[0] == B * world

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    fused_launch = (not args.separate) and model.kernel_variant.startswith("topo:") and 2 * ((B + 63) // 64) <= 256
    if args.separate:  # dominant kernel = the slower of the two launches of a step
        dom, dom_ms, dom_bytes = ("aba", ms_b, BYTES_ABA) if ms_b >= ms_a else ("rnea", ms_a, BYTES_RNEA)
        kernels_ms = {"rnea": ms_a, "aba": ms_b}
    else:              # one launch (fused at this batch size) computing both: 968 + 968 algorithmic bytes per configuration
        dom, dom_ms, dom_bytes = ("rnea+aba fused" if fused_launch else "rnea+aba (two launches)", ms_a, BYTES_RNEA + BYTES_ABA)
        kernels_ms = {"rnea_aba": ms_a}
    achieved = (dom_bytes * B) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    value = world * B * K / elapsed if elapsed > 0 else 0.0
    line = {
        "metric": "robot-configs/sec (RNEA+ABA), 30-DoF humanoid batch=4096",
        "value": value, "unit": "configs/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3 if K else None, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "30-DoF humanoid (SixDoF pelvis + 24 revolute), RNEA and ABA of every configuration per step, fp64, AoS [B][n] state",
                   "entry_point": "mh_rnea_f64 + mh_aba_f64" if args.separate else "mh_rnea_aba_f64",
                   "batch_per_gpu": B, "global_batch": B * world, "nq": desc.nq, "nv": desc.nv, "bodies": desc.n_joints,
                   "parallelism": f"dp{world} (batch sharded, no data-path collective)", "kernel_variant": model.kernel_variant,
                   "model_seed": MODEL_SEED, "state_seed": STATE_SEED},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(fused_launch, B),
                     "bytes_per_config": dom_bytes, "launch_ms": dom_ms},
        "kernels_ms": kernels_ms,
        "gather_ms": gather_ms,
    }
    # CPU baseline beside it: rank 0 at N = 1 only (a reported baseline, not the optimisation target)
    line["cpu_baseline"] = cpu_baseline(desc, q, qd, qdd, tau_in, gravity) if (world == 1 and not args.no_cpu_baseline) else None
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
