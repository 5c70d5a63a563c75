#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: robot-configs/sec (RNEA+ABA), 30-DoF humanoid, batch 4096 per GPU.

One "step" = one pass of the hot path over one batch of synthetic input.  Default workload (the metric's, BASELINE.json configs[2] with the
metric's RNEA+ABA pair): RNEA (q, qd, qdd -> tau) and ABA (q, qd, tau_in -> qdd) of the same 4096 configurations of the 30-DoF humanoid
(SixDoF pelvis + 24 revolute joints) through mh_rnea_aba_f64 -- one fused launch at this batch size (--separate: mh_rnea_f64 then
mh_aba_f64) -- with inputs resident in HBM before the timed region.  Multi-GPU: one process per GPU, no collective on the data path; the
model is broadcast once over RCCL before timing and the outputs are all-gathered once after it (reported separately as gather_ms).

  --config 3   configs[2]: humanoid RNEA + CRBA, fp64, 4096 per GPU                      (weak scaling)
  --config 4   configs[3]: humanoid ABA, fp64, 262 144 configurations sharded over the GPUs (strong scaling)
  --config 5   configs[4]: random 128-body tree, RNEA + ABA, fp32, 1 048 576 sharded       (strong scaling)
  (no flag)    the metric: humanoid RNEA + ABA, fp64, 4096 per GPU                       (weak scaling)

Timing: --ramp-ms (100) milliseconds of untimed steps for the device's clocks, W warm-up steps, then R (25; at least 5 for the long-running configurations) timed REGIONS of exactly K steps each, every region bracketed by barrier + synchronize on both
sides; `value` and `ms_per_step` come from the MEDIAN region (max over ranks per region), all regions are listed in `region_ms`.  Each is
followed by a region of the same K steps with HIP events recorded on the launch stream (`event_region_ms`): the roofline's kernel
durations come from those, `value` from the regions that carry nothing but the steps.  After
the timed regions the outputs the last step left in HBM are checked against the CPU oracle on a strided sample (`check`).

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
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MEASURED_COPY_GBS = 6290.0   # device copy kernel measured on this pool (profiles/r01_integrate_rates.txt)
CLOCK_HZ, N_SIMDS = 2.4e9, 1024  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs
REGIONS = 25  # 25 regions of the driver's 20 steps are 8 ms of the headline: the median of 25 rests on more than 1.6 ms of GPU time


def cpu_baseline(desc, q, qd, qdd, tau, gravity, jobs, target_s=20.0):
    """Oracle (CPU port) on the host cores: steps' worth of evaluations per second on a bounded sample of the same batch.  One thread per
    core, each walking its own passes (the C calls release the GIL; the oracle keeps its scratch thread-local); the single-thread rate is
    measured first and quoted in `sample`.  `jobs`: which oracle calls make one evaluation ("rnea", "aba", "crba")."""
    import threading
    from oracle.cpu_oracle import OracleModel
    om = OracleModel(desc)
    q, qd, qdd, tau = (np.asarray(x, dtype=np.float64) for x in (q, qd, qdd, tau))

    def one(lo, hi):
        if "rnea" in jobs:
            om.rnea(q[lo:hi], qd[lo:hi], qdd[lo:hi], gravity)
        if "aba" in jobs:
            om.aba(q[lo:hi], qd[lo:hi], tau[lo:hi], gravity)
        if "crba" in jobs:
            om.crba(q[lo:hi])

    n = min(256, len(q))
    one(0, n)  # warm
    t0 = time.perf_counter()
    reps1 = 0
    while time.perf_counter() - t0 < 2.0:
        one(0, n)
        reps1 += 1
    single = reps1 * n / (time.perf_counter() - t0)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a container's CPU quota (cgroup v2) is the number of cores it can really keep busy
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    # bounded: about target_s seconds of wall clock; every thread walks `per` configurations of the batch `reps` times
    per = min(len(q), 4096)
    reps = int(max(1, min(2048 // cores + 1, target_s * single / per)))
    done = [0] * cores

    def work(t):
        for _ in range(reps):
            one(0, per)
            done[t] += per

    threads = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    return {"value": total / dt, "unit": "configs/s", "cores": cores, "kind": "port",
            "sample": f"{total} evaluations of {'+'.join(jobs)} ({reps} passes over {per} configurations of the same batch per thread), "
                      f"oracle/mecano_oracle.c (C restatement in fp64, not Mecano/JVM), {cores} threads, {dt:.1f} s; one thread alone: {single:.0f} configs/s"}


def mecano_jvm_baseline(iterations=50000):
    """the REAL Mecano calculators timed on one host core by java/us/ihmc/mecano/hip/tools/MecanoGoldenVectorHarness.java
    (InverseDynamicsCalculatorTest.java:124-158's protocol on the 30-DoF humanoid).  Needs `java` / `javac` >= 17 on PATH and the Mecano,
    Euclid and EJML jars in $MECANO_CLASSPATH; returns None otherwise (this repository's image and its GPU box have no JVM)."""
    import shutil, subprocess, tempfile
    cp = os.environ.get("MECANO_CLASSPATH")
    if not (cp and shutil.which("java") and shutil.which("javac")):
        return None
    out = tempfile.mkdtemp(prefix="mh_mecano_")
    try:
        src = os.path.join(ROOT, "java", "us", "ihmc", "mecano", "hip", "tools", "MecanoGoldenVectorHarness.java")
        subprocess.run(["javac", "-cp", cp, "-d", out, src], check=True, timeout=300)
        p = subprocess.run(["java", "-cp", out + os.pathsep + cp, "us.ihmc.mecano.hip.tools.MecanoGoldenVectorHarness",
                            os.path.join(ROOT, "mecano_amd", "models", "humanoid30.json"), os.path.join(ROOT, "tests", "golden", "states_humanoid30.json"),
                            os.path.join(out, "mecano_humanoid30.json"), str(iterations)], check=True, timeout=600, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith('{"mecano_cpu_baseline"')][-1]
        b = json.loads(line)["mecano_cpu_baseline"]
        return {"value": b["rnea_aba_pair_including_frame_update"], "unit": "configs/s", "cores": 1, "kind": "reference",
                "sample": f"{iterations} fresh states of the 30-DoF humanoid, one thread, us.ihmc.mecano InverseDynamicsCalculator + ForwardDynamicsCalculator "
                          f"+ updateFramesRecursively per configuration (rnea alone {b['rnea']:.0f}/s, aba alone {b['aba']:.0f}/s)"}
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def committed_traffic(fused_launch, B):
    """Fallback for `roofline.traffic`: the rocprofv3 PMC passes committed under profiles/ (same workload, same correction); only quoted
    for the configuration they were collected on, else null."""
    for name in ("r03_final_zv_b4096_hbm_pmc.json",):  # (the bias-split launch that serves this call since round 3)
        path = os.path.join(ROOT, "profiles", name)
        if fused_launch and B == BATCH and os.path.exists(path):
            pmc = json.load(open(path))
            return (pmc.get("FETCH_SIZE_correction", 1.0) * pmc["FETCH_SIZE_KB_per_launch_mean"] + pmc["WRITE_SIZE_KB_per_launch_mean"]) * 1024.0, "profiles/" + name
    return None, None


def _pmc_pass(counters, B, pick, config=0):
    """One child process under `rocprofv3 --pmc <counters>` (nothing else enabled, the program directly after `--`) running a short form of
    this very command; returns {counter: mean value per launch} over the launches `pick(kernel_name, grid_size)` accepts, or None."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = tempfile.mkdtemp(prefix="mh_bench_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", MH_BENCH_PMC_INNER="1")
        subprocess.run([exe, "--pmc"] + list(counters) + ["-d", out, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                        "--steps", "20", "--warmup", "5", "--regions", "1", "--no-cpu-baseline", "--batch", str(B)] + (["--config", str(config)] if config else []),
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
        vals = {c: [] for c in counters}
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] in vals and pick(r["Kernel_Name"], int(r["Grid_Size"])):
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if any(len(v) < 10 for v in vals.values()):
            return None
        return {c: sum(v) / len(v) for c, v in vals.items()}
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _pmc_allowed():
    return not (os.environ.get("MH_BENCH_NO_PMC") or os.environ.get("MH_BENCH_PMC_INNER") or "rocprof" in os.environ.get("LD_PRELOAD", "")
                or any(k.startswith("ROCPROF") for k in os.environ))


def headline_kernel_pick(B):
    """The launch mh_rnea_aba_f64 issues for B configurations of the humanoid: the bias-split kernel (three workgroups of 256 threads per
    64 configurations, groups padded to blocks of eight) while they all fit the device, else the fused tree-split kernel (two)."""
    groups = (B + 63) // 64
    grids = {"spec_zv_kernel": 3 * ((groups + 7) // 8 * 8) * 256, "spec_fused_split_kernel": 2 * groups * 256}
    return lambda name, grid: any(k in name and grid == g for k, g in grids.items())


def config4_kernel_pick(B):
    """The launch mh_aba_f64 issues for a device-filling batch of the humanoid: the fused bias + inertia kernel, persistent workgroups (two per
    CU) looping over the groups of 64 configurations."""
    wgs = min((B + 63) // 64, 512)
    return lambda name, grid: "spec_zvf_kernel" in name and grid == wgs * 256


def live_traffic(B, config=0):
    """HBM bytes per launch of the dominant kernel, measured for THIS run: two child processes under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes, nothing else enabled) run a short form of this very command BEFORE this process touches the GPU; the
    launches of the benchmark's grid are picked out by kernel name and grid size (the create-time self-check launches the same kernel on
    197 configurations).  FETCH_SIZE counts half of the bytes read on this access pattern (profiles/r01_pmc_calibration_8B_per_lane.txt),
    WRITE_SIZE the bytes written.  None when rocprofv3 is missing, when this process already runs under a profiler, or on any failure."""
    if not _pmc_allowed():
        return None
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        got = _pmc_pass([counter], B, config4_kernel_pick(B) if config == 4 else headline_kernel_pick(B), config)
        if got is None:
            return None
        kb[counter] = got[counter]
    return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0


def _pmc_call_sum(counter, B, config, per_call, parts, min_grid=65536):
    """Bytes counter of a call that is several launches (config 5: one depth-first walk + six transposes): one child pass as in _pmc_pass;
    the counter summed over every BIG launch (>= min_grid threads: the create-time self-check runs the same kernels on a few hundred
    configurations) of the kernels in `parts`, divided by the number of big launches of `per_call` (one per call)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = tempfile.mkdtemp(prefix="mh_bench_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", MH_BENCH_PMC_INNER="1")
        subprocess.run([exe, "--pmc", counter, "-d", out, "-o", "pmc", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                        "--steps", "5", "--warmup", "2", "--regions", "1", "--ramp-ms", "0", "--no-cpu-baseline", "--batch", str(B), "--config", str(config)],
                       cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=200, check=True)
        total, calls = 0.0, 0
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter or int(r["Grid_Size"]) < min_grid:
                    continue
                if any(k in r["Kernel_Name"] for k in parts):
                    total += float(r["Counter_Value"])
                if per_call in r["Kernel_Name"]:
                    calls += 1
        return total / calls if calls >= 5 else None
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_traffic_of_call(B, config):
    """roofline.traffic of a configuration whose step is several launches or a launch not picked by grid size (3: the RNEA + CRBA launch; 5: the
    fused depth-first walk + the six AoS <-> SoA transposes): 2 x FETCH_SIZE + WRITE_SIZE over the call's kernels, per call."""
    if not _pmc_allowed():
        return None
    per_call, parts = {3: ("spec_rnea_crba_split_kernel", ("spec_rnea_crba_split_kernel",)),
                       5: ("aba_dfs_kernel", ("aba_dfs_kernel", "rnea_dfs_kernel", "rows_to_columns_kernel", "columns_to_rows_kernel", "transpose_kernel"))}[config]
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        got = _pmc_call_sum(counter, B, config, per_call, parts, 65536 if config == 5 else 64 * 256)
        if got is None:
            return None
        kb[counter] = got
    return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0


def live_issue_counters(B, config=0):
    """SQ counters of the dominant kernel for THIS run (one more child pass, same mechanism): wave-level VALU instructions, wave cycles and
    the cycles waves spent waiting, per launch.  None when unavailable."""
    if not _pmc_allowed():
        return None
    return _pmc_pass(["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAVES"], B, config4_kernel_pick(B) if config == 4 else headline_kernel_pick(B), config)


def self_launch(n):
    """One rank process per GPU, started from a parent that never touches the GPU (no torch import, no HIP call, no exec of an
    initialised process): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in each child's environment, rank 0's stdout (the JSON line) relayed,
    the first non-zero exit code returned.  An external launcher (torch.distributed.run) sets WORLD_SIZE itself and never gets here."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    for line in procs[0].stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = 0
    for p in procs:
        code = p.wait()
        rc = rc or code
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=0, choices=(0, 3, 4, 5), help="BASELINE.json configuration (1-based); default 0 = the metric")
    ap.add_argument("--batch", type=int, default=0, help="configurations per step: per GPU for the weak-scaling workloads, in total for --config 4 / 5")
    ap.add_argument("--ramp-ms", type=float, default=100.0, help="untimed steps for this long before the W warm-up steps (device clocks; 0: none)")
    ap.add_argument("--regions", type=int, default=0, help=f"timed regions of --steps steps each; the median is reported.  0 (default): {REGIONS}, fewer (>= 5) when 2 x regions x steps would take longer than ~12 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate", action="store_true", help="two calls per step (mh_rnea_f64, then mh_aba_f64 / mh_crba_f64) instead of mh_rnea_aba_f64 / mh_rnea_crba_f64")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` without a launcher: this process becomes a GPU-free parent of N fresh rank processes
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    # under an external launcher (torch.distributed.run) its world must be what was asked for -- checked before anything touches a GPU
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or unset WORLD_SIZE")

    # roofline.traffic measured for this run (N = 1, the metric's configuration), in child processes, before this one touches the GPU
    headline = args.gpus == 1 and args.config in (0, 4) and not args.separate
    pmc_batch = args.batch or (262144 if args.config == 4 else BATCH)
    pmc_bytes = live_traffic(pmc_batch, args.config) if headline else None
    call_bytes = None
    if args.gpus == 1 and args.config in (3, 5) and not args.separate:
        call_bytes = live_traffic_of_call(args.batch or (1048576 if args.config == 5 else BATCH), args.config)
    sq = live_issue_counters(pmc_batch, args.config) if headline else None

    import torch
    import torch.distributed as dist
    from mecano_amd import _lib, build, distributed as mdist, random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    from mecano_amd.multibody import MultiBodySystem

    # ---- N > 1 validates itself: one rank per GPU means as many visible devices as ranks, every rank on a device of
    # its own, and a communicator that counts --gpus ranks -- or the run exits non-zero instead of printing a line that measured one GPU
    # N times.  The one exception is explicit: MECANO_DIST_BACKEND=gloo (tests that rehearse the N > 1 path with ranks SHARING the box's GPU).
    shared_gpu_rehearsal = env_world > 1 and os.environ.get("MECANO_DIST_BACKEND") == "gloo"
    if env_world > 1 and not shared_gpu_rehearsal and torch.cuda.device_count() < env_world:  # (device_count() does not initialise the GPU)
        raise SystemExit(f"bench.py --gpus {env_world}: only {torch.cuda.device_count()} HIP device(s) visible; one rank per GPU needs {env_world} "
                         "(ranks folded onto one device would measure that device N times).  MECANO_DIST_BACKEND=gloo rehearses the path with shared devices.")
    rank, world, local_rank = mdist.init_from_env()
    assert world == args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if shared_gpu_rehearsal else local_rank)
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    pci = ":".join(f"{int(getattr(props, k)):02x}" for k in ("pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(props, k)) or None
    device_id = {"rank": rank, "host": os.uname().nodename, "device_index": torch.cuda.current_device(), "pci": pci,
                 "uuid": str(getattr(props, "uuid", "")) or None, "name": props.name}
    devices = [device_id]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, device_id)
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)  # what the communicator counts, not what the launcher was asked for
        if int(ones.item()) != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: the communicator counts {int(ones.item())} rank(s), world size {dist.get_world_size()}")
        where = [(d["host"], d["pci"] or d["uuid"] or d["device_index"]) for d in devices]
        if len(set(where)) != world and not shared_gpu_rehearsal:
            raise SystemExit(f"bench.py --gpus {args.gpus}: ranks share a device: {where}")
    if not os.path.exists(build.LIB):
        build.build_lib()

    # ---- workload
    cfg = args.config
    strong = cfg in (4, 5)
    if cfg == 5:
        sys_ = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
        total = args.batch or 1 << 20
        tdt, np_dt, dtype, word = torch.float32, np.float32, "f32", 4
    else:
        sys_ = rt.nextHumanoid(np.random.default_rng(MODEL_SEED))
        total = args.batch or (262144 if cfg == 4 else BATCH)
        tdt, np_dt, dtype, word = torch.float64, np.float64, "f64", 8
    # ---- model: built on rank 0, broadcast over RCCL (north_star: "RCCL broadcast of the model")
    # the humanoid is the committed model (mecano_amd/models/humanoid30.json = nextHumanoid(default_rng(43)) when it was generated)
    desc0 = (rt.humanoid30Desc() if cfg != 5 else sys_.toModelDesc()) if rank == 0 else None
    desc = mdist.broadcast_model_desc(desc0, src=0)
    model = HipModel(desc)
    if strong:
        lo, hi = mdist.shard_range(total, rank, world)
        B, B_total = hi - lo, total
    else:
        B, B_total = total, total * world
    model.reserve(B)
    # synthetic states (SURVEY.md section 8d distributions): at most 16384 distinct configurations are drawn on the host, bigger batches
    # tile them on the device (the kernels do not care; drawing 1 M x 362 doubles with numpy would take longer than the benchmark)
    base = min(B, 16384)
    q, qd, qdd, tau_in = rt.nextState(np.random.default_rng(STATE_SEED + rank), sys_, base)
    gravity = (0.0, 0.0, -9.81)
    reps = (B + base - 1) // base
    tq, tqd, tqdd, ttau = (torch.tensor(x, device="cuda", dtype=tdt).repeat(reps, 1)[:B].contiguous() for x in (q, qd, qdd, tau_in))
    stream = torch.cuda.current_stream().cuda_stream
    nq, nv = desc.nq, desc.nv
    bytes_rnea = bytes_aba = (nq + 3 * nv) * word  # inputs read once + outputs written once
    bytes_crba = (nq + nv * nv) * word

    # caller-owned output buffers, as the C-ABI prescribes (Mecano's calculators also write into pre-allocated matrices)
    tau, acc = torch.empty_like(tqd), torch.empty_like(tqd)
    fused = cfg in (0, 3, 5) and not args.separate
    fused_key = {0: "rnea_aba", 3: "rnea_crba", 5: "rnea_aba"}.get(cfg)
    Hm = torch.empty((B, nv, nv), dtype=tdt, device="cuda") if (fused and cfg == 3) else None
    fused_step = (model.bind_rnea_aba(tq, tqd, tqdd, ttau, tau, acc, gravity) if cfg != 3 else model.bind_rnea_crba(tq, tqd, tqdd, tau, Hm, gravity)) if fused else None
    jobs = {0: ("rnea", "aba"), 3: ("rnea", "crba"), 4: ("aba",), 5: ("rnea", "aba")}[cfg]
    outs = {}

    def run(job):
        if job == "rnea":
            outs["rnea"] = model.rnea(tq, tqd, tqdd, gravity)
        elif job == "aba":
            outs["aba"] = model.aba(tq, tqd, ttau, gravity)
        else:
            outs["crba"] = model.crba(tq)

    def step():
        if fused:
            fused_step()
        else:
            for job in jobs:
                run(job)

    # the device's clocks and caches first: the driver's five warm-up steps are 85 us of work, after which the regions still get faster
    # from one to the next (20 steps: 347 -> 337 us over nine regions; 331 flat after 80 ms of steps: profiles/r04_bench_warmup_sweep.txt)
    if args.ramp_ms > 0:
        t_end = time.perf_counter() + args.ramp_ms * 1e-3
        while time.perf_counter() < t_end:
            step()
        torch.cuda.synchronize()
    t_w = time.perf_counter()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    step_s_estimate = (time.perf_counter() - t_w) / max(1, args.warmup)
    # ---- timed regions: exactly K steps each between barrier + synchronize pairs.  They come in two kinds, interleaved: CLOCK regions carry
    # nothing but the K steps (`value`, `ms_per_step`, `region_ms`); EVENT regions additionally carry the HIP events on the launch stream
    # the roofline's kernel durations come from (`kernels_ms`, `event_region_ms`).  The events are instrumentation with a cost of their
    # own inside a region -- an event pair around 20 steps of the headline: 7-8 us of 340 (profiles/r04_region_overhead.txt) -- so the
    # figure the metric is quoted on is taken where they are absent, and the host clock of the event regions is printed beside it.
    K, R = args.steps, args.regions
    if R <= 0:  # every rank must loop the same number of times: rank 0 decides
        R = max(5, min(REGIONS, int(12.0 / max(1e-9, 2 * K * step_s_estimate))))
        if world > 1:
            r_t = torch.tensor([R], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(r_t, src=0)
            R = int(r_t.item())
    region_s, event_region_s, kernel_ms = [], [], {j: [] for j in ((fused_key,) if fused else jobs)}

    def region(with_events):
        # fused: ONE event pair brackets the K launches of a region (per-launch event pairs put ~5 us of markers between two ~25 us
        # kernels); average launch duration = elapsed / K, gaps included.  Otherwise one event pair per launch.
        t_all = HipTimer() if with_events else None
        per_launch = [[HipTimer() for _ in range(K)] for _ in jobs] if with_events and not fused else None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if with_events:
            t_all.start(stream)  # (ahead of the host clock: the event pair then brackets a superset of the K launches)
        t0 = time.perf_counter()
        for k in range(K):
            if fused or not with_events:
                step()
            else:
                for i, job in enumerate(jobs):
                    per_launch[i][k].start(stream)
                    run(job)
                    per_launch[i][k].stop(stream)
        if with_events:
            t_all.stop(stream)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0  # this rank's K steps, done; the closing barrier + synchronize follow, then the MAX over ranks
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if not with_events:
            region_s.append(elapsed)
            return
        event_region_s.append(elapsed)
        if fused:
            kernel_ms[fused_key].append(t_all.elapsed_ms() / K if K else 0.0)
        else:
            for i, job in enumerate(jobs):
                kernel_ms[job].append(float(np.mean([t.elapsed_ms() for t in per_launch[i]])) if K else 0.0)

    for r in range(R):
        region(False)
        region(True)
    elapsed = float(np.sort(region_s)[len(region_s) // 2])
    med = int(np.argsort(event_region_s)[len(event_region_s) // 2])
    kernels_ms = {j: v[med] for j, v in kernel_ms.items()}

    # ---- after the timed regions: one all-gather of the outputs over xGMI (north_star: "a final gather"), per-rank kernel times
    gather_ms, per_rank = None, None
    if world > 1:
        last = (acc if cfg == 0 else tau) if fused else outs.get("aba", outs.get("rnea"))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        full = mdist.all_gather_rows(last, B_total)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3
        assert full.shape[0] == B_total
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "batch": B, "kernels_ms": kernels_ms, "device": device_id})

    rccl_ranks = None
    if world > 1:  # what the communicator saw, not what the launcher was asked for
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl_ranks = {"world_size": dist.get_world_size(), "ranks_counted": int(ones.item())}
        if rccl_ranks["ranks_counted"] != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: {rccl_ranks}")
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- the outputs of the launches that were timed, against the CPU oracle on a strided sample (outside the timed regions).
    # fp64: ABSOLUTE 1e-10 on every entry (north_star's bar; what tests/test_gpu_parity.py asserts on configs 2-4).  fp32: the bounds of
    # tests/helpers.py (u = 2^-24, rounding errors of ~8 n operations adding up like a random walk): forward 4 sqrt(8 n) u max(1, |ref|)
    # for RNEA / CRBA; forward dynamics by its backward error, as the tests do: the fp64 inverse dynamics of the fp32 answer must
    # reproduce the given efforts within 16 sqrt(8 n) u (|tau| + |RNEA bias|) (its forward error is conditioning-bound and reported only).
    idx = np.arange(0, B, max(1, B // 64))[:64]
    check = {"rows": int(len(idx)), "tol": 1e-10 if word == 8 else None}
    got = {"rnea": tau if fused else outs.get("rnea"), "aba": acc if fused else outs.get("aba"), "crba": Hm if fused else outs.get("crba")}
    ok = True
    try:
        from oracle.cpu_oracle import OracleModel
        om = OracleModel(desc)
    except Exception as exc:  # no oracle build on this host (it is test infrastructure): the check is skipped, not failed
        om, check = None, {"ok": None, "skipped": True, "reason": f"oracle unavailable: {exc}"}  # null, never a truthy placeholder: an unverified run is not a passed one
    if om is not None:
        q64, qd64, qdd64, tau64 = (np.asarray(x[idx % base], dtype=np_dt).astype(np.float64) for x in (q, qd, qdd, tau_in))
        u32 = 2.0 ** -24
        for job in jobs:
            ref = {"rnea": lambda: om.rnea(q64, qd64, qdd64, gravity), "aba": lambda: om.aba(q64, qd64, tau64, gravity), "crba": lambda: om.crba(q64)}[job]()
            mine = got[job][torch.as_tensor(idx, device="cuda")].cpu().numpy().astype(np.float64)
            err = float(np.abs(mine - ref).max())
            check[f"max_err_{job}"] = err
            if word == 8:
                ok = ok and err <= 1e-10
            elif job == "aba":
                back = om.rnea(q64, qd64, mine, gravity)
                bias = om.rnea(q64, qd64, np.zeros_like(mine), gravity)
                check["backward_err_aba"] = float(np.abs(back - tau64).max())
                check["bound_aba"] = 16 * (8 * desc.n_joints) ** 0.5 * u32 * (float(np.abs(tau64).max()) + float(np.abs(bias).max()))
                ok = ok and np.isfinite(err) and check["backward_err_aba"] <= check["bound_aba"]
            else:
                check[f"bound_{job}"] = 4 * (8 * desc.n_joints) ** 0.5 * u32 * max(1.0, float(np.abs(ref).max()))
                ok = ok and err <= check[f"bound_{job}"]
        check["ok"] = bool(ok)

    fused_launch = fused and model.kernel_variant.startswith("topo:") and 2 * ((B + 63) // 64) <= 256  # one launch computes both outputs
    bytes_of = {"rnea": bytes_rnea, "aba": bytes_aba, "crba": bytes_crba, "rnea_aba": bytes_rnea + bytes_aba, "rnea_crba": bytes_rnea + bytes_crba}
    if fused:  # one launch computing both: 968 + 968 (config 3: 968 + 7448) algorithmic bytes per configuration
        dom, dom_name = fused_key, fused_key.replace("_", "+") + (" fused" if fused_launch else " (two launches)")
        if cfg == 5:  # mh_rnea_aba_f32 on a big AoS batch: transposed copies in, ONE depth-first walk for both algorithms, results transposed back
            dom_name = "rnea+aba, the whole call (one fused depth-first walk + six AoS <-> SoA transposes)"
    else:      # dominant kernel = the slowest launch of a step
        dom = max(kernels_ms, key=kernels_ms.get)
        dom_name = dom
    dom_ms, dom_bytes = kernels_ms[dom], bytes_of[dom]
    achieved = (dom_bytes * B) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    value = B_total * K / elapsed if elapsed > 0 else 0.0
    names = {0: "robot-configs/sec (RNEA+ABA), 30-DoF humanoid batch=4096", 3: "robot-configs/sec (RNEA+CRBA), 30-DoF humanoid batch=4096",
             4: "robot-configs/sec (ABA), 30-DoF humanoid batch=262144 sharded", 5: "robot-configs/sec (RNEA+ABA), random 128-body tree fp32 batch=1M sharded"}
    workloads = {0: "30-DoF humanoid (SixDoF pelvis + 24 revolute), RNEA and ABA of every configuration per step, fp64, AoS [B][n] state",
                 3: "30-DoF humanoid, RNEA and CRBA (30 x 30 mass matrix) of every configuration per step, fp64, AoS",
                 4: "30-DoF humanoid, ABA of every configuration per step, fp64, AoS, batch sharded over the GPUs",
                 5: "random 128-body tree (revolute / prismatic / 6-DoF joints), RNEA and ABA per step, fp32, AoS, batch sharded over the GPUs"}
    one_launch = fused_launch or (cfg == 4 and model.kernel_variant.startswith("topo:") and (B + 63) // 64 > 256)  # the counters were picked for that launch
    if pmc_bytes is not None and one_launch:
        traffic, traffic_source = pmc_bytes, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, child processes of this run), 2 x FETCH + WRITE"
    elif call_bytes is not None:
        traffic, traffic_source = call_bytes, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, child processes of this run), 2 x FETCH + WRITE summed over the kernels of one call"
    else:
        traffic, traffic_source = committed_traffic(fused_launch and cfg == 0, B)
    # The bounding roofline is HBM by north_star's reporting rule; what actually limits the launch is read off the counters: a launch
    # whose HBM traffic is near its algorithmic bytes and far below the bandwidth roof, with few waves per SIMD, is bound by one wave's
    # instruction issue.  valu_busy_frac prices a wave64 instruction at 4 cycles of its SIMD (1024 SIMDs, 2.4 GHz: MI355X_MICROARCH.md).
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy": achieved / MEASURED_COPY_GBS, "measured_copy": MEASURED_COPY_GBS,
                "traffic": traffic, "traffic_source": traffic_source, "bytes_per_config": dom_bytes, "launch_ms": dom_ms}
    if sq is not None and one_launch and dom_ms > 0:
        simd_cycles = dom_ms * 1e-3 * CLOCK_HZ * N_SIMDS
        roofline.update({"valu_insts_per_launch": sq["SQ_INSTS_VALU"], "waves_per_launch": sq["SQ_WAVES"],
                         "valu_busy_frac": 4.0 * sq["SQ_INSTS_VALU"] / simd_cycles,
                         "wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"] if sq["SQ_WAVE_CYCLES"] else None,
                         "issue_counters_source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES (child process of this run)"})
        few_waves = sq["SQ_WAVES"] <= N_SIMDS
        roofline["limited_by"] = ("instruction issue of single waves (at most one wave per SIMD, traffic far below the HBM roof)"
                                  if few_waves and roofline["frac"] < 0.25 else ("hbm" if roofline["frac"] >= 0.5 else "vector issue / latency"))
        if cfg == 4:  # launch geometry: persistent workgroups of four waves, two per CU (240 registers, 78 KB of LDS each), resident for the whole launch
            roofline["waves_per_simd"] = min((B + 63) // 64, 512) * 4 / N_SIMDS
    else:
        roofline["limited_by"] = None
    line = {
        "metric": names[cfg],
        "value": value, "unit": "configs/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3 if K else None, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": workloads[cfg],
                   "entry_point": f"mh_{fused_key}_{dtype}" if fused else " + ".join(f"mh_{j}_{dtype}" for j in jobs),
                   "batch_per_gpu": B, "global_batch": B_total, "nq": nq, "nv": nv, "bodies": desc.n_joints,
                   "parallelism": f"dp{world} (batch sharded, no data-path collective)", "kernel_variant": model.kernel_variant,
                   "model_seed": MODEL_SEED if cfg != 5 else 128, "state_seed": STATE_SEED},
        "ramp_ms": args.ramp_ms, "regions": R, "region_ms": [s * 1e3 for s in region_s], "reported_region": "median",
        "region_ms_min": min(region_s) * 1e3, "region_ms_max": max(region_s) * 1e3,
        "devices_distinct": (len({(d["host"], d["pci"] or d["uuid"] or d["device_index"]) for d in devices}) == world),
        "shared_gpu_rehearsal": bool(shared_gpu_rehearsal),
        "event_region_ms": [s * 1e3 for s in event_region_s],  # the same K steps with the HIP events of `kernels_ms` / `roofline` recorded around them
        "roofline": roofline,
        "kernels_ms": kernels_ms,
        "per_rank": per_rank,
        "rccl_ranks": rccl_ranks,
        "dist_backend": (dist.get_backend() if world > 1 else None),
        "gather_ms": gather_ms,
        "check": check,
    }
    # CPU baseline beside it: rank 0 at N = 1 only (a reported baseline, not the optimisation target)
    line["cpu_baseline"] = None
    if world == 1 and not args.no_cpu_baseline:
        jvm = mecano_jvm_baseline() if cfg == 0 else None  # the true Mecano figure when a JVM and the jars are on the box ...
        line["cpu_baseline"] = jvm or cpu_baseline(desc, q, qd, qdd, tau_in, gravity, jobs)  # ... else the C restatement ("port")
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"bench.py: the timed launches' outputs disagree with the oracle: {check}")


if __name__ == "__main__":
    main()
