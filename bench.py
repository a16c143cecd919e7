#!/usr/bin/env python3
"""Benchmark of the fused particle-filter update (predict by SimStep + observation
weight + systematic resample) on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one belief update of the whole particle population.  The population
is 10^8 particles (the size the headline metric is quoted at) for every N:
sharded by index over the N ranks (strong scaling).  Inputs are synthetic and
generated on the device before the timed region; the timed region holds K
updates between barriers + device synchronisation, max over ranks.

Rank 0 prints one JSON line with the whole-job particle-steps/s, the roofline
of the dominant kernel (fused step kernel, HIP-event timed) and a CPU baseline
(the oracle restatement timed on this host, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak
SIMD_CLOCK_HZ = 2.4e9           # nominal engine clock
N_SIMD = 256 * 4                # 256 CUs x 4 SIMDs
# issue cycles of one wave64 vector instruction on its SIMD with the SIMD's other waves issuing too, by counter class
# (tools/ubench.hip: binary64 add / mul / fma 5.2-5.6, conversions 3-4, binary32 fma 4.7 (add and multiply 2.4), 64-bit
# integer multiply-add 7.0, 32-bit integer 2.5-2.8 (high half of a product 5.2), binary32 transcendentals ~9, binary64
# reciprocal ~16, compares / selects / moves 2.5-3.2)
VALU_ISSUE_CYCLES = {"valu_add_f64": 5.2, "valu_mul_f64": 5.6, "valu_fma_f64": 5.3, "valu_trans_f64": 16.0, "valu_cvt": 3.2,
                     "valu_fma_f32": 4.7, "valu_int64": 7.0, "valu_int32": 3.2, "valu_trans_f32": 9.0, "other": 2.9}
SEED = 0x5EED
ACTION = 0.2
import datetime
# a rank that never arrives (or a collective that never completes) ends the others after this long
GROUP_TIMEOUT = datetime.timedelta(seconds=float(os.environ.get("EPI_BENCH_GROUP_TIMEOUT", "900")))
NODE_FILTER_TIMEOUT = float(os.environ.get("EPI_BENCH_NODE_FILTER_TIMEOUT", "420"))


def alg_bytes_per_particle_step(H: int, D: int) -> int:
    """Compulsory traffic of one fused update with int32 struct-of-arrays state and
    fp64 weights: read prior once + write posterior once + write weight + read weight."""
    W = 2 + H + D * (H - 1)
    return 2 * 4 * W + 16


def step_kernel_name(H: int, D: int) -> str:
    """The instantiation the library launches for the weighing step of an update (launch_step in epi_gfx950.hip)."""
    delay = "true" if D > 0 else "false"
    if H == 14 and D == 0:
        return "k_step_coop<true, false, 14, true, true, 1>"
    if H < 18:
        return f"k_step_coop<true, {delay}, 0, true, false, 1>"
    return f"k_step_coop<true, {delay}, 0, true, false, {2 if H <= 28 else 4}>"


def observations(pomdp, count):
    """Observation sequence from one true trajectory stepped through the library."""
    import epidemicmodeling_amd as em
    true = em.initialbelief(pomdp, 1, seed=SEED ^ 0xABCDEF)
    obs = []
    for t in range(count):
        o, _, _ = em.step_batch(true, ACTION, seed=SEED ^ 0x7275E, step=t, want_r=False)
        obs.append(int(o[0]))
    true.close()
    return obs


_OMP_DEFAULT = None


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pomdp, seconds_target=12.0, threads=None, n=500_000):
    """The oracle's complete update (port of the reference algorithm, OpenMP) on a bounded sample."""
    import numpy as np
    from oracle import pyoracle
    orc = pyoracle.Oracle(pomdp.to_c())
    L = pyoracle.lib()
    global _OMP_DEFAULT
    if _OMP_DEFAULT is None:
        _OMP_DEFAULT = L.orc_max_threads()          # OMP_NUM_THREADS of this host, before any override
    threads = threads or _OMP_DEFAULT
    L.orc_set_threads(threads)
    st = orc.init_normal(n, 10_000.0, 5_000.0, SEED)
    _, mu, _ = orc.log_weights(st, ACTION, 0)
    o = int(np.median(mu))
    # the same days of the epidemic as the timed GPU updates: the belief starts again from the initial one every 20
    # updates (later days, with the infection dying out under one repeated observation, draw less and cost less)
    st0 = st
    t0 = time.perf_counter()
    updates = 0
    while True:
        if updates % 20 == 0:
            st = st0
        st, info = orc.pf_update(st, ACTION, o, SEED, updates % 20)
        updates += 1
        if time.perf_counter() - t0 > seconds_target:
            break
    dt = time.perf_counter() - t0
    return {"value": n * updates / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "host_cpu": host_cpu_model(), "host_nproc": os.cpu_count(),
            "sample": f"{updates} complete updates of {n} particles in runs of 20 from the initial belief "
                      f"(oracle/epi_oracle.c, OpenMP {threads} threads, default parameters, a={ACTION})"}


def run_single(args):
    import torch
    import epidemicmodeling_amd as em
    from epidemicmodeling_amd._lib import check

    pomdp = em.make_pomdp()
    n = args.particles
    obs = observations(pomdp, args.warmup + args.steps)
    b = em.initialbelief(pomdp, n, seed=SEED)
    ctx, lib = b.ctx, b.ctx.lib
    for t in range(args.warmup):
        check(lib.epi_pf_update_async(ctx.handle, b.handle, ACTION, obs[t], SEED, t))
    ll, ess, fl = C.c_double(), C.c_double(), C.c_uint32()
    check(lib.epi_pf_collect(ctx.handle, b.handle, C.byref(ll), C.byref(ess), C.byref(fl)))
    check(lib.epi_ctx_timing(ctx.handle, 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        check(lib.epi_pf_update_async(ctx.handle, b.handle, ACTION, obs[t], SEED, t))
    check(lib.epi_pf_collect(ctx.handle, b.handle, C.byref(ll), C.byref(ess), C.byref(fl)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ksum, kcnt = C.c_double(), C.c_int64()
    check(lib.epi_ctx_timing_read(ctx.handle, C.byref(ksum), C.byref(kcnt)))
    check(lib.epi_ctx_timing(ctx.handle, 0))
    if args.calibrate:
        # one dispatch of the plain gather kernel (reads ancestor index + every row, writes every row: the step
        # kernel's own access pattern with a known byte count) for calibrating the memory-side PMC counters
        b2 = em.initialbelief(pomdp, n, seed=SEED + 1)
        check(lib.epi_belief_copy(b2.handle, b.handle))
        torch.cuda.synchronize()
        b2.close()
    return pomdp, n, dt, ksum.value / max(kcnt.value, 1), dict(loglik=ll.value, ess=ess.value, flags=fl.value)


def run_node_filter_child(args):
    """--node-filter-child: the same population and observations through the library's own sharded update
    (epi_mctx_create + epi_mpf_update of include/epi.h: one host thread drives every device; the form a single-process
    host such as the reference's Julia session binds).  One shard per device 0 .. N-1 with the small exchanges over
    RCCL; EPI_BENCH_NODE_DEVICES="0,0" names the shards' devices instead (a repeated device: exchanges without RCCL),
    for boxes with fewer devices than shards.  Prints one JSON object."""
    import epidemicmodeling_amd as em
    from epidemicmodeling_amd.distributed import NodeFilter
    names = os.environ.get("EPI_BENCH_NODE_DEVICES")
    devices = [int(x) for x in names.split(",")] if names else list(range(args.gpus))
    rccl = len(set(devices)) == len(devices)
    pomdp = em.make_pomdp()
    n = args.particles
    obs = observations(pomdp, args.warmup + args.steps)
    f = NodeFilter(pomdp, n, devices, seed=SEED, rccl=rccl)
    f.init_normal(10_000.0, 5_000.0)
    for t in range(args.warmup):
        f.update(ACTION, obs[t])
    f.sums()                                        # places what the last warm-up update left in flight, and waits
    over0 = f.overlapped_updates
    moved = 0
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        info = f.update(ACTION, obs[t])
        moved += info["moved"]
    f.sums()                                        # the last update's particle exchange belongs to the timed region
    dt = time.perf_counter() - t0
    res = {"entry_points": "epi_mctx_create + epi_mpf_update (one process, one host thread)", "shards": len(devices),
           "devices": devices, "rccl_devices": len(devices) if rccl else 0, "particles": n, "steps": args.steps,
           "ms_per_step": dt / args.steps * 1e3, "value": n * args.steps / dt, "unit": "particle-steps/s",
           "moved": moved / args.steps, "overlapped_updates": f.overlapped_updates - over0,
           "diagnostics": dict(loglik=info["loglik"], ess=info["ess"], flags=info["flags"])}
    f.close()
    return res


def node_filter_block(args):
    """Rank 0, before it joins the process group (and before it touches a device): the library's one-process sharded
    update, timed in a child process of its own on the same N devices while the other ranks wait at the rendezvous.
    A failure there is reported inside the block and does not stop the benchmark."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    cmd = [sys.executable, "-u", os.path.abspath(__file__), "--node-filter-child", "--gpus", str(args.gpus),
           "--particles", str(args.particles), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=NODE_FILTER_TIMEOUT)
    except subprocess.TimeoutExpired:
        return {"error": f"no result after {NODE_FILTER_TIMEOUT:.0f} s"}
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        return {"error": f"exit status {p.returncode}", "stderr_tail": p.stderr.decode(errors="replace")[-600:]}
    return json.loads(lines[0])


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`bench.py --gpus N` started without a launcher: this process starts the N ranks itself (fresh interpreters,
    before anything here has touched a GPU), one device each, and hands on rank 0's JSON line.  The ranks are watched
    together: when one of them fails, or the whole run takes longer than EPI_BENCH_TIMEOUT seconds, the others (which
    would wait for it in a collective) are ended too.  What the other ranks print goes to stderr.  Returns the exit
    status for the parent."""
    import subprocess
    import tempfile
    port = free_port()
    limit = float(os.environ.get("EPI_BENCH_TIMEOUT", "1800"))
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EPI_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    t0 = time.monotonic()
    why = None
    while why is None:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c for c in codes):
            why = f"rank {[i for i, c in enumerate(codes) if c][0]} failed"
        elif time.monotonic() - t0 > limit:
            why = f"no result after {limit:.0f} s"
        else:
            time.sleep(0.1)
    if why is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    codes = [p.returncode for p in procs]
    if why is not None or any(codes):
        sys.stderr.write(f"bench.py: {why or 'a rank failed'}; rank exit codes {codes}\n")
        return 1
    out0.seek(0)
    lines = [ln for ln in out0.read().decode().splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        sys.stderr.write("bench.py: rank 0 did not print exactly one JSON line\n")
        return 1
    print(lines[0], flush=True)
    return 0


def init_group(args):
    """Join the process group of the launcher (torchrun or spawn_ranks) and check that it is the one asked for."""
    import torch
    import torch.distributed as dist
    if "RANK" not in os.environ:          # --sharded on one GPU without a launcher: a group of one
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    rank = int(os.environ["RANK"])
    # EPI_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # tensors are staged through the host); the measured configuration is nccl (RCCL), one GPU per rank
    backend = os.environ.get("EPI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1)
    if backend == "nccl":
        if ndev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} devices, this host shows {ndev}")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=GROUP_TIMEOUT)
    else:
        if ndev:
            torch.cuda.set_device(local)
        dist.init_process_group(backend, timeout=GROUP_TIMEOUT)
    world = dist.get_world_size()
    if world != args.gpus:
        dist.destroy_process_group()
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    return rank, world, backend, local


def run_launch_check(args):
    """--launch-check: the ranks only meet, agree on a number and leave (no device work): what a host without GPUs
    can verify of the multi-rank start-up."""
    import torch
    import torch.distributed as dist
    rank, world, backend, _ = init_group(args)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    ok = float(t[0]) == world * (world + 1) / 2
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        return {"launch_check": ok, "n_gpus": args.gpus, "world_size": world, "backend": backend}
    return None


def run_sharded(args):
    import torch
    import torch.distributed as dist
    import epidemicmodeling_amd as em
    from epidemicmodeling_amd.distributed import ShardedFilter

    rank, world, backend, local = init_group(args)
    pomdp = em.make_pomdp()
    n = args.particles
    obs = observations(pomdp, args.warmup + args.steps) if rank == 0 else None
    box = [obs]
    dist.broadcast_object_list(box, src=0)
    obs = box[0]
    f = ShardedFilter(pomdp, n, seed=SEED, device=local)
    f.init_normal(10_000.0, 5_000.0)
    for t in range(args.warmup):
        f.update(ACTION, obs[t])
    f.timing(True)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        info = f.update(ACTION, obs[t])
    f.settle()                                      # the last update's particle exchange belongs to the timed region
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    # an update that steps around an exchange in flight launches the step kernel in up to three parts: their sum
    ksum, _ = f.timing_read()
    tt = torch.tensor([dt, ksum / args.steps], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, kms = float(tt[0]), float(tt[1])
    dist.destroy_process_group()
    return pomdp, n, dt, kms, dict(loglik=info["loglik"], ess=info["ess"], flags=info["flags"]), rank, world, backend


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=100_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--sharded", action="store_true",
                    help="use the sharded (torch.distributed) update even with one GPU, to measure its host overhead")
    ap.add_argument("--calibrate", action="store_true",
                    help="after the timed region, run one gather dispatch of known traffic (PMC calibration)")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, check the process group and leave without device work")
    ap.add_argument("--node-filter-child", action="store_true",
                    help="(internal) time the library's one-process sharded update on devices 0 .. N-1 and print its block")
    ap.add_argument("--no-node-filter", action="store_true", help="with --gpus N > 1: skip the node_filter block")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.node_filter_child:
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)                                   # RCCL's banner and the like: to stderr
        res = run_node_filter_child(args)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        return

    # no launcher around us: start the ranks here, before this process touches a GPU
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    # libraries underneath (RCCL prints a version banner at start-up) write to stdout too: send everything to stderr
    # while the benchmark runs, so that stdout carries the JSON line and nothing else
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank, world, backend = 0, 1, "none (one process, one device)"
    node_filter = None
    if args.gpus > 1 and not args.launch_check and not args.no_node_filter and int(os.environ.get("RANK", "0")) == 0:
        node_filter = node_filter_block(args)
    if args.launch_check:
        res = run_launch_check(args)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        if res is not None:
            print(json.dumps(res), flush=True)
        return
    if args.gpus > 1 or args.sharded:
        pomdp, n, dt, kms, diag, rank, world, backend = run_sharded(args)
    else:
        pomdp, n, dt, kms, diag = run_single(args)
    if rank != 0:
        return
    H, D = pomdp.horizon, pomdp.test_delay
    balg = alg_bytes_per_particle_step(H, D)
    value = n * args.steps / dt
    n_launch = n // world                           # particles one launch of the fused kernel processes
    achieved = balg * n_launch / (kms * 1e-3) / 1e9 if kms > 0 else None
    # traffic and the instruction mix below are NOT measured in this run: they come from counter passes of the same
    # kernel committed under profiles/ (rocprofv3 --pmc cannot run inside the timed benchmark); `source` says which
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and H == 14 and D == 0:
        try:
            tj = json.load(open(tpath))
            traffic = tj["k_step_hbm_bytes_per_particle"] * n_launch
            traffic_source = f"profiles/pmc_traffic.json (committed counter pass, build {tj.get('build', 'unrecorded')}): bytes per particle x particles per launch"
        except Exception:
            traffic = None
    # The step kernel is bound by vector-instruction issue, not by HBM.  Next to the HBM roofline: the SIMD cycles one
    # wavefront's step takes (live kernel time at the nominal 2.4 GHz, 1024 SIMDs) against the issue cycles of the
    # vector instructions it executes (per-class counts from the SQ counter pass of tools/pmc_step.sh, committed as
    # profiles/pmc_valu.json, times the issue cost of each class measured by tools/ubench.hip on this chip).
    issue = None
    vpath = os.path.join(ROOT, "profiles", "pmc_valu.json")
    if kms > 0 and os.path.exists(vpath) and H == 14 and D == 0:
        try:
            vj = json.load(open(vpath))
            mix = vj["insts_per_wave_step"]
            cyc = kms * 1e-3 * SIMD_CLOCK_HZ * N_SIMD / (n_launch / 64.0)
            classed = sum(mix[k] for k in VALU_ISSUE_CYCLES if k != "other")
            need = sum(mix[k] * c for k, c in VALU_ISSUE_CYCLES.items() if k != "other")
            need += (mix["valu"] - classed) * VALU_ISSUE_CYCLES["other"]
            issue = {"bound": "valu-issue", "valu_insts_per_wave_step": mix["valu"], "simd_cycles_per_wave_step": cyc,
                     "issue_cycles_of_the_mix": need, "frac": need / cyc,
                     "source": f"kernel time: this run; instruction mix: profiles/pmc_valu.json (committed counter pass, build {vj.get('build', 'unrecorded')}); "
                               "issue cost per class: tools/ubench.hip"}
        except Exception:
            issue = None
    out = {
        "metric": "particle-steps/sec (whole node) at 10^8 particles; % of HBM roofline",
        "value": value, "unit": "particle-steps/s", "n_gpus": args.gpus, "world_size": world, "backend": backend, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "dtype_note": "int32 compartment counts in HBM; observation weights, resampling and the samplers' defining acceptance tests in binary64, "
                                       "sampler proposals (inversion recurrence, rejection hats, normal strips) in binary32", "data": "synthetic",
        "config": {"workload": "bootstrap particle-filter update (predict+weight+systematic resample)",
                   "particles": n, "H": H, "test_delay": D, "N": pomdp.N, "action": ACTION, "seed": SEED,
                   "sharding": f"by index over {args.gpus} rank(s)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": step_kernel_name(H, D), "kernel_ms": kms, "alg_bytes_per_particle_step": balg},
        "issue": issue,
        "diagnostics": diag,
    }
    if node_filter is not None:
        # `value` above is the one-process-per-GPU form (torch.distributed ranks, ShardedFilter); this block is the same
        # update through the C ABI's own sharded entry points, one process driving all N devices
        out["node_filter"] = node_filter
    if args.gpus == 1 and not args.no_cpu:
        out["cpu_baseline_single_thread"] = cpu_baseline(pomdp, seconds_target=4.0, threads=1, n=100_000)
        out["cpu_baseline"] = cpu_baseline(pomdp)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
