"""Particle filter sharded by particle index over the GPUs of one node: one
process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI) for the
exchange steps of an update.

Global slot m lives on rank g iff start[g] <= m < start[g+1], start[g] =
floor(n_total * g / G).  Random streams and resampling positions are addressed
by the GLOBAL slot, and weights are summed in fixed point, so the posterior is
bit-identical for every G.

Per update (reference: update(BootstrapFilter, b, a, o), call sites
src/POMDPsInterface.jl:56, src/MPC.jl:393):
  1. local fused weight + step                          (no communication)
  2. all-reduce MAX of the local maximum log weight     (1 double)
  3. local fixed-point weights + prefix sums
  4. all-gather of the local weight totals, sums of squares, step flags and the
     global maximum (4 values/rank); with the device library steps 1-4 are only
     enqueued (library and collectives share torch's current stream) and the host
     waits here, once per update
  5. every rank derives the same plan: shard g's particles are the ancestors of
     global slots [lo[g], lo[g+1])
  6. slots of that range owned by another rank: the resampled particles are
     packed and sent directly to the owner (pairwise send/recv, no ring);
     slots owned locally only get an ancestor index (the gather is folded into
     the next step kernel), selected while the exchange is in flight.
  7. with the device library the update returns while the exchange is still in
     flight: the NEXT update first steps the slots that were filled locally,
     then waits for the received blocks, places them and steps the rest, so a
     transfer over xGMI (64 bytes per particle that changes shard) is hidden
     behind the step kernel.  Anything else that reads the population settles
     the exchange first.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import as_i64p, check
from .params import CovidPOMDP


def shard_starts(n_total: int, world: int) -> List[int]:
    return [(n_total * g) // world for g in range(world + 1)]


class LibShard:
    """The local shard, held by libepi_gfx950.so on one device."""

    def __init__(self, pomdp: CovidPOMDP, n_local: int, n_total: int, slot0: int, device: int):
        from .interface import Context, ParticleCollection
        import torch
        self.torch = torch
        # a context of its own (not the process-wide cached one): the shard moves the context onto torch's stream, and
        # other collections on the same device must not follow it there
        self.ctx = Context(pomdp, device)
        self.lib = self.ctx.lib
        self.b = ParticleCollection(self.ctx, n_local, halo_cap=n_local if n_total > n_local else 0)
        check(self.lib.epi_belief_set_shard(self.b.handle, n_total, slot0))
        self.rows = self.lib.epi_state_rows(pomdp.horizon, pomdp.test_delay)
        self.comm_device = torch.device("cuda", device)
        # the library works on torch's current stream, where the RCCL collectives are ordered too: the phases of
        # an update are then only enqueued, and the host waits once per update (for the weight totals)
        self.stream = None
        self.adopt_current_stream()
        self.shares_stream = True

    def adopt_current_stream(self):
        """Called at the top of every update: the library's kernels must run on the stream the collectives of this
        update are ordered on, i.e. torch's current one (it changes under `with torch.cuda.stream(...)`)."""
        cur = self.torch.cuda.current_stream(self.comm_device)
        if self.stream is None or cur.cuda_stream != self.stream.cuda_stream:
            self.stream = cur
            check(self.lib.epi_ctx_set_stream(self.ctx.handle, C.c_void_p(cur.cuda_stream), 1))

    def close(self):
        """Frees the shard; the context goes back to its own stream before it is destroyed."""
        if self.b is not None:
            self.b.close()
            self.b = None
        if self.ctx is not None:
            self.lib.epi_ctx_set_stream(self.ctx.handle, None, 0)
            self.ctx.close()
            self.ctx = None

    def init_normal(self, mu, sigma, seed):
        check(self.lib.epi_belief_init_normal(self.b.handle, mu, sigma, seed))

    def upload(self, S, I, R, T=None):
        self.b.upload(S, I, R, T)

    def download(self):
        return self.b.download()

    def predict(self, a, o, seed, step) -> float:
        v = C.c_double()
        check(self.lib.epi_pf_predict(self.ctx.handle, self.b.handle, a, int(o), seed, step, C.byref(v)))
        return v.value

    def predict_dev(self, a, o, seed, step, lwmax_out):
        """lwmax_out: 1-element float64 device tensor that receives the local maximum log weight."""
        check(self.lib.epi_pf_predict_dev(self.ctx.handle, self.b.handle, a, int(o), seed, step, C.c_void_p(lwmax_out.data_ptr())))

    def predict_part_dev(self, a, o, seed, step, col0, count, lwmax_out):
        """The step over the shard's columns [col0, col0 + count); the part that completes the cover writes lwmax_out."""
        check(self.lib.epi_pf_predict_part_dev(self.ctx.handle, self.b.handle, a, int(o), seed, step, col0, count,
                                               C.c_void_p(lwmax_out.data_ptr())))

    def weigh_dev(self, lwmax_global, totals_out):
        """lwmax_global: 1-element device tensor; totals_out: 4-element device tensor (w, sum w^2, flags, lwmax)."""
        check(self.lib.epi_pf_weigh_dev(self.ctx.handle, self.b.handle, C.c_void_p(lwmax_global.data_ptr()),
                                        C.c_void_p(totals_out.data_ptr())))

    def weigh(self, lwmax):
        w, sq, fl = C.c_uint64(), C.c_double(), C.c_uint32()
        check(self.lib.epi_pf_weigh(self.ctx.handle, self.b.handle, lwmax, C.byref(w), C.byref(sq), C.byref(fl)))
        return w.value, sq.value, fl.value

    def plan(self, n_total, w_shard, seed, step):
        G = len(w_shard)
        w = (C.c_uint64 * G)(*w_shard)
        lo = (C.c_int64 * (G + 1))()
        check(self.lib.epi_pf_plan(n_total, G, w, seed, step, lo))
        return list(lo)

    def select_local(self, g, w_shard, seed, step, s0, s1):
        G = len(w_shard)
        w = (C.c_uint64 * G)(*w_shard)
        check(self.lib.epi_pf_select_local(self.ctx.handle, self.b.handle, G, g, w, seed, step, s0, s1))

    def export(self, g, w_shard, seed, step, s0, s1):
        G = len(w_shard)
        w = (C.c_uint64 * G)(*w_shard)
        buf = self.torch.empty(self.rows * (s1 - s0), dtype=self.torch.int32, device=self.comm_device)
        check(self.lib.epi_pf_export(self.ctx.handle, self.b.handle, G, g, w, seed, step, s0, s1, C.c_void_p(buf.data_ptr())))
        return buf

    def recv_buffer(self, count):
        return self.torch.empty(self.rows * count, dtype=self.torch.int32, device=self.comm_device)

    def import_(self, s0, s1, buf):
        if not self.shares_stream:
            self.torch.cuda.current_stream().synchronize()
        check(self.lib.epi_pf_import(self.ctx.handle, self.b.handle, s0, s1, C.c_void_p(buf.data_ptr())))

    def import_late(self, s0, s1, buf):
        """import_ after finish(): for the slots an update left waiting for their particles."""
        if not self.shares_stream:
            self.torch.cuda.current_stream().synchronize()
        check(self.lib.epi_pf_import_late(self.ctx.handle, self.b.handle, s0, s1, C.c_void_p(buf.data_ptr())))

    def finish(self, want_flags: bool = True) -> int:
        fl = C.c_uint32()
        check(self.lib.epi_pf_finish(self.ctx.handle, self.b.handle, C.byref(fl) if want_flags else None))
        return fl.value

    def sums(self) -> np.ndarray:
        p = self.ctx.pomdp
        out = np.zeros(2 + p.horizon + (p.test_delay + 1) * p.horizon, dtype=np.int64)
        check(self.lib.epi_belief_sums(self.ctx.handle, self.b.handle, as_i64p(out)))
        return out

    def timing(self, on: bool):
        check(self.lib.epi_ctx_timing(self.ctx.handle, int(on)))

    def timing_read(self):
        s, c = C.c_double(), C.c_int64()
        check(self.lib.epi_ctx_timing_read(self.ctx.handle, C.byref(s), C.byref(c)))
        return s.value, c.value


class ShardedFilter:
    """BootstrapFilter over a population sharded across the ranks of the default
    process group.  `shard` may be supplied (tests use a CPU stand-in with the
    LibShard interface); by default it is the device library."""

    def __init__(self, pomdp: CovidPOMDP, n_total: int, seed: int = 0x5EED, device: int = 0, shard=None,
                 overlap: Optional[bool] = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.pomdp, self.n_total, self.seed = pomdp, int(n_total), seed
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.starts = shard_starts(self.n_total, self.world)
        self.slot0, self.n_local = self.starts[self.rank], self.starts[self.rank + 1] - self.starts[self.rank]
        if self.n_local < 1:
            raise ValueError("fewer particles than ranks")
        self.shard = shard if shard is not None else LibShard(pomdp, self.n_local, self.n_total, self.slot0, device)
        self.comm_device = self.shard.comm_device
        # the gloo backend moves host memory only: device tensors are staged through the host around each call
        # (rehearsals of the multi-rank path on a machine with fewer GPUs than ranks; RCCL takes them directly)
        self.stage = dist.get_backend() == "gloo" and torch.device(self.comm_device).type == "cuda"
        self.step = 0
        self.frac_bits = _frac_bits(self.n_total)
        # overlap: leave an update while its particle exchange is in flight and step around it in the next one
        self.overlap = hasattr(self.shard, "predict_part_dev") if overlap is None else bool(overlap)
        self._pending = None            # (requests, receive blocks, locally filled columns, send blocks kept alive)
        self.updates_in_parts = 0       # updates whose step ran around an exchange in flight

    # -- population ---------------------------------------------------------
    def init_normal(self, mu: float, sigma: float):
        self._pending = None
        self.shard.init_normal(mu, sigma, self.seed)

    def upload_local(self, S, I, R, T=None):
        self._pending = None
        self.shard.upload(S, I, R, T)

    def download_local(self):
        self.settle()
        return self.shard.download()

    def settle(self):
        """Completes the particle exchange the last update left in flight (a no-op otherwise)."""
        if self._pending is None:
            return
        reqs, recvs, _, _ = self._pending
        self._pending = None
        for req in reqs:
            req.wait()
        for r0, r1, buf in recvs:
            self.shard.import_late(r0, r1, buf.to(self.comm_device) if self.stage else buf)

    def _predict_around_the_exchange(self, a, o, seed, step, t):
        """The step of an update whose predecessor left its exchange in flight: the columns that were filled locally
        first (whole workgroups of 256), then the received blocks, then the rest."""
        sh, n = self.shard, self.n_local
        c0, c1 = self._pending[2]
        first = -(-c0 // 256) * 256
        last = c1 if c1 == n else (c1 // 256) * 256
        if last - first < max(256, n // 4):          # little to hide the transfer behind
            self.settle()
            sh.predict_dev(a, o, seed, step, t)
            return
        self.updates_in_parts += 1
        sh.predict_part_dev(a, o, seed, step, first, last - first, t)
        self.settle()
        if first > 0:
            sh.predict_part_dev(a, o, seed, step, 0, first, t)
        if last < n:
            sh.predict_part_dev(a, o, seed, step, last, n - last, t)

    def timing(self, on: bool):
        self.shard.timing(on)

    def timing_read(self):
        return self.shard.timing_read()

    # -- collectives (staged through the host for gloo) ---------------------
    def _all_reduce(self, t, op):
        if not self.stage:
            self.dist.all_reduce(t, op=op)
            return
        h = t.cpu()
        self.dist.all_reduce(h, op=op)
        t.copy_(h)

    def _all_gather(self, out, mine):
        if not self.stage:
            self.dist.all_gather_into_tensor(out, mine)
            return
        h = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(h, mine.cpu())
        out.copy_(h)

    # -- update -------------------------------------------------------------
    def update(self, a: float, o: int) -> dict:
        torch, dist = self.torch, self.dist
        G, g, seed, step = self.world, self.rank, self.seed, self.step
        sh = self.shard
        device_handover = hasattr(sh, "predict_dev")
        if hasattr(sh, "adopt_current_stream"):
            sh.adopt_current_stream()
        if device_handover:
            # 1-4 without a host round trip: the library's kernels and the collectives are ordered on one stream
            t = torch.empty(1, dtype=torch.float64, device=self.comm_device)
            if self._pending is not None:
                self._predict_around_the_exchange(a, o, seed, step, t)
            else:
                sh.predict_dev(a, o, seed, step, t)
            self._all_reduce(t, dist.ReduceOp.MAX)
            mine = torch.empty(4, dtype=torch.float64, device=self.comm_device)
            sh.weigh_dev(t, mine)
        else:
            # 1-2: fused weight + step, global maximum log weight
            t = torch.tensor([sh.predict(a, o, seed, step)], dtype=torch.float64, device=self.comm_device)
            self._all_reduce(t, dist.ReduceOp.MAX)
            lwmax = t.item()
            # 3-4: local fixed-point weights, totals of every shard
            w_local, sq_local, fl_local = sh.weigh(lwmax)
            # totals are < 2^53 and flags < 2^32: exact as doubles
            mine = torch.tensor([float(w_local), sq_local, float(fl_local), lwmax], dtype=torch.float64, device=self.comm_device)
        allv = torch.empty(4 * G, dtype=torch.float64, device=self.comm_device)
        self._all_gather(allv, mine)
        allv = allv.cpu().tolist()                                         # the update's one wait for the device
        w_shard = [int(allv[4 * h]) for h in range(G)]
        sumsq = math.fsum(allv[4 * h + 1] for h in range(G))
        lwmax = allv[3]
        flags = 0
        for h in range(G):
            flags |= int(allv[4 * h + 2])
        # 5: the plan (identical on every rank)
        lo = sh.plan(self.n_total, w_shard, seed, step)
        st = self.starts
        # 6: the ancestors of the slots that stay on this rank come first: with the device library that pass also
        # leaves the cumulative weights which the exports search
        s0, s1 = max(lo[g], st[g]), min(lo[g + 1], st[g + 1])
        if s1 > s0:
            sh.select_local(g, w_shard, seed, step, s0, s1)
        local = (s0 - st[g], s1 - st[g]) if s1 > s0 else (0, 0)
        # ... then the particles whose slot is owned by another rank
        ops, recvs, sends = [], [], []
        for h in range(G):
            if h == g:
                continue
            s0, s1 = max(lo[g], st[h]), min(lo[g + 1], st[h + 1])          # my particles -> slots of rank h
            if s1 > s0:
                blk = sh.export(g, w_shard, seed, step, s0, s1)
                sends.append(blk.cpu() if self.stage else blk)
                ops.append(dist.P2POp(dist.isend, sends[-1], h))
            r0, r1 = max(lo[h], st[g]), min(lo[h + 1], st[g + 1])          # rank h's particles -> my slots
            if r1 > r0:
                buf = sh.recv_buffer(r1 - r0)
                recvs.append((r0, r1, buf.cpu() if self.stage else buf))
                ops.append(dist.P2POp(dist.irecv, recvs[-1][2], h))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if self.overlap and device_handover and recvs:
            # leave with the exchange in flight (RCCL works on a stream of its own): the library records the slots that
            # still wait for their particles
            sh.finish(want_flags=False)
            self._pending = (reqs, recvs, local, sends)
        else:
            for req in reqs:
                req.wait()
            for r0, r1, buf in recvs:
                sh.import_(r0, r1, buf.to(self.comm_device) if self.stage else buf)
            if device_handover:
                sh.finish(want_flags=False)
            else:
                sh.finish()
        W = sum(w_shard)
        if W == 0:
            flags |= _lib.EPI_F_ZERO_W
        elif self.n_total > _lib.COARSE_W_BOUND * W:
            flags |= _lib.EPI_F_COARSE_W
        self.step += 1
        loglik = (lwmax + math.log(W / 2.0 ** self.frac_bits / self.n_total) - 0.5 * math.log(2 * math.pi)) if W else float("-inf")
        return {"loglik": loglik, "ess": (float(W) ** 2 / sumsq) if sumsq > 0 else 0.0, "flags": flags,
                "moved": sum(r1 - r0 for r0, r1, _ in recvs), "w_shard": w_shard, "lo": lo}

    def close(self):
        self.settle()
        if hasattr(self.shard, "close"):
            self.shard.close()

    # -- summaries ----------------------------------------------------------
    def mean(self):
        """mean(pc, pomdp) (src/updater.jl:5-25) of the whole sharded population:
        integer sums added over ranks, then rounded once."""
        torch, dist = self.torch, self.dist
        self.settle()
        sums = torch.from_numpy(self.shard.sums()).to(self.comm_device)
        self._all_reduce(sums, dist.ReduceOp.SUM)
        sums = sums.cpu().numpy()
        p = self.pomdp
        H, D1 = p.horizon, p.test_delay + 1
        n = self.n_total
        avgS = int(np.rint(np.float64(int(sums[0])) / n))
        avgI = np.rint(sums[2:2 + H].astype(np.float64) / n).astype(np.int64)
        avgR = p.N - (avgS + int(avgI.sum()))
        avgT = np.rint(sums[2 + H:].astype(np.float64) / n).astype(np.int64).reshape(H, D1).T
        if avgR < 0:
            raise ValueError("mean R < 0 (src/updater.jl:18 asserts)")
        return avgS, avgI, avgR, avgT


def _frac_bits(n_total: int) -> int:
    lg = 0
    while (1 << lg) < n_total:
        lg += 1
    return min(40, 53 - lg)


class NodeFilter:
    """BootstrapFilter over ONE population sharded across several devices of this host, driven by this process alone
    (epi_mctx of include/epi.h: the form a single-process host such as Julia uses; ShardedFilter above is the
    one-process-per-GPU form).  devices: one HIP device per shard; naming a device several times puts several shards on
    it.  rccl: the small exchanges go through RCCL instead of host memory (needs distinct devices)."""

    def __init__(self, pomdp: CovidPOMDP, n_total: int, devices, seed: int = 0x5EED, rccl: bool = False, overlap: bool = True):
        self.lib = _lib.load()
        self.pomdp, self.n_total, self.seed, self.step = pomdp, int(n_total), seed, 0
        self._cparams = pomdp.to_c()
        dev = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        check(self.lib.epi_mctx_create(dev, len(devices), C.byref(self._cparams), self.n_total, int(rccl), C.byref(self._h)))
        self.last_moved = 0
        if not overlap:         # every update places the particles it received before it returns
            check(self.lib.epi_mctx_set_overlap(self._h, 0))

    def init_normal(self, mu: float = 10_000.0, sigma: float = 5_000.0):
        check(self.lib.epi_mctx_init_normal(self._h, mu, sigma, self.seed))

    def upload(self, S, I, R, T=None):
        S, I, R = (np.ascontiguousarray(x, dtype=np.int64) for x in (S, I, R))
        T = None if T is None else np.ascontiguousarray(T, dtype=np.int64)
        check(self.lib.epi_mctx_upload(self._h, as_i64p(S), as_i64p(I), as_i64p(R), as_i64p(T)))

    def download(self):
        H, D1, n = self.pomdp.horizon, self.pomdp.test_delay + 1, self.n_total
        S, R = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
        I, T = np.empty((n, H), dtype=np.int64), np.empty((n, H, D1), dtype=np.int64)
        check(self.lib.epi_mctx_download(self._h, as_i64p(S), as_i64p(I), as_i64p(R), as_i64p(T)))
        return S, I, R, T

    def update(self, a: float, o: int) -> dict:
        """update(upd, b, a, o) (src/POMDPsInterface.jl:56) over all shards."""
        ll, ess, fl, mv = C.c_double(), C.c_double(), C.c_uint32(), C.c_int64()
        check(self.lib.epi_mpf_update(self._h, float(a), int(o), self.seed, self.step, C.byref(ll), C.byref(ess), C.byref(fl), C.byref(mv)))
        self.step += 1
        self.last_moved = mv.value
        return {"loglik": ll.value, "ess": ess.value, "flags": fl.value, "moved": mv.value}

    @property
    def overlapped_updates(self) -> int:
        """Updates so far whose step ran around particle blocks still in flight from the update before."""
        return int(self.lib.epi_mctx_overlapped_updates(self._h))

    def sums(self) -> np.ndarray:
        p = self.pomdp
        out = np.zeros(2 + p.horizon + (p.test_delay + 1) * p.horizon, dtype=np.int64)
        check(self.lib.epi_mctx_sums(self._h, as_i64p(out)))
        return out

    def mean(self):
        """mean(pc, pomdp) (src/updater.jl:5-25) of the whole population: (S, I[H], R, Tests[D+1, H])."""
        H, D1 = self.pomdp.horizon, self.pomdp.test_delay + 1
        S, R = C.c_int64(), C.c_int64()
        I, T = np.empty(H, dtype=np.int64), np.empty((H, D1), dtype=np.int64)
        check(self.lib.epi_mctx_mean(self._h, C.byref(S), as_i64p(I), C.byref(R), as_i64p(T)))
        return S.value, I, R.value, T.T.copy()

    def close(self):
        if self._h:
            self.lib.epi_mctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
