#!/usr/bin/env python3
"""The in-library sharded update (epi_mctx / epi_mpf_update) on the devices of this host: n particles as G shards,
K updates on the observations of bench.py.  With fewer devices than shards the shards share devices (what a one-GPU
box can show: the phases' own cost, not the transfers over xGMI).

usage: bench_node_filter.py [particles] [shards] [updates]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import epidemicmodeling_amd as em
from epidemicmodeling_amd.distributed import NodeFilter
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W = 3
ndev = max(torch.cuda.device_count(), 1)
pomdp = em.make_pomdp()
obs = bench.observations(pomdp, W + K)
f = NodeFilter(pomdp, n, [g % ndev for g in range(G)], seed=bench.SEED)
f.init_normal(10_000.0, 5_000.0)
for t in range(W):
    f.update(bench.ACTION, obs[t])
t0 = time.perf_counter()
moved = 0
for t in range(W, W + K):
    info = f.update(bench.ACTION, obs[t])
    moved += info["moved"]
dt = time.perf_counter() - t0
f.close()
print(f"node filter: {n} particles as {G} shard(s) on {min(G, ndev)} device(s): {dt / K * 1e3:.3f} ms per update = "
      f"{n * K / dt:.3e} particle-steps/s; {moved / K:.0f} particles changed shard per update; "
      f"loglik {info['loglik']:.12g} ess {info['ess']:.6g} flags {info['flags']}", flush=True)
