"""Developer tool: loss at the vertices of Optim.jl's initial Nelder-Mead simplex for the controlled ensemble fits
whose iteration-0 trace lines the reference's notebooks print (function value = lowest vertex, second column =
sqrt(var(f) * (n+1)/n))."""
import json, sys
import numpy as np
import epidemicmodeling_amd as em
from epidemicmodeling_amd.odefit import Ensemble
from epidemicmodeling_amd.interface import random_states

NSEED = int(sys.argv[1]) if len(sys.argv) > 1 else 10

def vertices(x0):
    x0 = np.asarray(x0, float)
    out = [x0]
    for i in range(x0.size):
        y = x0.copy(); y[i] = 1.5 * y[i] + 0.025
        out.append(y)
    return out

def run(label, pomdp, kind, T, n, guess, ref):
    rows = []
    for seed in range(NSEED):
        b = random_states(pomdp, n, seed=2000 + seed)
        ens = Ensemble(b, T, np.random.default_rng(seed).random(n), seed=2000 + seed)
        b.close()
        f = np.array([ens.loss(kind, v) for v in vertices(guess)])
        nvar = len(guess)
        rows.append(list(f) + [f.min(), np.sqrt(f.var(ddof=1) * (nvar + 1) / nvar)])
        ens.close()
    a = np.array(rows)
    print(json.dumps({"case": label, "ref": ref, "mean": np.round(a.mean(0), 4).tolist(), "std": np.round(a.std(0), 4).tolist()}), flush=True)

for lod in (6, 5, 3):
    p = em.make_pomdp(LOD=lod, symptomatic_isolation_prob=0.95, asymptomatic_prob=0.65)
    run(f"ODEFIT SIR_CTRL T30 N500 LOD{lod}", p, "SIR_CTRL", 30, 500, [0.1, 0.1, 0.1], [0.4019185, 0.1958017])
    run(f"ODEFIT SEIR_CTRL T20 N500 LOD{lod}", p, "SEIR_CTRL", 20, 500, [0.1, 0.1, 0.1, 1.0, 1.0], [0.4087847, 0.0749777])
p = em.make_pomdp(LOD=5, symptomatic_isolation_prob=0.95, asymptomatic_prob=0.50)
run("MPC SIR_CTRL T30 N500", p, "SIR_CTRL", 30, 500, [0.1, 0.1, 0.1], [0.4473752, 0.1004494])
run("MPC SEIR_CTRL T30 N500", p, "SEIR_CTRL", 30, 500, [0.1, 0.1, 0.1, 1.0, 1.0], [0.5945, 0.1210041])
