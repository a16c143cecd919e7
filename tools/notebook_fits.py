"""Developer tool: the ensemble fits of notebooks/ODEFIT_Demo.ipynb and MPC_Demo.ipynb (values the reference
printed under its native RNG) next to this build's fits over many seeds.  Run on a GPU box from the repository root."""
import json, sys
import numpy as np
import epidemicmodeling_amd as em
from epidemicmodeling_amd.odefit import Ensemble
from epidemicmodeling_amd.interface import random_states
from scipy.optimize import minimize

NSEED = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def simplex0(ens, kind, x0):
    x0 = np.asarray(x0, float)
    vals = [ens.loss(kind, x0)]
    for i in range(x0.size):
        y = x0.copy(); y[i] = 1.5 * y[i] + 0.025          # Optim.jl AffineSimplexer(a=0.025, b=0.5)
        vals.append(ens.loss(kind, y))
    return min(vals), vals[0]

def run(label, pomdp, kind, T, n, action, guess, ref):
    rows = []
    for seed in range(NSEED):
        b = random_states(pomdp, n, seed=1000 + seed)
        if action is None:
            act = np.random.default_rng(seed).random(n)
        else:
            act = action
        ens = Ensemble(b, T, act, seed=1000 + seed)
        b.close()
        s0, f0 = simplex0(ens, kind, guess)
        res = minimize(lambda x: ens.loss(kind, x), np.asarray(guess, float), method="Nelder-Mead",
                       options={"xatol": 1e-6, "fatol": 1e-10, "maxiter": 4000})
        rows.append([s0, f0, res.fun] + list(res.x))
        ens.close()
    a = np.array(rows)
    out = {"case": label, "ref": ref, "mean": a.mean(0).tolist(), "std": a.std(0).tolist(),
           "min": a.min(0).tolist(), "max": a.max(0).tolist(),
           "cols": ["simplex0_min", "f(x0)", "final"] + [f"p{i}" for i in range(len(guess))]}
    print(json.dumps(out), flush=True)

odefit = dict(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.65)
mpc = dict(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.50, LOD=5)
for lod in (6, 5):
    p = em.make_pomdp(LOD=lod, **odefit)
    run(f"ODEFIT SIR T20 N100 a0 LOD{lod}", p, "SIR", 20, 100, 0.0, [0.1, 0.1], {"final": 1.025157e-01, "p": [0.09972596, 0.32016542]})
    run(f"ODEFIT SEIR T20 N100 a0 LOD{lod}", p, "SEIR", 20, 100, 0.0, [0.1, 0.1, 0.1], {"final": 7.606587e-02, "p": [0.12889079, 0.42156765, 0.18205571]})
    run(f"ODEFIT SIR_CTRL T30 N500 LOD{lod}", p, "SIR_CTRL", 30, 500, None, [0.1, 0.1, 0.1], {"iter0": 4.019185e-01, "final": 1.355087e-01, "p": [0.08737892, 0.31515434, 0.33059201]})
    run(f"ODEFIT SEIR_CTRL T20 N500 LOD{lod}", p, "SEIR_CTRL", 20, 500, None, [0.1, 0.1, 0.1, 1.0, 1.0], {"iter0": 4.087847e-01, "final": 4.891439e-02, "p": [0.11194583, 0.37957389, 0.29718879, 0.42669492, -0.14235316]})
p = em.make_pomdp(**mpc)
run("MPC SIR_CTRL T30 N500", p, "SIR_CTRL", 30, 500, None, [0.1, 0.1, 0.1], {"iter0": 4.473752e-01, "iter50": 1.149938e-01})
run("MPC SEIR_CTRL T30? N500", p, "SEIR_CTRL", 30, 500, None, [0.1, 0.1, 0.1, 1.0, 1.0], {"iter0": 5.945e-01, "final": 7.541486e-02, "p": [0.11238919, 0.43659044, 0.29052273, 0.60854712, -0.18309989]})
