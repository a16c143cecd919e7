"""Second, independent restatement of the reference's deterministic step in
numpy, transliterated from the Julia source statement by statement.  It exists
to cross-check epi_oracle.c (which streams the same arithmetic differently).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference outputs exist).

Reference: WhiffleFish/EpidemicModeling src/simulate.jl:99-151 (UpdateIsolations,
SimStep), :37-45 (incident_infections), :153-158 (reward); variates are passed
in (the three `rand` calls at simulate.jl:40,58,79 are replaced by arguments).
Julia Int = int64, Float64 = float64; `floor(Int, x)` = int(np.floor(x)).
"""
from __future__ import annotations

import numpy as np


def update_isolations(I, R, tests, sympt, pos_tests):
    """simulate.jl:99-129 with sympt / pos_tests supplied.  I: int64[H]; tests: int64[D+1, H]."""
    I = I.astype(np.int64).copy()
    tests = tests.astype(np.int64).copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        sympt_prop = sympt.astype(np.float64) / I.astype(np.float64)      # :104
    sympt_prop[np.isnan(sympt_prop)] = 0.0                                 # :105
    R = R + int(sympt.sum())                                               # :107
    I = I - sympt                                                          # :108
    tests[-1, :] = pos_tests                                               # :110
    tests = np.floor((1.0 - sympt_prop)[None, :] * tests.astype(np.float64)).astype(np.int64)   # :112
    R = R + int(tests[0, :].sum())                                         # :114
    I = I - tests[0, :]                                                    # :115
    neg = bool((I < 0).any())                                              # :117
    tests = np.roll(tests, (-1, 1), axis=(0, 1))                           # :122 circshift(tests, (-1, 1))
    tests[:, 0] = 0                                                        # :125
    tests[-1, :] = 0                                                       # :126
    return I, R, tests, neg


def sim_step(S, I, R, tests, a, N, sympt, pos_tests, gam):
    """simulate.jl:137-151.  Returns (S', I', R', tests', new_infections, I-went-negative)."""
    I, R, tests, neg = update_isolations(I, R, tests, sympt, pos_tests)    # :141
    R = R + int(I[-1])                                                     # :144
    I = np.roll(I, 1)                                                      # :145
    inf_sum = 0                                                            # :38
    for g in gam:                                                          # :39-41 (one variate per age bin)
        inf_sum = inf_sum + float(g)
    susceptible_prop = np.float64(S) / np.float64(N - R)                   # :42
    new_inf = int(np.floor(susceptible_prop * np.float64(inf_sum)))        # :44
    I = I.copy()
    I[0] = new_inf                                                         # :147
    S = S - new_inf                                                        # :148
    return S, I, R, tests, new_inf, neg


def num_tested(I, tests, a):
    """floor(Int, (inf - sum(tests[:, i])) * a) per age (simulate.jl:77-78)."""
    return np.floor((I - tests.sum(axis=0)).astype(np.float64) * np.float64(a)).astype(np.int64)


def reward(inf_loss, test_loss, testrate_loss, N, I_after, a, a_prev):
    """simulate.jl:153-158"""
    il = inf_loss * np.float64(int(I_after.sum())) / np.float64(N)
    return -(il + test_loss * a + testrate_loss * abs(a - a_prev))


def belief_mean(S, I, T, N):
    """updater.jl:5-21; Julia round = ties to even."""
    n = len(S)
    avgS = int(np.rint(np.float64(int(S.sum())) / n))
    avgI = np.rint(I.sum(axis=0).astype(np.float64) / n).astype(np.int64)
    avgR = N - (avgS + int(avgI.sum()))
    avgT = np.rint(T.sum(axis=0).astype(np.float64) / n).astype(np.int64)
    return avgS, avgI, avgR, avgT
