"""Ensemble fitting of compartment ODE models (src/ODEFit.jl:166-340) with the ensemble and the loss
on the device: the stochastic trajectories never leave HBM, a loss evaluation is one kernel, and the
Nelder-Mead search on the host only moves the model parameters.

    FitRandEnsemble(kind, T, trajectories, params, action)            src/ODEFit.jl:166-213
    FitRandControlledEnsemble(kind, T, trajectories, params)          src/ODEFit.jl:225-273
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from ._lib import as_f64p, as_i64p, check
from .interface import ParticleCollection, random_states, _DEFAULT_SEED
from .params import Action, CovidPOMDP

KINDS = {"SIR": 0, "SIR_CTRL": 1, "SEIR": 2, "SEIR_CTRL": 3}
DEFAULT_SUBSTEPS = 8


class Ensemble:
    """Histories of n trajectories over T days, resident on the device (epi_ensemble)."""

    def __init__(self, b: ParticleCollection, T: int, actions, seed: int = _DEFAULT_SEED, step0: int = 1):
        self.ctx = b.ctx
        n = b.n
        act = np.asarray(actions, dtype=np.float64)
        if act.ndim == 0:
            per_p, per_d, act = 0, 0, act.reshape(1)
        elif act.ndim == 1 and act.size == n:
            per_p, per_d = 1, 0
        elif act.ndim == 2 and act.shape == (n, T):
            per_p, per_d = 1, 1
        else:
            raise ValueError("actions must be a scalar, [n] or [n, T]")
        act = np.ascontiguousarray(act)
        self._h = C.c_void_p()
        flags = C.c_uint32()
        check(self.ctx.lib.epi_ensemble_simulate(self.ctx.handle, b.handle, T, as_f64p(act), per_p, per_d,
                                                 b.prev_action.testing_prop, seed, step0, C.byref(self._h), C.byref(flags)))
        self.n, self.T, self.flags = n, T, flags.value

    def close(self):
        if self._h:
            self.ctx.lib.epi_ensemble_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Array(self) -> np.ndarray:
        """Array(::Vector{SimHist}) (src/simulate.jl:291-297): int64[3, T, n]."""
        h = np.empty((self.n, self.T, 3), dtype=np.int64)
        check(self.ctx.lib.epi_ensemble_download(self._h, as_i64p(h)))
        return np.ascontiguousarray(np.transpose(h, (2, 1, 0)))

    def loss(self, kind: str, p: Sequence[float], substeps: int = DEFAULT_SUBSTEPS, per_trajectory: bool = False):
        """SIR_param_loss & co. (src/ODEFit.jl:281-340) of parameter vector p."""
        q = np.ascontiguousarray(np.asarray(p, dtype=np.float64))
        out = C.c_double()
        pt = np.empty(self.n) if per_trajectory else None
        check(self.ctx.lib.epi_ensemble_loss(self.ctx.handle, self._h, KINDS[kind], as_f64p(q), q.size, substeps,
                                             C.byref(out), as_f64p(pt)))
        return (out.value, pt) if per_trajectory else out.value


def _fit(ens: Ensemble, kind: str, first_guess, show_trace: bool, substeps: int):
    from scipy.optimize import minimize
    res = minimize(lambda x: ens.loss(kind, x, substeps), np.asarray(first_guess, dtype=np.float64), method="Nelder-Mead",
                   options={"disp": show_trace})
    return res, tuple(float(v) for v in res.x)


def FitRandEnsemble(kind: str, T: int, trajectories: int, params: CovidPOMDP, action: Action, show_trace: bool = False,
                    seed: int = _DEFAULT_SEED, device: int = 0, substeps: int = DEFAULT_SUBSTEPS):
    """FitRandEnsemble (src/ODEFit.jl:166-213): kind "SIR" or "SEIR", one action for all trajectories."""
    if kind not in ("SIR", "SEIR"):
        raise ValueError("Unrecognized model kind. Must be SIR or SEIR")
    b = random_states(params, trajectories, seed=seed, device=device)
    ens = Ensemble(b, T, action.testing_prop, seed=seed)
    b.close()
    try:
        return _fit(ens, kind, [0.1, 0.1] if kind == "SIR" else [0.1, 0.1, 0.1], show_trace, substeps)
    finally:
        ens.close()


def FitRandControlledEnsemble(kind: str, T: int, trajectories: int, params: CovidPOMDP, show_trace: bool = False,
                              seed: int = _DEFAULT_SEED, device: int = 0, substeps: int = DEFAULT_SUBSTEPS,
                              actions: Optional[np.ndarray] = None):
    """FitRandControlledEnsemble (src/ODEFit.jl:225-273): a random action per trajectory (rand(trajectories),
    :230), controlled SIR / SEIR models."""
    if kind not in ("SIR", "SEIR"):
        raise ValueError("Unrecognized model kind. Must be SIR or SEIR")
    if actions is None:
        actions = np.random.default_rng(seed).random(trajectories)
    b = random_states(params, trajectories, seed=seed, device=device)
    ens = Ensemble(b, T, np.asarray(actions, dtype=np.float64), seed=seed)
    b.close()
    try:
        guess = [0.1, 0.1, 0.1] if kind == "SIR" else [0.1, 0.1, 0.1, 1.0, 1.0]
        return _fit(ens, kind + "_CTRL", guess, show_trace, substeps)
    finally:
        ens.close()
