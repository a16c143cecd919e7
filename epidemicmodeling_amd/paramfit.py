"""Deterministic parameter pipeline: from the individual-level viral-load model
to the per-age tables of CovidPOMDP (host side, numpy).

The reference constructor (src/typedef.jl:88-99) reads two CSV files that its
checkout does not contain and turns them into tables:

    Infdistributions = FitInfectionDistributions(df, horizon, sample_size)   src/init.jl:9-35
    pos_test_probs   = [prop_above_LOD(viral_loads, day, LOD) ...]           src/init.jl:46-48

The files were produced by the Monte-Carlo recipe of src/Python/distUtils.py
(driven from notebooks/DistTransformsForUnimodality.ipynb, cells 1, 6, 12):
per individual a piecewise-linear log10 viral-load curve with random knots
(distUtils.py:6-11), infectiousness proportional to the load above 10^6
(distUtils.py:37-76), scaled so that the mean total is R0 (distUtils.py:32-35);
"Sample50.csv" holds means of 50 random individuals per row.

This module regenerates both tables from a seed -- vectorised over individuals,
with the closed-form integral of the triangular infectiousness curve where the
reference loops a trapezoid rule over its knots (the same value: the curve is
piecewise linear) -- and fits them the way src/init.jl does.  The Gamma fit
restates the estimator of Distributions.jl (`fit_mle(Gamma, x)`: Minka's
generalised Newton iteration on the shape, started at 0.5/(log mean - mean log),
tolerance 1e-16, at most 1000 iterations), which is not part of the reference
checkout.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
from scipy.special import digamma, polygamma

__all__ = ["viral_load_knots", "infectiousness_scale", "secondary_infections", "viral_loads", "sample_means",
           "DomainError", "fit_gamma_mle", "fit_exponential_weighted", "FitInfectionDistributions", "prop_above_LOD",
           "generate_tables", "write_csvs", "read_csv", "tables_from_csv", "CovidPOMDP_generated", "CovidPOMDP_from_csv"]


# ---------------------------------------------------------------------------
# the individual-level model (src/Python/distUtils.py)
# ---------------------------------------------------------------------------
def viral_load_knots(n: int, rng: np.random.Generator) -> np.ndarray:
    """n draws of (t0, tpeak, tf, Vpeak) (distUtils.py:6-11): onset of detectable
    load, time and height of the peak (log10 copies/ml), time the load is back
    at 10^6.  Returns an (n, 4) array."""
    t0 = rng.uniform(2.5, 3.5, n)
    tpeak = t0 + 0.2 + rng.gamma(1.8, 1.0, n)
    tf = tpeak + rng.uniform(5.0, 10.0, n)
    vpeak = rng.uniform(7.0, 11.0, n)
    return np.stack([t0, tpeak, tf, vpeak], axis=1)


def _triangle(knots: np.ndarray, alpha: float):
    """Infectiousness curve of an individual (distUtils.py:49-50): zero up to
    t0n (where the load crosses 10^6), rising linearly to alpha*(Vpeak-6)^+ at
    tpeak, falling linearly to zero at tf."""
    t0, tpeak, tf, vpeak = knots.T
    t0n = t0 + 3.0 * (tpeak - t0) / (vpeak - 3.0)
    h = alpha * np.maximum(vpeak - 6.0, 0.0)
    return t0n, tpeak, tf, h


def _triangle_cumulative(t, a, b, c, h):
    """Integral of the triangular curve with feet a < c, apex (b, h), from -inf to t."""
    up = np.clip(t, a, b) - a
    dn = np.clip(t, b, c) - b
    return h * (up * up / (2.0 * (b - a)) + dn - dn * dn / (2.0 * (c - b)))


def infectiousness_scale(R0: float, n: int, rng: np.random.Generator) -> float:
    """get_alpha (distUtils.py:32-35): the factor that makes the mean area under
    the unscaled infectiousness curve equal R0 (area, distUtils.py:18-30)."""
    a, b, c, h = _triangle(viral_load_knots(n, rng), 1.0)
    area = 0.5 * (b - a) * h + 0.5 * (c - b) * h
    return float(R0 / area.mean())


def secondary_infections(knots: np.ndarray, alpha: float, start_day: int = 0, end_day: int = 20) -> np.ndarray:
    """gen_inf_data (distUtils.py:110-125): expected secondary infections of each
    individual during day [i, i+1), i = start_day..end_day; shape (n, days)."""
    a, b, c, h = _triangle(knots, alpha)
    edges = np.arange(start_day, end_day + 2, dtype=np.float64)
    cum = _triangle_cumulative(edges[None, :], a[:, None], b[:, None], c[:, None], h[:, None])
    return np.diff(cum, axis=1)


def viral_loads(knots: np.ndarray, start_day: int = 0, end_day: int = 20, alpha: float = 1.0) -> np.ndarray:
    """get_trajectory_data / get_beta (distUtils.py:89-108): log10 viral load on
    the integer days; 0 until t0, 3 -> Vpeak on (t0, tpeak), then falling through
    3 at tf along the same line, floored at 0."""
    t0, tpeak, tf, vpeak = (knots[:, j][:, None] for j in range(4))
    vpeak = np.minimum(alpha * np.maximum(vpeak - 6.0, 0.0) + 6.0, vpeak)
    t = np.arange(start_day, end_day + 1, dtype=np.float64)[None, :]
    rise = 3.0 + (vpeak - 3.0) / (tpeak - t0) * (t - t0)
    fall = np.maximum(vpeak + (3.0 - vpeak) / (tf - tpeak) * (t - tpeak), 0.0)
    return np.where(t <= t0, 0.0, np.where(t < tpeak, rise, fall))


def sample_means(data: np.ndarray, n_rows: int, sample_size: int, rng: np.random.Generator,
                 chunk: int = 10_000) -> np.ndarray:
    """Rows of means over `sample_size` individuals drawn with replacement
    (DistTransformsForUnimodality.ipynb cell 6): what "Sample50.csv" holds."""
    out = np.empty((n_rows, data.shape[1]))
    for lo in range(0, n_rows, chunk):
        hi = min(lo + chunk, n_rows)
        idx = rng.integers(0, data.shape[0], size=(hi - lo, sample_size))
        out[lo:hi] = data[idx].mean(axis=1)
    return out


# ---------------------------------------------------------------------------
# the fits of src/init.jl
# ---------------------------------------------------------------------------
class DomainError(ValueError):
    """Raised where the Julia estimators throw DomainError (src/init.jl:16,22)."""


def fit_gamma_mle(x: np.ndarray, maxiter: int = 1000, tol: float = 1e-16):
    """Maximum-likelihood Gamma(shape, scale) as `fit(Gamma, x)` computes it
    (call site src/init.jl:13).  Data containing a zero make the mean logarithm
    -inf and the estimator fail: DomainError, as in the reference's fallback
    chain."""
    x = np.asarray(x, dtype=np.float64)
    if x.size == 0 or not np.all(x > 0.0) or not np.all(np.isfinite(x)):
        raise DomainError("Gamma MLE needs strictly positive data")
    mx = float(x.mean())
    logmx = math.log(mx)
    mlogx = float(np.log(x).mean())
    gap = logmx - mlogx
    if not gap > 0.0:
        raise DomainError("degenerate sample")
    a = 0.5 / gap
    for _ in range(maxiter):
        a_old = a
        ia = 1.0 / a
        z = ia + (mlogx - logmx + math.log(a) - float(digamma(a))) / (a * a * (ia - float(polygamma(1, a))))
        a = 1.0 / z
        if not a > 0.0:
            raise DomainError("shape left the domain")
        if abs(a - a_old) <= tol:
            break
    return a, mx / a


def fit_exponential_weighted(x: np.ndarray, w: np.ndarray) -> float:
    """Scale of `fit(Exponential, x, w)` (call site src/init.jl:19): the weighted
    mean; a non-positive scale is a DomainError."""
    beta = float(np.sum(w * x) / np.sum(w))
    if not beta > 0.0:
        raise DomainError("Exponential scale must be positive")
    return beta


def FitInfectionDistributions(df: np.ndarray, horizon: int = 14, sample_size: int = 50) -> list:
    """src/init.jl:9-35.  df[:, day-1] is the column of infection age `day`.
    Returns [(shape, scale)] of the per-individual Gamma of each age:
    Gamma MLE of the sample means rescaled by the sample size; columns with
    zeros fall back to an Exponential fit weighted by min(log(1/x), 10) taken
    as Gamma(1/sample_size, .); all-zero columns to Gamma(1e-100, 1e-100)."""
    out = []
    for day in range(horizon):
        col = np.asarray(df[:, day], dtype=np.float64)
        try:
            shape, scale = fit_gamma_mle(col)
            out.append((shape / sample_size, scale * sample_size))
        except DomainError:
            try:
                with np.errstate(divide="ignore"):
                    w = np.minimum(np.log(1.0 / col), 10.0)
                beta = fit_exponential_weighted(col, w)
                out.append((1.0 / sample_size, beta * sample_size))
            except DomainError:
                out.append((1e-100, 1e-100))
    return out


def prop_above_LOD(viral: np.ndarray, day: int, LOD: float) -> float:
    """src/init.jl:46-48; `day` counts from 1 like the reference's column index."""
    return float(np.sum(viral[:, day - 1] > LOD) / viral.shape[0])


# ---------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------
def generate_tables(seed: int = 0, n: int = 100_000, R0: float = 2.5, horizon: int = 14, sample_size: int = 50,
                    LOD: float = 6, n_rows: Optional[int] = None, return_data: bool = False):
    """Seeded regeneration of the two inputs and their fits.

    Returns {"Infdistributions": [(shape, scale)]*horizon, "pos_test_probs": [p]*horizon, "alpha": a};
    with return_data also "infections" (n_rows x days sample means) and "viral_loads" (n x days)."""
    rng = np.random.default_rng(seed)
    days = max(20, horizon)
    alpha = infectiousness_scale(R0, n, rng)
    inf = secondary_infections(viral_load_knots(n, rng), alpha, 0, days)
    means = sample_means(inf, n_rows or n, sample_size, rng)
    vl = viral_loads(viral_load_knots(n, rng), 0, days)
    out = {"Infdistributions": FitInfectionDistributions(means, horizon, sample_size),
           "pos_test_probs": [prop_above_LOD(vl, d, LOD) for d in range(1, horizon + 1)],
           "alpha": alpha}
    if return_data:
        out["infections"], out["viral_loads"] = means, vl
    return out


def write_csvs(infections: np.ndarray, viral: np.ndarray, infections_path: str, viral_loads_path: str) -> None:
    """The two files in the layout the reference constructor reads (a header row
    of column numbers, one row per sample; DistTransformsForUnimodality.ipynb
    cell 12)."""
    for path, arr in ((infections_path, infections), (viral_loads_path, viral)):
        np.savetxt(path, arr, delimiter=",", header=",".join(str(j) for j in range(arr.shape[1])), comments="",
                   fmt="%.17g")


def read_csv(path: str) -> np.ndarray:
    return np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)


def tables_from_csv(infections_path: str, viral_loads_path: str, horizon: int = 14, sample_size: int = 50,
                    LOD: float = 6) -> dict:
    """The table part of the reference constructor (src/typedef.jl:95-99) for a
    user who has the two files."""
    df, vl = read_csv(infections_path), read_csv(viral_loads_path)
    return {"Infdistributions": FitInfectionDistributions(df, horizon, sample_size),
            "pos_test_probs": [prop_above_LOD(vl, d, LOD) for d in range(1, horizon + 1)]}


def CovidPOMDP_generated(seed: int = 0, n: int = 100_000, R0: float = 2.5, LOD: float = 6, horizon: int = 14,
                         sample_size: int = 50, **kw):
    """CovidPOMDP(; ...) (src/typedef.jl:78-116) with its two data files replaced
    by the seeded regeneration: any LOD and any horizon up to the 21 simulated
    days, where the built-in tables cover LOD 3, 5, 6 and 14 days."""
    from .params import make_pomdp
    t = generate_tables(seed=seed, n=n, R0=R0, horizon=horizon, sample_size=sample_size, LOD=LOD)
    return make_pomdp(LOD=LOD, horizon=horizon, pos_test_probs=t["pos_test_probs"],
                      Infdistributions=t["Infdistributions"], **kw)


def CovidPOMDP_from_csv(infections_path: str, viral_loads_path: str, sample_size: int = 50, LOD: float = 6,
                        horizon: int = 14, **kw):
    """CovidPOMDP(; infections_path, sample_size, viral_loads_path, ...) of
    src/typedef.jl:88-99, for a user who holds the reference's data files."""
    from .params import make_pomdp
    t = tables_from_csv(infections_path, viral_loads_path, horizon, sample_size, LOD)
    return make_pomdp(LOD=LOD, horizon=horizon, pos_test_probs=t["pos_test_probs"],
                      Infdistributions=t["Infdistributions"], **kw)
