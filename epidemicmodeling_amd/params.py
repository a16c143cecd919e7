"""Host-side model types: Action, State, CovidPOMDP.

Mirrors src/typedef.jl of WhiffleFish/EpidemicModeling (names, fields, keyword
arguments, error behaviour).  The reference constructor reads three CSV files
that are absent from its checkout (src/typedef.jl:88-96); the per-age tables
they produce are reconstructed here from the generator recipe the reference's
notebooks document (see DEFAULT_TABLES below) -- statistically equivalent
defaults, not the author's fitted values.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, replace
from typing import Optional, Sequence

import numpy as np

from . import _lib

# ---------------------------------------------------------------------------
# reconstructed per-age tables (infection age 1..14)
# ---------------------------------------------------------------------------
# pos_test_probs = prop_above_LOD(viral_loads, day, LOD) (src/init.jl:46-48).
# LOD=5: values printed by the reference in notebooks/InfectionSimDemo.ipynb
# (cell 1 output); LOD=6 (constructor default) and LOD=3: regenerated Monte
# Carlo of the same viral-load model.
POS_TEST_PROBS = {
    5: [0.0, 0.0, 0.0, 0.08586, 0.75496, 0.97208, 0.99638, 0.97604, 0.87264, 0.67308, 0.43986, 0.23942, 0.1146, 0.0513],
    6: [0.0, 0.0, 0.0, 0.03907, 0.5613, 0.8851, 0.93385, 0.82094, 0.60269, 0.37031, 0.19098, 0.08864, 0.03953, 0.01657],
    3: [0.0, 0.0, 0.0, 0.49845, 1.0, 1.0, 1.0, 1.0, 0.99994, 0.98088, 0.89607, 0.75397, 0.57889, 0.39119],
}
# Infdistributions = FitInfectionDistributions(df, 14, 50) (src/init.jl:9-35):
# Gamma(shape, scale) of secondary infections per infected individual and day.
INF_SHAPE = [1e-100, 1e-100, 0.02, 0.33005, 1.52492, 3.47748, 4.62062, 4.23519, 3.10213, 1.95824, 1.15777, 0.661063,
             0.353021, 0.168313]
INF_SCALE = [1e-100, 1e-100, 0.0465114, 0.299203, 0.213582, 0.122114, 0.0904923, 0.0863304, 0.0951885, 0.112771,
             0.131819, 0.146296, 0.156212, 0.165607]
SYMPTOM_MU, SYMPTOM_SIGMA = 1.644, 0.363          # LogNormal(1.644, 0.363), src/typedef.jl:100


def lognormal_cdf(x: float, mu: float = SYMPTOM_MU, sigma: float = SYMPTOM_SIGMA) -> float:
    if x <= 0.0:
        return 0.0
    return 0.5 * math.erfc(-(math.log(x) - mu) / (sigma * math.sqrt(2.0)))


@dataclass(frozen=True)
class Action:
    """src/typedef.jl:8-12"""
    testing_prop: float = 0.0


@dataclass
class State:
    """src/typedef.jl:23-29.  I has length horizon; Tests is (test_delay+1) x horizon."""
    S: int
    I: np.ndarray
    R: int
    Tests: np.ndarray
    prev_action: Action = field(default_factory=Action)

    def __post_init__(self):
        self.I = np.asarray(self.I, dtype=np.int64)
        self.Tests = np.asarray(self.Tests, dtype=np.int64)


def population(s: State) -> int:
    """src/typedef.jl:140-142"""
    return int(s.S + int(np.sum(s.I)) + s.R)


def Array(s: State) -> np.ndarray:
    """Base.Array(::State): [S, sum(I), R] (src/typedef.jl:147-149)"""
    return np.array([s.S, int(np.sum(s.I)), s.R], dtype=np.int64)


@dataclass(frozen=True)
class CovidPOMDP:
    """The parameter struct of src/typedef.jl:43-56.  Infdistributions is kept
    as the (shape, scale) pairs of its Gamma members; symptom_dist as the
    (mu, sigma) of its LogNormal."""
    symptom_dist: tuple
    Infdistributions: tuple            # ((shape, scale), ...) length = horizon
    symptomatic_isolation_prob: float
    asymptomatic_prob: float
    pos_test_probs: tuple
    test_delay: int
    N: int
    discount: float
    inf_loss: float
    test_loss: float
    testrate_loss: float
    test_period: int

    @property
    def horizon(self) -> int:
        return len(self.Infdistributions)

    def isolation_probs(self) -> np.ndarray:
        """q_i of src/simulate.jl:56-57."""
        mu, sg = self.symptom_dist
        out = np.empty(self.horizon)
        for i in range(1, self.horizon + 1):
            sp = lognormal_cdf(i, mu, sg) - lognormal_cdf(i - 1, mu, sg)
            out[i - 1] = sp * (1 - self.asymptomatic_prob) * self.symptomatic_isolation_prob
        return out

    def to_c(self) -> "_lib.epi_params":
        H = self.horizon
        if H > _lib.EPI_MAX_H:
            raise ValueError(f"horizon {H} > {_lib.EPI_MAX_H}")
        p = _lib.epi_params()
        p.H, p.D, p.N = H, self.test_delay, self.N
        p.inf_loss, p.test_loss, p.testrate_loss = self.inf_loss, self.test_loss, self.testrate_loss
        q = self.isolation_probs()
        for i in range(H):
            p.q[i] = float(q[i])
            p.p[i] = float(self.pos_test_probs[i])
            p.k[i] = float(self.Infdistributions[i][0])
            p.theta[i] = float(self.Infdistributions[i][1])
        return p


def make_pomdp(*, test_delay: int = 0, discount: float = 0.95, N: int = 1_000_000, inf_loss: float = 50.0,
               test_loss: float = 1.0, testrate_loss: float = 0.1, symptomatic_isolation_prob: float = 0.95,
               asymptomatic_prob: float = 0.40, LOD: float = 6, horizon: int = 14, test_period: int = 1,
               pos_test_probs: Optional[Sequence[float]] = None,
               Infdistributions: Optional[Sequence[tuple]] = None) -> CovidPOMDP:
    """Keyword constructor CovidPOMDP(; ...) of src/typedef.jl:78-116.

    infections_path / sample_size / viral_loads_path are replaced by the
    reconstructed tables; pass pos_test_probs / Infdistributions explicitly for
    any (LOD, horizon) the tables do not cover."""
    if Infdistributions is None:
        if horizon != 14:
            Infdistributions = wide_infdistributions(horizon)
        else:
            Infdistributions = list(zip(INF_SHAPE, INF_SCALE))
    if pos_test_probs is None:
        if LOD not in POS_TEST_PROBS:
            raise ValueError(f"no reconstructed pos_test_probs table for LOD={LOD}; pass pos_test_probs=")
        base = POS_TEST_PROBS[LOD]
        pos_test_probs = base if horizon == 14 else _extend_tail(base, horizon)
    if len(pos_test_probs) != len(Infdistributions):
        raise ValueError("pos_test_probs and Infdistributions differ in length")
    return CovidPOMDP((SYMPTOM_MU, SYMPTOM_SIGMA), tuple((float(a), float(b)) for a, b in Infdistributions),
                      float(symptomatic_isolation_prob), float(asymptomatic_prob),
                      tuple(float(x) for x in pos_test_probs), int(test_delay), int(N), float(discount),
                      float(inf_loss), float(test_loss), float(testrate_loss), int(test_period))


def unity_test_period(pomdp: CovidPOMDP) -> CovidPOMDP:
    """Same parameters with test_period = 1 (src/typedef.jl:122-138; the
    reference version passes a field that does not exist and cannot run)."""
    return replace(pomdp, test_period=1)


def _extend_tail(base: Sequence[float], horizon: int) -> list:
    """Extend a 14-day table to a longer horizon with a geometric tail."""
    out = list(base[:min(horizon, len(base))])
    ratio = base[-1] / base[-2] if base[-2] > 0 else 0.5
    while len(out) < horizon:
        out.append(out[-1] * ratio)
    return out


def wide_infdistributions(horizon: int) -> list:
    """Synthetic long-tail infectiousness for wide infection-age states: the
    14-day (shape, scale) table continued with geometrically decaying shape."""
    shapes = _extend_tail(INF_SHAPE, horizon)
    scales = list(INF_SCALE[:min(horizon, 14)])
    while len(scales) < horizon:
        scales.append(scales[-1])
    return list(zip(shapes, scales))


# ---------------------------------------------------------------------------
# initial states on the host (single states; device beliefs have their own
# generators).  src/typedef.jl:159-217
# ---------------------------------------------------------------------------
def State_from(I, params: CovidPOMDP, rng: Optional[np.random.Generator] = None) -> State:
    """State(I, params, rng) (src/typedef.jl:159-181): I is (mu, sigma) of a
    Normal whose truncation to [0, inf) every age bin is drawn from, a vector
    of ints, or a single int placed in the first bin."""
    N, H = params.N, params.horizon
    if isinstance(I, tuple) and len(I) == 2:
        rng = rng or np.random.default_rng()
        mu, sg = I
        draws = np.empty(H)
        for i in range(H):
            while True:
                x = rng.normal(mu, sg) if sg > 0 else mu
                if x >= 0:
                    break
            draws[i] = x
        I0 = np.rint(draws).astype(np.int64)
        assert I0.sum() <= N
    elif isinstance(I, (list, np.ndarray)):
        I0 = np.asarray(I, dtype=np.int64)
        assert (I0 >= 0).all() and I0.sum() <= N
    elif isinstance(I, (int, np.integer)):
        I0 = np.zeros(H, dtype=np.int64)
        I0[0] = I
    else:
        raise ValueError("`I` must be (mu, sigma), array of integers, or single integer")
    return State(int(N - I0.sum()), I0, 0, np.zeros((params.test_delay + 1, H), dtype=np.int64), Action(0.0))


def simplex_sample(n: int, m: float, rng: np.random.Generator) -> np.ndarray:
    """src/typedef.jl:183-188"""
    v = np.sort(np.concatenate([rng.random(n - 1) * m, [0.0, m]]))
    return np.diff(v)


def State_random(params: CovidPOMDP, rng: Optional[np.random.Generator] = None) -> State:
    """State(params, rng) (src/typedef.jl:196-217)."""
    rng = rng or np.random.default_rng()
    N, H = params.N, params.horizon
    S, inf, R = (int(x) for x in np.rint(simplex_sample(3, float(N), rng)))
    I = np.rint(simplex_sample(H, float(inf), rng)).astype(np.int64)
    leftover = N - (S + int(I.sum()) + R)
    mx = int(np.argmax([S, int(I.sum()), R]))
    if mx == 0:
        S += leftover
    elif mx == 1:
        I[-1] += leftover
    else:
        R += leftover
    return State(S, I, R, np.zeros((params.test_delay + 1, H), dtype=np.int64), Action(0.0))
