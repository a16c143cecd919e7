"""Host interface over the gfx950 library, shaped like the reference's
POMDPs.jl / ParticleFilters.jl surface (src/POMDPsInterface.jl, src/updater.jl,
src/simulate.jl of WhiffleFish/EpidemicModeling):

    gen(pomdp, s, a)                     POMDPs.gen            POMDPsInterface.jl:1-9
    observation(pomdp, s, a, sp)         POMDPs.observation    POMDPsInterface.jl:11-23
    BootstrapFilter(pomdp, n)            ParticleFilters       POMDPsInterface.jl:39
    update(upd, b, a, o)                 ParticleFilters       POMDPsInterface.jl:56
    ParticleCollection / n_particles / particles / rand
    initialbelief(pomdp, np)             updater.jl:1-3
    mean(pc, pomdp)                      updater.jl:5-25
    Simulate(T, state, params, action)   simulate.jl:175-196
    SimulateEnsemble(T, n, params, a)    simulate.jl:210-227
    Simulate(T, state, b, pomdp, planner) POMDPsInterface.jl:29-62

Every particle operation runs in libepi_gfx950.so; nothing here computes a
step or a weight on the host.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import as_f64p, as_i64p, check
from .params import Action, CovidPOMDP, State, unity_test_period

_DEFAULT_SEED = 0x5EED


class Context:
    """One device + stream + installed parameters (epi_ctx)."""

    def __init__(self, pomdp: CovidPOMDP, device: int = 0):
        self.lib = _lib.load()
        self.pomdp = pomdp
        self._h = C.c_void_p()
        check(self.lib.epi_ctx_create(device, C.byref(self._h)))
        self._cparams = pomdp.to_c()
        check(self.lib.epi_params_set(self._h, C.byref(self._cparams)))
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self.lib.epi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context_for(pomdp: CovidPOMDP, device: int = 0) -> Context:
    key = (pomdp, device)
    if key not in _contexts:
        _contexts[key] = Context(pomdp, device)
    return _contexts[key]


def _flatten_states(pomdp: CovidPOMDP, states: Sequence[State]):
    H, D1 = pomdp.horizon, pomdp.test_delay + 1
    n = len(states)
    S = np.fromiter((s.S for s in states), dtype=np.int64, count=n)
    R = np.fromiter((s.R for s in states), dtype=np.int64, count=n)
    I = np.empty((n, H), dtype=np.int64)
    T = np.empty((n, H, D1), dtype=np.int64)      # column-major (D+1) x H per particle
    for j, s in enumerate(states):
        if s.I.shape != (H,) or s.Tests.shape != (D1, H):
            raise ValueError("state shape does not match the model's horizon / test_delay")
        I[j] = s.I
        T[j] = s.Tests.T
    return S, np.ascontiguousarray(I), R, np.ascontiguousarray(T)


def _unflatten_states(pomdp: CovidPOMDP, S, I, R, T, prev_action: Action) -> List[State]:
    return [State(int(S[j]), I[j].copy(), int(R[j]), T[j].T.copy(), prev_action) for j in range(len(S))]


class ParticleCollection:
    """A belief: n particles resident on the device (ParticleCollection{State}).

    Value semantics: update() returns a new collection only if asked to
    (copy=True); by default the update is in place, because at 10^8 particles a
    history of beliefs (SimHist.beliefs, src/simulate.jl:21) does not fit."""

    def __init__(self, ctx: Context, n: int, halo_cap: int = 0):
        self.ctx = ctx
        self.n = int(n)
        self._h = C.c_void_p()
        self.prev_action = Action(0.0)
        check(ctx.lib.epi_belief_create(ctx.handle, self.n, int(halo_cap), C.byref(self._h)))

    # -- construction ------------------------------------------------------
    @classmethod
    def from_states(cls, pomdp: CovidPOMDP, states: Sequence[State], device: int = 0) -> "ParticleCollection":
        ctx = context_for(pomdp, device)
        b = cls(ctx, len(states))
        b.upload(*_flatten_states(pomdp, states))
        if states:
            b.prev_action = states[0].prev_action
        return b

    def upload(self, S, I, R, T=None):
        S = np.ascontiguousarray(S, dtype=np.int64)
        I = np.ascontiguousarray(I, dtype=np.int64)
        R = np.ascontiguousarray(R, dtype=np.int64)
        T = None if T is None else np.ascontiguousarray(T, dtype=np.int64)
        H, D1 = self.ctx.pomdp.horizon, self.ctx.pomdp.test_delay + 1
        if S.shape != (self.n,) or R.shape != (self.n,) or I.size != self.n * H or (T is not None and T.size != self.n * H * D1):
            raise ValueError("array sizes do not match the belief")
        check(self.ctx.lib.epi_belief_upload(self._h, as_i64p(S), as_i64p(I), as_i64p(R), as_i64p(T)))

    def download(self):
        """(S[n], I[n,H], R[n], T[n,H,D+1]) as int64 host arrays."""
        H, D1 = self.ctx.pomdp.horizon, self.ctx.pomdp.test_delay + 1
        S = np.empty(self.n, dtype=np.int64)
        R = np.empty(self.n, dtype=np.int64)
        I = np.empty((self.n, H), dtype=np.int64)
        T = np.empty((self.n, H, D1), dtype=np.int64)
        check(self.ctx.lib.epi_belief_download(self._h, as_i64p(S), as_i64p(I), as_i64p(R), as_i64p(T)))
        return S, I, R, T

    def copy(self) -> "ParticleCollection":
        out = ParticleCollection(self.ctx, self.n)
        check(self.ctx.lib.epi_belief_copy(out._h, self._h))
        out.prev_action = self.prev_action
        return out

    @property
    def particles(self) -> List[State]:
        return _unflatten_states(self.ctx.pomdp, *self.download(), self.prev_action)

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            self.ctx.lib.epi_belief_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def n_particles(b: ParticleCollection) -> int:
    return b.n


def particles(b: ParticleCollection) -> List[State]:
    return b.particles


def rand(b: ParticleCollection, seed: int = _DEFAULT_SEED, ctr: int = 0) -> State:
    """rand(rng, b): one particle uniformly at random."""
    p = b.ctx.pomdp
    H, D1 = p.horizon, p.test_delay + 1
    S, R = C.c_int64(), C.c_int64()
    I = np.empty(H, dtype=np.int64)
    T = np.empty((H, D1), dtype=np.int64)
    check(b.ctx.lib.epi_belief_sample(b.ctx.handle, b.handle, seed, ctr, C.byref(S), as_i64p(I), C.byref(R), as_i64p(T)))
    return State(S.value, I, R.value, T.T.copy(), b.prev_action)


def rand_many(b: ParticleCollection, count: int, seed: int = _DEFAULT_SEED, ctr0: int = 0) -> List[State]:
    """rand(rng, b, count): `count` particles uniformly at random in one device launch and one copy; element j is
    rand(b, seed, ctr0 + j).  (A tree search draws thousands of states from one belief, test/POMCPOWtest.jl:15-21.)"""
    p = b.ctx.pomdp
    H, D1 = p.horizon, p.test_delay + 1
    S = np.empty(count, dtype=np.int64); R = np.empty(count, dtype=np.int64)
    I = np.empty((count, H), dtype=np.int64)
    T = np.empty((count, H, D1), dtype=np.int64)
    check(b.ctx.lib.epi_belief_sample_many(b.ctx.handle, b.handle, seed, ctr0, count, as_i64p(S), as_i64p(I), as_i64p(R), as_i64p(T)))
    return [State(int(S[j]), I[j].copy(), int(R[j]), T[j].T.copy(), b.prev_action) for j in range(count)]


def initialbelief(pomdp: CovidPOMDP, np_: int, mu: float = 10_000.0, sigma: float = 5_000.0, seed: int = _DEFAULT_SEED,
                  device: int = 0) -> ParticleCollection:
    """initialbelief(pomdp, np) (src/updater.jl:1-3) with initialstate(pomdp)
    = State(Normal(mu, sigma), pomdp) as in test/POMCPOWtest.jl:8-10; particles
    are generated on the device."""
    b = ParticleCollection(context_for(pomdp, device), np_)
    check(b.ctx.lib.epi_belief_init_normal(b.handle, mu, sigma, seed))
    return b


def random_states(pomdp: CovidPOMDP, n: int, seed: int = _DEFAULT_SEED, device: int = 0) -> ParticleCollection:
    """n draws of State(params) (src/typedef.jl:196-217) generated on the device."""
    b = ParticleCollection(context_for(pomdp, device), n)
    check(b.ctx.lib.epi_belief_init_simplex(b.handle, seed))
    return b


def _prev_prop(b: "ParticleCollection") -> float:
    if b.prev_action is None:
        raise ValueError("the collection's last step used one action per particle, so it has no single previous action; "
                         "step it once with a common action, or pass a_prev where the call takes it")
    return b.prev_action.testing_prop


# ---------------------------------------------------------------------------
# generative model
# ---------------------------------------------------------------------------
def step_batch(b: ParticleCollection, a, a_prev=None, seed: int = _DEFAULT_SEED, step: int = 0,
               want_o: bool = True, want_r: bool = True):
    """continuous_gen (src/simulate.jl:160-166) over all particles of b, in place.
    a / a_prev: a float or an array of n floats.  Returns (o[n], r[n], flags)."""
    n = b.n
    a_arr = np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64)))
    per = int(a_arr.size == n and n > 1) if a_arr.size != 1 else 0
    if a_arr.size not in (1, n):
        raise ValueError("a must be a scalar or have one entry per particle")
    if a_prev is None:
        if b.prev_action is None:
            raise ValueError("the previous step used one action per particle: pass a_prev (one value per particle) explicitly")
        a_prev = b.prev_action.testing_prop
    ap_arr = np.ascontiguousarray(np.atleast_1d(np.asarray(a_prev, dtype=np.float64)))
    if ap_arr.size not in (1, n):
        raise ValueError("a_prev must be a scalar or have one entry per particle")
    pper = int(ap_arr.size == n and n > 1) if ap_arr.size != 1 else 0
    o = np.empty(n, dtype=np.int64) if want_o else None
    r = np.empty(n, dtype=np.float64) if want_r else None
    flags = C.c_uint32(0)
    check(b.ctx.lib.epi_step_batch(b.ctx.handle, b.handle, as_f64p(a_arr), per, as_f64p(ap_arr), pper, seed, step,
                                   as_i64p(o), as_f64p(r), None, C.byref(flags)))
    # after a step with one action per particle the collection has no single previous action any more
    b.prev_action = Action(float(a_arr[0])) if a_arr.size == 1 else None
    return o, r, flags.value


def gen(pomdp: CovidPOMDP, s: State, a: Action, seed: int = _DEFAULT_SEED, step: int = 0, device: int = 0):
    """POMDPs.gen(pomdp, s, a, rng) (src/POMDPsInterface.jl:1-9): test_period
    sub-steps under one action; returns (sp, o of the LAST sub-step, summed r).
    Sub-step i uses stream step * test_period + i."""
    b = ParticleCollection.from_states(pomdp, [s], device)
    o, rsum = _gen_in_place(pomdp, b, a, seed, step)
    sp = b.particles[0]
    b.close()
    return sp, o, rsum


def _gen_in_place(pomdp: CovidPOMDP, b: ParticleCollection, a: Action, seed: int, step: int):
    """gen on the first particle of a collection that stays on the device between calls:
    returns (o of the last sub-step, summed r)."""
    rsum, o = 0.0, 0
    for i in range(pomdp.test_period):
        ov, rv, _ = step_batch(b, a.testing_prop, seed=seed, step=step * pomdp.test_period + i)
        o, rsum = int(ov[0]), rsum + float(rv[0])
    return o, rsum


def observation(pomdp: CovidPOMDP, s: State, a: Action, sp: Optional[State] = None):
    """POMDPs.observation (src/POMDPsInterface.jl:11-23): (mean, std) of the
    Normal that moment-matches the sum of per-age Binomials, computed from the
    pre-step state s.  Host arithmetic on one state (used by planners); the
    filter computes the same moments on the device for every particle."""
    tot_mean, tot_var = 0.0, 0.0
    for i in range(pomdp.horizon):
        already = int(s.Tests[:, i].sum())
        nt = float(np.floor((int(s.I[i]) - already) * a.testing_prop))
        p = pomdp.pos_test_probs[i]
        tot_mean += nt * p
        tot_var += float(np.sqrt(nt * p * (1 - p))) ** 2
    return tot_mean, float(np.sqrt(tot_var))


# ---------------------------------------------------------------------------
# the bootstrap filter
# ---------------------------------------------------------------------------
@dataclass
class BootstrapFilter:
    """BootstrapFilter(pomdp, n) (ParticleFilters.jl; src/POMDPsInterface.jl:39)."""
    pomdp: CovidPOMDP
    n: int
    seed: int = _DEFAULT_SEED
    step: int = 0                      # advanced by every update
    last_loglik: float = float("nan")
    last_ess: float = float("nan")
    last_flags: int = 0


def update(upd: BootstrapFilter, b: ParticleCollection, a: Action, o: int, copy: bool = False) -> ParticleCollection:
    """update(upd, b, a, o): predict, weight by the observation likelihood,
    systematic resample (src/POMDPsInterface.jl:56).  In place unless copy=True."""
    if b.n != upd.n:
        raise ValueError("belief size differs from the filter's particle count")
    if upd.pomdp.test_period != 1:
        # the device update predicts with ONE SimStep; a filter on a model with test_period > 1 would predict through
        # gen's test_period sub-steps (src/POMDPsInterface.jl:1-9).  The reference's closed loop builds its filter on
        # unity_test_period(pomdp) (src/POMDPsInterface.jl:38-39): do the same.
        raise ValueError("BootstrapFilter needs a model with test_period == 1: build it on unity_test_period(pomdp)")
    if unity_test_period(upd.pomdp) != unity_test_period(b.ctx.pomdp):
        raise ValueError("the filter's model differs from the model the belief was created for")
    if copy:
        b = b.copy()
    ll, ess, flags = C.c_double(), C.c_double(), C.c_uint32()
    check(b.ctx.lib.epi_pf_update(b.ctx.handle, b.handle, a.testing_prop, int(o), upd.seed, upd.step,
                                  C.byref(ll), C.byref(ess), C.byref(flags)))
    upd.step += 1
    upd.last_loglik, upd.last_ess, upd.last_flags = ll.value, ess.value, flags.value
    b.prev_action = a
    return b


def mean(b: ParticleCollection, pomdp: Optional[CovidPOMDP] = None) -> State:
    """mean(pc, pomdp) (src/updater.jl:5-25)."""
    p = b.ctx.pomdp
    H, D1 = p.horizon, p.test_delay + 1
    S, R = C.c_int64(), C.c_int64()
    I = np.empty(H, dtype=np.int64)
    T = np.empty((H, D1), dtype=np.int64)
    check(b.ctx.lib.epi_belief_mean(b.ctx.handle, b.handle, C.byref(S), as_i64p(I), C.byref(R), as_i64p(T)))
    return State(S.value, I, R.value, T.T.copy(), b.prev_action)


def suminf_quantiles(b: ParticleCollection, q) -> np.ndarray:
    """Order statistics of sum(I) over the particles of b, computed on the device: for each q the smallest
    value with at least ceil(q n) particles at or below it."""
    qs = np.ascontiguousarray(np.atleast_1d(np.asarray(q, dtype=np.float64)))
    out = np.empty(qs.size, dtype=np.int64)
    for k in range(0, qs.size, 4):
        part = np.ascontiguousarray(qs[k:k + 4])
        res = np.empty(part.size, dtype=np.int64)
        check(b.ctx.lib.epi_belief_suminf_quantiles(b.ctx.handle, b.handle, as_f64p(part), part.size, as_i64p(res)))
        out[k:k + 4] = res
    return out


@dataclass
class BeliefSummary:
    """What a plot of a belief history needs of one day's belief (src/plots.jl:47): the mean state and a few
    order statistics of the infected total."""
    mean: State
    q: np.ndarray
    suminf_q: np.ndarray


# ---------------------------------------------------------------------------
# rollouts
# ---------------------------------------------------------------------------
@dataclass
class SimHist:
    """src/simulate.jl:12-22"""
    sus: np.ndarray
    inf: np.ndarray
    rec: np.ndarray
    N: int
    T: int
    pos_test: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))
    actions: list = field(default_factory=list)
    rewards: np.ndarray = field(default_factory=lambda: np.zeros(0))
    beliefs: list = field(default_factory=list)

    def Array(self) -> np.ndarray:
        """3 x T (src/simulate.jl:28-30)"""
        return np.stack([self.sus, self.inf, self.rec])


def rollout_batch(b: ParticleCollection, T: int, actions, seed: int = _DEFAULT_SEED, step0: int = 0,
                  want_hist: bool = True, want_rew: bool = True):
    """T days over every particle of b (in place).  actions: scalar, [T], [n] or [n, T].
    Returns hist int64[n, T, 4] (S, sum I, R before the day; sum(pos) of the day), rew[n, T], flags."""
    n = b.n
    act = np.asarray(actions, dtype=np.float64)
    if act.ndim == 0:
        per_p, per_d, act = 0, 0, act.reshape(1)
    elif act.ndim == 1 and act.size == T and not (act.size == n and n != T):
        per_p, per_d = 0, 1
    elif act.ndim == 1 and act.size == n:
        per_p, per_d = 1, 0
    elif act.ndim == 2 and act.shape == (n, T):
        per_p, per_d = 1, 1
    else:
        raise ValueError("actions must be a scalar, [T], [n] or [n, T]")
    act = np.ascontiguousarray(act)
    hist = np.empty((n, T, 4), dtype=np.int64) if want_hist else None
    rew = np.empty((n, T), dtype=np.float64) if want_rew else None
    flags = C.c_uint32()
    check(b.ctx.lib.epi_rollout_batch(b.ctx.handle, b.handle, T, as_f64p(act), per_p, per_d, _prev_prop(b),
                                      seed, step0, as_i64p(hist), as_f64p(rew), C.byref(flags)))
    return hist, rew, flags.value


def rollout_score(b: ParticleCollection, actions, rollouts_per_seq: int = 1, discount: Optional[float] = None,
                  seed: int = _DEFAULT_SEED, step0: int = 0):
    """Scores action sequences by rollouts from the belief b (unchanged): actions is [n_seq, T];
    every sequence gets rollouts_per_seq independent T-day rollouts, trajectory j starting from
    particle j mod n of b.  Returns (score[n_seq], flags): score = mean over the sequence's rollouts
    of sum_t discount^t reward_t (reward: src/simulate.jl:153-158; discount: src/typedef.jl:82).
    This is the batched form of the rollouts a tree-search or random-shooting planner makes
    (planner call site src/POMDPsInterface.jl:44)."""
    act = np.ascontiguousarray(np.asarray(actions, dtype=np.float64))
    if act.ndim != 2:
        raise ValueError("actions must be [n_seq, T]")
    n_seq, T = act.shape
    disc = b.ctx.pomdp.discount if discount is None else float(discount)
    score = np.empty(n_seq, dtype=np.float64)
    flags = C.c_uint32()
    check(b.ctx.lib.epi_rollout_score(b.ctx.handle, b.handle, n_seq, int(rollouts_per_seq), T, as_f64p(act),
                                      _prev_prop(b), disc, seed, step0, as_f64p(score), C.byref(flags)))
    return score, flags.value


class RandomShootingPlanner:
    """A planner in the sense of POMDPs.action(planner, b) (src/POMDPsInterface.jl:44): draws n_seq
    action sequences, piecewise constant over test_period days with levels 0:0.1:1, scores them on
    the device with rollouts from the belief and returns the first action of the best sequence."""

    def __init__(self, pomdp: CovidPOMDP, n_seq: int = 1000, rollouts_per_seq: int = 64, horizon: int = 20,
                 seed: int = _DEFAULT_SEED):
        self.pomdp, self.n_seq, self.rollouts_per_seq, self.horizon, self.seed = pomdp, n_seq, rollouts_per_seq, horizon, seed
        self.calls = 0
        self.last_scores = None
        self.last_sequences = None

    def sequences(self) -> np.ndarray:
        rng = np.random.default_rng([self.seed, self.calls])
        period = max(1, self.pomdp.test_period)
        blocks = -(-self.horizon // period)
        levels = rng.integers(0, 11, size=(self.n_seq, blocks)) / 10.0
        return np.repeat(levels, period, axis=1)[:, :self.horizon]

    def action(self, b: ParticleCollection) -> Action:
        seqs = self.sequences()
        score, _ = rollout_score(b, seqs, self.rollouts_per_seq, seed=self.seed ^ 0x9E3779B97F4A7C15, step0=self.calls * self.horizon)
        self.calls += 1
        self.last_scores, self.last_sequences = score, seqs
        return Action(float(seqs[int(np.argmax(score)), 0]))


def Simulate(T: int, state: State, params: CovidPOMDP, action: Action, seed: int = _DEFAULT_SEED, device: int = 0) -> SimHist:
    """Simulate(T, state, params, action) (src/simulate.jl:175-196)."""
    b = ParticleCollection.from_states(params, [state], device)
    hist, rew, _ = rollout_batch(b, T, action.testing_prop, seed=seed)
    b.close()
    h = hist[0]
    return SimHist(h[:, 0].copy(), h[:, 1].copy(), h[:, 2].copy(), params.N, T, h[:, 3].copy(), [action] * T, rew[0].copy())


def SimulateEnsemble(T: int, trajectories: int, params: CovidPOMDP, action, seed: int = _DEFAULT_SEED,
                     device: int = 0) -> np.ndarray:
    """SimulateEnsemble (src/simulate.jl:210-227) followed by Array(::Vector{SimHist})
    (src/simulate.jl:291-297): int64[3, T, trajectories] from random State(params)
    starts.  action: one Action or a sequence of `trajectories` Actions."""
    b = random_states(params, trajectories, seed=seed, device=device)
    if isinstance(action, Action):
        acts = action.testing_prop
    else:
        acts = np.array([a.testing_prop for a in action], dtype=np.float64)
        if acts.size != trajectories:
            raise ValueError("need one action per trajectory")
    hist, _, _ = rollout_batch(b, T, acts, seed=seed, step0=1, want_rew=False)
    b.close()
    return np.ascontiguousarray(np.transpose(hist[:, :, :3], (2, 1, 0)))


def _mean_from_sums(ctx: Context, n_total: int, sums: np.ndarray, prev_action) -> State:
    """mean(pc, pomdp) (src/updater.jl:5-25) from the integer column sums of a belief."""
    p = ctx.pomdp
    H, D1 = p.horizon, p.test_delay + 1
    S, R = C.c_int64(), C.c_int64()
    I = np.empty(H, dtype=np.int64)
    T = np.empty((H, D1), dtype=np.int64)
    sums = np.ascontiguousarray(sums, dtype=np.int64)
    check(ctx.lib.epi_mean_from_sums(C.byref(ctx._cparams), int(n_total), as_i64p(sums), C.byref(S), as_i64p(I), C.byref(R), as_i64p(T)))
    return State(S.value, I, R.value, T.T.copy(), prev_action)


def closed_loop_days(true: ParticleCollection, b: ParticleCollection, upd: BootstrapFilter, actions, seed_true: int, step_true0: int,
                     q=None, want_sums: bool = False) -> dict:
    """len(actions) consecutive days of the closed loop (src/POMDPsInterface.jl:41-60) in one call: per day the true
    system `true` (a collection of one particle on b's context) steps, b is updated on that step's observation and
    the posterior is summarised, all enqueued on the device with one host wait at the end.  Advances upd.step.
    Returns per-day arrays: o, r, true_sir[., 3] (S, sum I, R of the true state after the step), loglik, ess, flags
    and, if asked for, sums (column sums of the posterior) and quant (order statistics q of sum(I))."""
    if true.ctx is not b.ctx or true.n != 1:
        raise ValueError("the true state must be a collection of one particle on the belief's context")
    if b.n != upd.n:
        raise ValueError("belief size differs from the filter's particle count")
    acts = np.ascontiguousarray(np.atleast_1d(np.asarray(actions, dtype=np.float64)))
    k = acts.size
    p = b.ctx.pomdp
    ns = 2 + p.horizon + (p.test_delay + 1) * p.horizon
    qs = None if q is None else np.ascontiguousarray(np.atleast_1d(np.asarray(q, dtype=np.float64)))
    nq = 0 if qs is None else qs.size
    out = {"o": np.empty(k, dtype=np.int64), "r": np.empty(k), "true_sir": np.empty((k, 3), dtype=np.int64),
           "loglik": np.empty(k), "ess": np.empty(k), "flags": np.empty(k, dtype=np.uint32),
           "sums": np.empty((k, ns), dtype=np.int64) if want_sums else None,
           "quant": np.empty((k, nq), dtype=np.int64) if nq else None}
    check(b.ctx.lib.epi_closed_loop_days(
        b.ctx.handle, true.handle, b.handle, k, as_f64p(acts), _prev_prop(true), seed_true, step_true0, upd.seed, upd.step,
        nq, as_f64p(qs), as_i64p(out["o"]), as_f64p(out["r"]), as_i64p(out["true_sir"]), as_f64p(out["loglik"]), as_f64p(out["ess"]),
        out["flags"].ctypes.data_as(C.POINTER(C.c_uint32)), as_i64p(out["sums"]), as_i64p(out["quant"])))
    upd.step += k
    upd.last_loglik, upd.last_ess, upd.last_flags = float(out["loglik"][-1]), float(out["ess"][-1]), int(out["flags"][-1])
    true.prev_action = b.prev_action = Action(float(acts[-1]))
    return out


def SimulateClosedLoop(T: int, state: State, b: ParticleCollection, pomdp: CovidPOMDP,
                       planner: Callable[[ParticleCollection], Action], seed: int = _DEFAULT_SEED,
                       keep_beliefs: bool = False) -> SimHist:
    """Simulate(T, state, b, pomdp, planner) (src/POMDPsInterface.jl:29-62):
    plan every test_period days, step the true state, update the belief.  The days between two planner calls run as
    one device call (closed_loop_days).
    keep_beliefs: False, True (a copy of every day's belief, as the reference keeps) or "summary" (per day the
    mean state and the 5 / 50 / 95 % order statistics of sum(I), reduced on the device)."""
    single = unity_test_period(pomdp)
    upd = BootstrapFilter(single, n_particles(b), seed=seed)
    sus, inf, rec = (np.zeros(T, dtype=np.int64) for _ in range(3))
    tests = np.zeros(T, dtype=np.int64)
    rewards = np.zeros(T)
    actions: List[Action] = []
    beliefs = []
    summary = isinstance(keep_beliefs, str) and keep_beliefs == "summary"
    copies = keep_beliefs is True
    qs = np.array([0.05, 0.5, 0.95])
    period = max(1, pomdp.test_period)
    true = ParticleCollection(b.ctx, 1)                          # the true state lives on the device too
    true.upload(*_flatten_states(single, [state]))
    true.prev_action = state.prev_action
    cur = (state.S, int(state.I.sum()), state.R)
    prior = BeliefSummary(mean(b), qs, suminf_quantiles(b, qs)) if summary else None
    day, action = 0, None
    while day < T:
        if day % period == 0:
            action = planner(b)
        k = 1 if copies else min(period - day % period, T - day)
        if copies:
            beliefs.append(b.copy())
        res = closed_loop_days(true, b, upd, [action.testing_prop] * k, seed ^ 0x7275E, day, q=qs if summary else None, want_sums=summary)
        for j in range(k):
            d = day + j
            sus[d], inf[d], rec[d] = cur
            cur = tuple(int(v) for v in res["true_sir"][j])
            actions.append(action)
            rewards[d], tests[d] = res["r"][j], res["o"][j]
            if summary:
                beliefs.append(prior)
                prior = BeliefSummary(_mean_from_sums(b.ctx, b.n, res["sums"][j], action), qs, res["quant"][j].copy())
        day += k
    true.close()
    return SimHist(sus, inf, rec, pomdp.N, T, tests, actions, rewards, beliefs)
