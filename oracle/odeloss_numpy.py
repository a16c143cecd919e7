"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the ODE-model losses of the reference
(src/ODEFit.jl:1-37 the models, :281-340 the losses) in the form the device evaluates them: classical
Runge-Kutta with `substeps` steps per day from the trajectory's day-1 proportions, squared error against
the trajectory's proportions at every day, mean over trajectories.  Same operation order as the kernel, so
the per-trajectory sums agree bit for bit.  PARITY UNPINNED against the reference's adaptive solver (no
Julia here); `solve_ivp_loss` below integrates the same models with scipy at tight tolerance as a
cross-check of the restatement."""
import numpy as np

SIR, SIR_CTRL, SEIR, SEIR_CTRL = 0, 1, 2, 3


def rates(kind, p, tp):
    p = list(p) + [0.0] * (5 - len(p))
    alpha = np.full_like(tp, p[0])
    beta = p[1]
    gamma = np.zeros_like(tp)
    if kind == SIR_CTRL:
        alpha = p[0] + p[2] * tp
    if kind == SEIR:
        gamma = np.full_like(tp, p[2])
    if kind == SEIR_CTRL:
        gamma = p[2] + p[4] * tp
        alpha = p[0] + p[3] * tp
    return alpha, beta, gamma


def _rhs(seir, alpha, beta, gamma, u):
    inf = beta * u[2] * u[0]
    if seir:
        return [-inf, inf - gamma * u[1], gamma * u[1] - alpha * u[2], alpha * u[2]]
    return [-inf, np.zeros_like(inf), inf - alpha * u[2], alpha * u[2]]


def _rk4(seir, alpha, beta, gamma, u, h):
    k1 = _rhs(seir, alpha, beta, gamma, u)
    k2 = _rhs(seir, alpha, beta, gamma, [u[c] + (0.5 * h) * k1[c] for c in range(4)])
    k3 = _rhs(seir, alpha, beta, gamma, [u[c] + (0.5 * h) * k2[c] for c in range(4)])
    k4 = _rhs(seir, alpha, beta, gamma, [u[c] + h * k3[c] for c in range(4)])
    return [u[c] + (h / 6.0) * (((k1[c] + 2.0 * k2[c]) + 2.0 * k3[c]) + k4[c]) for c in range(4)]


def per_trajectory_loss(hist, N, kind, p, actions, substeps):
    """hist: int64[n, T, 3] (S, sum I, R before each day's step); actions: [n] first-day actions."""
    hist = np.asarray(hist)
    n, T, _ = hist.shape
    tp = np.asarray(actions, dtype=np.float64) if kind in (SIR_CTRL, SEIR_CTRL) else np.zeros(n)
    alpha, beta, gamma = rates(kind, p, tp)
    seir = kind >= SEIR
    Np = float(N)
    u = [hist[:, 0, 0] / Np, np.zeros(n), hist[:, 0, 1] / Np, hist[:, 0, 2] / Np]
    h = 1.0 / float(substeps)
    loss = np.zeros(n)
    for t in range(T):
        if t > 0:
            for _ in range(substeps):
                u = _rk4(seir, alpha, beta, gamma, u, h)
        e0 = u[0] - hist[:, t, 0] / Np
        e1 = (u[1] + u[2]) - hist[:, t, 1] / Np
        e2 = u[3] - hist[:, t, 2] / Np
        loss = loss + ((e0 * e0 + e1 * e1) + e2 * e2)
    return loss


def loss(hist, N, kind, p, actions, substeps):
    pt = per_trajectory_loss(hist, N, kind, p, actions, substeps)
    return float(np.sum(pt) / pt.size)


def solve_ivp_loss(hist, N, kind, p, actions, rtol=1e-10):
    """The same loss with scipy's adaptive integrator (cross-check, small ensembles only)."""
    from scipy.integrate import solve_ivp
    hist = np.asarray(hist)
    n, T, _ = hist.shape
    tp = np.asarray(actions, dtype=np.float64) if kind in (SIR_CTRL, SEIR_CTRL) else np.zeros(n)
    alpha, beta, gamma = rates(kind, p, tp)
    seir = kind >= SEIR
    out = np.zeros(n)
    for m in range(n):
        a, g = float(alpha[m]), float(gamma[m])

        def f(_, u):
            inf = beta * u[2] * u[0]
            if seir:
                return [-inf, inf - g * u[1], g * u[1] - a * u[2], a * u[2]]
            return [-inf, 0.0, inf - a * u[2], a * u[2]]
        u0 = [hist[m, 0, 0] / N, 0.0, hist[m, 0, 1] / N, hist[m, 0, 2] / N]
        if T > 1:
            sol = solve_ivp(f, (1.0, float(T)), u0, t_eval=np.arange(1, T + 1, dtype=float), rtol=rtol, atol=1e-13)
            y = sol.y
        else:
            y = np.array(u0).reshape(4, 1)
        ref = hist[m].T / N
        out[m] = np.sum((y[0] - ref[0]) ** 2 + ((y[1] + y[2]) - ref[1]) ** 2 + (y[3] - ref[2]) ** 2)
    return float(out.sum() / n)
