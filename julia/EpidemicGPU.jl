# EpidemicGPU.jl -- Julia binding of libepi_gfx950.so (include/epi.h) behind the interface the reference's callers
# use: POMDPs.gen / update / mean / rand on a particle belief (src/POMDPsInterface.jl:1-62, src/updater.jl:1-25 of
# WhiffleFish/EpidemicModeling), plus the batched entry points that have no single-state counterpart.
#
#   using EpidemicModeling; include("EpidemicGPU.jl"); using .EpidemicGPU
#   pomdp = CovidPOMDP()
#   up    = GPUBootstrapFilter(pomdp, 1_000_000)          # in place of BootstrapFilter(pomdp, n)
#   b     = initialbelief(up)                              # in place of initialbelief(pomdp, n)
#   b     = update(up, b, Action(0.2), o)                  # src/POMDPsInterface.jl:56
#   s     = mean(b)                                        # src/updater.jl:23, used by src/MPC.jl:322
#
# Written without a Julia installation at hand: every ccall below is checked, argument type by argument type, against
# the ctypes table the GPU test suite runs on, and every Module.name against the imports (tests/test_julia_glue.py), but
# the file itself has not been executed.
module EpidemicGPU

using Distributions, Random
import POMDPs, ParticleFilters
import Statistics: mean
using EpidemicModeling: CovidPOMDP, State, Action

export EpiCtx, GPUBelief, GPUBootstrapFilter, GPUNodeFilter, initialbelief, upload!, gen!, score_sequences, closed_loop_days!,
       SimulateClosedLoop, unity_test_period, step_variates, suminf_quantiles, gen_on_device!

const LIBEPI = get(ENV, "EPI_GFX950_LIB", joinpath(@__DIR__, "..", "epidemicmodeling_amd", "csrc", "libepi_gfx950.so"))
const EPI_MAX_H = 64

# ---- parameters: CovidPOMDP -> struct epi_params (include/epi.h) ---------------------------------------------
struct EpiParams
    H::Int32
    D::Int32
    N::Int64
    inf_loss::Float64
    test_loss::Float64
    testrate_loss::Float64
    q::NTuple{EPI_MAX_H,Float64}
    p::NTuple{EPI_MAX_H,Float64}
    k::NTuple{EPI_MAX_H,Float64}
    theta::NTuple{EPI_MAX_H,Float64}
end

pad(v) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, EPI_MAX_H)

function EpiParams(m::CovidPOMDP)
    H = length(m.Infdistributions)
    # src/simulate.jl:56-57: probability of developing symptoms on day i and isolating
    q = [(cdf(m.symptom_dist, i) - cdf(m.symptom_dist, i - 1)) * (1 - m.asymptomatic_prob) * m.symptomatic_isolation_prob for i in 1:H]
    kθ = Distributions.params.(m.Infdistributions)                 # src/init.jl:72
    EpiParams(H, m.test_delay, m.N, m.inf_loss, m.test_loss, m.testrate_loss,
              pad(q), pad(m.pos_test_probs), pad(first.(kθ)), pad(last.(kθ)))
end

last_error() = unsafe_string(ccall((:epi_last_error, LIBEPI), Cstring, ()))
check(rc) = rc == 0 || error("libepi_gfx950: ", last_error())

# The reference's unity_test_period (src/typedef.jl:122-138) passes a field the struct does not have and cannot run;
# this one copies the twelve fields that exist.
function unity_test_period(m::CovidPOMDP)
    CovidPOMDP(m.symptom_dist, m.Infdistributions, m.symptomatic_isolation_prob, m.asymptomatic_prob, m.pos_test_probs,
               m.test_delay, m.N, m.discount, m.inf_loss, m.test_loss, m.testrate_loss, 1)       # field order of src/typedef.jl:43-56
end

# ---- handles ----------------------------------------------------------------------------------------------------
mutable struct EpiCtx
    h::Ptr{Cvoid}
    m::CovidPOMDP
end
function EpiCtx(m::CovidPOMDP; device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:epi_ctx_create, LIBEPI), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h))
    p = Ref(EpiParams(m))
    check(ccall((:epi_params_set, LIBEPI), Cint, (Ptr{Cvoid}, Ref{EpiParams}), h[], p))
    c = EpiCtx(h[], m)
    finalizer(c -> ccall((:epi_ctx_destroy, LIBEPI), Cint, (Ptr{Cvoid},), c.h), c)
    c
end

# ParticleCollection{State} replacement (src/updater.jl:1-3, src/simulate.jl:21): the particles live in HBM.
mutable struct GPUBelief <: ParticleFilters.AbstractParticleBelief{State}
    h::Ptr{Cvoid}
    ctx::EpiCtx
    n::Int
    prev_action::Action
end
function GPUBelief(ctx::EpiCtx, n::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:epi_belief_create, LIBEPI), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Cvoid}}), ctx.h, n, 0, h))
    b = GPUBelief(h[], ctx, n, Action(0.0))
    finalizer(b -> ccall((:epi_belief_destroy, LIBEPI), Cint, (Ptr{Cvoid},), b.h), b)
    b
end
ParticleFilters.n_particles(b::GPUBelief) = b.n
dims(b::GPUBelief) = (length(b.ctx.m.Infdistributions), b.ctx.m.test_delay + 1)

# Vector{State} -> device: S[n], I[H, n], R[n], Tests[(D+1), H, n] (Julia's column-major arrays are the host layout)
function upload!(b::GPUBelief, ps::Vector{State})
    S = Int64[s.S for s in ps]
    R = Int64[s.R for s in ps]
    I = reduce(hcat, (Int64.(s.I) for s in ps))
    T = reduce(hcat, (vec(Int64.(s.Tests)) for s in ps))
    GC.@preserve S I R T check(ccall((:epi_belief_upload, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), b.h, S, I, R, T))
    isempty(ps) || (b.prev_action = ps[1].prev_action)
    b
end
function GPUBelief(ctx::EpiCtx, ps::Vector{State})
    upload!(GPUBelief(ctx, length(ps)), ps)
end

function ParticleFilters.particles(b::GPUBelief)                   # downloads: use mean / rand / quantiles where they do
    H, D1 = dims(b)
    S = Vector{Int64}(undef, b.n); R = similar(S)
    I = Matrix{Int64}(undef, H, b.n); T = Array{Int64}(undef, D1, H, b.n)
    check(ccall((:epi_belief_download, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), b.h, S, I, R, T))
    [State(S[j], I[:, j], R[j], T[:, :, j], b.prev_action) for j in 1:b.n]
end

function Base.copy(b::GPUBelief)                                    # beliefHist[day] = b (src/POMDPsInterface.jl:53)
    out = GPUBelief(b.ctx, b.n)
    check(ccall((:epi_belief_copy, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), out.h, b.h))
    out.prev_action = b.prev_action
    out
end

# ---- the drop-in methods ------------------------------------------------------------------------------------------
# BootstrapFilter(pomdp, n) / update(upd, b, a, o)   (src/POMDPsInterface.jl:39,56; src/MPC.jl:363,393)
mutable struct GPUBootstrapFilter <: POMDPs.Updater
    ctx::EpiCtx
    n::Int
    seed::UInt64
    step::UInt32
    loglik::Float64
    ess::Float64
    flags::UInt32
end
function GPUBootstrapFilter(m::CovidPOMDP, n::Integer; seed::Integer = 0x5EED, device::Integer = 0)
    m.test_period == 1 || error("build the filter on unity_test_period(pomdp), as src/POMDPsInterface.jl:38-39 does")
    GPUBootstrapFilter(EpiCtx(m; device = device), n, UInt64(seed), UInt32(0), NaN, NaN, UInt32(0))
end

function POMDPs.update(up::GPUBootstrapFilter, b::GPUBelief, a::Action, o::Integer)
    ll = Ref{Float64}(0.0); ess = Ref{Float64}(0.0); fl = Ref{UInt32}(0)
    check(ccall((:epi_pf_update, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int64, UInt64, UInt32, Ref{Float64}, Ref{Float64}, Ref{UInt32}),
        up.ctx.h, b.h, a.testing_prop, o, up.seed, up.step, ll, ess, fl))
    up.step += 1
    up.loglik, up.ess, up.flags = ll[], ess[], fl[]
    b.prev_action = a
    b                                                               # in place; copy(b) first to keep a history
end

# initialbelief(pomdp, np) (src/updater.jl:1-3) with initialstate(pomdp) = State(Normal(μ, σ), pomdp): the reference
# never defines initialstate (only the stale test/POMCPOWtest.jl:8-10 does, with Normal(10_000, 5_000))
function initialbelief(up::GPUBootstrapFilter; μ = 10_000.0, σ = 5_000.0)
    b = GPUBelief(up.ctx, up.n)
    check(ccall((:epi_belief_init_normal, LIBEPI), Cint, (Ptr{Cvoid}, Float64, Float64, UInt64), b.h, μ, σ, up.seed))
    b
end

# mean(pc, pomdp) (src/updater.jl:5-25; callers src/MPC.jl:322, src/plots.jl:47)
function mean(b::GPUBelief, m::CovidPOMDP = b.ctx.m)
    H, D1 = dims(b)
    S = Ref{Int64}(0); R = Ref{Int64}(0); I = Vector{Int64}(undef, H); T = Matrix{Int64}(undef, D1, H)
    check(ccall((:epi_belief_mean, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int64}),
                b.ctx.h, b.h, S, I, R, T))
    State(S[], I, R[], T, b.prev_action)
end

# rand(rng, b): POMCPOW samples states from the belief
function Base.rand(rng::AbstractRNG, b::GPUBelief)
    H, D1 = dims(b)
    S = Ref{Int64}(0); R = Ref{Int64}(0); I = Vector{Int64}(undef, H); T = Matrix{Int64}(undef, D1, H)
    check(ccall((:epi_belief_sample, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, UInt64, Ref{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int64}),
                b.ctx.h, b.h, rand(rng, UInt64), 0, S, I, R, T))
    State(S[], I, R[], T, b.prev_action)
end

# rand(rng, b, k): k states in one launch and one copy (a tree search draws thousands from one belief, test/POMCPOWtest.jl:15-21)
function Base.rand(rng::AbstractRNG, b::GPUBelief, k::Integer)
    H, D1 = dims(b)
    S = Vector{Int64}(undef, k); R = Vector{Int64}(undef, k); I = Matrix{Int64}(undef, H, k); T = Array{Int64,3}(undef, D1, H, k)
    check(ccall((:epi_belief_sample_many, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, UInt64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
                b.ctx.h, b.h, rand(rng, UInt64), 0, k, S, I, R, T))
    [State(S[j], I[:, j], R[j], T[:, :, j], b.prev_action) for j in 1:k]
end

# order statistics of sum(I) over the belief (what src/plots.jl:40-54 would draw of a belief history)
function suminf_quantiles(b::GPUBelief, q::Vector{Float64})
    length(q) <= 4 || error("at most 4 quantiles per call")
    out = Vector{Int64}(undef, length(q))
    check(ccall((:epi_belief_suminf_quantiles, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Int64}),
                b.ctx.h, b.h, q, length(q), out))
    out
end

# batched gen: continuous_gen over all particles (src/simulate.jl:160-166), in place
function gen!(b::GPUBelief, a::Action, seed::Integer, step::Integer)
    o = Vector{Int64}(undef, b.n); r = Vector{Float64}(undef, b.n); fl = Ref{UInt32}(0)
    ap = [b.prev_action.testing_prop]; av = [a.testing_prop]
    check(ccall((:epi_step_batch, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{Float64}, Cint, UInt64, UInt32, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ref{UInt32}),
        b.ctx.h, b.h, av, 0, ap, 0, seed, step, o, r, C_NULL, fl))
    b.prev_action = a
    (o = o, r = r, flags = fl[])
end

# POMDPs.gen on one state (src/POMDPsInterface.jl:1-9): test_period sub-steps, o of the last one, summed reward
function POMDPs.gen(ctx::EpiCtx, s::State, a::Action, seed::Integer, step::Integer)
    b = GPUBelief(ctx, [s])
    rsum = 0.0; o = 0
    for i in 0:ctx.m.test_period-1
        res = gen!(b, a, seed, step * ctx.m.test_period + i)
        o = res.o[1]; rsum += res.r[1]
    end
    (sp = ParticleFilters.particles(b)[1], o = o, r = rsum)
end

# The method the reference's callers dispatch on, POMDPs.gen(pomdp, s, a, rng) (src/POMDPsInterface.jl:1): defined for
# CovidPOMDP only after gen_on_device!(pomdp) has been called, since it replaces the package's own method.  The rng
# argument supplies the seed of the counter-based streams (the reference ignores it and draws from the global RNG,
# src/simulate.jl:40,58,79).
const GEN_CTX = Dict{UInt64,EpiCtx}()
function gen_on_device!(m::CovidPOMDP; device::Integer = 0)
    GEN_CTX[objectid(m)] = EpiCtx(m; device = device)
    @eval function POMDPs.gen(m::CovidPOMDP, s::State, a::Action, rng::AbstractRNG = Random.GLOBAL_RNG)
        ctx = get(GEN_CTX, objectid(m), nothing)
        ctx === nothing && error("gen_on_device!(pomdp) has not been called for this model")
        POMDPs.gen(ctx, s, a, rand(rng, UInt64), 0)
    end
    m
end

# Simulate / SimulateEnsemble (src/simulate.jl:175-227): hist[4, T, n] = S, sum(I), R before each day, sum(pos) of the day
function rollout!(b::GPUBelief, T::Integer, actions::Array{Float64}, per_particle::Bool, per_day::Bool, seed::Integer, step0::Integer)
    hist = Array{Int64}(undef, 4, T, b.n); rew = Matrix{Float64}(undef, T, b.n); fl = Ref{UInt32}(0)
    check(ccall((:epi_rollout_batch, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Cint, Float64, UInt64, UInt32, Ptr{Int64}, Ptr{Float64}, Ref{UInt32}),
        b.ctx.h, b.h, T, actions, per_particle, per_day, b.prev_action.testing_prop, seed, step0, hist, rew, fl))
    (hist = hist, rewards = rew, flags = fl[])
end

# A planner that evaluates candidate action sequences by rollouts from the belief (POMDPs.action(planner, b),
# src/POMDPsInterface.jl:44) gets all of them scored in one call.  seqs: T x n_seq (its memory is the n_seq x T
# row-major array the library reads).
function score_sequences(b::GPUBelief, seqs::Matrix{Float64}, rollouts_per_seq::Integer, seed::Integer, step0::Integer)
    T, nseq = size(seqs); score = Vector{Float64}(undef, nseq); fl = Ref{UInt32}(0)
    check(ccall((:epi_rollout_score, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint, Ptr{Float64}, Float64, Float64, UInt64, UInt32, Ptr{Float64}, Ref{UInt32}),
        b.ctx.h, b.h, nseq, rollouts_per_seq, T, seqs, b.prev_action.testing_prop, b.ctx.m.discount, seed, step0, score, fl))
    score
end

# ---- the closed loop (src/POMDPsInterface.jl:29-62) ---------------------------------------------------------------
# the days between two planner calls in one device call; `truth` is a GPUBelief of one particle on b's context
function closed_loop_days!(truth::GPUBelief, b::GPUBelief, up::GPUBootstrapFilter, actions::Vector{Float64},
                           seed_true::Integer, step_true0::Integer; q::Vector{Float64} = Float64[], want_sums::Bool = false)
    k = length(actions); H, D1 = dims(b); ns = 2 + H + D1 * H; nq = length(q)
    o = Vector{Int64}(undef, k); r = Vector{Float64}(undef, k); sir = Matrix{Int64}(undef, 3, k)
    ll = Vector{Float64}(undef, k); ess = Vector{Float64}(undef, k); fl = Vector{UInt32}(undef, k)
    sums = Matrix{Int64}(undef, ns, want_sums ? k : 0); quant = Matrix{Int64}(undef, nq, nq > 0 ? k : 0)
    check(ccall((:epi_closed_loop_days, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Float64, UInt64, UInt32, UInt64, UInt32, Cint, Ptr{Float64},
         Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{UInt32}, Ptr{Int64}, Ptr{Int64}),
        b.ctx.h, truth.h, b.h, k, actions, truth.prev_action.testing_prop, seed_true, step_true0, up.seed, up.step, nq,
        nq > 0 ? pointer(q) : Ptr{Float64}(C_NULL), o, r, sir, ll, ess, fl,
        want_sums ? pointer(sums) : Ptr{Int64}(C_NULL), nq > 0 ? pointer(quant) : Ptr{Int64}(C_NULL)))
    up.step += k
    up.loglik, up.ess, up.flags = ll[end], ess[end], fl[end]
    truth.prev_action = b.prev_action = Action(actions[end])
    (o = o, r = r, true_sir = sir, loglik = ll, ess = ess, flags = fl, sums = sums, quant = quant)
end

# Simulate(T, state, b, pomdp, planner) with the belief on the device.  `planner` is anything with
# POMDPs.action(planner, b::GPUBelief); an MPC controller plans on the belief mean as in src/MPC.jl:322:
#     POMDPs.action(mpc::MPC, b::GPUBelief) = POMDPs.action(mpc, mean(b))
# Instead of the reference's beliefHist (a ParticleCollection per day, src/POMDPsInterface.jl:53) the result carries,
# per day, what src/plots.jl:47 draws of it: the column sums of the posterior (belief_sums[:, d] ./ n is its mean: S, R,
# I[1..H], Tests) and the order statistics `q` of sum(I) over its particles (belief_quant[:, d]).
function SimulateClosedLoop(T::Int, state::State, b::GPUBelief, pomdp::CovidPOMDP, planner; seed::Integer = 0x5EED,
                            q::Vector{Float64} = [0.05, 0.5, 0.95])
    up = GPUBootstrapFilter(b.ctx, b.n, UInt64(seed), UInt32(0), NaN, NaN, UInt32(0))
    truth = GPUBelief(b.ctx, [state])
    sus = zeros(Int, T); inf = zeros(Int, T); rec = zeros(Int, T); tests = zeros(Int, T)
    acts = Vector{Action}(undef, T); rewards = zeros(Float64, T)
    H, D1 = dims(b)
    belief_sums = zeros(Int64, 2 + H + D1 * H, T); belief_quant = zeros(Int64, length(q), T)
    cur = (state.S, sum(state.I), state.R)
    day = 1
    while day <= T                                                  # src/POMDPsInterface.jl:41-60
        action = POMDPs.action(planner, b)                          # (day-1) % test_period == 0 here
        k = min(pomdp.test_period, T - day + 1)
        res = closed_loop_days!(truth, b, up, fill(action.testing_prop, k), xor(UInt64(seed), 0x7275E), day - 1; q = q, want_sums = true)
        for j in 1:k
            d = day + j - 1
            sus[d], inf[d], rec[d] = cur
            cur = (res.true_sir[1, j], res.true_sir[2, j], res.true_sir[3, j])
            acts[d] = action; rewards[d] = res.r[j]; tests[d] = res.o[j]
            belief_sums[:, d] = res.sums[:, j]
            isempty(q) || (belief_quant[:, d] = res.quant[:, j])
        end
        day += k
    end
    (sus = sus, inf = inf, rec = rec, N = pomdp.N, T = T, pos_test = tests, actions = acts, rewards = rewards,
     belief_sums = belief_sums, belief_quant = belief_quant, quantiles = q)
end

# ---- one population over the GPUs of a node (epi_mctx) ------------------------------------------------------------
# update(BootstrapFilter(pomdp, n), b, a, o) with the particles sharded by index over several devices, from this one
# Julia process: GPUNodeFilter is updater and belief in one (the particles never leave the devices).
mutable struct GPUNodeFilter <: POMDPs.Updater
    h::Ptr{Cvoid}
    m::CovidPOMDP
    n::Int
    seed::UInt64
    step::UInt32
    prev_action::Action
    loglik::Float64
    ess::Float64
    flags::UInt32
    moved::Int64
end
function GPUNodeFilter(m::CovidPOMDP, n::Integer; devices::Vector{<:Integer} = collect(0:7), seed::Integer = 0x5EED, rccl::Bool = true)
    m.test_period == 1 || error("build the filter on unity_test_period(pomdp), as src/POMDPsInterface.jl:38-39 does")
    h = Ref{Ptr{Cvoid}}(C_NULL); p = Ref(EpiParams(m)); dev = Cint.(devices)
    check(ccall((:epi_mctx_create, LIBEPI), Cint, (Ptr{Cint}, Cint, Ref{EpiParams}, Int64, Cint, Ref{Ptr{Cvoid}}),
                dev, length(dev), p, n, rccl, h))
    f = GPUNodeFilter(h[], m, n, UInt64(seed), UInt32(0), Action(0.0), NaN, NaN, UInt32(0), 0)
    finalizer(f -> ccall((:epi_mctx_destroy, LIBEPI), Cint, (Ptr{Cvoid},), f.h), f)
    f
end
ParticleFilters.n_particles(f::GPUNodeFilter) = f.n

# An update normally returns while the particles that change shard are still on their way, and the next one steps
# around them; overlap!(f, false) makes every update place them before it returns.
overlap!(f::GPUNodeFilter, on::Bool) = (check(ccall((:epi_mctx_set_overlap, LIBEPI), Cint, (Ptr{Cvoid}, Cint), f.h, on)); f)

function initialbelief(f::GPUNodeFilter; μ = 10_000.0, σ = 5_000.0)
    check(ccall((:epi_mctx_init_normal, LIBEPI), Cint, (Ptr{Cvoid}, Float64, Float64, UInt64), f.h, μ, σ, f.seed))
    f
end

function POMDPs.update(f::GPUNodeFilter, b::GPUNodeFilter, a::Action, o::Integer)
    b === f || error("a GPUNodeFilter updates its own particles")
    ll = Ref{Float64}(0.0); ess = Ref{Float64}(0.0); fl = Ref{UInt32}(0); mv = Ref{Int64}(0)
    check(ccall((:epi_mpf_update, LIBEPI), Cint,
        (Ptr{Cvoid}, Float64, Int64, UInt64, UInt32, Ref{Float64}, Ref{Float64}, Ref{UInt32}, Ref{Int64}),
        f.h, a.testing_prop, o, f.seed, f.step, ll, ess, fl, mv))
    f.step += 1
    f.loglik, f.ess, f.flags, f.moved, f.prev_action = ll[], ess[], fl[], mv[], a
    f
end

function mean(f::GPUNodeFilter)
    H = length(f.m.Infdistributions); D1 = f.m.test_delay + 1
    S = Ref{Int64}(0); R = Ref{Int64}(0); I = Vector{Int64}(undef, H); T = Matrix{Int64}(undef, D1, H)
    check(ccall((:epi_mctx_mean, LIBEPI), Cint, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int64}), f.h, S, I, R, T))
    State(S[], I, R[], T, f.prev_action)
end

function ParticleFilters.particles(f::GPUNodeFilter)
    H = length(f.m.Infdistributions); D1 = f.m.test_delay + 1
    S = Vector{Int64}(undef, f.n); R = similar(S); I = Matrix{Int64}(undef, H, f.n); T = Array{Int64}(undef, D1, H, f.n)
    check(ccall((:epi_mctx_download, LIBEPI), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), f.h, S, I, R, T))
    [State(S[j], I[:, j], R[j], T[:, :, j], f.prev_action) for j in 1:f.n]
end

# ---- for the parity harness (parity.jl) ---------------------------------------------------------------------------
# one step of every particle of b that also returns the 3H variates each particle drew
function step_variates(b::GPUBelief, a::Action, seed::Integer, step::Integer)
    H, _ = dims(b)
    sympt = Matrix{Int64}(undef, H, b.n); pos = Matrix{Int64}(undef, H, b.n); gam = Matrix{Float64}(undef, H, b.n); fl = Ref{UInt32}(0)
    check(ccall((:epi_step_variates, LIBEPI), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Float64, UInt64, UInt32, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ref{UInt32}),
        b.ctx.h, b.h, a.testing_prop, seed, step, sympt, pos, gam, fl))
    b.prev_action = a
    (sympt = sympt, pos = pos, gam = gam, flags = fl[])
end

end # module
