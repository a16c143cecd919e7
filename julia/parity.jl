# parity.jl -- drives the reference's own SimStep (src/simulate.jl:137-151) with the variates the device drew and
# compares every integer of the resulting states (the check "integer compartment counts bit-exact against the
# reference step function fed the same draws").  Needs Julia, the reference package and its dependencies, and a GPU:
#
#     julia --project=/path/to/EpidemicModeling parity.jl [n_particles] [days] [data_dir]
#
# data_dir holds the two files the reference constructor reads and its checkout lacks (src/typedef.jl:88-96;
# .MISSING_LARGE_BLOBS): Sample50.csv and raw_viral_load.csv.  epidemicmodeling_amd.paramfit.write_csvs(data_dir, seed)
# regenerates them from the model of src/Python/distUtils.py; without the argument the constructor's own default
# paths (src/data/) are used.
#
# The reference draws, per step, sympt[1..H] (simulate.jl:58), pos[1..H] (:79) and gam[1..H] (:40, through RVsum,
# init.jl:71-78, whose zero-shape branch draws from Normal(0, 0)), in that order, from the global RNG.  The three `rand`
# methods below pop those draws from a queue instead.
using EpidemicModeling, Distributions, Random
import ParticleFilters
using EpidemicModeling: SimStep, State, Action, CovidPOMDP
include(joinpath(@__DIR__, "EpidemicGPU.jl"))
using .EpidemicGPU

const QUEUE = Ref(Float64[])
Base.rand(::Random.AbstractRNG, ::Distributions.Binomial) = Int(popfirst!(QUEUE[]))
Base.rand(::Random.AbstractRNG, ::Distributions.Gamma)    = popfirst!(QUEUE[])
Base.rand(::Random.AbstractRNG, ::Distributions.Normal)   = popfirst!(QUEUE[])       # RVsum's k N <= 0 branch (init.jl:76)
Base.rand(d::Distributions.Binomial) = rand(Random.GLOBAL_RNG, d)
Base.rand(d::Distributions.Gamma)    = rand(Random.GLOBAL_RNG, d)
Base.rand(d::Distributions.Normal)   = rand(Random.GLOBAL_RNG, d)

function check_particle(m::CovidPOMDP, s::State, a::Action, sympt, pos, gam, sp_gpu::State)
    QUEUE[] = Float64[sympt; pos; gam]
    sp, new_infections, pos_tests = SimStep(deepcopy(s), m, a)     # deepcopy: SimStep writes into s.Tests (simulate.jl:110,112)
    isempty(QUEUE[]) || error("the reference consumed $(3length(sympt) - length(QUEUE[])) of $(3length(sympt)) variates")
    (sp.S, sp.I, sp.R, sp.Tests) == (sp_gpu.S, sp_gpu.I, sp_gpu.R, sp_gpu.Tests) || return false
    pos_tests == pos
end

function main(n = 2000, days = 20, data_dir = nothing; seed = 0x5EED)
    # keyword names of the constructor: src/typedef.jl:78-93
    m = data_dir === nothing ? CovidPOMDP() :
        CovidPOMDP(infections_path = joinpath(data_dir, "Sample50.csv"), viral_loads_path = joinpath(data_dir, "raw_viral_load.csv"))
    up = GPUBootstrapFilter(unity_test_period(m), n; seed = seed)
    b = initialbelief(up)
    bad = 0
    for day in 0:days-1
        a = Action(rand(0:0.1:1))
        before = ParticleFilters.particles(b)
        v = step_variates(b, a, seed, day)
        after = ParticleFilters.particles(b)
        for j in 1:n
            check_particle(m, before[j], a, v.sympt[:, j], v.pos[:, j], v.gam[:, j], after[j]) || (bad += 1)
        end
        println("day $day: action $(a.testing_prop), mismatching particles so far: $bad")
    end
    bad == 0 || error("$bad particle-steps differ from the reference's SimStep")
    println("reference SimStep reproduced on $(n * days) particle-steps")
end

if abspath(PROGRAM_FILE) == @__FILE__
    main(length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 2000, length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 20,
         length(ARGS) >= 3 ? ARGS[3] : nothing)
end
