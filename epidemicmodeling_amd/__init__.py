"""MI355X (gfx950) implementation of the InfectionSim particle step and the
bootstrap particle-filter belief update of WhiffleFish/EpidemicModeling,
behind the reference's own gen / update / mean interface."""
from .params import (Action, State, CovidPOMDP, make_pomdp, unity_test_period, population, Array,
                     State_from, State_random)
from .interface import (Context, ParticleCollection, BootstrapFilter, update, gen, observation, initialbelief,
                        random_states, mean, suminf_quantiles, BeliefSummary, n_particles, particles, rand, rand_many, step_batch, rollout_batch, rollout_score, RandomShootingPlanner, SimHist,
                        Simulate, SimulateEnsemble, SimulateClosedLoop)
from .odefit import Ensemble, FitRandEnsemble, FitRandControlledEnsemble
from .paramfit import FitInfectionDistributions, prop_above_LOD, CovidPOMDP_generated, CovidPOMDP_from_csv

__all__ = [
    "Action", "State", "CovidPOMDP", "make_pomdp", "unity_test_period", "population", "Array", "State_from",
    "State_random", "Context", "ParticleCollection", "BootstrapFilter", "update", "gen", "observation",
    "initialbelief", "random_states", "mean", "suminf_quantiles", "BeliefSummary", "n_particles", "particles", "rand", "rand_many", "step_batch", "rollout_batch", "rollout_score", "RandomShootingPlanner",
    "SimHist", "Simulate", "SimulateEnsemble", "SimulateClosedLoop",
    "Ensemble", "FitRandEnsemble", "FitRandControlledEnsemble",
    "FitInfectionDistributions", "prop_above_LOD", "CovidPOMDP_generated", "CovidPOMDP_from_csv",
]
