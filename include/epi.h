/*
 * epi.h -- C ABI of libepi_gfx950.so: the InfectionSim particle step and the
 * bootstrap particle-filter belief update on AMD Instinct MI355X (gfx950).
 *
 * This is the boundary a host in any language binds (Julia `ccall`, Python
 * ctypes, C/C++).  Plain C types only.  Every entry point names the piece of
 * the reference (WhiffleFish/EpidemicModeling, paths relative to its checkout)
 * whose work it takes over.
 *
 * Conventions
 *   - return value 0 = success, negative EPI_E* = failure; epi_last_error()
 *     returns a message for the last failure on the calling thread.
 *   - the caller owns every host array; the library owns device memory behind
 *     the opaque epi_ctx / epi_belief handles until *_destroy.
 *   - calls are synchronous unless stated (they return after the device work
 *     has completed); a context is not thread-safe.
 *   - host state layout = a Julia Vector{State} flattened:
 *       S[n], R[n]        int64
 *       I[n*H]            int64, particle p / age i at p*H + i
 *       T[n*(D+1)*H]      int64, per particle the column-major (D+1) x H Tests
 *                         matrix (row r, col c at p*(D+1)*H + c*(D+1) + r);
 *                         may be NULL (= all zero)
 *     Counts are narrowed to int32 on the device: N must be < 2^31.  Tests row
 *     D+1 and column 1 must be zero (they always are between SimStep calls,
 *     src/simulate.jl:125-126).
 *   - random variates are pure functions of (seed, global particle slot, step):
 *     see epi_rng.hpp.  `step` distinguishes successive calls; reuse of a
 *     (seed, step) pair reproduces the same draws.
 */
#ifndef EPI_H
#define EPI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPI_MAX_H 64   /* infection-age bins */
#define EPI_MAX_D 7    /* test_delay */

/* status codes */
#define EPI_OK          0
#define EPI_EINVAL     -1   /* bad argument (message says which)                  */
#define EPI_EHIP       -2   /* a HIP runtime call failed                          */
#define EPI_ENOMEM     -3   /* device or host allocation failed                   */
#define EPI_ERANGE     -4   /* a count does not fit the device representation     */
#define EPI_ESTATE     -5   /* call made in the wrong phase of an update          */

/* per-step condition flags (bit set, returned through `flags` arguments) */
#define EPI_F_NEG_I      1u /* an I[age] went negative (src/simulate.jl:117 assert) */
#define EPI_F_NEG_NT     2u /* a tested count was negative                          */
#define EPI_F_BAD_DENOM  4u /* N - R <= 0 in incident_infections (simulate.jl:42)   */
#define EPI_F_ZERO_W     8u /* every particle weight was zero: particles kept       */
#define EPI_F_COARSE_W  16u /* the fixed-point weights were coarse for this update (see below) */
/* Degenerate weights, where this library and ParticleFilters.jl differ (the reference's own behaviour there is NaN
 * arithmetic, not a specification):
 *   - Normal(mu, 0) (no one tested, or every test probability 0 or 1): a particle whose mu equals o gets weight 1
 *     (log weight 0, to which the constant -log sqrt(2 pi) of the log-likelihood still applies), any other weight 0;
 *     the reference's pdf is Inf / 0 there.
 *   - every weight zero: the particles are kept in place and EPI_F_ZERO_W is raised; the reference's low-variance
 *     resampler would return n copies of its first particle (its cumulative-weight walk never advances).
 * Resampling works on fixed-point weights floor(w_i / w_max 2^F) with F = 53 - ceil(log2 n) fractional bits (at most
 * 40), so that the sum over the population is exact in 53 bits whatever the order (and hence the number of shards).
 * A particle lighter than 2^-F w_max gets no offspring, and all truncations together remove at most n 2^-F w_max of
 * the weight; against the total, at most n / W (W = the fixed-point total).  That is below 10^-4 for every update of
 * up to 10^6 particles and for any size while the weights are spread; it grows when 10^7 or more particles collapse
 * onto a few (n = 10^8, F = 26: about 1.5 / (sum of w_i / w_max)).  EPI_F_COARSE_W reports n / W > 10^-3.
 * The samplers cap a rejection draw at 512 attempts (probability of reaching it below 10^-400 for any admissible
 * parameters); a draw that exhausted them ends as the binomial's mode, or as the Gamma's unit proposal, in every
 * step kernel. */

/* Model parameters the step needs: the numeric content of the reference's
 * CovidPOMDP struct (src/typedef.jl:43-56) after its constructor has run
 * (src/typedef.jl:78-116).
 *   q[i]     = (F(i+1) - F(i)) (1 - asymptomatic_prob) symptomatic_isolation_prob,
 *              F = cdf(symptom_dist)                      (src/simulate.jl:56-57)
 *   p[i]     = pos_test_probs[i+1]                        (src/simulate.jl:79)
 *   k, theta = params(Infdistributions[i+1])              (src/init.jl:72) */
typedef struct epi_params {
    int32_t H;              /* horizon = length(Infdistributions)        */
    int32_t D;              /* test_delay                                 */
    int64_t N;              /* population                                 */
    double inf_loss, test_loss, testrate_loss;
    double q[EPI_MAX_H];
    double p[EPI_MAX_H];
    double k[EPI_MAX_H];
    double theta[EPI_MAX_H];
} epi_params;

typedef struct epi_ctx epi_ctx;         /* one device, one stream, the parameters */
typedef struct epi_belief epi_belief;   /* a set of particles resident in HBM     */

/* ---- context ------------------------------------------------------------ */

/* Bind a context to HIP device `device` (creates a stream and scalar scratch). */
int epi_ctx_create(int device, epi_ctx **out);
int epi_ctx_destroy(epi_ctx *ctx);
/* Install the model parameters (replaces passing `params::CovidPOMDP` to every
 * call: src/simulate.jl:137, src/POMDPsInterface.jl:1). */
int epi_params_set(epi_ctx *ctx, const epi_params *p);
const char *epi_last_error(void);
/* Number of int32 rows of the device state for (H, D): 2 + H + D (H - 1). */
int epi_state_rows(int H, int D);

/* ---- beliefs (ParticleCollection{State}, src/updater.jl:1-3) -------------- */

/* n_local particles on this device; halo_cap extra columns for particles
 * received from other shards (0 for a single-device belief). */
int epi_belief_create(epi_ctx *ctx, int64_t n_local, int64_t halo_cap, epi_belief **out);
int epi_belief_destroy(epi_belief *b);
/* Place this belief in a sharded population: it holds global slots
 * [slot0, slot0 + n_local) of n_total.  Default: slot0 = 0, n_total = n_local. */
int epi_belief_set_shard(epi_belief *b, int64_t n_total, int64_t slot0);
int64_t epi_belief_size(const epi_belief *b);
/* Host <-> device transfer in the host layout above (ParticleCollection(particles),
 * particles(b)).  Download resolves pending resampling indirection first. */
int epi_belief_upload(epi_belief *b, const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T);
int epi_belief_download(epi_belief *b, int64_t *S, int64_t *I, int64_t *R, int64_t *T);
/* Copy src into dst (same shape): beliefs have value semantics in the reference
 * (beliefHist[day] = b, src/POMDPsInterface.jl:53).  The copy resolves src's pending
 * resampling indirection on the way; src itself is not modified. */
int epi_belief_copy(epi_belief *dst, epi_belief *src);

/* Initial particles generated on the device.
 * epi_belief_init_normal: State(Normal(mu, sigma), params) per particle
 *   (src/typedef.jl:159-181; the stale test/POMCPOWtest.jl:8-10 initialstate).
 *   Returns EPI_EINVAL if any particle drew sum(I0) > N (the reference asserts, src/typedef.jl:164).
 * epi_belief_init_simplex: State(params) per particle (src/typedef.jl:196-217). */
int epi_belief_init_normal(epi_belief *b, double mu, double sigma, uint64_t seed);
int epi_belief_init_simplex(epi_belief *b, uint64_t seed);

/* ---- the stochastic step ---------------------------------------------------- */

/* continuous_gen over every particle of b (src/simulate.jl:160-166 wrapping
 * SimStep, src/simulate.jl:137-151): b <- sp.  a / a_prev: one value
 * (per_particle = 0) or n_local values.  Optional host outputs, n_local each:
 * o = sum(pos_tests), r = reward (src/simulate.jl:153-158), new_inf. */
int epi_step_batch(epi_ctx *ctx, epi_belief *b,
                   const double *a, int a_per_particle, const double *a_prev, int a_prev_per_particle,
                   uint64_t seed, uint32_t step,
                   int64_t *o_out, double *r_out, int64_t *new_inf_out, uint32_t *flags);

/* SimStep with the 3H variates of every particle supplied by the caller
 * (sympt, pos: int64[n*H]; gam: double[n*H]; particle-major), i.e. the
 * deterministic bookkeeping of src/simulate.jl:99-151 alone.  pos_out: int64[n*H]. */
int epi_step_injected(epi_ctx *ctx, epi_belief *b, const double *a, int a_per_particle,
                      const int64_t *sympt, const int64_t *pos, const double *gam,
                      int64_t *new_inf_out, int64_t *pos_out, uint32_t *flags);

/* One step of every particle of b (as epi_step_batch with one action) that also
 * returns the 3H variates each particle drew (sympt, pos: int64[n*H]; gam:
 * double[n*H]; particle-major): the three `rand` calls of src/simulate.jl:40,58,79.
 * A harness feeds them to the reference's own SimStep to compare counts. */
int epi_step_variates(epi_ctx *ctx, epi_belief *b, double a, uint64_t seed, uint32_t step,
                      int64_t *sympt_out, int64_t *pos_out, double *gam_out, uint32_t *flags);

/* The bare samplers the step uses (rand(Binomial(n, p)), src/simulate.jl:58,79;
 * rand(Gamma(shape, scale)), src/simulate.jl:40 with src/init.jl:71-78), one
 * draw per element; element j uses the streams of particle slot j, age bin
 * `bin`; which = 0: the isolation draw's words, 1: the positive-test draw's. */
int epi_sample_binomial(epi_ctx *ctx, int64_t count, const int64_t *n, const double *p, uint64_t seed, uint32_t step,
                        uint32_t bin, int which, int64_t *out);
int epi_sample_gamma(epi_ctx *ctx, int64_t count, const double *shape, const double *scale, uint64_t seed, uint32_t step,
                     uint32_t bin, double *out);

/* Simulate / SimulateEnsemble (src/simulate.jl:175-227): T days from the
 * particles of b, which end at day T.  actions: [n or 1][T or 1] doubles.
 * hist (optional): int64[n][T][4] = S, sum(I), R before each day's step and
 * sum(pos_tests) of the day; rew (optional): double[n][T]. */
int epi_rollout_batch(epi_ctx *ctx, epi_belief *b, int T,
                      const double *actions, int act_per_particle, int act_per_day, double a_prev0,
                      uint64_t seed, uint32_t step0,
                      int64_t *hist, double *rew, uint32_t *flags);

/* Scenario scoring for planners (the batched form of the rollouts POMCPOW and random-shooting MPC
 * make from the belief: planner call sites src/POMDPsInterface.jl:44, test/POMCPOWtest.jl:15-21,
 * horizon src/MPC.jl:58): n_seq action sequences x rollouts_per_seq independent T-day rollouts
 * each.  Trajectory j (0 <= j < n_seq * rollouts_per_seq) follows sequence j / rollouts_per_seq
 * (actions is n_seq x T, row-major), starts from particle (j mod n) of `start` (unchanged) and
 * draws from the streams of slot j, steps step0 .. step0+T-1.
 * score[s] = mean over the sequence's rollouts (summed in rollout order) of
 * sum_t discount^t * reward_t (src/simulate.jl:153-158), discount^t by repeated multiplication. */
int epi_rollout_score(epi_ctx *ctx, epi_belief *start, int64_t n_seq, int rollouts_per_seq, int T,
                      const double *actions, double a_prev0, double discount, uint64_t seed, uint32_t step0,
                      double *score /* n_seq */, uint32_t *flags);

/* ---- the bootstrap filter update ---------------------------------------------- */

/* update(BootstrapFilter(pomdp, n), b, a, o) (ParticleFilters.jl; call sites
 * src/POMDPsInterface.jl:39,56 and src/MPC.jl:363,393): predict every
 * particle through SimStep, weight by pdf(observation(pomdp, s, a, sp), o)
 * (src/POMDPsInterface.jl:11-23), systematic resample to n particles.  b <- b'.
 * loglik (optional): log of the mean particle likelihood; ess (optional):
 * effective sample size of the weights. */
int epi_pf_update(epi_ctx *ctx, epi_belief *b, double a, int64_t o, uint64_t seed, uint32_t step,
                  double *loglik, double *ess, uint32_t *flags);

/* The same update in phases, for a population sharded over several devices
 * (one process per device; the caller exchanges the small results between
 * phases, e.g. with torch.distributed / RCCL):
 *   1. epi_pf_predict : fused weight + step of the local particles;
 *                       returns the local maximum log weight.
 *   2. epi_pf_weigh   : fixed-point weights relative to the GLOBAL maximum and
 *                       their local prefix sums; returns the local weight total,
 *                       sum of squares and the EPI_F_* flags the local step raised
 *                       (so that one all-gather carries everything the ranks
 *                       exchange).  frac_bits = epi_pf_frac_bits(n_total).
 *   3. epi_pf_plan    : given the weight totals of all G shards, the global
 *                       output-slot range [lo[g], lo[g+1]) whose ancestors live
 *                       in shard g (host arithmetic, identical on every rank).
 *   4. epi_pf_select_local : ancestors for the part of that range this shard
 *                       also owns (no data movement); at most once per update.  Called before the first
 *                       export of an update it also leaves the cumulative weights
 *                       that the exports search (one pass instead of two; any
 *                       order of the calls of this step gives the same result);
 *      epi_pf_export  : for a part owned by another shard, the resampled
 *                       particles themselves, packed as int32[rows][count];
 *      epi_pf_import  : place a packed block received from another shard.
 *   5. epi_pf_finish  : close the update (swap buffers, collect flags). */
int epi_pf_frac_bits(int64_t n_total);
int epi_pf_predict(epi_ctx *ctx, epi_belief *b, double a, int64_t o, uint64_t seed, uint32_t step, double *lwmax_local);
int epi_pf_weigh(epi_ctx *ctx, epi_belief *b, double lwmax_global, uint64_t *w_local, double *sumsq_local,
                 uint32_t *flags_local);
/* The first two phases without a host round trip, for a host that runs the exchange on the same HIP
 * stream (epi_ctx_set_stream): the local maximum is written to, the global maximum read from, device
 * memory, and d_totals receives {local weight total, local sum of squares, local flags, global maximum}
 * as four doubles (totals are < 2^53, exact).  The flags are delivered there: epi_pf_finish after epi_pf_weigh_dev
 * has none left to report. */
int epi_pf_predict_dev(epi_ctx *ctx, epi_belief *b, double a, int64_t o, uint64_t seed, uint32_t step,
                       double *d_lwmax_local /* device, 1 double */);
int epi_pf_weigh_dev(epi_ctx *ctx, epi_belief *b, const double *d_lwmax_global /* device, 1 double */,
                     double *d_totals /* device, 4 doubles */);
int epi_pf_plan(int64_t n_total, int n_shards, const uint64_t *w_shard, uint64_t seed, uint32_t step,
                int64_t *lo /* n_shards + 1 */);
int epi_pf_select_local(epi_ctx *ctx, epi_belief *b, int n_shards, int shard, const uint64_t *w_shard,
                        uint64_t seed, uint32_t step, int64_t slot_begin, int64_t slot_end);
int epi_pf_export(epi_ctx *ctx, epi_belief *b, int n_shards, int shard, const uint64_t *w_shard,
                  uint64_t seed, uint32_t step, int64_t slot_begin, int64_t slot_end, void *dev_buf);
int epi_pf_import(epi_ctx *ctx, epi_belief *b, int64_t slot_begin, int64_t slot_end, const void *dev_buf);
int epi_pf_finish(epi_ctx *ctx, epi_belief *b, uint32_t *flags /* NULL: do not fetch */);
/* Transfers hidden behind the next step.  Particles that change shard are 64 bytes each over a point-to-point link,
 * and a shard needs them only for the slots they fill.  A host may therefore call epi_pf_finish before its receives
 * have completed: the owned slots that neither epi_pf_select_local nor epi_pf_import has filled are then recorded as
 * waiting, and every call that reads the population refuses (EPI_ESTATE) until they are filled.  The next update of
 * the same call sites (update(up, b, a, o), src/POMDPsInterface.jl:56) steps the shard in parts:
 *   epi_pf_predict_part_dev over the columns that were filled locally, while the transfer is in flight;
 *   epi_pf_import_late for each received block (same packing as epi_pf_import);
 *   epi_pf_predict_part_dev over the remaining columns.
 * A part starts at a multiple of 256 columns and holds a multiple of 256, or ends with the shard; the parts of an
 * update agree in (a, o, seed, step) and cover the shard exactly once; the part that completes the cover writes the
 * local maximum log weight to d_lwmax_local as epi_pf_predict_dev does, and epi_pf_weigh_dev follows. */
int epi_pf_predict_part_dev(epi_ctx *ctx, epi_belief *b, double a, int64_t o, uint64_t seed, uint32_t step,
                            int64_t col0, int64_t count, double *d_lwmax_local /* device, 1 double */);
int epi_pf_import_late(epi_ctx *ctx, epi_belief *b, int64_t slot_begin, int64_t slot_end, const void *dev_buf);

/* ---- the same update over the GPUs of a node, behind one call ---------------------- */

/* An epi_mctx is ONE population of n_total particles sharded by index over n_shards devices (shard g holds global
 * slots [n_total g / G, n_total (g + 1) / G)), driven by the calling host thread: what a Julia host needs to run
 * update(BootstrapFilter(pomdp, n), b, a, o) (call sites src/POMDPsInterface.jl:56, src/MPC.jl:393) on the eight
 * GPUs of a node through plain ccall, with no process per GPU and no message passing of its own.  The particles, the
 * log-likelihood and the flags are bit-identical for every number of shards and equal to epi_pf_update's on one device
 * (integer weight totals and a maximum: no summation order enters); the effective sample size divides by a binary64
 * sum of squares whose order of addition follows the sharding, so it agrees to rounding (1e-12 relative), not bit for bit.
 *   device_ids  one HIP device per shard; a device may be named more than once (several shards share it: how this is
 *               tested on a machine with one GPU), unless use_rccl is set
 *   use_rccl    0: of the two small exchanges of an update, the maximum log weight is read by every shard from the other
 *               shards' device memory (devices with access to each other, or shards of one device; through page-locked
 *               host memory and one more host wait where the devices have no such access) and the four doubles per
 *               shard of the weight totals go through page-locked host memory, the update's one host wait;
 *               1: both through RCCL (ncclAllReduce MAX, ncclAllGather, one group call each, over xGMI), for which
 *               librccl.so is loaded at run time.  Particles that change shard always go device to device
 *               (hipMemcpyPeerAsync), in packed blocks, pairwise -- no ring.
 * Memory per shard: the belief with a halo of n_local columns plus one send and one receive buffer of n_local
 * particles. */
typedef struct epi_mctx epi_mctx;
int epi_mctx_create(const int *device_ids, int n_shards, const epi_params *p, int64_t n_total, int use_rccl, epi_mctx **out);
int epi_mctx_destroy(epi_mctx *m);
int epi_mctx_shards(const epi_mctx *m);
int64_t epi_mctx_size(const epi_mctx *m);
/* initialbelief / ParticleCollection(particles) / particles(b) of the whole population (host layout of this header) */
int epi_mctx_init_normal(epi_mctx *m, double mu, double sigma, uint64_t seed);
int epi_mctx_upload(epi_mctx *m, const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T);
int epi_mctx_download(epi_mctx *m, int64_t *S, int64_t *I, int64_t *R, int64_t *T);
/* update(upd, b, a, o) over all shards; moved (optional): particles that changed shard in this update.
 * Unless a shard sends more particles than it owns (collapsed weights), the call returns with the device-to-device
 * copies in flight on streams of their own; the next update steps the slots that each shard filled from its own
 * particles while they travel, then places the blocks and steps the rest (epi_pf_predict_part_dev /
 * epi_pf_import_late), and any other call that reads the population places them first.  The posterior is the same
 * either way; epi_mctx_set_overlap(m, 0) makes every update place its blocks before it returns. */
int epi_mpf_update(epi_mctx *m, double a, int64_t o, uint64_t seed, uint32_t step,
                   double *loglik, double *ess, uint32_t *flags, int64_t *moved);
int epi_mctx_set_overlap(epi_mctx *m, int enable);
int64_t epi_mctx_overlapped_updates(const epi_mctx *m);   /* updates so far whose step ran around blocks in flight */
/* epi_belief_sums / epi_belief_mean of the whole population (integer sums added over shards, rounded once) */
int epi_mctx_sums(epi_mctx *m, int64_t *sums);
int epi_mctx_mean(epi_mctx *m, int64_t *S, int64_t *I, int64_t *R, int64_t *T);

/* ---- ensembles kept on the device (ODE fitting, src/ODEFit.jl) ------------------ */

typedef struct epi_ensemble epi_ensemble;   /* the histories of n trajectories over T days, resident in HBM */

/* SimulateEnsemble (src/simulate.jl:210-227) with the histories left on the device: T days from the
 * particles of b (which end at day T); actions as for epi_rollout_batch.  The ensemble records S, sum(I), R
 * before each day's step (Array(::Vector{SimHist}), src/simulate.jl:291-297) and, per trajectory, its
 * first-day action (the T component of the controlled models' initial condition, src/ODEFit.jl:236-239). */
int epi_ensemble_simulate(epi_ctx *ctx, epi_belief *b, int T,
                          const double *actions, int act_per_particle, int act_per_day, double a_prev0,
                          uint64_t seed, uint32_t step0, epi_ensemble **out, uint32_t *flags);
int epi_ensemble_destroy(epi_ensemble *e);
int epi_ensemble_shape(const epi_ensemble *e, int64_t *n, int *T);
/* hist: int64[n][T][3] = S, sum(I), R */
int epi_ensemble_download(epi_ensemble *e, int64_t *hist);

/* SIR_param_loss / SIR_CTRL_param_loss / SEIR_param_loss / SEIR_CTRL_param_loss (src/ODEFit.jl:281-340):
 * for every trajectory the compartment model `kind` (src/ODEFit.jl:1-37) is integrated from the
 * trajectory's day-1 proportions over days 1..T and the squared error against the trajectory's
 * proportions is summed over days and compartments (SEIR: E + I against the infected count);
 * loss = that sum over trajectories / n.  p: alpha, beta | alpha, beta, delta | alpha, beta, gamma |
 * alpha, beta, gamma, delta, eps.  The reference calls an adaptive solver at its default tolerance
 * (1e-3 relative); here: classical Runge-Kutta with `substeps` steps per day.  per_traj (optional):
 * double[n], each trajectory's own sum. */
#define EPI_ODE_SIR       0
#define EPI_ODE_SIR_CTRL  1
#define EPI_ODE_SEIR      2
#define EPI_ODE_SEIR_CTRL 3
int epi_ensemble_loss(epi_ctx *ctx, epi_ensemble *e, int kind, const double *p, int n_p, int substeps,
                      double *loss, double *per_traj);

/* ---- belief summaries ------------------------------------------------------- */

/* Integer column sums over the particles of b: sums[0] = sum S, sums[1] = sum R,
 * sums[2 .. 2+H) = sum I[age], then the (D+1) x H column-major sums of Tests.
 * The sharded mean adds these over shards before rounding. */
int epi_belief_sums(epi_ctx *ctx, epi_belief *b, int64_t *sums /* 2 + H + (D+1) H */);
/* mean(pc, pomdp) (src/updater.jl:5-25): sums / n rounded half to even,
 * R = N - (S + sum I); returns EPI_EINVAL if that R is negative (the
 * reference asserts). */
int epi_belief_mean(epi_ctx *ctx, epi_belief *b, int64_t *S, int64_t *I, int64_t *R, int64_t *T);
int epi_mean_from_sums(const epi_params *p, int64_t n_total, const int64_t *sums,
                       int64_t *S, int64_t *I, int64_t *R, int64_t *T);
/* Order statistics of the infected total sum(I) over the particles of b (per-day belief summaries for
 * plotting a belief history, src/plots.jl:40-54, without keeping the ParticleCollections of
 * src/POMDPsInterface.jl:53): out[j] = the smallest value v such that at least ceil(q[j] n) particles have
 * sum(I) <= v (the minimum for q = 0).  1 to 4 quantiles per call. */
int epi_belief_suminf_quantiles(epi_ctx *ctx, epi_belief *b, const double *q, int nq, int64_t *out);
/* rand(rng, b): one particle, uniformly (src/updater.jl:2 pattern; POMCPOW
 * samples states from the belief). */
int epi_belief_sample(epi_ctx *ctx, epi_belief *b, uint64_t seed, uint64_t ctr,
                      int64_t *S, int64_t *I, int64_t *R, int64_t *T);
/* `count` such particles in one launch and one copy: element j is the particle epi_belief_sample returns for
 * (seed, ctr0 + j).  For planners that draw thousands of states from one belief (POMCPOW with tree_queries = 5000,
 * test/POMCPOWtest.jl:15-21): S[count], R[count], I[count * H] particle-major, T[count * (D+1) * H] per particle as in
 * epi_belief_download; any of them may be NULL.  count <= 2^24. */
int epi_belief_sample_many(epi_ctx *ctx, epi_belief *b, uint64_t seed, uint64_t ctr0, int64_t count,
                           int64_t *S, int64_t *I, int64_t *R, int64_t *T);

/* ---- the closed loop, several days per call -------------------------------------- */

/* `ndays` consecutive days of the closed loop of src/POMDPsInterface.jl:41-60 after the planner has chosen their
 * actions (the reference plans every test_period days and repeats the action in between, :43-47): per day the
 * true system steps (POMDPs.gen on the unit-period model, :55), the belief is updated on that step's observation
 * (:56), and what a plot of the belief history needs of the posterior (src/plots.jl:47; the reference keeps whole
 * ParticleCollections, :53) is reduced on the device.  Everything is enqueued back to back -- the observation goes
 * from the true step to the update in device memory -- and the host waits once, after the last day.
 *   truth    a population of ONE particle on the same context: the true state (advanced ndays days)
 *   b        the belief, a whole population (advanced by ndays updates)
 *   actions  ndays testing proportions; a_prev: the action of the day before the first
 *   true system draws: streams (seed_true, step_true0 + d); filter: (seed, step0 + d)
 * Outputs, each optional (NULL), per day d:
 *   o[d], r[d]      sum(pos_tests) and reward of the true step (src/simulate.jl:160-166)
 *   true_sir[3 d..] S, sum(I), R of the true state AFTER the day's step
 *   loglik[d], ess[d], flags[d]   as epi_pf_update returns them (flags also carry those of the true step)
 *   sums[d (2 + H + (D+1) H) ..]  column sums of the posterior as epi_belief_sums returns them
 *   quant[d nq ..]  order statistics q[0..nq) of sum(I) over the posterior as epi_belief_suminf_quantiles (nq <= 4) */
int epi_closed_loop_days(epi_ctx *ctx, epi_belief *truth, epi_belief *b, int ndays, const double *actions, double a_prev,
                         uint64_t seed_true, uint32_t step_true0, uint64_t seed, uint32_t step0, int nq, const double *q,
                         int64_t *o, double *r, int64_t *true_sir, double *loglik, double *ess, uint32_t *flags,
                         int64_t *sums, int64_t *quant);

/* ---- for callers that cannot allocate ------------------------------------------- */
/* epi_pf_update for hosts that must not call hipMalloc while they run (tight serving loops, harnesses that forbid
 * allocation).  No counterpart in the reference.  The call is synchronous like epi_pf_update (it waits for the
 * stream and reads the diagnostics back), so it cannot run under HIP graph capture.  Same arguments, same results bit
 * for bit; the scratch of the update (scalars, per-tile weight totals, the list of long child ranges: 1.4 MB)
 * lives in a file-scope device array of the library instead of buffers owned by the belief, so the call makes no
 * hipMalloc / hipFree / hipEventCreate of any kind; results come back through the context's page-locked buffer.
 * Limit: at most 2^27 = 134 217 728 particles (EPI_ERANGE beyond).  Every device holds one such array for the life of
 * the process and every call on a device uses it: calls that overlap in time on one device (two contexts, two
 * threads) race; calls in sequence do not (a call leaves nothing queued on the array when it returns).  The belief
 * must be a whole population (not a shard).  epi_pf_collect does not apply to this call: it reads the scalars of the
 * belief's own last epi_pf_update / epi_pf_update_async, which this call does not touch. */
int epi_pf_update_noalloc(epi_ctx *ctx, epi_belief *b, double a, int64_t o, uint64_t seed, uint32_t step,
                          double *loglik, double *ess, uint32_t *flags);

/* ---- introspection for benchmarks ------------------------------------------- */
/* The context's HIP stream (hipStream_t) so that callers can bracket work with
 * their own events. */
void *epi_ctx_stream(epi_ctx *ctx);
/* adopt != 0: use the caller's HIP stream (hipStream_t; a null handle is the default stream; e.g. torch's
 * current stream, on which its RCCL collectives are ordered) for all further work of the context;
 * adopt == 0 restores the context's own stream.  On an
 * adopted stream epi_pf_export / epi_pf_import / epi_pf_select_local / epi_pf_finish(flags = NULL) only
 * enqueue work: ordering with the caller's kernels and collectives is the stream's. */
int epi_ctx_set_stream(epi_ctx *ctx, void *hip_stream, int adopt);
/* Asynchronous variant of epi_pf_update: enqueues the whole update on the
 * context's stream and returns without synchronising; diagnostics stay on the
 * device until epi_pf_collect. */
int epi_pf_update_async(epi_ctx *ctx, epi_belief *b, double a, int64_t o, uint64_t seed, uint32_t step);
int epi_pf_collect(epi_ctx *ctx, epi_belief *b, double *loglik, double *ess, uint32_t *flags);
/* With timing enabled every filter update brackets its fused step kernel with a
 * pair of HIP events on the context's stream; epi_ctx_timing_read waits for the
 * stream, returns the summed kernel time (ms) and the number of launches since
 * the last read, and clears the list. */
int epi_ctx_timing(epi_ctx *ctx, int enable);
int epi_ctx_timing_read(epi_ctx *ctx, double *sum_ms, int64_t *count);
/* Time of the fused predict kernel of the last update, measured with HIP events
 * on the context's stream (milliseconds); negative if unavailable. */
float epi_last_predict_ms(epi_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* EPI_H */
