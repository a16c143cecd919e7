// libepi_gfx950: HIP kernels for the InfectionSim particle step and the
// bootstrap particle-filter update on gfx950 (MI355X), and the C ABI of
// include/epi.h on top of them.
//
// Kernels (one thread per particle or per output slot, 256-thread workgroups =
// 4 wave64, coalesced struct-of-arrays rows):
//   k_step_coop   fused [gather by ancestor] + observation weight + SimStep, phase
//                 by phase over a workgroup with rejected sampler attempts queued
//                 in LDS (states up to H = 21)
//   k_step        the same step with every rejection loop in its own thread (wide
//                 states, injected / recorded variates)
//   k_wsum        fixed-point weights: per-workgroup totals
//   k_wscan       single-workgroup exclusive scan of those totals
//   k_wcum        fixed-point weights again + workgroup prefix scan -> C[i]
//   k_scatter     systematic resampling from the ancestors' side: every particle
//                 writes its index into the contiguous output slots it owns
//                 (k_scatter_big: the long ranges of collapsed weights)
//   k_export / k_import / k_gather / k_replicate   move resampled particles
//   k_sums, k_suminf, k_summary16, k_qsel_pass, k_qsel_narrow   belief mean and order statistics
//   k_score_acc / k_score_mean      discounted reward of scenario rollouts
//   k_ode_loss / k_sum_partials     ODE-model loss of an ensemble kept on the device
//   k_init_*      initial particles;  k_lfact_fill  the log-factorial table
//
// The filter keeps the posterior as (propagated particles, ancestor index):
// the gather of resampled particles is folded into the next step kernel, so one
// update reads each prior particle once and writes each posterior particle
// once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is looked up at run time, when a caller asks for it
#include <new>
#include <string>
#include <vector>
#include <algorithm>
#include <utility>
#include "epi_step.hpp"

using namespace epi;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;
// The testing build (-DEPI_TESTING, libepi_gfx950_testing.so, declared in include/epi_testing.h) can switch the samplers'
// binary32 screens off and force either implementation of the step kernel, to show that neither changes a result;
// in the product both are constants and the alternative kernels are not even instantiated.
#ifdef EPI_TESTING
static int g_screen = 1;     // see epi_debug_screen
static int g_coop = 1;       // see epi_debug_step_impl: 0 per-thread, 1 by state width, 2 cooperative
static int g_node_exchange = 0;   // see epi_debug_node_exchange: bit 0 peer copies between shards of one device, bit 1 maximum through the host
#else
constexpr int g_screen = 1;
constexpr int g_coop = 1;
constexpr int g_node_exchange = 0;
#endif

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// An entry point whose body allocates host memory (std::vector, new) runs inside guarded(): an exception must not
// unwind through the C ABI into a ctypes or ccall frame, it becomes a status code.
template <class F>
static int guarded(const char* who, F&& body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc&) {
        try { return fail(EPI_ENOMEM, "%s: host allocation failed", who); } catch (...) { return EPI_ENOMEM; }
    }
    catch (...) {
        try { return fail(EPI_EINVAL, "%s: unexpected exception", who); } catch (...) { return EPI_EINVAL; }
    }
}

#define HIPCHECK(expr)                                                                  \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(EPI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char* epi_last_error(void) { return g_err.c_str(); }

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------
struct DevScalars {
    unsigned long long lwmax_bits;  // order-preserving code of the local max log weight
    double lwmax;                   // max log weight the fixed-point weights are relative to
    unsigned long long w_local;     // local total of fixed-point weights
    unsigned long long w_off;       // cumulative weight of the shards before this one
    unsigned long long w_total;     // total over all shards
    double sumsq;                   // local sum of squared fixed-point weights
    unsigned int flags;
    unsigned int pad;
};
static_assert(sizeof(DevScalars) <= 256, "the scalars of an update live in a 256-byte slot");

struct epi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;   // created with the context; `stream` is this one unless a caller's was adopted
    bool external = false;              // stream adopted from the caller: the phased update then never blocks the host
    epi_params P;
    bool have_params = false;
    bool tiny_p = false;                // some test probability lies in (0, 2^-700): see launch_step
    DevParams* d_params = nullptr;      // parameters + derived per-bin constants, as the kernels read them
    double* d_lf = nullptr;             // lfact(k), k < LF_TABLE_N
    long long* d_sums = nullptr;        // 2 + H + (D+1) H column sums
    // Small buffers that live as long as the context, so that scalar read-backs do not go through pageable memory and
    // calls on small populations (the true state of a closed loop is a population of one) do not allocate: a context
    // runs one call at a time, and every call that uses them has waited for the stream before it returns.
    unsigned char* h_pin = nullptr;     // PIN_BYTES of page-locked host memory
    unsigned char* d_small = nullptr;   // SMALL_BYTES of device scratch
    unsigned char* d_days = nullptr;    // per-day records of epi_closed_loop_days (grown on demand)
    size_t days_cap = 0;
    unsigned char* d_qsel = nullptr;    // histograms (QSEL_HIST_BYTES, zero between calls) and state of the order-statistic selection
    unsigned char* d_static = nullptr;  // this device's copy of the file-scope scratch array of epi_pf_update_noalloc
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    bool timing = false;                // record an event pair around every fused step kernel of an update
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
};

// PH_PENDING: an update has finished, but some owned slots still wait for particles of other shards (late imports);
// PH_PARTIAL: the step of the next update has run over some of the shard's columns
enum Phase { PH_IDLE = 0, PH_PREDICTED = 1, PH_WEIGHED = 2, PH_PENDING = 3, PH_PARTIAL = 4 };
typedef std::pair<int64_t, int64_t> Span;                   // columns [first, second) of a shard

struct epi_belief {
    epi_ctx* ctx = nullptr;
    int64_t n = 0, halo_cap = 0, stride = 0;
    int64_t n_total = 0, slot0 = 0;
    int rows = 0;
    int32_t* buf[2] = {nullptr, nullptr};
    int cur = 0;
    int32_t* anc = nullptr;
    bool has_anc = false;
    double* lw = nullptr;               // log weights, then (same memory) uint64 cumulative weights
    unsigned long long* blocksum = nullptr;
    double* blocksq = nullptr;
    unsigned long long* big = nullptr;  // [0] = number of listed long child ranges, then the ranges (k_scatter)
    DevScalars* scal = nullptr;         // the scalars of this belief's updates (maximum log weight, weight totals, step flags)
    int64_t nblocks = 0;
    int phase = PH_IDLE;
    int64_t halo_used = 0;
    int frac_bits = 0;
    uint64_t seed = 0; uint32_t step = 0;
    // bookkeeping of the phased update: columns that received an ancestor in the update in progress, columns that
    // still wait for one after epi_pf_finish, columns the step of the next update has covered so far
    bool cum_ready = false;             // the update in progress has written its cumulative weights (over the log weights)
    bool selected = false;              // epi_pf_select_local has run in the update in progress (once per update)
    bool flags_delivered = false;       // ... and handed its step flags over with the weight totals (device flags are clear again)
    std::vector<Span> filled, unfilled, done;
    double part_a = 0.0; int64_t part_o = 0;                // arguments of the first epi_pf_predict_part of an update
};

// back to "no update in progress" (after the population has been replaced)
static inline void protocol_reset(epi_belief* b)
{
    b->phase = PH_IDLE;
    b->filled.clear(); b->unfilled.clear(); b->done.clear(); b->selected = false;
}

static bool spans_overlap(const std::vector<Span>& v, int64_t a, int64_t b)
{
    for (const Span& s : v) if (a < s.second && s.first < b) return true;
    return false;
}

// the parts of [0, n) that no span covers
static std::vector<Span> spans_complement(std::vector<Span> v, int64_t n)
{
    std::sort(v.begin(), v.end());
    std::vector<Span> out;
    int64_t at = 0;
    for (const Span& s : v) {
        if (s.first > at) out.emplace_back(at, s.first);
        if (s.second > at) at = s.second;
    }
    if (at < n) out.emplace_back(at, n);
    return out;
}

static bool spans_contain(const std::vector<Span>& v, int64_t a, int64_t b)
{
    for (const Span& s : v) if (s.first <= a && b <= s.second) return true;
    return false;
}

// takes [a, b) out of the spans; false unless one of them contains it
static bool spans_remove(std::vector<Span>& v, int64_t a, int64_t b)
{
    for (size_t k = 0; k < v.size(); k++) {
        if (v[k].first <= a && b <= v[k].second) {
            const Span s = v[k];
            v.erase(v.begin() + (long)k);
            if (s.first < a) v.emplace_back(s.first, a);
            if (b < s.second) v.emplace_back(b, s.second);
            return true;
        }
    }
    return false;
}

struct epi_ensemble {
    epi_ctx* ctx = nullptr;
    int64_t n = 0;
    int T = 0;
    int32_t* hist = nullptr;            // [T][4][n]: S, sum I, R before each day's step, sum pos of the day
    double* act = nullptr;              // [n]: the trajectory's (first-day) action, for the controlled ODE models
    double* partial = nullptr;          // per-workgroup partial sums of a loss evaluation, then the total
};

static const int TPB = 256;            // threads per workgroup
static const int LF_TABLE_N = 1 << 20; // tabulated log factorials (8 MB per context); larger arguments are computed
static const int SCAN_ITEMS = 8;       // fixed-point weights per thread in the scan kernels
static const int SCAN_TILE = TPB * SCAN_ITEMS;

static const size_t QSEL_HIST_BYTES = 4 * 2048 * sizeof(unsigned long long);   // histograms of the order-statistic selection (k_qsel_pass)
static const size_t PIN_BYTES = 128 << 10;
static const size_t SMALL_BYTES = 128 << 10;

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
// the list of child ranges longer than 8192 slots that the scatter kernels leave to k_scatter_big: a count, then at most
// n / 8192 + 1 ranges of 24 bytes
static constexpr size_t big_list_bytes(int64_t n) { return 16 + 24 * ((size_t)n / 8192 + 2); }

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long order_code(double x)
{
    unsigned long long b = d2u(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ __forceinline__ double order_decode(unsigned long long c)
{
    unsigned long long b = (c >> 63) ? (c & 0x7fffffffffffffffULL) : ~c;
    return u2d(b);
}

__device__ __forceinline__ unsigned long long shfl_down_u64(unsigned long long v, int d)
{
    unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
    lo = __shfl_down(lo, d, 64); hi = __shfl_down(hi, d, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d)
{
    unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
    lo = __shfl_up(lo, d, 64); hi = __shfl_up(hi, d, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int from)
{
    unsigned int lo = (unsigned int)v, hi = (unsigned int)(v >> 32);
    lo = __shfl(lo, from, 64); hi = __shfl(hi, from, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_down_f64(double v, int d) { return u2d(shfl_down_u64(d2u(v), d)); }

// ---------------------------------------------------------------------------
// k_step: [gather] + [weight] + SimStep
// ---------------------------------------------------------------------------
struct StepArgs {
    int64_t n;                  // particles handled by this launch
    const int32_t* in; int64_t in_stride;
    const int32_t* anc;         // source column of each particle in `in`, or null = identity
    int32_t* out; int64_t out_stride;
    double a; const double* a_dev; int64_t a_row_stride;   // action: scalar or a_dev[row(m) * a_row_stride + a_col]
    int64_t a_col;
    int64_t a_div;              // row(m) = m / a_div when a_div > 1 (consecutive particles share an action row), else m
    double a_prev; const double* a_prev_dev; int64_t a_prev_col;   // previous action: scalar or a_prev_dev[row(m) * a_row_stride + a_prev_col]
    uint64_t seed; uint32_t step; uint64_t slot0;
    // filter
    double o; double* lw; DevScalars* scal;
    const int32_t* o_dev;       // when set, the observation is read from device memory (written by an earlier kernel on the stream) instead of `o`
    // optional per-particle outputs
    int32_t* o_out; double* r_out; int32_t* newinf_out;
    // optional history row of a rollout: hist[c * n + m], c = 0..3
    int32_t* hist;
    // injected variates
    const int32_t* inj_sympt; const int32_t* inj_pos; const double* inj_gam;
    int32_t* pos_out;
    // recorded variates (MODE_RECORD)
    int32_t* rec_sympt; int32_t* rec_pos; double* rec_gam;
    int screen;                 // 1: the rejection samplers may use their binary32 screens
    int64_t block0;             // k_step_coop: index of the launch's first workgroup within the population
    int padded;                 // the rows of `out` and `lw` extend to a multiple of 256 particles: the build without bounds checks may
                                // run the last, partial workgroup too (its surplus threads step copies of the last particle into the padding)
};

enum { MODE_PHILOX = 0, MODE_INJECT = 1, MODE_RECORD = 2 };

// Workgroups are dealt to the 8 XCDs in turn (b and b + 8 share one, with its L2).  The step kernels give the
// workgroups of one XCD a contiguous share of the tiles, so that the lines of the ancestor gather which two neighbouring
// tiles both touch are fetched into one L2 instead of two: a bijection of [0, gridDim.x) for every grid size.  (Which
// tile a workgroup steps has no bearing on the result: variates are addressed by slot.)
__device__ __forceinline__ unsigned int xcd_tile()
{
    const unsigned int nb = gridDim.x, x = blockIdx.x & 7u, q = nb >> 3, r = nb & 7u;
    return x * q + (x < r ? x : r) + (blockIdx.x >> 3);
}

constexpr int THREAD_WAVES = 4;        // waves per SIMD the per-thread step kernel is allocated for (128 VGPRs; 13 % faster than 3 waves)
// SCREEN = true: the samplers' binary32 screens are on at compile time (production); false: the run-time flag decides
template <bool WEIGH, int MODE, bool DELAY = true, bool SCREEN = false>
__global__ __launch_bounds__(TPB, THREAD_WAVES) void k_step(const StepArgs A, const DevParams* __restrict__ Dp)
{
    const epi_params& P = Dp->P;
    __shared__ unsigned long long s_max[TPB / 64];
    const int64_t m = (int64_t)xcd_tile() * TPB + threadIdx.x;
    unsigned long long code = 0ULL;       // below every real code
    unsigned int flags = 0;
    if (m < A.n) {
        const int64_t src = A.anc ? (int64_t)A.anc[m] : m;
        const int64_t arow = A.a_div > 1 ? m / A.a_div : m;
        const double a = A.a_dev ? A.a_dev[arow * A.a_row_stride + A.a_col] : A.a;
        StepResult res;
        int32_t s_pre = 0, r_pre = 0, suminf_pre = 0;
        if (A.hist) {
            s_pre = A.in[src]; r_pre = A.in[A.in_stride + src];
            for (int i = 0; i < P.H; i++) suminf_pre += A.in[(int64_t)(2 + i) * A.in_stride + src];
        }
        if constexpr (MODE == MODE_INJECT) {
            InjectedDraws d{A.inj_sympt, A.inj_pos, A.inj_gam, A.n, m};
            res = particle_step(P, A.in, A.in_stride, src, A.out, A.out_stride, m, a, d, A.pos_out, A.n);
        } else if constexpr (MODE == MODE_RECORD) {
            RecordingDraws d{PhiloxDraws{rng_make(A.seed, A.slot0 + (uint64_t)m, A.step), A.screen != 0, Dp, 0u, 0u}, A.rec_sympt, A.rec_pos, A.rec_gam, A.n, m};
            res = particle_step(P, A.in, A.in_stride, src, A.out, A.out_stride, m, a, d, A.pos_out, A.n);
        } else {
            PhiloxDraws d{rng_make(A.seed, A.slot0 + (uint64_t)m, A.step), SCREEN ? true : A.screen != 0, Dp, 0u, 0u};
            res = particle_step<DELAY>(P, A.in, A.in_stride, src, A.out, A.out_stride, m, a, d, A.pos_out, A.n);
        }
        flags = res.flags;
        if constexpr (WEIGH) {
            double lw = log_weight(res.mu, res.var, A.o_dev ? (double)*A.o_dev : A.o);
            A.lw[m] = lw;
            code = order_code(lw);
        }
        if (A.o_out) A.o_out[m] = res.o;
        if (A.newinf_out) A.newinf_out[m] = res.new_inf;
        if (A.r_out) {
            double ap = A.a_prev_dev ? A.a_prev_dev[arow * A.a_row_stride + A.a_prev_col] : A.a_prev;
            A.r_out[m] = reward_of(P, res.sum_inf, a, ap);
        }
        if (A.hist) {
            A.hist[m] = s_pre; A.hist[A.n + m] = suminf_pre; A.hist[2 * A.n + m] = r_pre; A.hist[3 * A.n + m] = res.o;
        }
    }
    // flags: one atomic per wave that saw any
    unsigned int wf = flags;
    for (int d = 32; d >= 1; d >>= 1) wf |= __shfl_down(wf, d, 64);
    if ((threadIdx.x & 63) == 0 && wf) atomicOr(&A.scal->flags, wf);
    if constexpr (WEIGH) {
        for (int d = 32; d >= 1; d >>= 1) {
            unsigned long long o = shfl_down_u64(code, d);
            code = o > code ? o : code;
        }
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = code;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long c = s_max[0];
            for (int w = 1; w < TPB / 64; w++) c = s_max[w] > c ? s_max[w] : c;
            atomicMax(&A.scal->lwmax_bits, c);
        }
    }
}

// ---------------------------------------------------------------------------
// k_step_one: the step of a population of ONE particle (the true state of a closed loop, POMDPs.gen on a single state,
// Simulate), one lane per infection age in a single wavefront.  The two other kernels take a lone particle through its
// 3 H draws one after the other (40-50 us of dependent latency); here every age bin draws at once.  Same variates
// (addressed by slot, step, bin), same operations per bin, the infection sum added in bin order: the same bits.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_step_one(const StepArgs A, const DevParams* __restrict__ Dp)
{
    const epi_params& P = Dp->P;
    const int H = P.H, D = P.D;
    const int i = threadIdx.x;                                  // age bin of this lane (and its infection term)
    const bool act = i < H;
    const int64_t src = A.anc ? (int64_t)A.anc[0] : 0;
    const double a = A.a_dev ? A.a_dev[A.a_col] : A.a;
    const Rng g = rng_make(A.seed, A.slot0, A.step);
    const bool screen = A.screen != 0;
    unsigned int flags = 0;
    int32_t r_add = 0, pos = 0, ipost = 0, I_pre = 0;
    if (act) {
        int32_t Ii = A.in[(int64_t)(2 + i) * A.in_stride + src];
        I_pre = Ii;
        int32_t already = 0;
        if (D > 0 && i >= 1)
            for (int r = 0; r < D; r++) already += A.in[(int64_t)tests_row(H, r, i) * A.in_stride + src];
        const double ntf = __builtin_floor((double)(Ii - already) * a);                    // simulate.jl:78
        const int32_t nt = (int32_t)ntf;
        if (nt < 0) flags |= EPI_F_NEG_NT;
        uint32_t w[4];
        rng_block(g, K_BINOM0, (uint32_t)i, 0, w);
        const int32_t sympt = bin_binomial(*Dp, i, 0, Ii, w[0], w[1], g, screen);        // simulate.jl:58
        pos = bin_binomial(*Dp, i, 1, nt, w[2], w[3], g, screen);                        // simulate.jl:79
        double pr = div_unscaled((double)sympt, (double)Ii);                             // simulate.jl:104
        if (pr != pr) pr = 0.0;                                                          // simulate.jl:105
        const double om = 1.0 - pr;
        r_add = sympt; Ii -= sympt;                                                      // simulate.jl:107-108
        for (int r = 0; r <= D; r++) {                                                   // simulate.jl:110-126
            int32_t tv;
            if (r == D) tv = pos;
            else tv = (i >= 1) ? A.in[(int64_t)tests_row(H, r, i) * A.in_stride + src] : 0;
            const int32_t ts = (int32_t)__builtin_floor(om * (double)tv);
            if (r == 0) { r_add += ts; Ii -= ts; }
            else if (i + 1 < H) A.out[(int64_t)tests_row(H, r - 1, i + 1) * A.out_stride] = ts;
        }
        if (Ii < 0) flags |= EPI_F_NEG_I;                                                // simulate.jl:117
        ipost = Ii;
        if (A.pos_out) A.pos_out[i] = pos;
    }
    // infection term `it` is seeded by the survivors of bin it - 1, term 0 by those of the last bin (simulate.jl:144-145)
    const int from = i == 0 ? H - 1 : i - 1;
    const int32_t seedling = __shfl(ipost, from, 64);
    double gval = 0.0;
    if (act) {
        if (i >= 1) A.out[(int64_t)(2 + i) * A.out_stride] = seedling;
        uint32_t wz, wu;
        gamma_first(g, (uint32_t)i, wz, wu);
        gval = gamma_draw(P.k[i] * (double)seedling, P.theta[i], wz, wu, &Dp->zig, g, (uint32_t)i, screen);   // simulate.jl:40, init.jl:71-78
    }
    int32_t r_sum = r_add, o_sum = pos, pre_sum = I_pre, post_sum = (act && i >= 1) ? seedling : 0;
    for (int d = 32; d >= 1; d >>= 1) {
        r_sum += __shfl_down(r_sum, d, 64); o_sum += __shfl_down(o_sum, d, 64);
        pre_sum += __shfl_down(pre_sum, d, 64); post_sum += __shfl_down(post_sum, d, 64);
        flags |= __shfl_down(flags, d, 64);
    }
    const int32_t last_post = __shfl(ipost, H - 1, 64);
    double infsum = 0.0;
    for (int it = 0; it < H; it++) infsum += u2d(((uint64_t)__shfl((unsigned int)(d2u(gval) >> 32), it, 64) << 32) | __shfl((unsigned int)d2u(gval), it, 64));   // in term order
    if (i == 0) {
        const int32_t S = A.in[src], R_pre = A.in[A.in_stride + src];
        const int32_t R = R_pre + r_sum + last_post;                                     // simulate.jl:107,115,144
        const int64_t denom = P.N - (int64_t)R;
        int32_t newi = 0;
        if (denom <= 0) flags |= EPI_F_BAD_DENOM;
        else newi = (int32_t)__builtin_floor(div_unscaled((double)S, (double)denom) * infsum);     // simulate.jl:42-44
        A.out[(int64_t)2 * A.out_stride] = newi;                                         // simulate.jl:147
        A.out[0] = S - newi;                                                             // simulate.jl:148
        A.out[A.out_stride] = R;
        if (A.o_out) A.o_out[0] = o_sum;
        if (A.newinf_out) A.newinf_out[0] = newi;
        if (A.r_out) {
            const double ap = A.a_prev_dev ? A.a_prev_dev[A.a_prev_col] : A.a_prev;
            A.r_out[0] = reward_of(P, post_sum + newi, a, ap);
        }
        if (A.hist) { A.hist[0] = S; A.hist[1] = pre_sum; A.hist[2] = R_pre; A.hist[3] = o_sum; }
        if (flags) atomicOr(&A.scal->flags, flags);
    }
}

// ---------------------------------------------------------------------------
// k_step_coop: the same fused step as k_step<WEIGH, MODE_PHILOX>, evaluated phase
// by phase over a workgroup with failed sampler attempts handed to whichever
// thread is free.  A rejection sampler run per thread costs a wavefront the
// MAXIMUM attempt count of its 64 lanes; here every draw makes its first attempt
// in its own thread, draws that were rejected are queued in LDS, and each retry
// round spreads the queued draws evenly over the 256 threads.  Variates are
// addressed by (slot, step, bin, attempt), so the result is bit-identical to the
// per-thread kernel.
//   A  the particle's infection ages (requested at the kernel's start) staged as (I_i, tested_i) per age bin in LDS, log weight
//   B  attempt 0 of both binomials of every bin without the log-factorial table (squeeze and series bounds);
//      draws it rejects or leaves open -> queue
//   C  three retry rounds (the next attempt in full, with the table), then a draw is finished by the thread that holds it
//   D  isolation bookkeeping per bin (loads two bins ahead), attempt 0 of the gamma term (ziggurat strip and squeeze); the rest -> queue
//   E  one gamma retry round, then finished in thread
//   F  S and R, infection sum in bin order, new infections, outputs
// LDS: one 8-byte slot per (bin, particle) + a 2048- and a 1024-entry queue + three rotating entry counts + the ziggurat
// strips + the per-bin records of the retry rounds.
// ---------------------------------------------------------------------------
#ifndef EPI_CTPB
#define EPI_CTPB 256
#endif
constexpr int CTPB = EPI_CTPB;                   // threads per workgroup of the cooperative step kernel (queue entries hold 8 bits of thread index)
constexpr unsigned int QCAP = 8 * CTPB;          // entries of the first retry queue (what the first-attempt phases leave open; rounds 2, 4, ... write it again)
constexpr unsigned int QSEG = QCAP / (CTPB / 64); // entries of one wave's segment of the first queue
constexpr unsigned int QCAP1 = 4 * CTPB;         // entries of the second one (what round 1, 3, ... leave open: a fraction of their input)
static_assert(CTPB == 64 || CTPB == 128 || CTPB == 256, "whole wavefronts, 8 bits of thread index, rows padded to 256 particles");

// Binary64 square root for arguments that are zero or at least 2^-767: the compiler's own expansion (reciprocal
// square root estimate, one coupled Goldschmidt step, two residual corrections) without the pre- and post-scaling by
// powers of two that it wraps around the same steps for arguments below 2^-767, where the estimate would lose bits.
// The steps and their order are those of the expansion, so the result is the same bit for bit.
__device__ __forceinline__ double sqrt_unscaled(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y, h0 = y * 0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    const double g3 = __builtin_fma(d1, h1, g2);
    return x == 0.0 ? x : g3;
}

// a rejection draw from attempt 0 on in one thread (a full retry queue)
__device__ __noinline__ int32_t btrs_from0(int32_t n, const BtrsBin B, const LfTable lf, bool screen, const Rng g, int j, uint32_t bin)
{
    uint32_t w[4];
    rng_block(g, K_BINOM0, bin, 0, w);
    const BtrsConsts C = btrs_consts(n, B);
    const int32_t k = btrs_attempt(C, B.lpq, lf, screen, j ? w[2] : w[0], j ? w[3] : w[1]);
    return k >= 0 ? k : btrs_finish(C, B.lpq, lf, screen, g, j ? K_POS : K_SYMPT, bin, 1u);
}
// the remaining attempts of a rejection draw in one thread; one attempt for an n beyond the log-factorial table
__device__ __noinline__ int32_t btrs_finish_thread(const BtrsConsts C, double lpq, const LfTable lf, bool screen, const Rng g, uint32_t kind, uint32_t bin, uint32_t att)
{
    return btrs_finish(C, lpq, lf, screen, g, kind, bin, att);
}
__device__ __noinline__ int32_t btrs_attempt_far(const BtrsConsts C, double lpq, const LfTable lf, bool screen, uint32_t wa, uint32_t wb)
{
    return btrs_attempt(C, lpq, lf, screen, wa, wb);
}
// an infection term from attempt 0 on in one thread
__device__ __noinline__ double gamma_from0(double shape, double scale, const ZigTable* __restrict__ Z, const Rng g, uint32_t term, bool screen)
{
    uint32_t wz, wu;
    gamma_first(g, term, wz, wu);
    return gamma_draw(shape, scale, wz, wu, Z, g, term, screen);
}

// what a retry round reads of a (draw, bin) record, kept in LDS
struct alignas(8) CoopRec { double lpq; uint32_t ppx; float pq, a0x2; int32_t flip; };
static_assert(sizeof(CoopRec) == 24, "three 8-byte words");

// the number of lanes below this one whose bit is set in `mask` (two instructions, no lane mask to keep)
__device__ __forceinline__ unsigned int rank_in(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
}

// push `entry` for every lane that wants to (one LDS atomic per wave); false = queue full
__device__ __forceinline__ bool q_push(bool want, unsigned short entry, unsigned short* q, unsigned int* qn, unsigned int cap)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ULL) return true;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(qn, (unsigned int)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (!want) return true;
    const unsigned int pos = base + rank_in(mask);
    if (pos >= cap) return false;
    q[pos] = entry;
    return true;
}

// The posterior state and the log weights are written once and read by a later kernel: stored with the
// non-temporal hint so that they do not displace the workgroups' spill lines and the tables from the L2.
#define EPI_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))

// -DEPI_PHASE_STAMPS (variant builds for measurements only): thread 0 of every workgroup adds the clock ticks it spent
// in each phase of k_step_coop to g_phase[]; epi_debug_phase_times reads and clears them.
#ifdef EPI_PHASE_STAMPS
// (g_phase[] itself is defined beside the samplers, which count into it as well)
// (the counter is read by an instruction of its own with a memory clobber: the compiler may not move a phase's LDS and
// global accesses across it, which it does with the plain builtin)
__device__ __forceinline__ unsigned long long epi_ticks()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define EPI_STAMP(k) do { const unsigned long long now_ = epi_ticks(); if (threadIdx.x == 0) atomicAdd(&g_phase[k], now_ - stamp_); stamp_ = now_; } while (0)
#else
#define EPI_STAMP(k) do { } while (0)
#endif

#ifndef EPI_D_AHEAD
#define EPI_D_AHEAD 2
#endif
constexpr int D_AHEAD = EPI_D_AHEAD;
#ifndef EPI_COOP_BINOM_ROUNDS
#define EPI_COOP_BINOM_ROUNDS 3
#endif
constexpr uint32_t COOP_BINOM_ROUNDS = EPI_COOP_BINOM_ROUNDS;   // cooperative rounds of the binomial retries; what is still open after them (about one draw in
                                            // a thousand) is finished by the thread that holds it, without further barriers
#ifndef EPI_COOP_GAMMA_ROUNDS
#define EPI_COOP_GAMMA_ROUNDS 1
#endif
constexpr uint32_t COOP_GAMMA_ROUNDS = EPI_COOP_GAMMA_ROUNDS;   // the same for the infection terms (a term in a thousand is open after one round)
#ifndef EPI_COOP_WAVES
#define EPI_COOP_WAVES 4
#endif
constexpr int COOP_WAVES = EPI_COOP_WAVES;   // waves per SIMD the register allocation aims at (128 VGPRs): the table loads of the
                                // binomial acceptance test want the occupancy; 3 waves (138 VGPRs, no spills) is 19 % slower
// DELAY = false: the build for test_delay = 0 (the reference's default), without the pending-test rows and their loops.
// HC > 0: the number of infection ages is this constant (the reference's default horizon has a build of its own).
// SCREEN: the rejection samplers use their binary32 screens (a compile-time constant in the production builds: the
// flag only selects code, never a result; the builds without screens exist for the test that shows exactly that).
// FULL: no thread needs the bounds check: either the population fills every workgroup of the grid, or the output rows
// are padded to whole workgroups and the surplus threads of the last one step copies of the last particle into the
// padding (their optional outputs and flags are dropped; their log weight equals the last particle's, so the maximum
// is unchanged).
// NC > 1 (wide states): a particle is stepped by NC threads of the workgroup, each owning every NC-th pair of
// infection terms (terms 2j and 2j+1 share a block of variates) and the age bins that seed them, so that a workgroup
// holds 256 / NC particles and its LDS footprint -- 8 H bytes per particle -- stays that of a narrow state.  Thread tid
// steps particle tid mod (256 / NC) with share tid / (256 / NC) (one share per wave or pair of waves: bin indices stay
// wave-uniform; dealing the pairs out in turn gives every wave young and old ages alike, whose draws differ in cost).
// Draws are addressed by (slot, bin), so which thread makes them is immaterial; the integer sums over bins are
// combined through LDS, and everything whose order of evaluation enters the result (the observation moments, the
// infection sum) is taken in bin order by the thread of share 0.
template <bool WEIGH, bool DELAY, int HC = 0, bool SCREEN = true, bool FULL = false, int NC = 1>
__global__ __launch_bounds__(CTPB, COOP_WAVES) void k_step_coop(const StepArgs A, const DevParams* __restrict__ Dp)
{
    static_assert(NC == 1 || (HC == 0 && !FULL), "chunked particles: run-time H, bounds-checked");
    static_assert(NC == 1 || NC == 2 || NC == 4, "one share per whole number of waves");
    // The weighing step of an update has no per-particle outputs (observation, reward, new infections, history row: those
    // belong to plain steps and rollouts), so its build does not carry their sums through the sampler phases.
    constexpr bool OUTS = !WEIGH;
    constexpr int PW = CTPB / NC;                                                  // particles per workgroup
    const epi_params& P = Dp->P;
    // Arguments needed only at the end of the kernel are re-read from the kernel-argument segment there (scalar
    // loads) instead of occupying scalar registers across the sampler phases, which are short of them.
    typedef const StepArgs __attribute__((address_space(4)))* KernArgs;
    const KernArgs Ak = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = HC ? HC : P.H, D = DELAY ? P.D : 0;
    int2* slots = reinterpret_cast<int2*>(smem);                                   // [H][PW]: (I_i, tested_i) -> (sympt_i, pos_i)
    int* slot_words = reinterpret_cast<int*>(smem);
    double* gslots = reinterpret_cast<double*>(smem);                              // [H][PW]: gamma term of the bin's survivors
    unsigned short* q0 = reinterpret_cast<unsigned short*>(smem + (size_t)H * PW * 8);
    unsigned short* q1 = q0 + QCAP;
    unsigned int* qn = reinterpret_cast<unsigned int*>(q1 + QCAP1);                 // qn[0..2]: entry counts, rotating (+ 1 pad)
    unsigned int* qw = qn + 4;                                                     // [CTPB/64]: entries of each wave's segment of the first queue
    unsigned long long* s_max = reinterpret_cast<unsigned long long*>(qw + CTPB / 64);   // [CTPB/64]
    ZigEntry* ztab = reinterpret_cast<ZigEntry*>(s_max + CTPB / 64);               // [ZIG_N]: strips of the normal generator
    CoopRec* crec = reinterpret_cast<CoopRec*>(ztab + ZIG_N);                       // [2][H]: per-(draw, bin) constants for the retry rounds
    int* acc = reinterpret_cast<int*>(crec + 2 * H);                               // NC > 1, [4][PW]: the chunks' shares of R, o, sum(I') and sum(I)

    const int tid = threadIdx.x;
    const int lp = NC == 1 ? tid : (tid & (PW - 1));                               // the thread's particle within the workgroup
    const int ch = NC == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid / PW);         // its share (wave-uniform)
    // its terms, in order: term_of(0 .. nown); term `it` is seeded by bin (it == 0 ? H-1 : it-1) (simulate.jl:145)
    const int J = (H + 1) >> 1;                                                    // pairs of terms
    const int npair = NC == 1 ? J : (ch < J ? (J - ch + NC - 1) / NC : 0);
    const int nown = NC == 1 ? H : 2 * npair - (((H & 1) && npair > 0 && (J - 1) % NC == ch) ? 1 : 0);
    auto term_of = [&](int k) -> int { return NC == 1 ? k : 2 * (ch + (k >> 1) * NC) + (k & 1); };
    auto bin_of = [&](int k) -> int { const int it = term_of(k); return it == 0 ? H - 1 : it - 1; };
    auto owns = [&](int i) -> bool { return NC == 1 || (((i + 1 == H ? 0 : i + 1) >> 1) % NC) == ch; };
    const bool lead = NC == 1 || ch == 0;                                          // the thread that holds S and R and closes the step
    const int64_t base = ((int64_t)xcd_tile() + A.block0 * NC) * PW;
    const int64_t m = base + lp;
    const bool active = FULL ? true : m < A.n;
    const int64_t mc = FULL ? (m < A.n ? m : A.n - 1) : m;                         // the particle this thread reads
    const int64_t src = active ? (A.anc ? (int64_t)A.anc[mc] : mc) : 0;
    const int64_t arow = A.a_div > 1 ? mc / A.a_div : mc;
    const double a = active ? (A.a_dev ? A.a_dev[arow * A.a_row_stride + A.a_col] : A.a) : 0.0;
    const Rng g = rng_make(A.seed, A.slot0 + (uint64_t)m, A.step);
    const bool screen = SCREEN;
    // The row strides and base pointers arrive in one wide scalar load; copied into register pairs of their own,
    // a reload after a scalar-register spill restores two registers instead of the whole eight-register tuple.
    uint64_t in_u, out_u, is_u, os_u;
    asm volatile("s_mov_b64 %0, %1" : "=s"(in_u) : "s"((uint64_t)A.in));
    asm volatile("s_mov_b64 %0, %1" : "=s"(out_u) : "s"((uint64_t)A.out));
    asm volatile("s_mov_b64 %0, %1" : "=s"(is_u) : "s"((uint64_t)A.in_stride));
    asm volatile("s_mov_b64 %0, %1" : "=s"(os_u) : "s"((uint64_t)A.out_stride));
    const int32_t* __restrict__ in = (const int32_t*)in_u;
    int32_t* __restrict__ out = (int32_t*)out_u;
    const int64_t is = (int64_t)is_u, os = (int64_t)os_u;
    // state rows: one coalesced 4-byte access per lane and row
    auto ld_in = [&](int64_t row) -> int32_t { return in[row * is + src]; };
    auto st_out = [&](int64_t row, int32_t v) { EPI_STREAM_STORE(&(out + (row * os + base))[lp], v); };
    // Phase A's loads of the infection ages are requested here, all of them when their number is a constant of the build
    // (else the first sixteen -- a thread's share of a wide state -- and the rest running that far ahead of their use): a bin's arithmetic there is a few dozen
    // instructions, a load from HBM takes the time of several hundred, a wave has at most three others to hide behind, and
    // the set-up below and its barrier pass in the meantime.  (One bin ahead: 19.2 ms; four or eight: the same; all: 18.8.)
    constexpr int A_AHEAD = HC > 0 ? HC : 16;
    const bool a_every = NC == 1 || (WEIGH && lead);                               // this thread visits every bin in phase A
    const int a_bins = a_every ? H : nown;
    int32_t Iq[A_AHEAD];
#pragma unroll
    for (int k = 0; k < A_AHEAD; k++) Iq[k] = (active && k < a_bins) ? ld_in(2 + (a_every ? k : bin_of(k))) : 0;
    if (tid == 0) { qn[0] = 0u; qn[1] = 0u; qn[2] = 0u; }
    if (WEIGH && tid < CTPB / 64) s_max[tid] = 0ULL;
    if constexpr (NC > 1) for (int r = tid; r < 4 * PW; r += CTPB) acc[r] = 0;
    for (int r = tid; r < ZIG_N; r += CTPB) ztab[r] = Dp->zig.e[r];
    for (int r = tid; r < 2 * H; r += CTPB) {
        const BinConsts& bc = Dp->bc[r >= H][r >= H ? r - H : r];
        crec[r] = CoopRec{bc.lpq, bc.ppx, bc.pq, bc.a0x2, bc.flip};
    }
    // the cleared counts and the strips must be visible before any wave pushes to the queue (after phase B): the barrier stands here,
    // where the waves of the workgroup still run together, and not between phases A and B, which only exchange data
    // within a thread
    __syncthreads();
    unsigned int flags = 0;
    int32_t S = 0, R = 0, R_pre = 0, o_sum = 0, suminf_pre = 0;
    unsigned long long code = 0ULL;
#ifdef EPI_PHASE_STAMPS
    unsigned long long stamp_ = epi_ticks();
#endif

    // ---- A: stage the bins, observation moments -----------------------------------
    if (active) {
        // (S and R are read at the end, where they are needed: R counts the step's recoveries from zero until then)
        double tm = 0.0, tv = 0.0;
        // the observation moments are sums in bin order: with chunks, the lead thread visits every bin for them
        // (re-reading the other chunks' rows, which the cache holds) and stages only its own
        const bool moments = WEIGH && lead;
        const bool every = NC == 1 || moments;                                     // all bins, in order
        const int na = every ? H : nown;
        // (the default horizon's fourteen ages in two unrolled halves, a run-time horizon's four at a time: the queue of requested ages then moves by renaming, not by
        // thirteen copies per age -- 18.36 -> 18.09 ms; unrolled in full the loop spills: 19.0)
        constexpr int A_UNROLL = HC == 14 ? 7 : 4;
#pragma unroll A_UNROLL
        for (int q = 0; q < na; q++) {
            const int i = every ? q : bin_of(q);
            const int32_t Ii = Iq[0];
#pragma unroll
            for (int k = 0; k + 1 < A_AHEAD; k++) Iq[k] = Iq[k + 1];
            if (q + A_AHEAD < na) Iq[A_AHEAD - 1] = ld_in(2 + (every ? q + A_AHEAD : bin_of(q + A_AHEAD)));
            int32_t already = 0;
            if (D > 0 && i >= 1)
                for (int r = 0; r < D; r++) already += ld_in(tests_row(H, r, i));
            const double ntf = __builtin_floor((double)(Ii - already) * a);        // simulate.jl:78
            const int32_t nt = (int32_t)ntf;
            if (nt < 0) flags |= EPI_F_NEG_NT;
            if (owns(i)) {
                slots[i * PW + lp] = make_int2(Ii, nt);
                if constexpr (OUTS) suminf_pre += Ii;
            }
            if (WEIGH && moments) {
                const double pi = P.p[i];
                tm += ntf * pi;                                                    // POMDPsInterface.jl:19
                if (pi > 0.0 && pi < 1.0) {                                        // else std = 0 exactly: nothing to add
                    // the launcher keeps parameter sets with a test probability below 2^-700 away from this kernel
                    // (a uniform branch on pi here costs more than the shorter form saves: 24.1 vs 23.4 ms)
                    const double sd = sqrt_unscaled(ntf * pi * (1.0 - pi));
                    tv += sd * sd;                                                 // POMDPsInterface.jl:20
                }
            }
        }
        if (WEIGH && moments) {
            const double lw = log_weight(tm, tv, Ak->o_dev ? (double)*Ak->o_dev : Ak->o);
            EPI_STREAM_STORE(&(Ak->lw + base)[lp], lw);
            code = order_code(lw);
        }
    }
    // the largest log weight of the wave is collected in LDS here (the barriers of the later phases order it before
    // thread 0 reads it at the end), so that the code does not occupy a register pair through the sampler phases; by LDS
    // atomics rather than shuffles, whose lane addresses the compiler would keep for the reduction of the flags at the end
    if constexpr (WEIGH) { if (code) atomicMax(&s_max[tid >> 6], code); }

    EPI_STAMP(0);
    // ---- B: the table-free part of attempt 0 of both binomials of every bin --------------------------------
    // A rejection draw is settled here when its candidate falls inside the squeeze region (about six in seven);
    // every other draw is remembered in a bit mask (bit 2 i + j) and queued in one go after the loop: one prefix sum
    // and one LDS atomic per wave instead of one per bin.  The queued draws repeat attempt 0 in full in round 1 of
    // phase C (the words are a function of the counter: nothing has to be kept).
    // Every wave appends its open draws to a segment of its own of the first queue (wave-uniform running count, one
    // ballot per draw: no atomics, no prefix sums); round 1 of phase C reads the segments as one list.  A draw that
    // finds its segment full is remembered (bit 2 i + j) and finished by its thread after the loop.
    unsigned int wcount = 0u;                                                      // entries of this wave's segment (wave-uniform)
    constexpr bool SMALLH = HC > 0 && HC <= 16;                                    // the masks of leftover draws / terms fit 32 bits
    unsigned long long over0 = 0ULL, over1 = 0ULL;                                 // bins 0..31 / 32..63
    unsigned int over_s = 0u;                                                      // SMALLH: all bins
    const int lane = tid & 63;
    unsigned short* const qseg = q0 + (unsigned int)(tid >> 6) * QSEG;
#pragma unroll 1
    for (int kb = 0; kb < nown; kb++) {
        const int i = NC == 1 ? kb : bin_of(kb);
        if (active) {
            int2 nn = slots[i * PW + lp];
            uint32_t w[4];
            rng_block(g, K_BINOM0, (uint32_t)i, 0, w);
            // both draws of the bin as straight code: no selects on j, and the scheduler can overlap the two
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int32_t n = j ? nn.y : nn.x;
                const uint32_t wa = j ? w[2] : w[0], wb = j ? w[3] : w[1];
                int32_t x = 0;
                const BinConsts bc = Dp->bc[j][i];                                 // wave-uniform: scalar loads
                // (the generator's block is complete before the record is waited for: its instructions pass the loads' latency)
                if (j == 0) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
                // all of the record before the first branch: the loads go out together and are waited for once
                asm volatile("" :: "s"(bc.kind), "s"(bc.nthr), "s"(bc.flip), "s"(bc.ppx), "s"(bc.pq), "s"(bc.a0x2), "s"(bc.ppf), "s"(bc.hqp), "s"(bc.c3u), "s"(bc.c3d));
                int st = BTRS_ACCEPT;
#ifdef EPI_PHASE_STAMPS
                if (bc.kind == 2) {
                    const unsigned long long mb = __ballot(n > 0 && (uint32_t)n >= bc.nthr), mi = __ballot(n > 0 && (uint32_t)n < bc.nthr);
                    if (lane == 0) { atomicAdd(&g_phase[16], mb ? 1ULL : 0ULL); atomicAdd(&g_phase[17], mi ? 1ULL : 0ULL); atomicAdd(&g_phase[18], (mb && mi) ? 1ULL : 0ULL); atomicAdd(&g_phase[19], (unsigned long long)__popcll(mi)); atomicAdd(&g_phase[21], 1ULL); }
                }
#endif
                if (bc.kind == 1) x = n > 0 ? n : 0;                               // p >= 1
                else if (bc.kind == 2) {
                    // (without the screens -- the testing build's cross-check -- nothing is settled without the table)
                    if (n > 0 && (uint32_t)n >= bc.nthr) st = SCREEN ? btrs_quick(btrs_consts_quick(n, bc.btrs()), bc.side(), wa, wb, x) : BTRS_OPEN;
                    else if (n > 0) x = binom_inversion(n, bc.lnq, bc.sr, wa);
                }
                const int32_t res = st != BTRS_ACCEPT ? n : (bc.flip != 0 ? n - x : x);   // an unsettled draw keeps n for phase C
                if (j) nn.y = res; else nn.x = res;
                const unsigned long long mask = __ballot(st != BTRS_ACCEPT);
                if (mask) {                                                        // wave-uniform
                    const unsigned int pos = wcount + rank_in(mask);
                    if (st != BTRS_ACCEPT) {
                        // bit 15: attempt 0 is settled (rejected), the draw goes on with attempt 1
                        if (pos < QSEG) qseg[pos] = (unsigned short)(lp | (i << 8) | (j << 14) | ((st == BTRS_REJECT) << 15));
                        else if (SMALLH) over_s |= 1u << (2 * i + j);
                        else if (i < 32) over0 |= 1ULL << (2 * i + j);
                        else over1 |= 1ULL << (2 * (i - 32) + j);
                    }
                    wcount += (unsigned int)__popcll(mask);
                }
            }
            slots[i * PW + lp] = nn;
        }
    }
    if (lane == 0) qw[tid >> 6] = wcount < QSEG ? wcount : QSEG;
    if constexpr (SMALLH) over0 = over_s;
    for (int half = 0; half < (SMALLH ? 1 : 2); half++) {
        unsigned long long om = half ? over1 : over0;
        while (om) {                                                               // segment full: finish here
            const int bit = __ffsll((long long)om) - 1;
            om &= om - 1ULL;
            const int i = (bit >> 1) + 32 * half, j = bit & 1;
            const int32_t n = slot_words[(i * PW + lp) * 2 + j];
            const int32_t k = btrs_from0(n, Dp->bc[j][i].btrs(), Dp->lf, screen, g, j, (uint32_t)i);
            slot_words[(i * PW + lp) * 2 + j] = Dp->flip[j][i] ? n - k : k;
        }
    }

    EPI_STAMP(1);
    // ---- C: binomial rounds ----------------------------------------------------------------
    // Round r runs, in full, attempt r - 1 of the queued draws whose attempt 0 was left open and attempt r of those whose
    // attempt 0 was rejected (bit 15 of the entry).
    // One barrier per round: round r reads the entries of queue (r-1) mod 2, counted in qn[(r-1) mod 3], and pushes
    // its rejected draws to queue r mod 2, counted in qn[r mod 3]; thread 0 clears qn[(r+1) mod 3] for the round after
    // (it was last read in round r-1, before this round's barrier).
    // `spare` is the count the rounds leave at zero and that no thread of the last round reads (a thread that is through
    // must not clear the count a slower one has yet to read): the infection terms' queue starts there.
    int spare = 1;
    {
        int cur = 0;                                                               // (round - 1) mod 3
        const double* __restrict__ lf_t = Dp->lf.t;
        const int32_t lf_n = Dp->lf.n;
#pragma unroll 1
        for (uint32_t round = 1; round <= COOP_BINOM_ROUNDS; round++) {
            __syncthreads();
            const unsigned int qcap_cur = (round & 1u) ? QCAP : QCAP1, qcap_next = (round & 1u) ? QCAP1 : QCAP;
            // round 1: the waves' segments, end to end (up to four waves: three comparisons map a position to its segment)
            unsigned int seg_end[CTPB / 64];
            unsigned int cnt = 0u;
            if (round == 1u) {
#pragma unroll
                for (int w = 0; w < CTPB / 64; w++) { cnt += qw[w]; seg_end[w] = cnt; }
            } else cnt = qn[cur] < qcap_cur ? qn[cur] : qcap_cur;
            const int nxt = cur == 2 ? 0 : cur + 1;
            spare = nxt;
            if (cnt == 0u) {                                                       // uniform
                if (tid == 0) qn[nxt == 2 ? 0 : nxt + 1] = 0u;                     // read a round ago: every thread is past it
                break;
            }
#ifdef EPI_PHASE_STAMPS
            if (tid == 0) { atomicAdd(&g_phase[8], 1ULL); atomicAdd(&g_phase[9], (unsigned long long)cnt); if (round == 1u) atomicAdd(&g_phase[10], (unsigned long long)cnt); if (round == 2u) atomicAdd(&g_phase[11], (unsigned long long)cnt); }
#endif
            if (tid == 0) qn[nxt == 2 ? 0 : nxt + 1] = 0u;
            const unsigned short* q = (round & 1u) ? q0 : q1;
            unsigned short* qnext = (round & 1u) ? q1 : q0;
            auto at_of = [&](unsigned int e) -> unsigned int {
                if (round != 1u) return e;
                unsigned int seg = 0u, first = 0u;
#pragma unroll
                for (int w = 0; w + 1 < CTPB / 64; w++) if (e >= seg_end[w]) { seg = (unsigned int)w + 1u; first = seg_end[w]; }
                return seg * QSEG + (e - first);
            };
#pragma unroll 1
            for (unsigned int e = tid; e < cnt; e += CTPB) {
                const unsigned short entry = q[at_of(e)];
                const int t = entry & 255, i = (entry >> 8) & 63, j = (entry >> 14) & 1;
#ifdef EPI_PHASE_STAMPS
                if (round == 1u && (entry >> 15)) atomicAdd(&g_phase[15], 1ULL);
#endif
                const uint32_t att = round - 1u + (uint32_t)(entry >> 15);
                const int32_t n = slot_words[(i * PW + t) * 2 + j];
                const CoopRec rec = crec[j * H + i];
                const BtrsBin B{rec.ppx, rec.pq, rec.a0x2, rec.lpq};
                const bool flip = rec.flip != 0;                                   // a queued draw is a rejection draw
                const Rng gt = rng_make(A.seed, A.slot0 + (uint64_t)(base + t), A.step);
                // attempt 0 takes its half of the bin's first block, attempt a >= 1 half (a - 1) & 1 of block (a - 1) / 2 of the draw's own stream
                const bool first = att == 0u;
                const uint32_t kind = first ? K_BINOM0 : (j ? K_POS : K_SYMPT);
                const bool hi = first ? j != 0 : ((att - 1u) & 1u) != 0u;
                uint32_t w[4];
                rng_block(gt, kind, (uint32_t)i, first ? 0u : (att - 1u) >> 1, w);
                const uint32_t wa = hi ? w[2] : w[0], wb = hi ? w[3] : w[1];
                const BtrsConsts C = btrs_consts(n, B);
                int32_t k;
                if (n < lf_n) k = btrs_attempt_tab(C, B.lpq, lf_t, screen, wa, wb);
                else k = btrs_attempt_far(C, B.lpq, LfTable{lf_t, lf_n}, screen, wa, wb);
                const bool fail = k < 0;
                // the first rounds hand rejected draws on; after them (and when the queue is full) a draw is finished by the thread that holds it
                const bool pushed = round < COOP_BINOM_ROUNDS && q_push(fail, entry, qnext, &qn[nxt], qcap_next);
                if (fail && !pushed) k = btrs_finish_thread(C, B.lpq, LfTable{lf_t, lf_n}, screen, gt, j ? K_POS : K_SYMPT, (uint32_t)i, att + 1u);
                if (k >= 0) slot_words[(i * PW + t) * 2 + j] = flip ? n - k : k;
            }
            cur = nxt;
        }
    }
    // (phase D's first loads -- the ancestor index again, then the first infection ages -- are requested ahead of the barrier
    // that ends the rounds: it passes while they arrive)
    // The particle's source column and its generator are formed again from the thread index here (one more read of the
    // ancestor index, from the cache): carried over from phase A they would sit in registers through the binomial rounds,
    // which need every one of them, or go through the stack.
    int lpd = lp;
    asm volatile("" : "+v"(lpd));
    const int64_t md = base + lpd;
    const int64_t mcd = FULL ? (md < A.n ? md : A.n - 1) : md;
    const int64_t srcd = active ? (A.anc ? (int64_t)A.anc[mcd] : mcd) : 0;
    const Rng gd = rng_make(A.seed, A.slot0 + (uint64_t)md, A.step);
    auto ld_d = [&](int64_t row) -> int32_t { return in[row * is + srcd]; };
    const int nterm = nown;
    const int i_first = nterm > 0 ? bin_of(0) : 0;
    // the loads of the infection ages run D_AHEAD iterations ahead of their use
    auto bin_of_iter = [&](int k) -> int { const int it = term_of(k); return it == 0 ? H - 1 : it - 1; };
    int32_t Id[D_AHEAD];
#pragma unroll
    for (int k = 0; k < D_AHEAD; k++) Id[k] = (active && k < nterm) ? ld_d(2 + bin_of_iter(k)) : 0;
    // the last round's readers of the first queue's entries are through before phase D pushes to it
    __syncthreads();

    EPI_STAMP(2);
    // ---- D: isolation bookkeeping and the table-and-squeeze part of attempt 0 of the gamma terms ----------------
    // iteration `it` handles bin (it == 0 ? H-1 : it-1), whose survivors seed infection term `it`
    // (the circular shift of simulate.jl:145).  A term is settled here when its normal comes from the inside of a
    // strip and the squeeze accepts it; the others are queued and repeat attempt 0 in full in round 1 of phase E.
    int32_t sum_inf = 0;
    typename std::conditional<SMALLH, unsigned int, unsigned long long>::type g_over = 0;  // terms that found the queue full
    uint32_t z_odd = 0u, w_odd = 0u;   // words of the odd term of the pair block
    int2 sp_next = slots[i_first * PW + lp];                                       // so does the read of the bin's slot
    // (the default horizon's loop by pairs of terms: which term of a pair generates the pair's block is then known in the code)
    constexpr int D_UNROLL = (HC == 14 && NC == 1) ? 2 : 1;
#pragma unroll D_UNROLL
    for (int k = 0; k < nterm; k++) {
        const int it = term_of(k);
        const int i = it == 0 ? H - 1 : it - 1;
        const int i_next = NC == 1 ? it : term_of(k + 1) - 1;                      // bin of the next iteration (never bin H-1)
        bool gfail = false;
        // the term's Gamma parameters (wave-uniform)
        const double k_it = P.k[it], theta_it = P.theta[it];
        if (active) {
            const int2 sp = sp_next;
            if (k + 1 < nterm) sp_next = slots[i_next * PW + lp];                  // bin of the next iteration (another slot than this bin's)
            const int32_t sympt = sp.x, pos = sp.y;
            int32_t Ii = Id[0];
#pragma unroll
            for (int r = 0; r + 1 < D_AHEAD; r++) Id[r] = Id[r + 1];
            if (k + D_AHEAD < nterm) Id[D_AHEAD - 1] = ld_d(2 + bin_of_iter(k + D_AHEAD));
            double pr = div_unscaled((double)sympt, (double)Ii);                   // simulate.jl:104
            if (pr != pr) pr = 0.0;                                                // simulate.jl:105
            const double om = 1.0 - pr;
            R += sympt; Ii -= sympt;                                               // simulate.jl:107-108
            for (int r = 0; r <= D; r++) {                                         // simulate.jl:110-126
                int32_t tv;
                if (r == D) tv = pos;
                else tv = (i >= 1) ? ld_d(tests_row(H, r, i)) : 0;
                const int32_t ts = (int32_t)__builtin_floor(om * (double)tv);
                if (r == 0) { R += ts; Ii -= ts; }
                else if (i + 1 < H) st_out(tests_row(H, r - 1, i + 1), ts);
            }
            if (Ii < 0) flags |= EPI_F_NEG_I;                                      // simulate.jl:117
            if constexpr (OUTS) o_sum += pos;
            if (it == 0) R += Ii;                                                  // simulate.jl:144
            else { st_out(2 + it, Ii); if constexpr (OUTS) sum_inf += Ii; }        // simulate.jl:145
            const double shape = k_it * (double)Ii;                                // init.jl:71-78
            uint32_t wz = z_odd, wu = w_odd;
            if ((it & 1) == 0) {                                                   // an even term generates the block of its pair
                uint32_t w[4];
                rng_block(gd, K_GAMMA, (uint32_t)it >> 1, 0, w);
                wz = w[0]; z_odd = w[1]; wu = w[2]; w_odd = w[3];
            }
            if (gamma_is_zero(shape)) gslots[i * PW + lp] = 0.0;
            else {
                const GammaConsts G = gamma_consts(shape, screen);
                float xf;
                double v;
                const bool inside = zig_fast(wz, ztab[wz & 255u], xf);
                // (a shape below 1 needs the boost's extra uniform: left to the rounds, which are not this loop's straight code --
                // except in the builds for wide states, whose old ages have shapes below 1 in every lane: there the boost of
                // a term the squeeze has settled is taken here, in line, when a good part of the wave needs it)
                const bool fastok = gamma_fast(G, xf, wu, v) && inside;
                bool boost_here = false;
                if constexpr (NC > 1) boost_here = __popcll(__ballot(fastok && G.boost)) >= 16;   // (1, 4, 16 lanes measure the same; 40 is slower)
                if (fastok && !G.boost) gslots[i * PW + lp] = (G.d * v) * theta_it;
                else if (NC > 1 && fastok && boost_here) {
                    uint32_t wb[4];
                    rng_block(gd, K_BOOST, (uint32_t)it, 0, wb);
                    gslots[i * PW + lp] = ((G.d * v) * s_exp(s_log(u01(wb[0])) / G.shape)) * theta_it;   // gamma_finish's expression
                } else {
                    slots[i * PW + lp] = make_int2(Ii, 0); gfail = true;
                }
            }
        }
        const bool gpushed = q_push(gfail, (unsigned short)(lp | (it << 8)), q0, &qn[spare], QCAP);
        if (gfail && !gpushed) g_over |= (decltype(g_over))1 << it;                // queue full: finished after the loop
    }
    while (g_over) {
        const int it = __ffsll((long long)g_over) - 1;
        g_over &= g_over - 1;
        const int i = it == 0 ? H - 1 : it - 1;
        const int32_t Ii = slots[i * PW + lp].x;
        gslots[i * PW + lp] = gamma_from0(P.k[it] * (double)Ii, P.theta[it], &Dp->zig, gd, (uint32_t)it, screen);
    }

    EPI_STAMP(3);
    // the other chunks' shares of the particle's integer sums (order is immaterial), ahead of the barriers of phase E
    if constexpr (NC > 1) {
        if (active && !lead) {
            if (R) atomicAdd(&acc[lp], R);
            if constexpr (OUTS) {
                if (o_sum) atomicAdd(&acc[PW + lp], o_sum);
                if (sum_inf) atomicAdd(&acc[2 * PW + lp], sum_inf);
                if (suminf_pre) atomicAdd(&acc[3 * PW + lp], suminf_pre);
            }
        }
    }

    // ---- E: gamma rounds (round r runs attempt r - 1 in full) ----------------------------------------------------
    {
        int cur = spare;
#pragma unroll 1
        for (uint32_t round = 1; round <= COOP_GAMMA_ROUNDS; round++) {
            __syncthreads();
            const unsigned int qcap_cur = (round & 1u) ? QCAP : QCAP1, qcap_next = (round & 1u) ? QCAP1 : QCAP;
            const unsigned int cnt = qn[cur] < qcap_cur ? qn[cur] : qcap_cur;
            if (cnt == 0u) break;
#ifdef EPI_PHASE_STAMPS
            if (tid == 0) { atomicAdd(&g_phase[12], 1ULL); atomicAdd(&g_phase[13], (unsigned long long)cnt); if (round == 1u) atomicAdd(&g_phase[14], (unsigned long long)cnt); }
#endif
            const int nxt = cur == 2 ? 0 : cur + 1;
            if (tid == 0) qn[nxt == 2 ? 0 : nxt + 1] = 0u;
            const unsigned short* q = (round & 1u) ? q0 : q1;
            unsigned short* qnext = (round & 1u) ? q1 : q0;
            const uint32_t att = round - 1u;
#pragma unroll 1
            for (unsigned int e = tid; e < cnt; e += CTPB) {
                const unsigned short entry = q[e];
                const int t = entry & 255, it = entry >> 8;
                const int i = it == 0 ? H - 1 : it - 1;
                const int32_t Ii = slots[i * PW + t].x;
                const GammaConsts G = gamma_consts(P.k[it] * (double)Ii, screen);
                const Rng gt = rng_make(A.seed, A.slot0 + (uint64_t)(base + t), A.step);
                uint32_t wz, wu;
                if (att == 0u) gamma_first(gt, (uint32_t)it, wz, wu);
                else gamma_retry(gt, (uint32_t)it, att, wz, wu);
                float xf;
                if (!zig_fast(wz, ztab[wz & 255u], xf)) xf = zig_slow(&Dp->zig, gt, (att << 6) | (uint32_t)it, wz);
                double v;
                const bool ok = gamma_attempt(G, xf, wu, v);
                const bool pushed = round < COOP_GAMMA_ROUNDS && q_push(!ok, entry, qnext, &qn[nxt], qcap_next);   // after the cooperative rounds: finished by this thread
                if (ok) gslots[i * PW + t] = gamma_finish(G, v, P.theta[it], gt, (uint32_t)it);
                else if (!pushed) gslots[i * PW + t] = gamma_finish_from(G, P.theta[it], &Dp->zig, gt, (uint32_t)it, att + 1u);
            }
            cur = nxt;
        }
    }
    __syncthreads();                                                               // the terms other threads have drawn for this one

    EPI_STAMP(4);
    // ---- F: infection sum in bin order, new infections, outputs --------------------------------
    if (active && lead) {
        // the addresses of this phase's loads and stores are formed here, not ahead of the sampler phases (where the
        // compiler would otherwise compute them, to carry them through every phase in registers it does not have)
        int lpf = lp;
        asm volatile("" : "+v"(lpf));
        auto st_end = [&](int64_t row, int32_t v) { EPI_STREAM_STORE(&(out + (row * os + base))[lpf], v); };
        S = ld_d(0); R_pre = ld_d(1);
        R += R_pre;
        if constexpr (NC > 1) {
            R += acc[lp];
            if constexpr (OUTS) { o_sum += acc[PW + lp]; sum_inf += acc[2 * PW + lp]; suminf_pre += acc[3 * PW + lp]; }
        }
        double infsum = 0.0;
        if constexpr (HC > 0) {
            // (a constant number of terms: their reads go out together, the sum keeps its order)
            double term[HC];
#pragma unroll
            for (int it = 0; it < HC; it++) term[it] = gslots[(it == 0 ? HC - 1 : it - 1) * PW + lp];
#pragma unroll
            for (int it = 0; it < HC; it++) infsum += term[it];                                  // simulate.jl:38-41
        } else {
#pragma unroll 1
            for (int it = 0; it < H; it++) infsum += gslots[(it == 0 ? H - 1 : it - 1) * PW + lp];   // simulate.jl:38-41
        }
        const int64_t denom = P.N - (int64_t)R;
        int32_t newi = 0;
        if (denom <= 0) flags |= EPI_F_BAD_DENOM;
        else newi = (int32_t)__builtin_floor(div_unscaled((double)S, (double)denom) * infsum);  // simulate.jl:42-44
        st_end(2, newi);                                                                        // simulate.jl:147
        st_end(0, S - newi);                                                                    // simulate.jl:148
        st_end(1, R);
        const bool real = !FULL || md < Ak->n;                                                   // not a surplus thread of a padded launch
        if (!real) flags = 0u;
        if constexpr (OUTS) {
            if (Ak->o_out && real) Ak->o_out[md] = o_sum;
            if (Ak->newinf_out && real) Ak->newinf_out[md] = newi;
            if (Ak->r_out && real) {
                const int64_t prow = Ak->a_div > 1 ? md / Ak->a_div : md;
                const double ap = Ak->a_prev_dev ? Ak->a_prev_dev[prow * Ak->a_row_stride + Ak->a_prev_col] : Ak->a_prev;
                const double af = Ak->a_dev ? Ak->a_dev[prow * Ak->a_row_stride + Ak->a_col] : Ak->a;   // read again rather than kept since phase A
                Ak->r_out[md] = reward_of(P, sum_inf + newi, af, ap);
            }
            if (Ak->hist && real) {
                int32_t* hist = Ak->hist;
                const int64_t hn = Ak->n;
                hist[md] = S; hist[hn + md] = suminf_pre; hist[2 * hn + md] = R_pre; hist[3 * hn + md] = o_sum;
            }
        }
    }
    // (flags are the exception: a wave without any skips the reduction and its six exchanges)
    if (__any(flags != 0u)) {
        unsigned int wf = flags;
        for (int d = 32; d >= 1; d >>= 1) wf |= __shfl_down(wf, d, 64);
        if ((tid & 63) == 0) atomicOr(&Ak->scal->flags, wf);
    }
    if constexpr (WEIGH) {
        // (s_max was written after phase A; the barrier after the infection-term rounds lies in between)
        if (tid == 0) {
            unsigned long long c = s_max[0];
            for (int w = 1; w < CTPB / 64; w++) c = s_max[w] > c ? s_max[w] : c;
            atomicMax(&Ak->scal->lwmax_bits, c);
        }
    }
    EPI_STAMP(5);
}

static size_t coop_lds_bytes(int H, int NC = 1)
{
    return (size_t)H * (CTPB / NC) * 8 + (QCAP + QCAP1) * sizeof(unsigned short) + (4 + CTPB / 64) * sizeof(unsigned int) + (CTPB / 64) * sizeof(unsigned long long) +
           ZIG_N * sizeof(ZigEntry) + 2 * (size_t)H * sizeof(CoopRec) + (NC > 1 ? 4 * (CTPB / NC) * sizeof(int) : 0);
}

// launch the Philox-driven step with the selected implementation
template <bool WEIGH>
static hipError_t launch_step(epi_ctx* c, const StepArgs& A)
{
    // The cooperative kernel needs 8 H bytes of LDS per particle: with one thread per particle four workgroups share a CU
    // up to H = 14 and three up to H = 21 (wider states would leave two workgroups or one: H = 28 1.08 vs 1.33e9
    // particle-steps/s for the per-thread kernel, H = 56 0.26 vs 0.79e9), which is why wider states split a particle over
    // several threads (below).
    // The cooperative kernel takes the square root of the observation variance without the power-of-two scaling that
    // arguments below 2^-767 need; the argument is a positive integer times p (1 - p), so a parameter set with a test
    // probability in (0, 2^-700) always runs the per-thread kernel.
    if (WEIGH && (A.o_out || A.r_out || A.newinf_out || A.hist)) return hipErrorInvalidValue;   // the weighing builds have no per-particle outputs
    if (!WEIGH && A.n == 1 && g_coop == 1) {                    // a lone particle: one lane per age bin
        k_step_one<<<dim3(1), dim3(64), 0, c->stream>>>(A, c->d_params);
        return hipGetLastError();
    }
    // States of 18 ages and more run the cooperative kernel with every particle split over two threads (up to H = 28) or four,
    // which brings the footprint per workgroup back to a narrow state's (ensembles, 10^9 particle-steps/s, against the
    // per-thread kernel: H = 21 1.85 vs 1.55, H = 28 1.59 vs 1.26, H = 42 1.10 vs 0.89, H = 56 0.95 vs 0.75, H = 64 0.76 vs 0.71).
    // What made the difference for the widest states is where the boost of a small shape is taken: old ages have Gamma
    // shapes below 1 in every lane, and as long as all of those terms went through the retry queue (as they do for narrow
    // states, where they are rare) the gamma round drowned in them and four threads per particle lost to one.
    if (!c->tiny_p && g_coop == 1 && c->P.H >= 18 && c->P.H <= 28 && A.screen) {
        const size_t ldw = coop_lds_bytes(c->P.H, 2);
        const dim3 grid((unsigned)ceil_div(A.n, CTPB / 2));
        if (c->P.D > 0) k_step_coop<WEIGH, true, 0, true, false, 2><<<grid, dim3(CTPB), ldw, c->stream>>>(A, c->d_params);
        else k_step_coop<WEIGH, false, 0, true, false, 2><<<grid, dim3(CTPB), ldw, c->stream>>>(A, c->d_params);
        return hipGetLastError();
    }
    if (!c->tiny_p && g_coop == 1 && c->P.H > 28 && A.screen) {
        const size_t ldw = coop_lds_bytes(c->P.H, 4);
        const dim3 grid((unsigned)ceil_div(A.n, CTPB / 4));
        if (c->P.D > 0) k_step_coop<WEIGH, true, 0, true, false, 4><<<grid, dim3(CTPB), ldw, c->stream>>>(A, c->d_params);
        else k_step_coop<WEIGH, false, 0, true, false, 4><<<grid, dim3(CTPB), ldw, c->stream>>>(A, c->d_params);
        return hipGetLastError();
    }
    const size_t lds = coop_lds_bytes(c->P.H);
    if (!c->tiny_p && (g_coop == 2 || (g_coop == 1 && lds <= 53 * 1024))) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_coop<WEIGH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_coop<WEIGH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef EPI_TESTING
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_coop<WEIGH, true, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
            if (e != hipSuccess) return e;
        }
        const dim3 grid((unsigned)ceil_div(A.n, CTPB));
#ifdef EPI_TESTING
        if (!A.screen) k_step_coop<WEIGH, true, 0, false><<<grid, dim3(CTPB), lds, c->stream>>>(A, c->d_params);
        else
#endif
        if (c->P.D > 0) k_step_coop<WEIGH, true><<<grid, dim3(CTPB), lds, c->stream>>>(A, c->d_params);
        else if (c->P.H == 14) {
            // the workgroups that the population fills run the build without the per-thread bounds check; the last,
            // partial one (if any) is a launch of its own
            const int64_t full = A.padded ? ceil_div(A.n, CTPB) : A.n / CTPB;
            if (full > 0) k_step_coop<WEIGH, false, 14, true, true><<<dim3((unsigned)full), dim3(CTPB), lds, c->stream>>>(A, c->d_params);
            if (!A.padded && A.n % CTPB != 0) {
                StepArgs T = A;
                T.block0 = full;
                k_step_coop<WEIGH, false, 14><<<dim3(1), dim3(CTPB), lds, c->stream>>>(T, c->d_params);
            }
        }
        else k_step_coop<WEIGH, false><<<grid, dim3(CTPB), lds, c->stream>>>(A, c->d_params);
    } else {
        const dim3 grid((unsigned)ceil_div(A.n, TPB));
#ifdef EPI_TESTING
        if (!A.screen) k_step<WEIGH, MODE_PHILOX, true, false><<<grid, dim3(TPB), 0, c->stream>>>(A, c->d_params);
        else
#endif
        if (c->P.D > 0) k_step<WEIGH, MODE_PHILOX, true, true><<<grid, dim3(TPB), 0, c->stream>>>(A, c->d_params);
        else k_step<WEIGH, MODE_PHILOX, false, true><<<grid, dim3(TPB), 0, c->stream>>>(A, c->d_params);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fixed-point weights and their prefix sums
//   wq_i = floor(exp(lw_i - lwmax) * 2^F)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long quantize(double lw, double lwmax, double scale)
{
    return (unsigned long long)__builtin_floor(s_exp(lw - lwmax) * scale);
}

// The code of the maximum log weight (lwmax_bits) is zero between updates: the belief's scalars are created zero and
// k_wscan, the one kernel of an update that runs as a single workgroup after every reader of the code, clears it again.
__device__ __forceinline__ double lwmax_decode(unsigned long long bits) { return bits ? order_decode(bits) : -__builtin_inf(); }
// the device-side hand-over of the phased update: local maximum out, global maximum in, local totals out
__global__ void k_scal_export_lwmax(const DevScalars* s, double* out) { out[0] = lwmax_decode(s->lwmax_bits); }
// (global maximum in and local totals out ride on k_wsum and k_wscan)

// LOCAL: the population lives on this device alone, so the local maximum log weight (still coded, as the step kernel's
// atomics left it) is the global one; otherwise the global maximum is read from lwmax_src (device memory, where the
// caller's exchange left it) or, when that is null, from scal->lwmax, where the caller has installed it.
template <bool LOCAL>
__global__ __launch_bounds__(TPB) void k_wsum(int64_t n, const double* __restrict__ lw, const DevScalars* __restrict__ scal, int F,
                                               unsigned long long* __restrict__ blocksum, double* __restrict__ blocksq,
                                               const double* __restrict__ lwmax_src = nullptr)
{
    __shared__ unsigned long long s_sum[TPB / 64];
    __shared__ double s_sq[TPB / 64];
    const double lwmax = LOCAL ? lwmax_decode(scal->lwmax_bits) : (lwmax_src ? *lwmax_src : scal->lwmax);
    const double scale = u2d((uint64_t)(F + 1023) << 52);
    // (the tile's weights in any order: consecutive lanes read consecutive log weights)
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x;
    unsigned long long sum = 0ULL;
    double sq = 0.0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        int64_t i = base + (int64_t)j * TPB;
        if (i < n) {
            const double wd = __builtin_floor(s_exp(lw[i] - lwmax) * scale);     // quantize(), before its conversion
            sum += (unsigned long long)wd;
            sq += wd * wd;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) { sum += shfl_down_u64(sum, d); sq += shfl_down_f64(sq, d); }
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_sq[threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ULL; double q = 0.0;
        for (int w = 0; w < TPB / 64; w++) { t += s_sum[w]; q += s_sq[w]; }
        blocksum[blockIdx.x] = t; blocksq[blockIdx.x] = q;
    }
}

// one workgroup: exclusive scan of blocksum in place, totals to scal; also the update's housekeeping: the list of long
// child ranges is emptied for the scatter kernel that follows, and the code of the maximum log weight is cleared for
// the next update (local: the maximum is decoded first and the local weight total is the global one)
// lwmax_src / totals_out (both or neither): the device-memory hand-over of a sharded update -- the global maximum is
// installed in scal for the kernels that follow, and {local weight total, sum of squares, step flags, global maximum} go
// out as four doubles; the flags have then been delivered and are cleared.
__global__ __launch_bounds__(1024) void k_wscan(int64_t nb, unsigned long long* __restrict__ blocksum, const double* __restrict__ blocksq,
                                                DevScalars* __restrict__ scal, unsigned long long* __restrict__ big_count, int local,
                                                const double* __restrict__ lwmax_src = nullptr, double* __restrict__ totals_out = nullptr)
{
    __shared__ unsigned long long s_w[16];
    __shared__ double s_q[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    // Every wave takes a run of consecutive tiles.  First sweep: the run's total (independent loads, one reduction); the
    // waves' totals are exchanged once; second sweep: the prefixes, 256 tiles at a time -- four per lane, scanned in the
    // lane, then one scan of the lanes' totals -- with the running total in a register (no barrier inside the loop).
    constexpr int PER_LANE = 4, PER_TRIP = 64 * PER_LANE;
    const int64_t per = (((nb + 15) / 16) + PER_TRIP - 1) / PER_TRIP * PER_TRIP, first = (int64_t)wv * per, last = first + per < nb ? first + per : nb;
    unsigned long long mine = 0ULL;
    double sq = 0.0;
#pragma unroll 8
    for (int64_t i = first + lane; i < last; i += 64) { mine += blocksum[i]; sq += blocksq[i]; }
    for (int d = 32; d >= 1; d >>= 1) { mine += shfl_down_u64(mine, d); sq += shfl_down_f64(sq, d); }
    if (lane == 0) { s_w[wv] = mine; s_q[wv] = sq; }
    __syncthreads();
    unsigned long long carry = 0ULL, total = 0ULL;
    for (int w = 0; w < 16; w++) { if (w < wv) carry += s_w[w]; total += s_w[w]; }
    for (int64_t b = first; b < last; b += PER_TRIP) {
        const int64_t i0 = b + (int64_t)lane * PER_LANE;
        unsigned long long v[PER_LANE];
        unsigned long long run = 0ULL;
#pragma unroll
        for (int k = 0; k < PER_LANE; k++) {
            const unsigned long long x = i0 + k < last ? blocksum[i0 + k] : 0ULL;
            v[k] = run;                                                            // exclusive within the lane
            run += x;
        }
        unsigned long long incl = run;
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = shfl_up_u64(incl, d);
            if (lane >= d) incl += o;
        }
        const unsigned long long lane_before = carry + incl - run;
#pragma unroll
        for (int k = 0; k < PER_LANE; k++) if (i0 + k < last) blocksum[i0 + k] = lane_before + v[k];
        carry += shfl_u64(incl, 63);
    }
    if (t == 0) {
        double q = 0.0;
        for (int w = 0; w < 16; w++) q += s_q[w];
        scal->w_local = total;
        scal->sumsq = q;
        if (local) { scal->lwmax = lwmax_decode(scal->lwmax_bits); scal->w_off = 0ULL; scal->w_total = total; }
        scal->lwmax_bits = 0ULL;
        *big_count = 0ULL;
        if (totals_out) {
            const double g = *lwmax_src;
            scal->lwmax = g;
            totals_out[0] = (double)total; totals_out[1] = q; totals_out[2] = (double)scal->flags; totals_out[3] = g;
            scal->flags = 0u;
        }
    }
}

__global__ __launch_bounds__(TPB) void k_wcum(int64_t n, double* __restrict__ lw_then_cum, const DevScalars* __restrict__ scal, int F,
                                               const unsigned long long* __restrict__ blockoff)
{
    __shared__ unsigned long long s_w[TPB / 64];
    const double lwmax = scal->lwmax;
    const double scale = u2d((uint64_t)(F + 1023) << 52);
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS];
    unsigned long long run = 0ULL;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        int64_t i = base + j;
        unsigned long long w = (i < n) ? quantize(lw_then_cum[i], lwmax, scale) : 0ULL;
        run += w;
        v[j] = run;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long incl = run;
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = shfl_up_u64(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    unsigned long long off = blockoff[blockIdx.x] + (incl - run);
    for (int w = 0; w < wv; w++) off += s_w[w];
    unsigned long long* cum = reinterpret_cast<unsigned long long*>(lw_then_cum);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        int64_t i = base + j;
        if (i < n) cum[i] = off + v[j];
    }
}

// ---------------------------------------------------------------------------
// systematic resampling
//   T_m = fl(fl(m + u) * fl(W / n_total)); ancestor = #{ i : C_i <= floor(T_m) }
// ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t find_ancestor(const unsigned long long* __restrict__ cum, int64_t n_local,
                                                 unsigned long long W, unsigned long long off, int64_t n_total, double u, int64_t m_global,
                                                 int64_t slot0)
{
    if (W == 0ULL) {                        // every weight zero: keep the particle in place
        int64_t i = m_global - slot0;
        return i < 0 ? 0 : (i >= n_local ? n_local - 1 : i);
    }
    const double stepw = (double)W / (double)n_total;
    const double Tm = ((double)m_global + u) * stepw;
    unsigned long long tq = (unsigned long long)__builtin_floor(Tm);
    tq = tq > off ? tq - off : 0ULL;
    int64_t lo = 0, hi = n_local;           // first index with cum > tq
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (cum[mid] > tq) hi = mid; else lo = mid + 1;
    }
    return lo > n_local - 1 ? n_local - 1 : lo;
}

// The same assignment from the ancestors' side, without a search: particle i is the ancestor of the
// contiguous slots m with C_{i-1} <= floor(T_m) < C_i, i.e. of [first_slot(C_{i-1}), first_slot(C_i)),
// first_slot(c) = the first slot whose position reaches c.  Every owned slot in [slot_lo, slot_hi) is
// written exactly once.  A particle writes a few children itself, a wave shares the work of a particle
// with up to BIG_RANGE children, and longer ranges (collapsed weights) are listed for k_scatter_big.
struct BigRange { long long a, b; long long idx; };
constexpr int SMALL_RANGE = 8, BIG_RANGE = 8192;

// a non-negative integer below 2^53 as a double (exact; three instructions instead of the general 64-bit conversion)
__device__ __forceinline__ double exact_f64(unsigned long long x)
{
    return __builtin_fma((double)(uint32_t)(x >> 32), 4294967296.0, (double)(uint32_t)x);
}

// (inv_stepw ~ 1 / stepw only seeds the guess; the two loops make the result exact whatever its rounding.)
// Cumulative weights and slot numbers are below 2^53, so the comparisons of floor(...) with c are made on doubles:
// the same decisions as on the integers, without converting back.
__device__ __forceinline__ int64_t first_slot(unsigned long long c, double stepw, double inv_stepw, double u, int64_t n_total)
{
    const double cd = exact_f64(c);
    const double guess = __builtin_ceil(cd * inv_stepw - u);
    int64_t m = guess < 0.0 ? 0 : (guess > (double)n_total ? n_total : (int64_t)guess);
    // the guess is nearly always right: each loop evaluates its condition once and leaves (not to be unrolled: an
    // unrolled body evaluates several candidates at once, all but one of them for nothing)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    while (m > 0 && __builtin_floor((exact_f64((unsigned long long)(m - 1)) + u) * stepw) >= cd) m--;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    while (m < n_total && __builtin_floor((exact_f64((unsigned long long)m) + u) * stepw) < cd) m++;
    return m;
}

// The same for a population of fewer than 2^31 slots (every population one device holds): slot numbers in 32-bit
// registers, converted by single instructions.
__device__ __forceinline__ int32_t first_slot32(unsigned long long c, double stepw, double inv_stepw, double u, int32_t n_total)
{
    const double cd = exact_f64(c);
    const double guess = __builtin_ceil(cd * inv_stepw - u);
    int32_t m = guess < 0.0 ? 0 : (guess > (double)n_total ? n_total : (int32_t)guess);
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    while (m > 0 && __builtin_floor(((double)(m - 1) + u) * stepw) >= cd) m--;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    while (m < n_total && __builtin_floor(((double)m + u) * stepw) < cd) m++;
    return m;
}

// W, off: the total weight of all shards and the cumulative weight of the shards before this one (the host knows
// both when it launches this kernel); stepw = fl(W / n_total) and inv_stepw = fl(n_total / W) come from the host too
// (the same two correctly rounded divisions), so that no thread repeats them.
__global__ __launch_bounds__(TPB) void k_scatter(int64_t n_local, const unsigned long long* __restrict__ cum, unsigned long long W, unsigned long long off,
                                                  double stepw, double inv_stepw,
                                                  int64_t n_total, double u, int64_t slot_lo, int64_t slot_hi, int64_t slot0,
                                                  int32_t* __restrict__ anc, unsigned long long* __restrict__ big_count, BigRange* __restrict__ big)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    const bool act = i < n_local;
    if (W == 0ULL) {                        // every weight zero: keep the particle in place
        if (act && slot0 + i >= slot_lo && slot0 + i < slot_hi) anc[i] = (int32_t)i;
        return;
    }
    const int lane = threadIdx.x & 63;
    int64_t b = slot_hi;
    if (act && i < n_local - 1) b = first_slot(off + cum[i], stepw, inv_stepw, u, n_total);
    int64_t a = (int64_t)shfl_up_u64((unsigned long long)b, 1);
    if (lane == 0) a = (act && i > 0) ? first_slot(off + cum[i - 1], stepw, inv_stepw, u, n_total) : slot_lo;
    if (i == 0) a = slot_lo;
    a = a < slot_lo ? slot_lo : a;
    b = b > slot_hi ? slot_hi : b;
    const int64_t len = act ? b - a : 0;
    int32_t* dst = anc + (a - slot0);
    if (len > 0 && len <= SMALL_RANGE)
        for (int64_t k = 0; k < len; k++) dst[k] = (int32_t)i;
    unsigned long long mask = __ballot(len > SMALL_RANGE && len <= BIG_RANGE);
    while (mask) {
        const int l = __ffsll((long long)mask) - 1;
        mask &= mask - 1ULL;
        const int64_t a_l = (int64_t)__shfl((unsigned int)(a >> 32), l, 64) << 32 | (int64_t)__shfl((unsigned int)a, l, 64);
        const int len_l = __shfl((int)len, l, 64);
        const int idx_l = __shfl((int)i, l, 64);
        for (int k = lane; k < len_l; k += 64) anc[a_l - slot0 + k] = idx_l;
    }
    if (len > BIG_RANGE) {
        const unsigned long long pos = atomicAdd(big_count, 1ULL);
        big[pos] = BigRange{(long long)a, (long long)b, (long long)i};
    }
}

// k_wcum and k_scatter in one pass for a shard: the cumulative weights are written (k_export searches them) and go
// straight from the registers into the ancestor scatter of the owned slots [slot_lo, slot_hi).  W, off, stepw,
// inv_stepw as for k_scatter.
__global__ __launch_bounds__(TPB) void k_wcum_scatter_shard(int64_t n, double* __restrict__ lw_then_cum, const DevScalars* __restrict__ scal, int F,
                                                             const unsigned long long* __restrict__ blockoff, unsigned long long W, unsigned long long off,
                                                             double stepw, double inv_stepw, int64_t n_total, double u, int64_t slot_lo, int64_t slot_hi,
                                                             int64_t slot0, int32_t* __restrict__ anc, unsigned long long* __restrict__ big_count,
                                                             BigRange* __restrict__ big)
{
    __shared__ unsigned long long s_w[TPB / 64];
    const double lwmax = scal->lwmax;
    const double scale = u2d((uint64_t)(F + 1023) << 52);
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS];
    unsigned long long run = 0ULL;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        int64_t i = base + j;
        unsigned long long w = (i < n) ? quantize(lw_then_cum[i], lwmax, scale) : 0ULL;
        run += w;
        v[j] = run;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long incl = run;
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = shfl_up_u64(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    unsigned long long before = blockoff[blockIdx.x] + (incl - run);       // local cumulative weight before this thread's particles
    for (int w = 0; w < wv; w++) before += s_w[w];
    unsigned long long* cum = reinterpret_cast<unsigned long long*>(lw_then_cum);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) cum[base + j] = before + v[j];
    if (W == 0ULL) {                                                        // every weight zero: keep the particles in place
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const int64_t i = base + j;
            if (i < n && slot0 + i >= slot_lo && slot0 + i < slot_hi) anc[i] = (int32_t)i;
        }
        return;
    }
    int64_t a = base == 0 ? slot_lo : first_slot(off + before, stepw, inv_stepw, u, n_total);
    a = a < slot_lo ? slot_lo : (a > slot_hi ? slot_hi : a);
#pragma unroll 1
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const int64_t i = base + j;
        int64_t b = a;
        if (i < n) {
            b = i == n - 1 ? slot_hi : first_slot(off + before + v[j], stepw, inv_stepw, u, n_total);
            b = b < a ? a : (b > slot_hi ? slot_hi : b);
        }
        const int64_t len = b - a;
        if (len > 0 && len <= SMALL_RANGE)
            for (int64_t k = 0; k < len; k++) anc[a - slot0 + k] = (int32_t)i;
        unsigned long long mask = __ballot(len > SMALL_RANGE && len <= BIG_RANGE);
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1ULL;
            const int64_t a_l = (int64_t)__shfl((unsigned int)(a >> 32), l, 64) << 32 | (int64_t)__shfl((unsigned int)a, l, 64);
            const int len_l = __shfl((int)len, l, 64);
            const int idx_l = __shfl((int)i, l, 64);
            for (int k = lane; k < len_l; k += 64) anc[a_l - slot0 + k] = idx_l;
        }
        if (len > BIG_RANGE) {
            const unsigned long long pos = atomicAdd(big_count, 1ULL);
            big[pos] = BigRange{(long long)a, (long long)b, (long long)i};
        }
        a = b;
    }
}

// k_wcum and k_scatter in one pass for a population held by one device (slot m = particle m before resampling,
// no shard offset): the cumulative weights stay in registers and are never written.
__global__ __launch_bounds__(TPB) void k_wcum_scatter(int64_t n, const double* __restrict__ lw, const DevScalars* __restrict__ scal, int F,
                                                       const unsigned long long* __restrict__ blockoff, double u,
                                                       int32_t* __restrict__ anc, unsigned long long* __restrict__ big_count, BigRange* __restrict__ big)
{
    __shared__ unsigned long long s_w[TPB / 64];
    __shared__ unsigned long long s_tile[SCAN_TILE + SCAN_TILE / SCAN_ITEMS];  // a pad word per thread's run: its lanes read distinct banks
    const double lwmax = scal->lwmax;
    const double scale = u2d((uint64_t)(F + 1023) << 52);
    const int64_t tile = (int64_t)blockIdx.x * SCAN_TILE;
    const int64_t base = tile + (int64_t)threadIdx.x * SCAN_ITEMS;
    // the tile's log weights are read with consecutive lanes on consecutive particles and turned into weights; a thread
    // then takes its run of consecutive weights from LDS
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const int e = j * TPB + (int)threadIdx.x;
        const int64_t i = tile + e;
        s_tile[e + e / SCAN_ITEMS] = (i < n) ? quantize(lw[i], lwmax, scale) : 0ULL;
    }
    __syncthreads();
    unsigned long long v[SCAN_ITEMS];
    unsigned long long run = 0ULL;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        run += s_tile[(int)threadIdx.x * (SCAN_ITEMS + 1) + j];
        v[j] = run;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long incl = run;
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long o = shfl_up_u64(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    unsigned long long off = blockoff[blockIdx.x] + (incl - run);
    for (int w = 0; w < wv; w++) off += s_w[w];
    const unsigned long long W = scal->w_local;             // the total: k_wscan has run
    if (W == 0ULL) {                                        // every weight zero: keep the particles in place
        for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) anc[base + j] = (int32_t)(base + j);
        return;
    }
    const double stepw = (double)W / (double)n;
    const double inv_stepw = (double)n / (double)W;
    typedef int32_t Slot;                                   // n < 2^31: the ancestor index is 32 bits wide
    const Slot ns = (Slot)n;
    Slot a = base == 0 ? 0 : first_slot32(off, stepw, inv_stepw, u, ns);
    a = a > ns ? ns : a;
#pragma unroll 1
    for (int j = 0; j < SCAN_ITEMS; j++) {
        const int64_t i = base + j;
        Slot b = a;
        if (i < n) b = i == n - 1 ? ns : first_slot32(off + v[j], stepw, inv_stepw, u, ns);
        const Slot len = b - a;
        if (len > 0 && len <= SMALL_RANGE)
            for (Slot k = 0; k < len; k++) anc[(int64_t)a + k] = (int32_t)i;
        unsigned long long mask = __ballot(len > SMALL_RANGE && len <= BIG_RANGE);
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1ULL;
            const int64_t a_l = (int64_t)__shfl((unsigned int)((int64_t)a >> 32), l, 64) << 32 | (int64_t)__shfl((unsigned int)a, l, 64);
            const int len_l = __shfl((int)len, l, 64);
            const int idx_l = __shfl((int)i, l, 64);
            for (int k = lane; k < len_l; k += 64) anc[a_l + k] = idx_l;
        }
        if (len > BIG_RANGE) {
            const unsigned long long pos = atomicAdd(big_count, 1ULL);
            big[pos] = BigRange{(long long)a, (long long)b, (long long)i};
        }
        a = b;
    }
}

// the listed long ranges, each spread over the whole grid
__global__ __launch_bounds__(TPB) void k_scatter_big(const unsigned long long* __restrict__ big_count, const BigRange* __restrict__ big,
                                                      int64_t slot0, int32_t* __restrict__ anc)
{
    const unsigned long long cnt = *big_count;
    for (unsigned long long e = 0; e < cnt; e++) {
        const BigRange r = big[e];
        for (long long m = r.a + (long long)blockIdx.x * TPB + threadIdx.x; m < r.b; m += (long long)gridDim.x * TPB)
            anc[m - slot0] = (int32_t)r.idx;
    }
}

// resampled particles of slots [slot_begin, slot_begin + count) packed as dst[row * count + j]
__global__ __launch_bounds__(TPB) void k_export(int64_t count, int64_t slot_begin, int64_t slot0, const unsigned long long* __restrict__ cum,
                                                 int64_t n_local, unsigned long long W, unsigned long long off, int64_t n_total, double u,
                                                 const int32_t* __restrict__ src, int64_t stride, int rows, int32_t* __restrict__ dst)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= count) return;
    const int64_t a = find_ancestor(cum, n_local, W, off, n_total, u, slot_begin + j, slot0);
    for (int r = 0; r < rows; r++) dst[(int64_t)r * count + j] = src[(int64_t)r * stride + a];
}

__global__ __launch_bounds__(TPB) void k_import(int64_t count, int64_t local_begin, int64_t halo_begin, const int32_t* __restrict__ src,
                                                 int32_t* __restrict__ dst, int64_t stride, int rows, int32_t* __restrict__ anc)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= count) return;
    for (int r = 0; r < rows; r++) dst[(int64_t)r * stride + halo_begin + j] = src[(int64_t)r * count + j];
    anc[local_begin + j] = (int32_t)(halo_begin + j);
}

// dst column m = src column anc[m]
__global__ __launch_bounds__(TPB) void k_gather(int64_t n, const int32_t* __restrict__ src, int64_t stride, const int32_t* __restrict__ anc,
                                                 int32_t* __restrict__ dst, int rows)
{
    const int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (m >= n) return;
    const int64_t a = anc ? (int64_t)anc[m] : m;
    for (int r = 0; r < rows; r++) dst[(int64_t)r * stride + m] = src[(int64_t)r * stride + a];
}

// trajectory m starts from particle (m mod n_start) of a belief (through its ancestor index)
__global__ __launch_bounds__(TPB) void k_replicate(int64_t n, const int32_t* __restrict__ src, int64_t src_stride, const int32_t* __restrict__ anc,
                                                    int64_t n_start, int32_t* __restrict__ dst, int64_t dst_stride, int rows)
{
    const int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (m >= n) return;
    const int64_t p = m % n_start;
    const int64_t s = anc ? (int64_t)anc[p] : p;
    for (int r = 0; r < rows; r++) dst[(int64_t)r * dst_stride + m] = src[(int64_t)r * src_stride + s];
}

// acc[m] = (first ? 0 : acc[m]) + scale * r[m]   (two roundings: product, then sum)
__global__ __launch_bounds__(TPB) void k_score_acc(int64_t n, double* __restrict__ acc, const double* __restrict__ r, double scale, int first)
{
    const int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (m >= n) return;
    const double t = scale * r[m];
    acc[m] = first ? t : acc[m] + t;
}

// score[s] = (acc[s R] + acc[s R + 1] + ... in that order) / R
__global__ __launch_bounds__(TPB) void k_score_mean(int64_t n_seq, int R, const double* __restrict__ acc, double* __restrict__ score)
{
    const int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (s >= n_seq) return;
    double t = 0.0;
    for (int r = 0; r < R; r++) t += acc[s * R + r];
    score[s] = t / (double)R;
}

// sums[row] += sum over particles of state[row][anc[m]]; blockIdx.y = row
__global__ __launch_bounds__(TPB) void k_sums(int64_t n, const int32_t* __restrict__ src, int64_t stride, const int32_t* __restrict__ anc,
                                               long long* __restrict__ sums)
{
    __shared__ long long s_p[TPB / 64];
    const int row = blockIdx.y;
    long long acc = 0;
    for (int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x; m < n; m += (int64_t)gridDim.x * TPB) {
        const int64_t a = anc ? (int64_t)anc[m] : m;
        acc += (long long)src[(int64_t)row * stride + a];
    }
    unsigned long long v = (unsigned long long)acc;
    for (int d = 32; d >= 1; d >>= 1) v += shfl_down_u64(v, d);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = (long long)v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < TPB / 64; w++) t += s_p[w];
        atomicAdd(reinterpret_cast<unsigned long long*>(&sums[row]), (unsigned long long)t);
    }
}

__global__ __launch_bounds__(TPB) void k_lfact_fill(double* __restrict__ t, int n)
{
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k < n) t[k] = lfact((double)k);
}

// ---------------------------------------------------------------------------
// bare samplers (element j draws from the streams of slot j)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_sample_binomial(int64_t count, const int32_t* __restrict__ n, const double* __restrict__ p,
                                                          uint64_t seed, uint32_t step, uint32_t bin, int which, int32_t* __restrict__ out, bool screen, LfTable lf)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= count) return;
    Rng g = rng_make(seed, (uint64_t)j, step);
    uint32_t w[4];
    rng_block(g, K_BINOM0, bin, 0, w);
    out[j] = which ? binomial_draw(n[j], p[j], w[2], w[3], g, K_POS, bin, lf, screen) : binomial_draw(n[j], p[j], w[0], w[1], g, K_SYMPT, bin, lf, screen);
}

__global__ __launch_bounds__(TPB) void k_sample_gamma(int64_t count, const double* __restrict__ shape, const double* __restrict__ scale,
                                                       uint64_t seed, uint32_t step, uint32_t bin, double* __restrict__ out, bool screen,
                                                       const DevParams* __restrict__ Dp)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= count) return;
    Rng g = rng_make(seed, (uint64_t)j, step);
    uint32_t wz, wu;
    gamma_first(g, bin, wz, wu);
    out[j] = gamma_draw(shape[j], scale[j], wz, wu, &Dp->zig, g, bin, screen);
}

// ---------------------------------------------------------------------------
// initial particles
// ---------------------------------------------------------------------------
__device__ __forceinline__ int32_t round_half_even(double x) { return (int32_t)__builtin_rint(x); }

// State(Normal(mu, sigma), params): I_i = round(truncated-normal draw), S = N - sum I, R = 0
__global__ __launch_bounds__(TPB) void k_init_normal(int64_t n, int32_t* __restrict__ out, int64_t stride, int H, int rows, long long N,
                                                      double mu, double sigma, uint64_t seed, uint64_t slot0, unsigned int* __restrict__ over)
{
    const int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (m >= n) return;
    Rng g = rng_make(seed, slot0 + (uint64_t)m, 0);
    long long tot = 0;
    for (int i = 0; i < H; i++) {
        double x = 0.0;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (uint32_t att = 0; att < 512u; att++) {
            if ((att & 1u) == 0u) rng_block(g, K_INIT, (uint32_t)i, att >> 1, w);
            float z0, z1;
            if ((att & 1u) == 0u) box_muller(w[0], w[1], z0, z1); else box_muller(w[2], w[3], z0, z1);
            double cand = mu + sigma * (double)z0;
            if (cand >= 0.0) { x = cand; break; }
        }
        int32_t v = round_half_even(x);
        out[(int64_t)(2 + i) * stride + m] = v;
        tot += v;
    }
    out[m] = (int32_t)(N - tot);
    out[stride + m] = 0;
    for (int r = 2 + H; r < rows; r++) out[(int64_t)r * stride + m] = 0;
    if (tot > N) atomicOr(over, 1u);                         // typedef.jl:164 asserts sum(I0) <= N
}

// State(params): Bayesian-bootstrap split of N into (S, inf, R) and of inf into I[1..H]
__global__ __launch_bounds__(TPB) void k_init_simplex(int64_t n, int32_t* __restrict__ out, int64_t stride, int H, int rows, long long N,
                                                       uint64_t seed, uint64_t slot0)
{
    const int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (m >= n) return;
    Rng g = rng_make(seed, slot0 + (uint64_t)m, 0);
    uint32_t w[4];
    rng_block(g, K_INIT, 0x100u, 0, w);
    double e0 = -s_log(u01(w[0])), e1 = -s_log(u01(w[1])), e2 = -s_log(u01(w[2]));
    double es = (e0 + e1) + e2;
    double Nd = (double)N;
    long long s0 = (long long)__builtin_rint(Nd * (e0 / es));
    long long inf = (long long)__builtin_rint(Nd * (e1 / es));
    long long r0 = (long long)__builtin_rint(Nd * (e2 / es));
    // first pass: the sum of the H exponentials (in index order); second pass regenerates them
    double esum = 0.0;
    for (int i = 0; i < H; i++) {
        if ((i & 3) == 0) rng_block(g, K_INIT, 0x200u + (uint32_t)(i >> 2), 0, w);
        double e = -s_log(u01(w[i & 3]));
        esum = (i == 0) ? e : esum + e;
    }
    long long tot = 0;
    int32_t lastv = 0;
    for (int i = 0; i < H; i++) {
        if ((i & 3) == 0) rng_block(g, K_INIT, 0x200u + (uint32_t)(i >> 2), 0, w);
        double e = -s_log(u01(w[i & 3]));
        int32_t v = (int32_t)__builtin_rint((double)inf * (e / esum));
        tot += v; lastv = v;
        out[(int64_t)(2 + i) * stride + m] = v;
    }
    long long left = N - (s0 + tot + r0);
    if (s0 >= tot && s0 >= r0) s0 += left;
    else if (tot >= r0) out[(int64_t)(2 + H - 1) * stride + m] = (int32_t)(lastv + left);
    else r0 += left;
    out[m] = (int32_t)s0;
    out[stride + m] = (int32_t)r0;
    for (int r = 2 + H; r < rows; r++) out[(int64_t)r * stride + m] = 0;
}

// ---------------------------------------------------------------------------
// order statistics of the infected total sum(I) over the particles of a belief
//
// Exact selection by narrowing windows, entirely on the device.  A window [lo, lo + 2^W) holds the wanted value of a
// quantile; the first one, common to all quantiles, is [0, 2^bits(N)) for the default state shape (an infected total
// is at most the population) and the smallest power-of-two span from the minimum that holds every value otherwise.
// Each pass histograms the values of the window into 2048 bins of 2^max(W - 11, 0), finds per quantile the bin in
// which the wanted rank falls and makes that bin the next window; W = 0 ends it (at most three passes for 31-bit
// values, two for the default population).  For the default shape the first pass rides on the kernel that computes the
// totals (and the column sums).  A pass's workgroups add their LDS histograms to the global one; a single workgroup
// launched after them narrows the windows and clears the histogram, so the buffers are zero between calls.
// ---------------------------------------------------------------------------
constexpr int QSEL_MAX = 4;            // quantiles per call
constexpr int QSEL_BINS = 2048;
constexpr int QSEL_LOG_BINS = 11;
constexpr int QSEL_MAX_PASSES = 3;     // ceil(31 / 11)
constexpr int QTPB = 1024;             // threads per workgroup of the histogram passes

struct QselState {
    int32_t lo[QSEL_MAX];              // start of the quantile's window
    long long rank[QSEL_MAX];          // 1-based rank among the values inside the window
    unsigned int done;                 // ticket: workgroups of the running kernel that have delivered
    int nq;
    int W;                             // log2 of the window size; 0 = finished, lo[] are the answers
    int shared;                        // all quantiles still share one window (first pass)
};
struct QselRanks { long long rank[QSEL_MAX]; int nq; };

// Every workgroup calls this after its global writes; true (in all its threads) in the workgroup that arrives last.
// Atomics on one address cost about 70 ns each when they queue up (measured: the histogram kernels slowed down by that
// much per additional workgroup), so kernels that end in same-address atomics run with few, large workgroups.
__device__ __forceinline__ bool last_workgroup(unsigned int* ticket)
{
    __shared__ unsigned int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
    __syncthreads();
    return s_last != 0u;
}

// start of a selection: called by thread 0 of the last workgroup of the kernel that produced the totals
__device__ __forceinline__ void qsel_begin(QselState* __restrict__ st, const QselRanks& R, int32_t vmin, int32_t vmax)
{
    const unsigned int span = (unsigned int)(vmax - vmin);             // values are non-negative 31-bit numbers
    int W = 0;
    while (W < 31 && (span >> W) != 0u) W++;                           // smallest W with span < 2^W
    for (int j = 0; j < QSEL_MAX; j++) { st->lo[j] = vmin; st->rank[j] = j < R.nq ? R.rank[j] : 1; }
    st->nq = R.nq; st->W = W; st->shared = 1;
}

// One histogram increment per run of equal bins: neighbouring particles of a belief are alike (after resampling,
// copies of one ancestor), so counting a run in a register and adding it once replaces most of the LDS atomics.
struct DigitRun {
    unsigned int digit = 0u, count = 0u;
    __device__ __forceinline__ void add(unsigned int d, unsigned int* s_h)
    {
        if (d != digit && count) { atomicAdd(&s_h[digit], count); count = 0u; }
        digit = d; count++;
    }
    __device__ __forceinline__ void flush(unsigned int* s_h) { if (count) atomicAdd(&s_h[digit], count); count = 0u; }
};

// One workgroup of THREADS threads: every quantile's window becomes the bin of the pass's histogram that holds its rank
// (the state's lo, rank, W are the pass's); clears the histogram.  s_part: THREADS / 64 words of LDS.
template <int THREADS>
__device__ __forceinline__ void qsel_narrow(QselState* __restrict__ st, unsigned long long* __restrict__ hist, unsigned long long* s_part)
{
    constexpr int per = QSEL_BINS / THREADS;                            // bins per thread
    const int W = st->W, nq = st->nq, shared = st->shared, nh = shared ? 1 : nq;
    const int shift = W > QSEL_LOG_BINS ? W - QSEL_LOG_BINS : 0;
    int32_t lo[QSEL_MAX];
    long long rank[QSEL_MAX];
    for (int j = 0; j < QSEL_MAX; j++) { lo[j] = st->lo[j]; rank[j] = st->rank[j]; }
    __syncthreads();                                                    // every thread has the pass's state before one of them replaces it
    for (int j = 0; j < nq; j++) {
        const unsigned long long* h = hist + (shared ? 0 : (size_t)j * QSEL_BINS);
        const long long want = rank[j];
        unsigned long long c[per], tot = 0ULL;
        for (int k = 0; k < per; k++) { c[k] = h[threadIdx.x * per + k]; tot += c[k]; }
        // inclusive prefix sum of the threads' totals: waves by shuffles, then the wave totals through LDS
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned long long incl = tot;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = shfl_up_u64(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        unsigned long long before = incl - tot;
        for (int w = 0; w < wv; w++) before += s_part[w];
        __syncthreads();
        if ((long long)before < want && want <= (long long)(before + tot)) {          // exactly one thread
            unsigned long long acc = before;
            int k = 0;
            for (; k < per - 1; k++) { if ((long long)(acc + c[k]) >= want) break; acc += c[k]; }
            st->lo[j] = lo[shared ? 0 : j] + (int32_t)((unsigned int)(threadIdx.x * per + k) << shift);
            st->rank[j] = want - (long long)acc;
        }
        // fewer values inside the window than the rank asks for (a total outside [0, 2^bits(N)): a state that breaks the
        // population invariant): the selection is marked unfinished, which the host reports
        if (threadIdx.x == THREADS - 1 && want > (long long)(before + tot)) st->rank[j] = -1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nh * QSEL_BINS; i += THREADS) hist[i] = 0ULL;
    if (threadIdx.x == 0) {
        bool short_of = false;
        for (int j = 0; j < nq; j++) short_of |= st->rank[j] < 0;
        st->W = short_of ? 31 : shift; st->shared = 0;
    }
}

// One pass: the histograms of the values inside the windows.  A pass that finds the selection finished returns at once.
__global__ __launch_bounds__(QTPB) void k_qsel_pass(int64_t n, const int32_t* __restrict__ v, const QselState* __restrict__ st,
                                                     unsigned long long* __restrict__ hist)
{
    __shared__ unsigned int s_h[QSEL_MAX * QSEL_BINS];
    const int W = st->W;
    if (W == 0) return;
    const int nq = st->nq, shared = st->shared, nh = shared ? 1 : nq;
    const int shift = W > QSEL_LOG_BINS ? W - QSEL_LOG_BINS : 0;
    int32_t lo[QSEL_MAX];
    for (int j = 0; j < QSEL_MAX; j++) lo[j] = st->lo[j];
    for (int i = threadIdx.x; i < nh * QSEL_BINS; i += QTPB) s_h[i] = 0u;
    __syncthreads();
    DigitRun run[QSEL_MAX];
    // eight values per trip, loaded together
    constexpr int U = 8;
    const int64_t sweep = (int64_t)gridDim.x * QTPB * U;
    for (int64_t base = (int64_t)blockIdx.x * QTPB * U + threadIdx.x; base < n; base += sweep) {
        int32_t x[U];
        bool have[U];
#pragma unroll
        for (int k = 0; k < U; k++) { have[k] = base + (int64_t)k * QTPB < n; x[k] = have[k] ? v[base + (int64_t)k * QTPB] : 0; }
#pragma unroll
        for (int k = 0; k < U; k++) {
#pragma unroll
            for (int j = 0; j < QSEL_MAX; j++) {
                if (j >= nh || !have[k]) continue;
                const unsigned int off = (unsigned int)(x[k] - lo[j]);  // below the window: wraps to a huge number
                if (x[k] >= lo[j] && (off >> W) == 0u) run[j].add(off >> shift, s_h + j * QSEL_BINS);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < QSEL_MAX; j++) run[j].flush(s_h + j * QSEL_BINS);
    __syncthreads();
    for (int i = threadIdx.x; i < nh * QSEL_BINS; i += QTPB)
        if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}

// after a pass (a kernel of its own rather than a ticket among the pass's workgroups: same-address atomics queue up at
// about 70 ns each, which kept the passes to a few dozen workgroups)
__global__ __launch_bounds__(QTPB) void k_qsel_narrow(QselState* __restrict__ st, unsigned long long* __restrict__ hist)
{
    __shared__ unsigned long long s_part[QTPB / 64];
    if (st->W == 0) return;
    qsel_narrow<QTPB>(st, hist, s_part);
}

// Infected total of every particle (through the ancestor index), their minimum and maximum, and the start of a
// selection over them.  part: one (min, max) pair per workgroup.
__global__ __launch_bounds__(TPB) void k_suminf(int64_t n, const int32_t* __restrict__ src, int64_t stride, const int32_t* __restrict__ anc,
                                                 int H, int32_t* __restrict__ out, int32_t* __restrict__ part, QselRanks R, QselState* __restrict__ st)
{
    __shared__ int32_t s_mm[2][TPB / 64];
    int32_t vmin = 0x7fffffff, vmax = 0;
    for (int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x; m < n; m += (int64_t)gridDim.x * TPB) {
        const int64_t a = anc ? (int64_t)anc[m] : m;
        int32_t t = 0;
        for (int i = 0; i < H; i++) t += src[(int64_t)(2 + i) * stride + a];
        out[m] = t;
        vmin = t < vmin ? t : vmin; vmax = t > vmax ? t : vmax;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const int32_t a = __shfl_down(vmin, d, 64), b = __shfl_down(vmax, d, 64);
        vmin = a < vmin ? a : vmin; vmax = b > vmax ? b : vmax;
    }
    if ((threadIdx.x & 63) == 0) { s_mm[0][threadIdx.x >> 6] = vmin; s_mm[1][threadIdx.x >> 6] = vmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; w++) { vmin = s_mm[0][w] < vmin ? s_mm[0][w] : vmin; vmax = s_mm[1][w] > vmax ? s_mm[1][w] : vmax; }
        part[2 * blockIdx.x] = vmin; part[2 * blockIdx.x + 1] = vmax;
    }
    if (!last_workgroup(&st->done)) return;
    if (threadIdx.x == 0) {
        for (unsigned int g = 0; g < gridDim.x; g++) {
            const int32_t a = __hip_atomic_load(&part[2 * g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int32_t b = __hip_atomic_load(&part[2 * g + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vmin = a < vmin ? a : vmin; vmax = b > vmax ? b : vmax;
        }
        qsel_begin(st, R, vmin, vmax);
        st->done = 0u;
    }
}

// The same for the default state shape (H = 14, no pending tests: 16 rows) together with the column sums of the
// belief mean and the first pass of the selection: every row is read once, through the ancestor index, for all
// results.  part: per workgroup 16 column sums (k_summary_finish adds them up).  W0 > 0: the totals are also counted
// into the histogram of the window [0, 2^W0) (2^W0 above the population, so every total falls inside).
constexpr int SUMMARY_ROWS = 16;
constexpr int SUMMARY_SLOT = SUMMARY_ROWS;
constexpr int SUMMARY_MAX_GROUPS = 512;
__global__ __launch_bounds__(TPB) void k_summary16(int64_t n, const int32_t* __restrict__ src, int64_t stride, const int32_t* __restrict__ anc,
                                                    int32_t* __restrict__ out, long long* __restrict__ part, int W0, unsigned long long* __restrict__ hist)
{
    __shared__ long long s_sum[TPB / 64][SUMMARY_SLOT];
    __shared__ unsigned int s_h[QSEL_BINS];
    long long acc[SUMMARY_ROWS];
#pragma unroll
    for (int r = 0; r < SUMMARY_ROWS; r++) acc[r] = 0;
    const int want_quant = W0 > 0;
    const int shift0 = W0 > QSEL_LOG_BINS ? W0 - QSEL_LOG_BINS : 0;
    if (want_quant) {
        for (int i = threadIdx.x; i < QSEL_BINS; i += TPB) s_h[i] = 0u;
        __syncthreads();
    }
    DigitRun run;
    // two particles per trip: their 34 loads are in flight together
    const int64_t sweep = (int64_t)gridDim.x * TPB * 2;
    for (int64_t base = (int64_t)blockIdx.x * TPB * 2 + threadIdx.x; base < n; base += sweep) {
        const int64_t m0 = base, m1 = base + TPB;
        const bool have1 = m1 < n;
        const int64_t a0 = anc ? (int64_t)anc[m0] : m0;
        const int64_t a1 = have1 ? (anc ? (int64_t)anc[m1] : m1) : a0;
        int32_t v0[SUMMARY_ROWS], v1[SUMMARY_ROWS];
#pragma unroll
        for (int r = 0; r < SUMMARY_ROWS; r++) { v0[r] = src[(int64_t)r * stride + a0]; v1[r] = src[(int64_t)r * stride + a1]; }
        int32_t t0 = 0, t1 = 0;
#pragma unroll
        for (int r = 0; r < SUMMARY_ROWS; r++) {
            acc[r] += v0[r] + (have1 ? (long long)v1[r] : 0LL);
            if (r >= 2) { t0 += v0[r]; t1 += v1[r]; }
        }
        if (want_quant) {
            out[m0] = t0;
            if (t0 >= 0 && ((unsigned int)t0 >> W0) == 0u) run.add((unsigned int)t0 >> shift0, s_h);
            if (have1) {
                out[m1] = t1;
                if (t1 >= 0 && ((unsigned int)t1 >> W0) == 0u) run.add((unsigned int)t1 >> shift0, s_h);
            }
        }
    }
    if (want_quant) {
        run.flush(s_h);
        __syncthreads();
        for (int i = threadIdx.x; i < QSEL_BINS; i += TPB)
            if (s_h[i]) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
    }
    // wave, then workgroup, then this workgroup's slot
#pragma unroll
    for (int r = 0; r < SUMMARY_ROWS; r++) {
        unsigned long long v = (unsigned long long)acc[r];
        for (int d = 32; d >= 1; d >>= 1) v += shfl_down_u64(v, d);
        if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6][r] = (long long)v;
    }
    __syncthreads();
    if (threadIdx.x < SUMMARY_SLOT) {
        long long t = s_sum[0][threadIdx.x];
        for (int w = 1; w < TPB / 64; w++) t += s_sum[w][threadIdx.x];
        part[(int64_t)blockIdx.x * SUMMARY_SLOT + threadIdx.x] = t;
    }
}

// One workgroup: adds up the slots k_summary16's workgroups left (thread t takes slots t, t + 256, ..., then the 256
// partial results are combined in thread order) and adds the column sums to `sums`; W0 > 0: starts the selection with the
// window [0, 2^W0) and narrows it by the histogram k_summary16 has filled.  A kernel of its own rather than a ticket in
// k_summary16: with hundreds of workgroups the ticket atomics alone took 30 us.
__global__ __launch_bounds__(TPB) void k_summary_finish(int groups, const long long* __restrict__ part, long long* __restrict__ sums,
                                                         QselRanks R, QselState* __restrict__ st, int W0, unsigned long long* __restrict__ hist)
{
    __shared__ long long s_all[TPB][SUMMARY_SLOT];
    __shared__ unsigned long long s_part[TPB / 64];
    long long t[SUMMARY_SLOT];
    for (int r = 0; r < SUMMARY_ROWS; r++) t[r] = 0;
    for (int g = threadIdx.x; g < groups; g += TPB) {
#pragma unroll
        for (int r = 0; r < SUMMARY_SLOT; r++) t[r] += part[(int64_t)g * SUMMARY_SLOT + r];
    }
    for (int r = 0; r < SUMMARY_SLOT; r++) s_all[threadIdx.x][r] = t[r];
    __syncthreads();
    // 256 -> 8 -> 1 per column, in index order
    constexpr int CHUNKS = 8, CHUNK = TPB / CHUNKS;
    static_assert(CHUNKS * SUMMARY_SLOT <= TPB, "one thread per (chunk, column)");
    const bool stage1 = threadIdx.x < CHUNKS * SUMMARY_SLOT;
    long long u = 0;
    if (stage1) {
        const int r = threadIdx.x % SUMMARY_SLOT, chunk = threadIdx.x / SUMMARY_SLOT;
        u = s_all[chunk * CHUNK][r];
        for (int k = 1; k < CHUNK; k++) u += s_all[chunk * CHUNK + k][r];
    }
    __syncthreads();
    if (stage1) s_all[threadIdx.x / SUMMARY_SLOT][threadIdx.x % SUMMARY_SLOT] = u;
    __syncthreads();
    if (threadIdx.x < SUMMARY_SLOT) {
        const int r = threadIdx.x;
        u = s_all[0][r];
        for (int k = 1; k < CHUNKS; k++) u += s_all[k][r];
        sums[r] += u;
    }
    if (W0 > 0) {
        if (threadIdx.x == 0) {
            for (int j = 0; j < QSEL_MAX; j++) { st->lo[j] = 0; st->rank[j] = j < R.nq ? R.rank[j] : 1; }
            st->nq = R.nq; st->W = W0; st->shared = 1;
        }
        __syncthreads();
        qsel_narrow<TPB>(st, hist, s_part);
    }
}

// ---------------------------------------------------------------------------
// ODE-model loss of an ensemble kept on the device
//   hist: int32 [T][4][n] (S, sum I, R, sum pos before / of each day), one thread per trajectory.
//   The compartment model (src/ODEFit.jl:1-37) starts from the trajectory's day-1 proportions and is
//   integrated with classical Runge-Kutta, `sub` steps per day; the squared error against the
//   trajectory's proportions is taken at every day (src/ODEFit.jl:281-340).
// ---------------------------------------------------------------------------
enum { ODE_SIR = 0, ODE_SIR_CTRL = 1, ODE_SEIR = 2, ODE_SEIR_CTRL = 3 };

struct OdeRates { double alpha, beta, gamma; };       // effective rates of one trajectory

// u = (S, E, I, R); SIR models keep E = 0 and gamma unused
__host__ __device__ __forceinline__ void ode_rhs(int seir, const OdeRates& k, const double u[4], double du[4])
{
    const double inf = k.beta * u[2] * u[0];
    du[0] = -inf;
    if (seir) {
        du[1] = inf - k.gamma * u[1];
        du[2] = k.gamma * u[1] - k.alpha * u[2];
    } else {
        du[1] = 0.0;
        du[2] = inf - k.alpha * u[2];
    }
    du[3] = k.alpha * u[2];
}

__host__ __device__ __forceinline__ void ode_rk4_step(int seir, const OdeRates& k, double u[4], double h)
{
    double k1[4], k2[4], k3[4], k4[4], t[4];
    ode_rhs(seir, k, u, k1);
    for (int c = 0; c < 4; c++) t[c] = u[c] + (0.5 * h) * k1[c];
    ode_rhs(seir, k, t, k2);
    for (int c = 0; c < 4; c++) t[c] = u[c] + (0.5 * h) * k2[c];
    ode_rhs(seir, k, t, k3);
    for (int c = 0; c < 4; c++) t[c] = u[c] + h * k3[c];
    ode_rhs(seir, k, t, k4);
    for (int c = 0; c < 4; c++) u[c] = u[c] + (h / 6.0) * (((k1[c] + 2.0 * k2[c]) + 2.0 * k3[c]) + k4[c]);
}

__global__ __launch_bounds__(TPB) void k_ode_loss(int64_t n, int T, const int32_t* __restrict__ hist, const double* __restrict__ act,
                                                   int kind, double p0, double p1, double p2, double p3, double p4, int sub, double Npop,
                                                   double* __restrict__ per_traj, double* __restrict__ partial)
{
    __shared__ double s_p[TPB / 64];
    const int64_t m = (int64_t)blockIdx.x * TPB + threadIdx.x;
    double loss = 0.0;
    if (m < n) {
        const int seir = kind >= ODE_SEIR;
        const double tp = (kind == ODE_SIR_CTRL || kind == ODE_SEIR_CTRL) ? act[m] : 0.0;
        OdeRates k;
        k.alpha = p0; k.beta = p1; k.gamma = 0.0;
        if (kind == ODE_SIR_CTRL) k.alpha = p0 + p2 * tp;                         // alpha + delta T
        if (kind == ODE_SEIR) k.gamma = p2;
        if (kind == ODE_SEIR_CTRL) { k.gamma = p2 + p4 * tp; k.alpha = p0 + p3 * tp; }   // gamma + eps T, alpha + delta T
        double u[4];
        u[0] = (double)hist[m] / Npop; u[1] = 0.0; u[2] = (double)hist[n + m] / Npop; u[3] = (double)hist[2 * n + m] / Npop;
        const double h = 1.0 / (double)sub;
        for (int t = 0; t < T; t++) {
            if (t > 0) for (int j = 0; j < sub; j++) ode_rk4_step(seir, k, u, h);
            const int32_t* row = hist + (int64_t)t * 4 * n;
            const double rs = (double)row[m] / Npop, ri = (double)row[n + m] / Npop, rr = (double)row[2 * n + m] / Npop;
            const double e0 = u[0] - rs, e1 = (u[1] + u[2]) - ri, e2 = u[3] - rr;
            loss += (e0 * e0 + e1 * e1) + e2 * e2;
        }
        if (per_traj) per_traj[m] = loss;
    }
    for (int d = 32; d >= 1; d >>= 1) loss += shfl_down_f64(loss, d);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((s_p[0] + s_p[1]) + s_p[2]) + s_p[3];
}

// sum of the per-workgroup partial sums in a fixed order (one workgroup)
__global__ __launch_bounds__(1024) void k_sum_partials(int64_t count, const double* __restrict__ partial, double* __restrict__ out)
{
    __shared__ double s_p[16];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < count; i += 1024) t += partial[i];
    for (int d = 32; d >= 1; d >>= 1) t += shfl_down_f64(t, d);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int w = 0; w < 16; w++) r += s_p[w];
        out[0] = r;
    }
}

struct DevBuf {          // device scratch freed on scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

struct Arena {           // device scratch of one call: carved from the context's small buffer while it lasts, else allocated
    epi_ctx* c;
    size_t used = 0;
    std::vector<void*> owned;
    explicit Arena(epi_ctx* ctx) : c(ctx) {}
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;
    ~Arena() { for (void* p : owned) (void)hipFree(p); }
    void* get(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        if (used + bytes <= SMALL_BYTES) { void* p = c->d_small + used; used += bytes; return p; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        owned.push_back(p);
        return p;
    }
};

static int static_scratch_address(unsigned char** out);   // device address of epi_pf_update_noalloc's scratch array (below)

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int epi_state_rows(int H, int D) { return state_rows(H, D); }

int epi_ctx_create(int device, epi_ctx** out)
{
    return guarded("epi_ctx_create", [&]() -> int {
        if (!out) return fail(EPI_EINVAL, "epi_ctx_create: out is NULL");
        int ndev = 0;
        HIPCHECK(hipGetDeviceCount(&ndev));
        if (device < 0 || device >= ndev) return fail(EPI_EINVAL, "epi_ctx_create: device %d not in [0, %d)", device, ndev);
        HIPCHECK(hipSetDevice(device));
        epi_ctx* c = new epi_ctx();
        c->device = device;
        const int rc = [c]() -> int {
            HIPCHECK(hipStreamCreate(&c->own_stream));
            c->stream = c->own_stream;
            HIPCHECK(hipMalloc((void**)&c->d_params, sizeof(DevParams)));
            {                                   // the strips of the normal generator do not depend on the parameters
                DevParams* dp = new DevParams();
                memset(dp, 0, sizeof *dp);
                zig_build(dp->zig);
                const hipError_t e = hipMemcpy(c->d_params, dp, sizeof *dp, hipMemcpyHostToDevice);
                delete dp;
                HIPCHECK(e);
            }
            HIPCHECK(hipMalloc((void**)&c->d_lf, sizeof(double) * (size_t)LF_TABLE_N));
            k_lfact_fill<<<dim3(LF_TABLE_N / TPB), dim3(TPB), 0, c->stream>>>(c->d_lf, LF_TABLE_N);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipStreamSynchronize(c->stream));
            HIPCHECK(hipMalloc((void**)&c->d_sums, sizeof(long long) * (2 + EPI_MAX_H + (EPI_MAX_D + 1) * EPI_MAX_H)));
            HIPCHECK(hipHostMalloc((void**)&c->h_pin, PIN_BYTES, hipHostMallocDefault));
            HIPCHECK(hipMalloc((void**)&c->d_small, SMALL_BYTES));
            HIPCHECK(hipMalloc((void**)&c->d_qsel, QSEL_HIST_BYTES + 256));
            HIPCHECK(hipMemset(c->d_qsel, 0, QSEL_HIST_BYTES + 256));
            HIPCHECK(hipEventCreate(&c->ev0));
            HIPCHECK(hipEventCreate(&c->ev1));
            return static_scratch_address(&c->d_static);
        }();
        if (rc != EPI_OK) {                 // release what was created before the failure (the message is already set)
            (void)epi_ctx_destroy(c);
            return rc;
        }
        *out = c;
        return EPI_OK;
    });
}

int epi_ctx_destroy(epi_ctx* c)
{
    if (!c) return EPI_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& pr : c->timed) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    c->timed.clear();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->d_sums) (void)hipFree(c->d_sums);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->d_lf) (void)hipFree(c->d_lf);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->d_qsel) (void)hipFree(c->d_qsel);
    if (c->d_days) (void)hipFree(c->d_days);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return EPI_OK;
}

void* epi_ctx_stream(epi_ctx* c) { return c ? (void*)c->stream : nullptr; }

int epi_ctx_set_stream(epi_ctx* c, void* hip_stream, int adopt)
{
    if (!c) return fail(EPI_EINVAL, "epi_ctx_set_stream: ctx is NULL");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    c->external = adopt != 0;
    c->stream = c->external ? static_cast<hipStream_t>(hip_stream) : c->own_stream;     // a null handle is the default stream
    return EPI_OK;
}

int epi_params_set(epi_ctx* c, const epi_params* p)
{
    return guarded("epi_params_set", [&]() -> int {
        if (!c || !p) return fail(EPI_EINVAL, "epi_params_set: NULL argument");
        if (p->H < 1 || p->H > EPI_MAX_H) return fail(EPI_EINVAL, "epi_params_set: H = %d not in [1, %d]", p->H, EPI_MAX_H);
        if (p->D < 0 || p->D > EPI_MAX_D) return fail(EPI_EINVAL, "epi_params_set: test_delay D = %d not in [0, %d]", p->D, EPI_MAX_D);
        if (p->N < 1 || p->N >= (1LL << 31)) return fail(EPI_ERANGE, "epi_params_set: N = %lld must be in [1, 2^31)", (long long)p->N);
        for (int i = 0; i < p->H; i++) {
            if (!(p->q[i] >= 0.0 && p->q[i] <= 1.0)) return fail(EPI_EINVAL, "epi_params_set: q[%d] = %g not a probability", i, p->q[i]);
            if (!(p->p[i] >= 0.0 && p->p[i] <= 1.0)) return fail(EPI_EINVAL, "epi_params_set: p[%d] = %g not a probability", i, p->p[i]);
            if (!(p->k[i] >= 0.0) || !(p->theta[i] >= 0.0)) return fail(EPI_EINVAL, "epi_params_set: Gamma parameters of bin %d negative", i);
        }
        HIPCHECK(hipSetDevice(c->device));
        HIPCHECK(hipStreamSynchronize(c->stream));
        DevParams dp;
        memset(&dp, 0, sizeof dp);
        dp.P = *p;
        derive_params(dp);
        dp.lf = LfTable{c->d_lf, LF_TABLE_N};
        pack_bin_consts(dp);
        zig_build(dp.zig);
        HIPCHECK(hipMemcpy(c->d_params, &dp, sizeof dp, hipMemcpyHostToDevice));
        c->P = *p;
        c->have_params = true;
        c->tiny_p = false;
        for (int i = 0; i < p->H; i++) c->tiny_p |= p->p[i] > 0.0 && p->p[i] < 0x1p-700;
        return EPI_OK;
    });
}

// ---- beliefs -----------------------------------------------------------------

static int belief_check(const epi_belief* b, const char* who)
{
    if (!b) return fail(EPI_EINVAL, "%s: belief is NULL", who);
    if (!b->ctx->have_params) return fail(EPI_ESTATE, "%s: epi_params_set has not been called", who);
    if (b->rows != state_rows(b->ctx->P.H, b->ctx->P.D))
        return fail(EPI_ESTATE, "%s: belief was created for a different (H, D)", who);
    return EPI_OK;
}

// the phased calls and the summaries name the context next to the belief: it must be the belief's own
static int shard_check(const epi_ctx* c, const epi_belief* b, const char* who)
{
    int rc = belief_check(b, who);
    if (rc) return rc;
    if (!c || b->ctx != c) return fail(EPI_EINVAL, "%s: belief belongs to another context", who);
    return EPI_OK;
}

int epi_belief_create(epi_ctx* c, int64_t n_local, int64_t halo_cap, epi_belief** out)
{
    return guarded("epi_belief_create", [&]() -> int {
        if (!c || !out) return fail(EPI_EINVAL, "epi_belief_create: NULL argument");
        if (!c->have_params) return fail(EPI_ESTATE, "epi_belief_create: epi_params_set has not been called");
        if (n_local < 1 || halo_cap < 0) return fail(EPI_EINVAL, "epi_belief_create: n_local = %lld, halo_cap = %lld", (long long)n_local, (long long)halo_cap);
        if (n_local + halo_cap >= (1LL << 31)) return fail(EPI_ERANGE, "epi_belief_create: n_local + halo_cap must be < 2^31");
        HIPCHECK(hipSetDevice(c->device));
        epi_belief* b = new epi_belief();
        b->ctx = c; b->n = n_local; b->halo_cap = halo_cap;
        b->stride = ceil_div(n_local + halo_cap, CTPB) * CTPB;      // rows start on 1 KB boundaries and hold whole workgroups (StepArgs::padded)
        b->n_total = n_local; b->slot0 = 0;
        b->rows = state_rows(c->P.H, c->P.D);
        b->nblocks = ceil_div(n_local, SCAN_TILE);
        size_t bytes = sizeof(int32_t) * (size_t)b->rows * (size_t)b->stride;
        hipError_t e = hipMalloc((void**)&b->buf[0], bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&b->buf[1], bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&b->anc, sizeof(int32_t) * (size_t)n_local);
        if (e == hipSuccess) e = hipMalloc((void**)&b->lw, sizeof(double) * (size_t)(ceil_div(n_local, CTPB) * CTPB));
        if (e == hipSuccess) e = hipMalloc((void**)&b->blocksum, sizeof(unsigned long long) * (size_t)b->nblocks);
        if (e == hipSuccess) e = hipMalloc((void**)&b->blocksq, sizeof(double) * (size_t)b->nblocks);
        if (e == hipSuccess) e = hipMalloc((void**)&b->big, big_list_bytes(n_local));
        if (e == hipSuccess) e = hipMalloc((void**)&b->scal, 256);
        if (e != hipSuccess) {
            epi_belief_destroy(b);
            return fail(EPI_ENOMEM, "epi_belief_create: hipMalloc failed: %s", hipGetErrorString(e));
        }
        hipError_t z = hipMemsetAsync(b->buf[0], 0, bytes, c->stream);
        if (z == hipSuccess) z = hipMemsetAsync(b->buf[1], 0, bytes, c->stream);
        if (z == hipSuccess) z = hipMemsetAsync(b->scal, 0, 256, c->stream);
        if (z == hipSuccess) z = hipMemsetAsync(b->big, 0, sizeof(unsigned long long), c->stream);
        if (z == hipSuccess) z = hipStreamSynchronize(c->stream);
        if (z != hipSuccess) {
            epi_belief_destroy(b);
            return fail(EPI_EHIP, "epi_belief_create: clearing the buffers failed: %s", hipGetErrorString(z));
        }
        *out = b;
        return EPI_OK;
    });
}

int epi_belief_destroy(epi_belief* b)
{
    if (!b) return EPI_OK;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->buf[0]) (void)hipFree(b->buf[0]);
    if (b->buf[1]) (void)hipFree(b->buf[1]);
    if (b->anc) (void)hipFree(b->anc);
    if (b->lw) (void)hipFree(b->lw);
    if (b->blocksum) (void)hipFree(b->blocksum);
    if (b->big) (void)hipFree(b->big);
    if (b->scal) (void)hipFree(b->scal);
    if (b->blocksq) (void)hipFree(b->blocksq);
    delete b;
    return EPI_OK;
}

int epi_belief_set_shard(epi_belief* b, int64_t n_total, int64_t slot0)
{
    if (!b) return fail(EPI_EINVAL, "epi_belief_set_shard: belief is NULL");
    if (slot0 < 0 || n_total < slot0 + b->n) return fail(EPI_EINVAL, "epi_belief_set_shard: slots [%lld, %lld) not inside [0, %lld)",
                                                         (long long)slot0, (long long)(slot0 + b->n), (long long)n_total);
    b->n_total = n_total; b->slot0 = slot0;
    return EPI_OK;
}

int64_t epi_belief_size(const epi_belief* b) { return b ? b->n : 0; }

// fold a pending ancestor indirection into the stored particles
static int materialize(epi_belief* b)
{
    if (!b->has_anc) return EPI_OK;
    epi_ctx* c = b->ctx;
    k_gather<<<dim3((unsigned)ceil_div(b->n, TPB)), dim3(TPB), 0, c->stream>>>(b->n, b->buf[b->cur], b->stride, b->anc, b->buf[b->cur ^ 1], b->rows);
    HIPCHECK(hipGetLastError());
    b->cur ^= 1;
    b->has_anc = false;
    return EPI_OK;
}

int epi_belief_upload(epi_belief* b, const int64_t* S, const int64_t* I, const int64_t* R, const int64_t* T)
{
    return guarded("epi_belief_upload", [&]() -> int {
        int rc = belief_check(b, "epi_belief_upload");
        if (rc) return rc;
        if (!S || !I || !R) return fail(EPI_EINVAL, "epi_belief_upload: S, I, R must not be NULL");
        epi_ctx* c = b->ctx;
        const int H = c->P.H, D = c->P.D, D1 = D + 1;
        const int64_t n = b->n;
        HIPCHECK(hipSetDevice(c->device));
        std::vector<int32_t> h((size_t)b->rows * (size_t)n);
        const int64_t lim = (1LL << 31) - 1;
        for (int64_t p = 0; p < n; p++) {
            if (S[p] < -lim || S[p] > lim || R[p] < 0 || R[p] > lim) return fail(EPI_ERANGE, "epi_belief_upload: particle %lld: S or R out of int32 range", (long long)p);
            h[(size_t)p] = (int32_t)S[p];
            h[(size_t)n + (size_t)p] = (int32_t)R[p];
            for (int i = 0; i < H; i++) {
                int64_t v = I[p * H + i];
                if (v < 0 || v > lim) return fail(EPI_ERANGE, "epi_belief_upload: particle %lld: I[%d] = %lld out of range", (long long)p, i, (long long)v);
                h[(size_t)(2 + i) * (size_t)n + (size_t)p] = (int32_t)v;
            }
            if (T) {
                const int64_t* Tp = T + p * D1 * H;
                for (int cidx = 0; cidx < H; cidx++)
                    for (int r = 0; r < D1; r++) {
                        int64_t v = Tp[cidx * D1 + r];
                        bool live = (r < D) && (cidx >= 1);
                        if (!live) {
                            if (v != 0) return fail(EPI_EINVAL, "epi_belief_upload: particle %lld: Tests[%d,%d] = %lld but row D+1 and column 1 must be zero",
                                                    (long long)p, r + 1, cidx + 1, (long long)v);
                        } else {
                            if (v < 0 || v > lim) return fail(EPI_ERANGE, "epi_belief_upload: particle %lld: Tests entry out of range", (long long)p);
                            h[(size_t)tests_row(H, r, cidx) * (size_t)n + (size_t)p] = (int32_t)v;
                        }
                    }
            } else {
                for (int r = 2 + H; r < b->rows; r++) h[(size_t)r * (size_t)n + (size_t)p] = 0;
            }
        }
        HIPCHECK(hipStreamSynchronize(c->stream));
        HIPCHECK(hipMemcpy2D(b->buf[b->cur], sizeof(int32_t) * (size_t)b->stride, h.data(), sizeof(int32_t) * (size_t)n, sizeof(int32_t) * (size_t)n, (size_t)b->rows,
                             hipMemcpyHostToDevice));
        b->has_anc = false;
        protocol_reset(b);
        return EPI_OK;
    });
}

int epi_belief_download(epi_belief* b, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    return guarded("epi_belief_download", [&]() -> int {
        int rc = belief_check(b, "epi_belief_download");
        if (rc) return rc;
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_belief_download: a phased update is in progress");
        epi_ctx* c = b->ctx;
        const int H = c->P.H, D = c->P.D, D1 = D + 1;
        const int64_t n = b->n;
        HIPCHECK(hipSetDevice(c->device));
        rc = materialize(b);
        if (rc) return rc;
        HIPCHECK(hipStreamSynchronize(c->stream));
        // all rows in one strided copy; a small population comes back through the page-locked buffer
        const size_t row_bytes = sizeof(int32_t) * (size_t)n;
        std::vector<int32_t> hv;
        const int32_t* h;
        if (row_bytes * (size_t)b->rows <= PIN_BYTES) {
            HIPCHECK(hipMemcpy2DAsync(c->h_pin, row_bytes, b->buf[b->cur], sizeof(int32_t) * (size_t)b->stride, row_bytes, (size_t)b->rows, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            h = reinterpret_cast<const int32_t*>(c->h_pin);
        } else {
            hv.resize((size_t)b->rows * (size_t)n);
            HIPCHECK(hipMemcpy2D(hv.data(), row_bytes, b->buf[b->cur], sizeof(int32_t) * (size_t)b->stride, row_bytes, (size_t)b->rows, hipMemcpyDeviceToHost));
            h = hv.data();
        }
        for (int64_t p = 0; p < n; p++) {
            if (S) S[p] = h[(size_t)p];
            if (R) R[p] = h[(size_t)n + (size_t)p];
            if (I) for (int i = 0; i < H; i++) I[p * H + i] = h[(size_t)(2 + i) * (size_t)n + (size_t)p];
            if (T) {
                int64_t* Tp = T + p * D1 * H;
                for (int j = 0; j < D1 * H; j++) Tp[j] = 0;
                for (int cidx = 1; cidx < H; cidx++)
                    for (int r = 0; r < D; r++) Tp[cidx * D1 + r] = h[(size_t)tests_row(H, r, cidx) * (size_t)n + (size_t)p];
            }
        }
        return EPI_OK;
    });
}

int epi_belief_copy(epi_belief* dst, epi_belief* src)
{
    int rc = belief_check(src, "epi_belief_copy");
    if (rc) return rc;
    if (!dst || dst->n != src->n || dst->rows != src->rows || dst->stride != src->stride || dst->ctx != src->ctx)
        return fail(EPI_EINVAL, "epi_belief_copy: beliefs differ in shape or context");
    if (src->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_belief_copy: a phased update of the source is in progress");
    epi_ctx* c = src->ctx;
    HIPCHECK(hipSetDevice(c->device));
    if (src->has_anc) {
        // the copy resolves the source's pending ancestor indirection on the way (one gather pass, source untouched)
        k_gather<<<dim3((unsigned)ceil_div(src->n, TPB)), dim3(TPB), 0, c->stream>>>(src->n, src->buf[src->cur], src->stride, src->anc,
                                                                                   dst->buf[dst->cur], src->rows);
        HIPCHECK(hipGetLastError());
    } else {
        size_t bytes = sizeof(int32_t) * (size_t)src->rows * (size_t)src->stride;
        HIPCHECK(hipMemcpyAsync(dst->buf[dst->cur], src->buf[src->cur], bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    dst->has_anc = false; dst->n_total = src->n_total; dst->slot0 = src->slot0; protocol_reset(dst);
    return EPI_OK;
}

int epi_belief_init_normal(epi_belief* b, double mu, double sigma, uint64_t seed)
{
    int rc = belief_check(b, "epi_belief_init_normal");
    if (rc) return rc;
    if (!(sigma >= 0.0)) return fail(EPI_EINVAL, "epi_belief_init_normal: sigma = %g", sigma);
    epi_ctx* c = b->ctx;
    HIPCHECK(hipSetDevice(c->device));
    unsigned int* over = reinterpret_cast<unsigned int*>(c->d_small);
    HIPCHECK(hipMemsetAsync(over, 0, sizeof(unsigned int), c->stream));
    k_init_normal<<<dim3((unsigned)ceil_div(b->n, TPB)), dim3(TPB), 0, c->stream>>>(b->n, b->buf[b->cur], b->stride, c->P.H, b->rows, (long long)c->P.N, mu, sigma, seed, (uint64_t)b->slot0, over);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(c->h_pin, over, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    b->has_anc = false; protocol_reset(b);
    if (*reinterpret_cast<const unsigned int*>(c->h_pin))
        return fail(EPI_EINVAL, "epi_belief_init_normal: a particle drew more initial infections than the population N = %lld "
                                "(src/typedef.jl:164 asserts sum(I0) <= N)", (long long)c->P.N);
    return EPI_OK;
}

int epi_belief_init_simplex(epi_belief* b, uint64_t seed)
{
    int rc = belief_check(b, "epi_belief_init_simplex");
    if (rc) return rc;
    epi_ctx* c = b->ctx;
    HIPCHECK(hipSetDevice(c->device));
    k_init_simplex<<<dim3((unsigned)ceil_div(b->n, TPB)), dim3(TPB), 0, c->stream>>>(b->n, b->buf[b->cur], b->stride, c->P.H, b->rows, (long long)c->P.N, seed, (uint64_t)b->slot0);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(c->stream));
    b->has_anc = false; protocol_reset(b);
    return EPI_OK;
}

// ---- the step ------------------------------------------------------------------


// the context's scalars, read back through the page-locked staging buffer; waits for the stream
static int read_scalars(epi_ctx* c, const DevScalars* d, DevScalars* h)
{
    HIPCHECK(hipMemcpyAsync(c->h_pin, d, sizeof(DevScalars), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    memcpy(h, c->h_pin, sizeof(DevScalars));
    return EPI_OK;
}

static int fetch_flags(epi_ctx* c, DevScalars* d, uint32_t* flags, bool reset)
{
    DevScalars h;
    if (int rs = read_scalars(c, d, &h)) return rs;
    if (flags) *flags = h.flags;
    if (reset) HIPCHECK(hipMemsetAsync(&d->flags, 0, sizeof(unsigned int), c->stream));
    return EPI_OK;
}

static void widen(const std::vector<int32_t>& h, int64_t n, int64_t* out) { for (int64_t i = 0; i < n; i++) out[i] = h[(size_t)i]; }

int epi_step_batch(epi_ctx* c, epi_belief* b, const double* a, int a_per_particle, const double* a_prev, int a_prev_per_particle,
                   uint64_t seed, uint32_t step, int64_t* o_out, double* r_out, int64_t* new_inf_out, uint32_t* flags)
{
    return guarded("epi_step_batch", [&]() -> int {
        int rc = belief_check(b, "epi_step_batch");
        if (rc) return rc;
        if (!c || b->ctx != c || !a) return fail(EPI_EINVAL, "epi_step_batch: bad context or NULL action");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_step_batch: a phased update is in progress");
        HIPCHECK(hipSetDevice(c->device));
        const int64_t n = b->n;
        Arena scratch(c);
        const char* nomem = "epi_step_batch: hipMalloc failed";
        StepArgs A;
        memset(&A, 0, sizeof A);
        A.screen = g_screen; A.padded = 1;
        A.n = n; A.in = b->buf[b->cur]; A.in_stride = b->stride; A.anc = b->has_anc ? b->anc : nullptr;
        A.out = b->buf[b->cur ^ 1]; A.out_stride = b->stride;
        A.a = a[0]; A.a_row_stride = 1; A.a_col = 0;
        if (a_per_particle) {
            double* da = static_cast<double*>(scratch.get(sizeof(double) * (size_t)n));
            if (!da) return fail(EPI_ENOMEM, nomem);
            HIPCHECK(hipMemcpyAsync(da, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
            A.a_dev = da;
        }
        A.a_prev = a_prev ? a_prev[0] : 0.0;
        if (a_prev && a_prev_per_particle) {
            double* dap = static_cast<double*>(scratch.get(sizeof(double) * (size_t)n));
            if (!dap) return fail(EPI_ENOMEM, nomem);
            HIPCHECK(hipMemcpyAsync(dap, a_prev, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
            A.a_prev_dev = dap;
        }
        A.seed = seed; A.step = step; A.slot0 = (uint64_t)b->slot0; A.scal = b->scal;
        if (o_out && !(A.o_out = static_cast<int32_t*>(scratch.get(sizeof(int32_t) * (size_t)n)))) return fail(EPI_ENOMEM, nomem);
        if (r_out && !(A.r_out = static_cast<double*>(scratch.get(sizeof(double) * (size_t)n)))) return fail(EPI_ENOMEM, nomem);
        if (new_inf_out && !(A.newinf_out = static_cast<int32_t*>(scratch.get(sizeof(int32_t) * (size_t)n)))) return fail(EPI_ENOMEM, nomem);
        HIPCHECK(launch_step<false>(c, A));
        if ((size_t)n * 16 + 256 <= PIN_BYTES) {
            // small population: outputs and scalars come back through the page-locked buffer with one wait
            unsigned char* hp = c->h_pin;
            int32_t* ho = reinterpret_cast<int32_t*>(hp + 256);
            int32_t* hn = ho + n;
            double* hr = reinterpret_cast<double*>(hp + 256 + 8 * (size_t)n);
            HIPCHECK(hipMemcpyAsync(hp, b->scal, sizeof(DevScalars), hipMemcpyDeviceToHost, c->stream));
            if (o_out) HIPCHECK(hipMemcpyAsync(ho, A.o_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            if (new_inf_out) HIPCHECK(hipMemcpyAsync(hn, A.newinf_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            if (r_out) HIPCHECK(hipMemcpyAsync(hr, A.r_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipMemsetAsync(&b->scal->flags, 0, sizeof(unsigned int), c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            if (flags) *flags = reinterpret_cast<const DevScalars*>(hp)->flags;
            for (int64_t i = 0; i < n; i++) {
                if (o_out) o_out[i] = ho[i];
                if (new_inf_out) new_inf_out[i] = hn[i];
                if (r_out) r_out[i] = hr[i];
            }
        } else {
            std::vector<int32_t> h((size_t)n);
            if (o_out) { HIPCHECK(hipMemcpyAsync(h.data(), A.o_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream)); HIPCHECK(hipStreamSynchronize(c->stream)); widen(h, n, o_out); }
            if (new_inf_out) { HIPCHECK(hipMemcpyAsync(h.data(), A.newinf_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream)); HIPCHECK(hipStreamSynchronize(c->stream)); widen(h, n, new_inf_out); }
            if (r_out) HIPCHECK(hipMemcpyAsync(r_out, A.r_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            rc = fetch_flags(c, b->scal, flags, true);
            if (rc) return rc;
            HIPCHECK(hipStreamSynchronize(c->stream));
        }
        b->cur ^= 1; b->has_anc = false;
        return EPI_OK;
    });
}

int epi_step_injected(epi_ctx* c, epi_belief* b, const double* a, int a_per_particle,
                      const int64_t* sympt, const int64_t* pos, const double* gam,
                      int64_t* new_inf_out, int64_t* pos_out, uint32_t* flags)
{
    return guarded("epi_step_injected", [&]() -> int {
        int rc = belief_check(b, "epi_step_injected");
        if (rc) return rc;
        if (!c || b->ctx != c || !a || !sympt || !pos || !gam) return fail(EPI_EINVAL, "epi_step_injected: NULL argument");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_step_injected: a phased update is in progress");
        HIPCHECK(hipSetDevice(c->device));
        const int64_t n = b->n;
        const int H = c->P.H;
        // particle-major host arrays -> one row per age bin
        std::vector<int32_t> hs((size_t)n * H), hp((size_t)n * H);
        std::vector<double> hg((size_t)n * H);
        const int64_t lim = (1LL << 31) - 1;
        for (int64_t p = 0; p < n; p++)
            for (int i = 0; i < H; i++) {
                int64_t s = sympt[p * H + i], q = pos[p * H + i];
                if (s < 0 || s > lim || q < 0 || q > lim) return fail(EPI_ERANGE, "epi_step_injected: variate out of int32 range at particle %lld", (long long)p);
                hs[(size_t)i * (size_t)n + (size_t)p] = (int32_t)s;
                hp[(size_t)i * (size_t)n + (size_t)p] = (int32_t)q;
                hg[(size_t)i * (size_t)n + (size_t)p] = gam[p * H + i];
            }
        DevBuf ds, dp, dg, da, dni, dpo;
        size_t cnt = (size_t)n * (size_t)H;
        if (ds.alloc(4 * cnt) != hipSuccess || dp.alloc(4 * cnt) != hipSuccess || dg.alloc(8 * cnt) != hipSuccess ||
            dni.alloc(4 * (size_t)n) != hipSuccess || dpo.alloc(4 * cnt) != hipSuccess)
            return fail(EPI_ENOMEM, "epi_step_injected: hipMalloc failed");
        HIPCHECK(hipMemcpyAsync(ds.p, hs.data(), 4 * cnt, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(hipMemcpyAsync(dp.p, hp.data(), 4 * cnt, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(hipMemcpyAsync(dg.p, hg.data(), 8 * cnt, hipMemcpyHostToDevice, c->stream));
        StepArgs A;
        memset(&A, 0, sizeof A);
        A.screen = g_screen; A.padded = 1;
        A.n = n; A.in = b->buf[b->cur]; A.in_stride = b->stride; A.anc = b->has_anc ? b->anc : nullptr;
        A.out = b->buf[b->cur ^ 1]; A.out_stride = b->stride;
        A.a = a[0]; A.a_row_stride = 1; A.a_col = 0;
        if (a_per_particle) {
            if (da.alloc(sizeof(double) * (size_t)n) != hipSuccess) return fail(EPI_ENOMEM, "epi_step_injected: hipMalloc failed");
            HIPCHECK(hipMemcpyAsync(da.p, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
            A.a_dev = da.as<double>();
        }
        A.scal = b->scal; A.slot0 = (uint64_t)b->slot0;
        A.inj_sympt = ds.as<int32_t>(); A.inj_pos = dp.as<int32_t>(); A.inj_gam = dg.as<double>();
        A.newinf_out = dni.as<int32_t>(); A.pos_out = dpo.as<int32_t>();
        k_step<false, MODE_INJECT><<<dim3((unsigned)ceil_div(n, TPB)), dim3(TPB), 0, c->stream>>>(A, c->d_params);
        HIPCHECK(hipGetLastError());
        if (new_inf_out) {
            std::vector<int32_t> h((size_t)n);
            HIPCHECK(hipMemcpyAsync(h.data(), dni.p, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            widen(h, n, new_inf_out);
        }
        if (pos_out) {
            HIPCHECK(hipMemcpyAsync(hp.data(), dpo.p, 4 * cnt, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            for (int64_t p = 0; p < n; p++)
                for (int i = 0; i < H; i++) pos_out[p * H + i] = hp[(size_t)i * (size_t)n + (size_t)p];
        }
        rc = fetch_flags(c, b->scal, flags, true);
        if (rc) return rc;
        HIPCHECK(hipStreamSynchronize(c->stream));
        b->cur ^= 1; b->has_anc = false;
        return EPI_OK;
    });
}

int epi_step_variates(epi_ctx* c, epi_belief* b, double a, uint64_t seed, uint32_t step,
                      int64_t* sympt_out, int64_t* pos_out, double* gam_out, uint32_t* flags)
{
    return guarded("epi_step_variates", [&]() -> int {
        int rc = belief_check(b, "epi_step_variates");
        if (rc) return rc;
        if (!c || b->ctx != c || !sympt_out || !pos_out || !gam_out) return fail(EPI_EINVAL, "epi_step_variates: NULL argument");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_step_variates: a phased update is in progress");
        HIPCHECK(hipSetDevice(c->device));
        const int64_t n = b->n;
        const int H = c->P.H;
        const size_t cnt = (size_t)n * (size_t)H;
        DevBuf ds, dp, dg;
        if (ds.alloc(4 * cnt) != hipSuccess || dp.alloc(4 * cnt) != hipSuccess || dg.alloc(8 * cnt) != hipSuccess)
            return fail(EPI_ENOMEM, "epi_step_variates: hipMalloc failed");
        StepArgs A;
        memset(&A, 0, sizeof A);
        A.screen = g_screen; A.padded = 1;
        A.n = n; A.in = b->buf[b->cur]; A.in_stride = b->stride; A.anc = b->has_anc ? b->anc : nullptr;
        A.out = b->buf[b->cur ^ 1]; A.out_stride = b->stride;
        A.a = a; A.a_row_stride = 1;
        A.seed = seed; A.step = step; A.slot0 = (uint64_t)b->slot0; A.scal = b->scal;
        A.rec_sympt = ds.as<int32_t>(); A.rec_pos = dp.as<int32_t>(); A.rec_gam = dg.as<double>();
        k_step<false, MODE_RECORD><<<dim3((unsigned)ceil_div(n, TPB)), dim3(TPB), 0, c->stream>>>(A, c->d_params);
        HIPCHECK(hipGetLastError());
        std::vector<int32_t> hs(cnt), hp(cnt);
        std::vector<double> hg(cnt);
        HIPCHECK(hipMemcpyAsync(hs.data(), ds.p, 4 * cnt, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipMemcpyAsync(hp.data(), dp.p, 4 * cnt, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipMemcpyAsync(hg.data(), dg.p, 8 * cnt, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        for (int64_t p = 0; p < n; p++)
            for (int i = 0; i < H; i++) {
                sympt_out[p * H + i] = hs[(size_t)i * (size_t)n + (size_t)p];
                pos_out[p * H + i] = hp[(size_t)i * (size_t)n + (size_t)p];
                gam_out[p * H + i] = hg[(size_t)i * (size_t)n + (size_t)p];
            }
        rc = fetch_flags(c, b->scal, flags, true);
        if (rc) return rc;
        HIPCHECK(hipStreamSynchronize(c->stream));
        b->cur ^= 1; b->has_anc = false;
        return EPI_OK;
    });
}

int epi_sample_binomial(epi_ctx* c, int64_t count, const int64_t* n, const double* p, uint64_t seed, uint32_t step,
                        uint32_t bin, int which, int64_t* out)
{
    return guarded("epi_sample_binomial", [&]() -> int {
        if (!c || !n || !p || !out || count < 1 || bin > 0xffffu) return fail(EPI_EINVAL, "epi_sample_binomial: bad argument");
        HIPCHECK(hipSetDevice(c->device));
        std::vector<int32_t> hn((size_t)count);
        for (int64_t j = 0; j < count; j++) {
            if (n[j] >= (1LL << 31) || n[j] <= -(1LL << 31)) return fail(EPI_ERANGE, "epi_sample_binomial: n[%lld] out of int32 range", (long long)j);
            hn[(size_t)j] = (int32_t)n[j];
        }
        DevBuf dn, dp, dout;
        if (dn.alloc(4 * (size_t)count) != hipSuccess || dp.alloc(8 * (size_t)count) != hipSuccess || dout.alloc(4 * (size_t)count) != hipSuccess)
            return fail(EPI_ENOMEM, "epi_sample_binomial: hipMalloc failed");
        HIPCHECK(hipMemcpyAsync(dn.p, hn.data(), 4 * (size_t)count, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(hipMemcpyAsync(dp.p, p, 8 * (size_t)count, hipMemcpyHostToDevice, c->stream));
        k_sample_binomial<<<dim3((unsigned)ceil_div(count, TPB)), dim3(TPB), 0, c->stream>>>(count, dn.as<int32_t>(), dp.as<double>(), seed, step, bin, which, dout.as<int32_t>(), g_screen != 0, LfTable{c->d_lf, LF_TABLE_N});
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(hn.data(), dout.p, 4 * (size_t)count, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        for (int64_t j = 0; j < count; j++) out[j] = hn[(size_t)j];
        return EPI_OK;
    });
}

int epi_sample_gamma(epi_ctx* c, int64_t count, const double* shape, const double* scale, uint64_t seed, uint32_t step,
                     uint32_t bin, double* out)
{
    return guarded("epi_sample_gamma", [&]() -> int {
        if (!c || !shape || !scale || !out || count < 1 || bin > 0xffffu) return fail(EPI_EINVAL, "epi_sample_gamma: bad argument");
        HIPCHECK(hipSetDevice(c->device));
        DevBuf dk, ds, dout;
        if (dk.alloc(8 * (size_t)count) != hipSuccess || ds.alloc(8 * (size_t)count) != hipSuccess || dout.alloc(8 * (size_t)count) != hipSuccess)
            return fail(EPI_ENOMEM, "epi_sample_gamma: hipMalloc failed");
        HIPCHECK(hipMemcpyAsync(dk.p, shape, 8 * (size_t)count, hipMemcpyHostToDevice, c->stream));
        HIPCHECK(hipMemcpyAsync(ds.p, scale, 8 * (size_t)count, hipMemcpyHostToDevice, c->stream));
        k_sample_gamma<<<dim3((unsigned)ceil_div(count, TPB)), dim3(TPB), 0, c->stream>>>(count, dk.as<double>(), ds.as<double>(), seed, step, bin, dout.as<double>(), g_screen != 0, c->d_params);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(out, dout.p, 8 * (size_t)count, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        return EPI_OK;
    });
}

int epi_rollout_batch(epi_ctx* c, epi_belief* b, int T, const double* actions, int act_per_particle, int act_per_day, double a_prev0,
                      uint64_t seed, uint32_t step0, int64_t* hist, double* rew, uint32_t* flags)
{
    return guarded("epi_rollout_batch", [&]() -> int {
        int rc = belief_check(b, "epi_rollout_batch");
        if (rc) return rc;
        if (!c || b->ctx != c || !actions || T < 1) return fail(EPI_EINVAL, "epi_rollout_batch: bad argument");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_rollout_batch: a phased update is in progress");
        HIPCHECK(hipSetDevice(c->device));
        const int64_t n = b->n;
        const int64_t arow = act_per_day ? T : 1;
        const size_t acount = (size_t)(act_per_particle ? n : 1) * (size_t)arow;
        DevBuf dact, dhist, drew;
        if (dact.alloc(sizeof(double) * acount) != hipSuccess) return fail(EPI_ENOMEM, "epi_rollout_batch: hipMalloc failed");
        HIPCHECK(hipMemcpyAsync(dact.p, actions, sizeof(double) * acount, hipMemcpyHostToDevice, c->stream));
        if (hist && dhist.alloc(sizeof(int32_t) * 4 * (size_t)n * (size_t)T) != hipSuccess) return fail(EPI_ENOMEM, "epi_rollout_batch: hipMalloc failed (history)");
        if (rew && drew.alloc(sizeof(double) * (size_t)n * (size_t)T) != hipSuccess) return fail(EPI_ENOMEM, "epi_rollout_batch: hipMalloc failed (rewards)");
        for (int t = 0; t < T; t++) {
            StepArgs A;
            memset(&A, 0, sizeof A);
        A.screen = g_screen; A.padded = 1;
            A.n = n; A.in = b->buf[b->cur]; A.in_stride = b->stride; A.anc = b->has_anc ? b->anc : nullptr;
            A.out = b->buf[b->cur ^ 1]; A.out_stride = b->stride;
            const int64_t tcol = act_per_day ? t : 0;
            const int64_t pcol = act_per_day ? t - 1 : 0;          // column of the previous day's action
            A.a_row_stride = 1;
            if (act_per_particle) { A.a_dev = dact.as<double>(); A.a_row_stride = arow; A.a_col = tcol; }
            else A.a = actions[tcol];
            if (t == 0) A.a_prev = a_prev0;                         // prev_action of the start state (typedef.jl:180,216)
            else if (act_per_particle) { A.a_prev_dev = dact.as<double>(); A.a_prev_col = pcol; }
            else A.a_prev = actions[pcol];
            A.seed = seed; A.step = step0 + (uint32_t)t; A.slot0 = (uint64_t)b->slot0; A.scal = b->scal;
            if (hist) A.hist = dhist.as<int32_t>() + (size_t)t * 4 * (size_t)n;
            if (rew) A.r_out = drew.as<double>() + (size_t)t * (size_t)n;
            HIPCHECK(launch_step<false>(c, A));
            b->cur ^= 1; b->has_anc = false;
        }
        if (hist) {
            std::vector<int32_t> h((size_t)4 * (size_t)n * (size_t)T);
            HIPCHECK(hipMemcpyAsync(h.data(), dhist.p, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            for (int64_t p = 0; p < n; p++)
                for (int t = 0; t < T; t++)
                    for (int k = 0; k < 4; k++) hist[(p * T + t) * 4 + k] = h[((size_t)t * 4 + (size_t)k) * (size_t)n + (size_t)p];
        }
        if (rew) {
            std::vector<double> h((size_t)n * (size_t)T);
            HIPCHECK(hipMemcpyAsync(h.data(), drew.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            for (int64_t p = 0; p < n; p++)
                for (int t = 0; t < T; t++) rew[p * T + t] = h[(size_t)t * (size_t)n + (size_t)p];
        }
        rc = fetch_flags(c, b->scal, flags, true);
        if (rc) return rc;
        HIPCHECK(hipStreamSynchronize(c->stream));
        return EPI_OK;
    });
}

int epi_rollout_score(epi_ctx* c, epi_belief* start, int64_t n_seq, int rollouts_per_seq, int T, const double* actions, double a_prev0,
                      double discount, uint64_t seed, uint32_t step0, double* score, uint32_t* flags)
{
    return guarded("epi_rollout_score", [&]() -> int {
        int rc = belief_check(start, "epi_rollout_score");
        if (rc) return rc;
        if (!c || start->ctx != c || !actions || !score || T < 1 || n_seq < 1 || rollouts_per_seq < 1)
            return fail(EPI_EINVAL, "epi_rollout_score: bad argument");
        if (start->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_rollout_score: a phased update is in progress");
        const int64_t n = n_seq * (int64_t)rollouts_per_seq;
        if (n >= (1LL << 31)) return fail(EPI_ERANGE, "epi_rollout_score: n_seq * rollouts_per_seq must be < 2^31");
        HIPCHECK(hipSetDevice(c->device));
        const int rows = start->rows;
        const int64_t stride = ceil_div(n, CTPB) * CTPB;
        DevBuf dact, dst[2], drew, dacc, dscore;
        const size_t sbytes = sizeof(int32_t) * (size_t)rows * (size_t)stride;
        if (dact.alloc(sizeof(double) * (size_t)n_seq * (size_t)T) != hipSuccess || dst[0].alloc(sbytes) != hipSuccess || dst[1].alloc(sbytes) != hipSuccess ||
            drew.alloc(sizeof(double) * (size_t)n) != hipSuccess || dacc.alloc(sizeof(double) * (size_t)n) != hipSuccess ||
            dscore.alloc(sizeof(double) * (size_t)n_seq) != hipSuccess)
            return fail(EPI_ENOMEM, "epi_rollout_score: hipMalloc failed");
        HIPCHECK(hipMemcpyAsync(dact.p, actions, sizeof(double) * (size_t)n_seq * (size_t)T, hipMemcpyHostToDevice, c->stream));
        const dim3 grid((unsigned)ceil_div(n, TPB)), blk(TPB);
        k_replicate<<<grid, blk, 0, c->stream>>>(n, start->buf[start->cur], start->stride, start->has_anc ? start->anc : nullptr, start->n,
                                                 dst[0].as<int32_t>(), stride, rows);
        HIPCHECK(hipGetLastError());
        double scale = 1.0;
        for (int t = 0; t < T; t++) {
            StepArgs A;
            memset(&A, 0, sizeof A);
            A.screen = g_screen; A.padded = 1;
            A.n = n; A.in = dst[t & 1].as<int32_t>(); A.in_stride = stride;
            A.out = dst[(t & 1) ^ 1].as<int32_t>(); A.out_stride = stride;
            A.a_dev = dact.as<double>(); A.a_row_stride = T; A.a_col = t; A.a_div = rollouts_per_seq;
            if (t == 0) A.a_prev = a_prev0;
            else { A.a_prev_dev = dact.as<double>(); A.a_prev_col = t - 1; }
            A.seed = seed; A.step = step0 + (uint32_t)t; A.slot0 = 0; A.scal = start->scal;
            A.r_out = drew.as<double>();
            HIPCHECK(launch_step<false>(c, A));
            k_score_acc<<<grid, blk, 0, c->stream>>>(n, dacc.as<double>(), drew.as<double>(), scale, t == 0);
            scale *= discount;
        }
        k_score_mean<<<dim3((unsigned)ceil_div(n_seq, TPB)), blk, 0, c->stream>>>(n_seq, rollouts_per_seq, dacc.as<double>(), dscore.as<double>());
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(score, dscore.p, sizeof(double) * (size_t)n_seq, hipMemcpyDeviceToHost, c->stream));
        rc = fetch_flags(c, start->scal, flags, true);
        if (rc) return rc;
        HIPCHECK(hipStreamSynchronize(c->stream));
        return EPI_OK;
    });
}

// ---- ensembles kept on the device (ODE fitting) ------------------------------------

int epi_ensemble_destroy(epi_ensemble* e)
{
    if (!e) return EPI_OK;
    (void)hipSetDevice(e->ctx->device);
    (void)hipStreamSynchronize(e->ctx->stream);
    if (e->hist) (void)hipFree(e->hist);
    if (e->act) (void)hipFree(e->act);
    if (e->partial) (void)hipFree(e->partial);
    delete e;
    return EPI_OK;
}

int epi_ensemble_simulate(epi_ctx* c, epi_belief* b, int T, const double* actions, int act_per_particle, int act_per_day, double a_prev0,
                          uint64_t seed, uint32_t step0, epi_ensemble** out, uint32_t* flags)
{
    return guarded("epi_ensemble_simulate", [&]() -> int {
        int rc = belief_check(b, "epi_ensemble_simulate");
        if (rc) return rc;
        if (!c || b->ctx != c || !actions || !out || T < 1) return fail(EPI_EINVAL, "epi_ensemble_simulate: bad argument");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_ensemble_simulate: a phased update is in progress");
        HIPCHECK(hipSetDevice(c->device));
        const int64_t n = b->n;
        const int64_t arow = act_per_day ? T : 1;
        const size_t acount = (size_t)(act_per_particle ? n : 1) * (size_t)arow;
        epi_ensemble* e = new epi_ensemble();
        e->ctx = c; e->n = n; e->T = T;
        struct Owner {            // frees the half-built ensemble on every early return
            epi_ensemble* e;
            ~Owner() { if (e) epi_ensemble_destroy(e); }
        } owner{e};
        DevBuf dact;
        hipError_t he = dact.alloc(sizeof(double) * acount);
        if (he == hipSuccess) he = hipMalloc((void**)&e->hist, sizeof(int32_t) * 4 * (size_t)n * (size_t)T);
        if (he == hipSuccess) he = hipMalloc((void**)&e->act, sizeof(double) * (size_t)n);
        if (he == hipSuccess) he = hipMalloc((void**)&e->partial, sizeof(double) * ((size_t)ceil_div(n, TPB) + 1));
        if (he != hipSuccess) return fail(EPI_ENOMEM, "epi_ensemble_simulate: hipMalloc failed: %s", hipGetErrorString(he));
        HIPCHECK(hipMemcpyAsync(dact.p, actions, sizeof(double) * acount, hipMemcpyHostToDevice, c->stream));
        {   // the action the controlled ODE models see: the trajectory's first-day action
            std::vector<double> a0((size_t)n);
            for (int64_t m = 0; m < n; m++) a0[(size_t)m] = actions[act_per_particle ? (size_t)m * (size_t)arow : 0];
            HIPCHECK(hipMemcpyAsync(e->act, a0.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
        }
        for (int t = 0; t < T; t++) {
            StepArgs A;
            memset(&A, 0, sizeof A);
            A.screen = g_screen; A.padded = 1;
            A.n = n; A.in = b->buf[b->cur]; A.in_stride = b->stride; A.anc = b->has_anc ? b->anc : nullptr;
            A.out = b->buf[b->cur ^ 1]; A.out_stride = b->stride;
            const int64_t tcol = act_per_day ? t : 0;
            A.a_row_stride = 1;
            if (act_per_particle) { A.a_dev = dact.as<double>(); A.a_row_stride = arow; A.a_col = tcol; }
            else A.a = actions[tcol];
            A.a_prev = a_prev0;                                   // rewards are not computed here
            A.seed = seed; A.step = step0 + (uint32_t)t; A.slot0 = (uint64_t)b->slot0; A.scal = b->scal;
            A.hist = e->hist + (size_t)t * 4 * (size_t)n;
            HIPCHECK(launch_step<false>(c, A));
            b->cur ^= 1; b->has_anc = false;
        }
        rc = fetch_flags(c, b->scal, flags, true);
        if (rc) return rc;
        HIPCHECK(hipStreamSynchronize(c->stream));
        owner.e = nullptr;
        *out = e;
        return EPI_OK;
    });
}

int epi_ensemble_shape(const epi_ensemble* e, int64_t* n, int* T)
{
    if (!e) return fail(EPI_EINVAL, "epi_ensemble_shape: ensemble is NULL");
    if (n) *n = e->n;
    if (T) *T = e->T;
    return EPI_OK;
}

int epi_ensemble_download(epi_ensemble* e, int64_t* hist)
{
    return guarded("epi_ensemble_download", [&]() -> int {
        if (!e || !hist) return fail(EPI_EINVAL, "epi_ensemble_download: NULL argument");
        epi_ctx* c = e->ctx;
        HIPCHECK(hipSetDevice(c->device));
        const int64_t n = e->n;
        const int T = e->T;
        std::vector<int32_t> h((size_t)4 * (size_t)n * (size_t)T);
        HIPCHECK(hipMemcpyAsync(h.data(), e->hist, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        for (int64_t p = 0; p < n; p++)
            for (int t = 0; t < T; t++)
                for (int k = 0; k < 3; k++) hist[(p * T + t) * 3 + k] = h[((size_t)t * 4 + (size_t)k) * (size_t)n + (size_t)p];
        return EPI_OK;
    });
}

int epi_ensemble_loss(epi_ctx* c, epi_ensemble* e, int kind, const double* p, int n_p, int substeps, double* loss, double* per_traj)
{
    return guarded("epi_ensemble_loss", [&]() -> int {
        if (!c || !e || e->ctx != c || !p || !loss) return fail(EPI_EINVAL, "epi_ensemble_loss: bad argument");
        static const int want[4] = {2, 3, 3, 5};              // parameters of SIR, SIR_CTRL, SEIR, SEIR_CTRL (src/ODEFit.jl:40-55)
        if (kind < 0 || kind > 3) return fail(EPI_EINVAL, "epi_ensemble_loss: kind = %d", kind);
        if (n_p != want[kind]) return fail(EPI_EINVAL, "epi_ensemble_loss: model %d takes %d parameters, got %d", kind, want[kind], n_p);
        if (substeps < 1 || substeps > 4096) return fail(EPI_EINVAL, "epi_ensemble_loss: substeps = %d", substeps);
        HIPCHECK(hipSetDevice(c->device));
        double q[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < n_p; i++) q[i] = p[i];
        DevBuf dper;
        if (per_traj && dper.alloc(sizeof(double) * (size_t)e->n) != hipSuccess) return fail(EPI_ENOMEM, "epi_ensemble_loss: hipMalloc failed");
        const int64_t nb = ceil_div(e->n, TPB);
        k_ode_loss<<<dim3((unsigned)nb), dim3(TPB), 0, c->stream>>>(e->n, e->T, e->hist, e->act, kind, q[0], q[1], q[2], q[3], q[4], substeps,
                                                                  (double)c->P.N, per_traj ? dper.as<double>() : nullptr, e->partial);
        k_sum_partials<<<1, 1024, 0, c->stream>>>(nb, e->partial, e->partial + nb);
        HIPCHECK(hipGetLastError());
        double total = 0.0;
        HIPCHECK(hipMemcpyAsync(&total, e->partial + nb, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (per_traj) HIPCHECK(hipMemcpyAsync(per_traj, dper.p, sizeof(double) * (size_t)e->n, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        *loss = total / (double)e->n;                           // scaled by the number of simulations (src/ODEFit.jl:289)
        return EPI_OK;
    });
}

// ---- the filter update ---------------------------------------------------------

int epi_pf_frac_bits(int64_t n_total)
{
    int lg = 0;
    while (((int64_t)1 << lg) < n_total) lg++;
    int F = 53 - lg;
    return F > 40 ? 40 : F;
}

// The device memory an update uses beyond the particles, their ancestor index and their log weights: the scalars, the
// per-tile weight totals and the list of long child ranges.  Normally the belief's own (allocated with it).
struct UpdateScratch {
    DevScalars* scal;
    unsigned long long* blocksum;
    double* blocksq;
    unsigned long long* big;
};
static inline UpdateScratch own_scratch(epi_belief* b) { return UpdateScratch{b->scal, b->blocksum, b->blocksq, b->big}; }

// The fused step + weight kernel over the columns [col0, col0 + count) of the population (count < 0: all of it).  A
// range either ends with the population or holds whole workgroups: the surplus threads of a padded launch write the
// columns after the range's last one.
static int launch_predict(epi_ctx* c, epi_belief* b, const UpdateScratch& sc, double a, int64_t o, uint64_t seed, uint32_t step, bool timed = true,
                          const int32_t* o_dev = nullptr, int64_t col0 = 0, int64_t count = -1)
{
    if (count < 0) count = b->n - col0;
    StepArgs A;
    memset(&A, 0, sizeof A);
    A.screen = g_screen; A.padded = 1;
    A.n = count; A.in_stride = b->stride;
    // ancestors are columns of the whole input buffer; without them column m is read for slot m
    A.in = b->has_anc ? b->buf[b->cur] : b->buf[b->cur] + col0;
    A.anc = b->has_anc ? b->anc + col0 : nullptr;
    A.out = b->buf[b->cur ^ 1] + col0; A.out_stride = b->stride;
    A.a = a; A.a_row_stride = 1;
    A.seed = seed; A.step = step; A.slot0 = (uint64_t)(b->slot0 + col0);
    A.o = (double)o; A.o_dev = o_dev; A.lw = b->lw + col0; A.scal = sc.scal;
    hipEvent_t e0 = c->ev0, e1 = c->ev1;
    if (c->timing && timed) {
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventCreate(&e1));
        c->timed.emplace_back(e0, e1);
    }
    HIPCHECK(hipEventRecord(e0, c->stream));
    HIPCHECK(launch_step<true>(c, A));                                  // (no per-particle outputs: the weighing build has none)
    HIPCHECK(hipEventRecord(e1, c->stream));
    if (!(c->timing && timed)) c->ev_valid = true;
    b->seed = seed; b->step = step;
    b->frac_bits = epi_pf_frac_bits(b->n_total);
    return EPI_OK;
}

// d_lwmax_global / d_totals: the device-memory hand-over (both, or neither: the host has installed the maximum in scal)
static int launch_weigh(epi_ctx* c, epi_belief* b, const double* d_lwmax_global = nullptr, double* d_totals = nullptr)
{
    b->halo_used = 0;                   // the halo of the buffer this update writes starts empty
    b->filled.clear();
    b->selected = false;
    k_wsum<false><<<dim3((unsigned)b->nblocks), dim3(TPB), 0, c->stream>>>(b->n, b->lw, b->scal, b->frac_bits, b->blocksum, b->blocksq, d_lwmax_global);
    k_wscan<<<1, 1024, 0, c->stream>>>(b->nblocks, b->blocksum, b->blocksq, b->scal, b->big, 0, d_lwmax_global, d_totals);
    HIPCHECK(hipGetLastError());
    b->flags_delivered = d_totals != nullptr;
    b->cum_ready = false;               // written by whichever comes first: the ancestors of the owned slots, or an export
    return EPI_OK;
}

// the cumulative fixed-point weights of the shard, for k_export's search, unless the update has them already
static int ensure_cum(epi_ctx* c, epi_belief* b)
{
    if (b->cum_ready) return EPI_OK;
    k_wcum<<<dim3((unsigned)b->nblocks), dim3(TPB), 0, c->stream>>>(b->n, b->lw, b->scal, b->frac_bits, b->blocksum);
    HIPCHECK(hipGetLastError());
    b->cum_ready = true;
    return EPI_OK;
}

// ancestors of the owned slots [slot_lo, slot_hi) from the local cumulative weights (once per update: the list of long
// child ranges is the one k_wscan emptied)
static int launch_scatter(epi_ctx* c, epi_belief* b, unsigned long long W, unsigned long long off, double u, int64_t slot_lo, int64_t slot_hi)
{
    BigRange* list = reinterpret_cast<BigRange*>(b->big + 2);
    const double stepw = W ? (double)W / (double)b->n_total : 0.0, inv_stepw = W ? (double)b->n_total / (double)W : 0.0;
    if (!b->cum_ready) {
        // nothing has needed the cumulative weights yet: they are computed, written and used in one pass
        k_wcum_scatter_shard<<<dim3((unsigned)b->nblocks), dim3(TPB), 0, c->stream>>>(b->n, b->lw, b->scal, b->frac_bits, b->blocksum, W, off, stepw, inv_stepw,
                                                                                    b->n_total, u, slot_lo, slot_hi, b->slot0, b->anc, b->big, list);
        b->cum_ready = true;
    } else
    k_scatter<<<dim3((unsigned)ceil_div(b->n, TPB)), dim3(TPB), 0, c->stream>>>(b->n, reinterpret_cast<const unsigned long long*>(b->lw), W, off, stepw, inv_stepw,
                                                                               b->n_total, u, slot_lo, slot_hi, b->slot0, b->anc, b->big, list);
    int64_t gx = ceil_div(slot_hi - slot_lo, (int64_t)TPB * 8);
    gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
    k_scatter_big<<<dim3((unsigned)gx), dim3(TPB), 0, c->stream>>>(b->big, list, b->slot0, b->anc);
    HIPCHECK(hipGetLastError());
    return EPI_OK;
}

static int update_checks(epi_ctx* c, epi_belief* b, const char* who)
{
    int rc = shard_check(c, b, who);
    if (rc) return rc;
    if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "%s: a phased update is in progress", who);
    if (b->n_total != b->n || b->slot0 != 0) return fail(EPI_ESTATE, "%s: belief is a shard; use the phased calls", who);
    return EPI_OK;
}

// One whole update of a population held by one device, enqueued on the context's stream: step + weight, tile totals,
// their scan, cumulative weights and ancestors in one pass, the long child ranges.  Five launches (six when the last
// workgroup of the population is partial); nothing is allocated and the host does not wait.
static int enqueue_update(epi_ctx* c, epi_belief* b, const UpdateScratch& sc, double a, int64_t o, uint64_t seed, uint32_t step, bool timed = true,
                          const int32_t* o_dev = nullptr)
{
    int rc = launch_predict(c, b, sc, a, o, seed, step, timed, o_dev);
    if (rc) return rc;
    const double u = resample_offset(seed, step);
    BigRange* list = reinterpret_cast<BigRange*>(sc.big + 2);
    k_wsum<true><<<dim3((unsigned)b->nblocks), dim3(TPB), 0, c->stream>>>(b->n, b->lw, sc.scal, b->frac_bits, sc.blocksum, sc.blocksq);
    k_wscan<<<1, 1024, 0, c->stream>>>(b->nblocks, sc.blocksum, sc.blocksq, sc.scal, sc.big, 1);
    k_wcum_scatter<<<dim3((unsigned)b->nblocks), dim3(TPB), 0, c->stream>>>(b->n, b->lw, sc.scal, b->frac_bits, sc.blocksum, u, b->anc, sc.big, list);
    int64_t gx = ceil_div(b->n, (int64_t)TPB * 8);
    gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
    k_scatter_big<<<dim3((unsigned)gx), dim3(TPB), 0, c->stream>>>(sc.big, list, 0, b->anc);
    HIPCHECK(hipGetLastError());
    b->cur ^= 1; b->has_anc = true;
    return EPI_OK;
}

int epi_pf_update_async(epi_ctx* c, epi_belief* b, double a, int64_t o, uint64_t seed, uint32_t step)
{
    return guarded("epi_pf_update_async", [&]() -> int {
        int rc = update_checks(c, b, "epi_pf_update");
        if (rc) return rc;
        HIPCHECK(hipSetDevice(c->device));
        return enqueue_update(c, b, own_scratch(b), a, o, seed, step);
    });
}

// what the fixed-point weight total says about an update: every weight zero, or weights so collapsed that the
// truncation of the fixed-point weights is no longer negligible (bound n / W above 10^-3; see include/epi.h)
static inline uint32_t weight_flags(unsigned long long w_total, int64_t n_total)
{
    if (w_total == 0ULL) return EPI_F_ZERO_W;
    return (double)n_total > 1e-3 * (double)w_total ? EPI_F_COARSE_W : 0u;
}

// diagnostics of the update whose scalars are `d`; waits for the stream; clear: leaves the device flags zero for the
// next update that accumulates into the same scalars (not wanted for scalars the next user zeroes itself)
static int collect_from(epi_ctx* c, epi_belief* b, DevScalars* d, double* loglik, double* ess, uint32_t* flags, bool clear = true)
{
    DevScalars h;
    if (int rs = read_scalars(c, d, &h)) return rs;
    const uint32_t f = h.flags | weight_flags(h.w_total, b->n_total);
    if (flags) *flags = f;
    if (clear) HIPCHECK(hipMemsetAsync(&d->flags, 0, sizeof(unsigned int), c->stream));
    const double scale = ldexp(1.0, b->frac_bits);
    if (loglik) *loglik = h.w_total ? h.lwmax + log((double)h.w_total / scale / (double)b->n_total) - 0.91893853320467274178 : -INFINITY;
    if (ess) *ess = h.sumsq > 0.0 ? ((double)h.w_total * (double)h.w_total) / h.sumsq : 0.0;
    return EPI_OK;
}

int epi_pf_collect(epi_ctx* c, epi_belief* b, double* loglik, double* ess, uint32_t* flags)
{
    if (!c || !b || b->ctx != c) return fail(EPI_EINVAL, "epi_pf_collect: bad argument");
    HIPCHECK(hipSetDevice(c->device));
    return collect_from(c, b, b->scal, loglik, ess, flags);
}

int epi_pf_update(epi_ctx* c, epi_belief* b, double a, int64_t o, uint64_t seed, uint32_t step, double* loglik, double* ess, uint32_t* flags)
{
    int rc = epi_pf_update_async(c, b, a, o, seed, step);
    if (rc) return rc;
    return epi_pf_collect(c, b, loglik, ess, flags);
}

// ---- the update for callers that cannot allocate ---------------------------------------
// epi_pf_update_noalloc runs the kernels of epi_pf_update with the update's scratch (scalars, per-tile weight totals
// and sums of squares, the list of long child ranges) in a file-scope device array instead of the buffers a belief
// allocates for them.  Every device the library runs on holds its own copy of the array, at its own address, until the
// process ends; epi_ctx_create asks for that address once per context (with the context's device current) and host
// code only ever uses that pointer, never the array's name.
constexpr int64_t NOALLOC_MAX_PARTICLES = 1LL << 27;                 // 134 217 728: covers the 10^8-particle configuration
constexpr size_t NA_TILES = (size_t)(NOALLOC_MAX_PARTICLES / SCAN_TILE);
constexpr size_t NA_OFF_SUM = 256;                                   // after the DevScalars slot
constexpr size_t NA_OFF_SQ = NA_OFF_SUM + 8 * NA_TILES;
constexpr size_t NA_OFF_BIG = NA_OFF_SQ + 8 * NA_TILES;
constexpr size_t NA_BYTES = (NA_OFF_BIG + big_list_bytes(NOALLOC_MAX_PARTICLES) + 255) & ~(size_t)255;   // 1.4 MB
static __device__ __align__(256) unsigned char scratch[NA_BYTES];

static int static_scratch_address(unsigned char** out)
{
    HIPCHECK(hipGetSymbolAddress((void**)out, HIP_SYMBOL(scratch)));
    return EPI_OK;
}

int epi_pf_update_noalloc(epi_ctx* c, epi_belief* b, double a, int64_t o, uint64_t seed, uint32_t step, double* loglik, double* ess, uint32_t* flags)
{
    int rc = update_checks(c, b, "epi_pf_update_noalloc");
    if (rc) return rc;
    if (b->n > NOALLOC_MAX_PARTICLES)
        return fail(EPI_ERANGE, "epi_pf_update_noalloc: %lld particles exceed the static scratch's maximum of %lld; use epi_pf_update",
                    (long long)b->n, (long long)NOALLOC_MAX_PARTICLES);
    if (!c->d_static) return fail(EPI_ESTATE, "epi_pf_update_noalloc: the context has no address for the static scratch");
    HIPCHECK(hipSetDevice(c->device));
    unsigned char* const p = c->d_static;
    const UpdateScratch sc{reinterpret_cast<DevScalars*>(p), reinterpret_cast<unsigned long long*>(p + NA_OFF_SUM),
                           reinterpret_cast<double*>(p + NA_OFF_SQ), reinterpret_cast<unsigned long long*>(p + NA_OFF_BIG)};
    // the array keeps what the last call on this device left: the scalars (maximum code, flags) start from zero; the
    // tile totals are overwritten before they are read and k_wscan empties the range list
    HIPCHECK(hipMemsetAsync(p, 0, 256, c->stream));
    rc = enqueue_update(c, b, sc, a, o, seed, step, false);
    if (rc) return rc;
    // nothing may be left queued on the shared array when this returns (the next call on this device, from any
    // context, zeroes the scalars itself)
    return collect_from(c, b, sc.scal, loglik, ess, flags, false);
}

#ifdef EPI_PHASE_STAMPS
int epi_debug_phase_times(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
#ifdef EPI_TESTING
int epi_debug_node_exchange(int mode)
{
    g_node_exchange = mode & 3;
    return EPI_OK;
}

int epi_debug_step_impl(int cooperative)
{
    g_coop = cooperative < 0 ? 0 : (cooperative > 2 ? 2 : cooperative);
    return EPI_OK;
}

// epi_debug_f32_forms: the short device forms of the binary32 square root and quotient against the compiler's
// full expansions, over a run of consecutive bit patterns.
__global__ __launch_bounds__(256) void k_f32_forms(int mode, uint32_t first_bits, unsigned long long count, unsigned long long* mismatches)
{
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const float x = u2f(first_bits + (uint32_t)i);
        float got, want;
        if (mode == 0) { got = sqrt_rn(x); want = __builtin_sqrtf(x); }
        else if (mode == 1) { got = div_rn(1.0f, x); want = 1.0f / x; }
        else { const float f = x - 1.0f; got = div_rn(f, 2.0f + f); want = f / (2.0f + f); }
        bad += f2u(got) != f2u(want);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// The table-free attempt (btrs_quick) against the defining one on random (n, words): out[0] attempts, out[1] settled as
// accepted, out[2] settled as rejected, out[3] left open, out[4] settled differently from the defining test (must be 0).
__global__ __launch_bounds__(256) void k_btrs_quick_check(uint64_t seed, int32_t n_lo, uint32_t n_span, double pp, LfTable lf, unsigned long long* out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const Rng g = rng_make(seed, (uint64_t)i, 0);
    uint32_t h[4];
    rng_block(g, 9, 0, 0, h);
    const int32_t n = n_lo + (int32_t)(h[0] % n_span);
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    if ((double)n * pp >= 10.0) {
        const BtrsBin B = btrs_bin(pp);
        const BtrsSide S = btrs_side(pp);
        const BtrsConsts Cq = btrs_consts_quick(n, B), C = btrs_consts(n, B);
        for (uint32_t r = 0; r < 32; r++) {
            uint32_t w[4];
            rng_block(g, 1, r, 0, w);
            for (int half = 0; half < 2; half++) {
                int32_t kq;
                const int st = btrs_quick(Cq, S, w[2 * half], w[2 * half + 1], kq);
                const int32_t kd = btrs_attempt(C, B.lpq, lf, false, w[2 * half], w[2 * half + 1]);
                cnt[0]++;
                cnt[1] += st == BTRS_ACCEPT; cnt[2] += st == BTRS_REJECT; cnt[3] += st == BTRS_OPEN;
                cnt[4] += (st == BTRS_ACCEPT && kd != kq) || (st == BTRS_REJECT && kd >= 0);
            }
        }
    }
    for (int q = 0; q < 5; q++) {
        unsigned long long v = cnt[q];
        for (int d = 32; d >= 1; d >>= 1) v += shfl_down_u64(v, d);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[q], v);
    }
}

int epi_debug_btrs_quick(epi_ctx* c, uint64_t seed, int32_t n_lo, uint32_t n_span, double pp, int blocks, uint64_t* out5)
{
    if (!c || !out5 || n_lo < 1 || n_span < 1 || !(pp > 0.0 && pp <= 0.5) || blocks < 1) return fail(EPI_EINVAL, "epi_debug_btrs_quick: bad argument");
    HIPCHECK(hipSetDevice(c->device));
    DevBuf d;
    if (d.alloc(5 * sizeof(unsigned long long)) != hipSuccess) return fail(EPI_ENOMEM, "epi_debug_btrs_quick: hipMalloc failed");
    HIPCHECK(hipMemsetAsync(d.p, 0, 5 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_btrs_quick_check, dim3((unsigned)blocks), dim3(256), 0, c->stream, seed, n_lo, n_span, pp, LfTable{c->d_lf, LF_TABLE_N}, (unsigned long long*)d.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(out5, d.p, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return EPI_OK;
}

int epi_debug_f32_forms(epi_ctx* c, int mode, float first, float last, uint64_t* mismatches)
{
    if (!c || !mismatches || mode < 0 || mode > 2 || !(first > 0.0f) || !(last >= first) || !(last < INFINITY))
        return fail(EPI_EINVAL, "epi_debug_f32_forms: bad argument");
    HIPCHECK(hipSetDevice(c->device));
    DevBuf d;
    if (d.alloc(sizeof(unsigned long long)) != hipSuccess) return fail(EPI_ENOMEM, "epi_debug_f32_forms: hipMalloc failed");
    HIPCHECK(hipMemsetAsync(d.p, 0, sizeof(unsigned long long), c->stream));
    const uint32_t b0 = f2u(first);
    const unsigned long long count = (unsigned long long)(f2u(last) - b0) + 1ull;      // positive floats order like their bits
    hipLaunchKernelGGL(k_f32_forms, dim3(4096), dim3(256), 0, c->stream, mode, b0, count, (unsigned long long*)d.p);
    HIPCHECK(hipGetLastError());
    unsigned long long h = 0;
    HIPCHECK(hipMemcpyAsync(&h, d.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    *mismatches = h;
    return EPI_OK;
}

int epi_debug_screen(int enable)
{
    g_screen = enable != 0;
    return EPI_OK;
}
#endif // EPI_TESTING

int epi_ctx_timing(epi_ctx* c, int enable)
{
    if (!c) return fail(EPI_EINVAL, "epi_ctx_timing: ctx is NULL");
    c->timing = enable != 0;
    return EPI_OK;
}

int epi_ctx_timing_read(epi_ctx* c, double* sum_ms, int64_t* count)
{
    if (!c) return fail(EPI_EINVAL, "epi_ctx_timing_read: ctx is NULL");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (auto& pr : c->timed) {
        float ms = 0.0f;
        HIPCHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (sum_ms) *sum_ms = tot;
    if (count) *count = (int64_t)c->timed.size();
    c->timed.clear();
    return EPI_OK;
}

float epi_last_predict_ms(epi_ctx* c)
{
    if (!c || !c->ev_valid) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0f;
    return ms;
}

// ---- phased update for sharded populations ---------------------------------------

int epi_pf_predict(epi_ctx* c, epi_belief* b, double a, int64_t o, uint64_t seed, uint32_t step, double* lwmax_local)
{
    return guarded("epi_pf_predict", [&]() -> int {
        int rc = shard_check(c, b, "epi_pf_predict");
        if (rc) return rc;
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_pf_predict: previous update not finished");
        HIPCHECK(hipSetDevice(c->device));
        rc = launch_predict(c, b, own_scratch(b), a, o, seed, step);
        if (rc) return rc;
        DevScalars h;
        if (int rs = read_scalars(c, b->scal, &h)) return rs;
        if (lwmax_local) *lwmax_local = h.lwmax_bits ? order_decode(h.lwmax_bits) : -INFINITY;
        b->phase = PH_PREDICTED;
        return EPI_OK;
    });
}

int epi_pf_weigh(epi_ctx* c, epi_belief* b, double lwmax_global, uint64_t* w_local, double* sumsq_local, uint32_t* flags_local)
{
    int rc = shard_check(c, b, "epi_pf_weigh");
    if (rc) return rc;
    if (b->phase != PH_PREDICTED) return fail(EPI_ESTATE, "epi_pf_weigh: call epi_pf_predict first");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipMemcpyAsync(&b->scal->lwmax, &lwmax_global, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    rc = launch_weigh(c, b);
    if (rc) return rc;
    DevScalars h;
    if (int rs = read_scalars(c, b->scal, &h)) return rs;
    if (w_local) *w_local = h.w_local;
    if (sumsq_local) *sumsq_local = h.sumsq;
    if (flags_local) *flags_local = h.flags;
    b->phase = PH_WEIGHED;
    return EPI_OK;
}

int epi_pf_predict_dev(epi_ctx* c, epi_belief* b, double a, int64_t o, uint64_t seed, uint32_t step, double* d_lwmax_local)
{
    return guarded("epi_pf_predict_dev", [&]() -> int {
        int rc = shard_check(c, b, "epi_pf_predict_dev");
        if (rc) return rc;
        if (!d_lwmax_local) return fail(EPI_EINVAL, "epi_pf_predict_dev: NULL device pointer");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_pf_predict_dev: previous update not finished");
        HIPCHECK(hipSetDevice(c->device));
        rc = launch_predict(c, b, own_scratch(b), a, o, seed, step);
        if (rc) return rc;
        k_scal_export_lwmax<<<1, 1, 0, c->stream>>>(b->scal, d_lwmax_local);
        HIPCHECK(hipGetLastError());
        b->phase = PH_PREDICTED;
        return EPI_OK;
    });
}

// One part of the step of a sharded update: the columns [col0, col0 + count) of the shard.  The parts of an update
// may come in any order and must cover the shard exactly once; columns that still wait for a late import cannot be
// stepped.  The part that completes the cover hands the local maximum log weight over (as epi_pf_predict_dev).
int epi_pf_predict_part_dev(epi_ctx* c, epi_belief* b, double a, int64_t o, uint64_t seed, uint32_t step, int64_t col0, int64_t count,
                            double* d_lwmax_local)
{
    return guarded("epi_pf_predict_part_dev", [&]() -> int {
        int rc = shard_check(c, b, "epi_pf_predict_part_dev");
        if (rc) return rc;
        if (!d_lwmax_local) return fail(EPI_EINVAL, "epi_pf_predict_part_dev: NULL device pointer");
        if (b->phase != PH_IDLE && b->phase != PH_PENDING && b->phase != PH_PARTIAL)
            return fail(EPI_ESTATE, "epi_pf_predict_part_dev: previous update not finished");
        if (col0 < 0 || count < 1 || col0 + count > b->n)
            return fail(EPI_EINVAL, "epi_pf_predict_part_dev: columns [%lld, %lld) outside the shard's %lld", (long long)col0, (long long)(col0 + count), (long long)b->n);
        if (col0 % CTPB != 0 || (count % CTPB != 0 && col0 + count != b->n))
            return fail(EPI_EINVAL, "epi_pf_predict_part_dev: a part starts at a multiple of %d columns and holds a multiple of %d, or ends with the shard", CTPB, CTPB);
        if (b->phase == PH_PARTIAL && (a != b->part_a || o != b->part_o || seed != b->seed || step != b->step))
            return fail(EPI_EINVAL, "epi_pf_predict_part_dev: the parts of one update must agree in action, observation, seed and step");
        if (spans_overlap(b->done, col0, col0 + count)) return fail(EPI_ESTATE, "epi_pf_predict_part_dev: columns [%lld, %lld) were stepped already", (long long)col0, (long long)(col0 + count));
        if (spans_overlap(b->unfilled, col0, col0 + count))
            return fail(EPI_ESTATE, "epi_pf_predict_part_dev: columns in [%lld, %lld) still wait for particles of another shard (epi_pf_import_late)", (long long)col0, (long long)(col0 + count));
        HIPCHECK(hipSetDevice(c->device));
        rc = launch_predict(c, b, own_scratch(b), a, o, seed, step, true, nullptr, col0, count);
        if (rc) return rc;
        b->part_a = a; b->part_o = o;
        b->done.emplace_back(col0, col0 + count);
        b->phase = PH_PARTIAL;
        if (spans_complement(b->done, b->n).empty()) {
            k_scal_export_lwmax<<<1, 1, 0, c->stream>>>(b->scal, d_lwmax_local);
            HIPCHECK(hipGetLastError());
            b->done.clear();
            b->phase = PH_PREDICTED;
        }
        return EPI_OK;
    });
}

int epi_pf_weigh_dev(epi_ctx* c, epi_belief* b, const double* d_lwmax_global, double* d_totals)
{
    int rc = shard_check(c, b, "epi_pf_weigh_dev");
    if (rc) return rc;
    if (!d_lwmax_global || !d_totals) return fail(EPI_EINVAL, "epi_pf_weigh_dev: NULL device pointer");
    if (b->phase != PH_PREDICTED) return fail(EPI_ESTATE, "epi_pf_weigh_dev: call epi_pf_predict_dev first");
    HIPCHECK(hipSetDevice(c->device));
    rc = launch_weigh(c, b, d_lwmax_global, d_totals);
    if (rc) return rc;
    b->phase = PH_WEIGHED;
    return EPI_OK;
}

static inline int64_t shard_start(int64_t n_total, int G, int g) { return (int64_t)(((__int128)n_total * g) / G); }

int epi_pf_plan(int64_t n_total, int G, const uint64_t* w_shard, uint64_t seed, uint32_t step, int64_t* lo)
{
    if (n_total < 1 || G < 1 || !w_shard || !lo) return fail(EPI_EINVAL, "epi_pf_plan: bad argument");
    uint64_t W = 0;
    for (int g = 0; g < G; g++) W += w_shard[g];
    if (W == 0) {                       // nothing to resample: every shard keeps its own slots
        for (int g = 0; g <= G; g++) lo[g] = shard_start(n_total, G, g);
        return EPI_OK;
    }
    const double u = resample_offset(seed, step);
    const double stepw = (double)W / (double)n_total;
    uint64_t off = 0;
    lo[0] = 0;
    for (int g = 1; g < G; g++) {
        off += w_shard[g - 1];
        // first slot m with floor(fl(fl(m + u) * stepw)) >= off
        double guess = ceil((double)off / stepw - u);
        int64_t m = guess < 0.0 ? 0 : (guess > (double)n_total ? n_total : (int64_t)guess);
        auto tq = [&](int64_t mm) { return (uint64_t)floor(((double)mm + u) * stepw); };
        while (m > 0 && tq(m - 1) >= off) m--;
        while (m < n_total && tq(m) < off) m++;
        lo[g] = m;
    }
    lo[G] = n_total;
    return EPI_OK;
}

static int epi_pf_select_local_body(epi_ctx* c, epi_belief* b, int G, int shard, const uint64_t* w_shard, uint64_t seed, uint32_t step,
                        int64_t slot_begin, int64_t slot_end)
{
    int rc = shard_check(c, b, "epi_pf_select_local");
    if (rc) return rc;
    if (!w_shard || G < 1 || shard < 0 || shard >= G) return fail(EPI_EINVAL, "epi_pf_select_local: bad shard arguments");
    if (b->phase != PH_WEIGHED) return fail(EPI_ESTATE, "epi_pf_select_local: call epi_pf_weigh first");
    // the pass leaves the update's cumulative weights and its list of long child ranges behind: a second one would
    // append to that list and replay it
    if (b->selected) return fail(EPI_ESTATE, "epi_pf_select_local: once per update (one range of owned slots)");
    if (slot_begin < b->slot0 || slot_end > b->slot0 + b->n || slot_begin > slot_end)
        return fail(EPI_EINVAL, "epi_pf_select_local: slots [%lld, %lld) not owned by this shard", (long long)slot_begin, (long long)slot_end);
    if (slot_begin == slot_end) return EPI_OK;
    HIPCHECK(hipSetDevice(c->device));
    const double u = resample_offset(seed, step);
    unsigned long long off = 0ULL, total = 0ULL;
    for (int g = 0; g < G; g++) { if (g < shard) off += w_shard[g]; total += w_shard[g]; }
    rc = launch_scatter(c, b, total, off, u, slot_begin, slot_end);
    if (rc) return rc;
    b->selected = true;
    b->filled.emplace_back(slot_begin - b->slot0, slot_end - b->slot0);
    return EPI_OK;
}

int epi_pf_select_local(epi_ctx* c, epi_belief* b, int G, int shard, const uint64_t* w_shard, uint64_t seed, uint32_t step,
                        int64_t slot_begin, int64_t slot_end)
{
    return guarded("epi_pf_select_local", [&]() -> int { return epi_pf_select_local_body(c, b, G, shard, w_shard, seed, step, slot_begin, slot_end); });
}

int epi_pf_export(epi_ctx* c, epi_belief* b, int G, int shard, const uint64_t* w_shard, uint64_t seed, uint32_t step,
                  int64_t slot_begin, int64_t slot_end, void* dev_buf)
{
    int rc = shard_check(c, b, "epi_pf_export");
    if (rc) return rc;
    if (!w_shard || G < 1 || shard < 0 || shard >= G) return fail(EPI_EINVAL, "epi_pf_export: bad shard arguments");
    if (b->phase != PH_WEIGHED) return fail(EPI_ESTATE, "epi_pf_export: call epi_pf_weigh first");
    if (slot_begin > slot_end || slot_begin < 0 || slot_end > b->n_total || !dev_buf) return fail(EPI_EINVAL, "epi_pf_export: bad slot range or NULL buffer");
    if (slot_begin == slot_end) return EPI_OK;
    HIPCHECK(hipSetDevice(c->device));
    rc = ensure_cum(c, b);
    if (rc) return rc;
    const double u = resample_offset(seed, step);
    const int64_t count = slot_end - slot_begin;
    unsigned long long off = 0ULL, total = 0ULL;
    for (int g = 0; g < G; g++) { if (g < shard) off += w_shard[g]; total += w_shard[g]; }
    // the propagated particles of this update live in the buffer k_step wrote
    k_export<<<dim3((unsigned)ceil_div(count, TPB)), dim3(TPB), 0, c->stream>>>(count, slot_begin, b->slot0, reinterpret_cast<const unsigned long long*>(b->lw),
                                                                             b->n, total, off, b->n_total, u, b->buf[b->cur ^ 1], b->stride, b->rows,
                                                                             static_cast<int32_t*>(dev_buf));
    HIPCHECK(hipGetLastError());
    if (!c->external) HIPCHECK(hipStreamSynchronize(c->stream));
    return EPI_OK;
}

// late: after epi_pf_finish, into the buffer that has become the current one
static int import_block(epi_ctx* c, epi_belief* b, int64_t slot_begin, int64_t slot_end, const void* dev_buf, bool late, const char* who)
{
    if (slot_begin < b->slot0 || slot_end > b->slot0 + b->n || slot_begin > slot_end || !dev_buf)
        return fail(EPI_EINVAL, "%s: slots [%lld, %lld) not owned by this shard", who, (long long)slot_begin, (long long)slot_end);
    const int64_t count = slot_end - slot_begin;
    if (count == 0) return EPI_OK;
    const int64_t col0 = slot_begin - b->slot0;
    if (late && !spans_contain(b->unfilled, col0, col0 + count))
        return fail(EPI_ESTATE, "%s: slots [%lld, %lld) do not wait for an import", who, (long long)slot_begin, (long long)slot_end);
    if (b->halo_used + count > b->halo_cap) return fail(EPI_ERANGE, "%s: halo capacity %lld exceeded", who, (long long)b->halo_cap);
    if (late) spans_remove(b->unfilled, col0, col0 + count);
    HIPCHECK(hipSetDevice(c->device));
    k_import<<<dim3((unsigned)ceil_div(count, TPB)), dim3(TPB), 0, c->stream>>>(count, col0, b->n + b->halo_used, static_cast<const int32_t*>(dev_buf),
                                                                             b->buf[late ? b->cur : b->cur ^ 1], b->stride, b->rows, b->anc);
    HIPCHECK(hipGetLastError());
    if (!c->external) HIPCHECK(hipStreamSynchronize(c->stream));
    b->halo_used += count;
    if (!late) b->filled.emplace_back(col0, col0 + count);
    else if (b->phase == PH_PENDING && b->unfilled.empty()) b->phase = PH_IDLE;
    return EPI_OK;
}

static int epi_pf_import_body(epi_ctx* c, epi_belief* b, int64_t slot_begin, int64_t slot_end, const void* dev_buf)
{
    int rc = shard_check(c, b, "epi_pf_import");
    if (rc) return rc;
    if (b->phase != PH_WEIGHED) return fail(EPI_ESTATE, "epi_pf_import: call epi_pf_weigh first");
    return import_block(c, b, slot_begin, slot_end, dev_buf, false, "epi_pf_import");
}

int epi_pf_import(epi_ctx* c, epi_belief* b, int64_t slot_begin, int64_t slot_end, const void* dev_buf)
{
    return guarded("epi_pf_import", [&]() -> int { return epi_pf_import_body(c, b, slot_begin, slot_end, dev_buf); });
}

static int epi_pf_import_late_body(epi_ctx* c, epi_belief* b, int64_t slot_begin, int64_t slot_end, const void* dev_buf)
{
    int rc = shard_check(c, b, "epi_pf_import_late");
    if (rc) return rc;
    if (b->phase != PH_PENDING && b->phase != PH_PARTIAL) return fail(EPI_ESTATE, "epi_pf_import_late: no slot waits for an import");
    return import_block(c, b, slot_begin, slot_end, dev_buf, true, "epi_pf_import_late");
}

int epi_pf_import_late(epi_ctx* c, epi_belief* b, int64_t slot_begin, int64_t slot_end, const void* dev_buf)
{
    return guarded("epi_pf_import_late", [&]() -> int { return epi_pf_import_late_body(c, b, slot_begin, slot_end, dev_buf); });
}

static int epi_pf_finish_body(epi_ctx* c, epi_belief* b, uint32_t* flags)
{
    int rc = shard_check(c, b, "epi_pf_finish");
    if (rc) return rc;
    if (b->phase != PH_WEIGHED) return fail(EPI_ESTATE, "epi_pf_finish: no update in progress");
    HIPCHECK(hipSetDevice(c->device));
    if (flags) {
        rc = fetch_flags(c, b->scal, flags, true);
        if (rc) return rc;
    } else if (!b->flags_delivered) HIPCHECK(hipMemsetAsync(&b->scal->flags, 0, sizeof(unsigned int), c->stream));
    if (!c->external) HIPCHECK(hipStreamSynchronize(c->stream));
    b->cur ^= 1; b->has_anc = true;
    // owned slots that neither epi_pf_select_local nor epi_pf_import has given an ancestor wait for epi_pf_import_late
    b->unfilled = spans_complement(b->filled, b->n);
    b->filled.clear();
    b->selected = false;
    b->phase = b->unfilled.empty() ? PH_IDLE : PH_PENDING;
    return EPI_OK;
}

int epi_pf_finish(epi_ctx* c, epi_belief* b, uint32_t* flags)
{
    return guarded("epi_pf_finish", [&]() -> int { return epi_pf_finish_body(c, b, flags); });
}

// ---- summaries ---------------------------------------------------------------------

static int enqueue_summary(epi_ctx* c, epi_belief* b, bool want_sums, const double* q, int nq);

// column sums of b added to c->d_sums (which the caller has cleared)
static int enqueue_sums(epi_ctx* c, epi_belief* b)
{
    int64_t gx = ceil_div(b->n, TPB);
    if (gx > 1024) gx = 1024;
    k_sums<<<dim3((unsigned)gx, (unsigned)b->rows), dim3(TPB), 0, c->stream>>>(b->n, b->buf[b->cur], b->stride, b->has_anc ? b->anc : nullptr, c->d_sums);
    HIPCHECK(hipGetLastError());
    return EPI_OK;
}

int epi_belief_sums(epi_ctx* c, epi_belief* b, int64_t* sums)
{
    int rc = shard_check(c, b, "epi_belief_sums");
    if (rc) return rc;
    if (!sums) return fail(EPI_EINVAL, "epi_belief_sums: sums is NULL");
    if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_belief_sums: a phased update is in progress");
    HIPCHECK(hipSetDevice(c->device));
    const int H = c->P.H, D = c->P.D, D1 = D + 1;
    HIPCHECK(hipMemsetAsync(c->d_sums, 0, sizeof(long long) * (size_t)b->rows, c->stream));
    rc = enqueue_summary(c, b, true, nullptr, 0);
    if (rc) return rc;
    const long long* h = reinterpret_cast<const long long*>(c->h_pin);          // rows <= 2 + H + D (H - 1): a few KB
    HIPCHECK(hipMemcpyAsync(c->h_pin, c->d_sums, sizeof(long long) * (size_t)b->rows, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    sums[0] = h[0]; sums[1] = h[1];
    for (int i = 0; i < H; i++) sums[2 + i] = h[(size_t)(2 + i)];
    int64_t* ts = sums + 2 + H;
    for (int j = 0; j < D1 * H; j++) ts[j] = 0;
    for (int cidx = 1; cidx < H; cidx++)
        for (int r = 0; r < D; r++) ts[cidx * D1 + r] = h[(size_t)tests_row(H, r, cidx)];
    return EPI_OK;
}

static inline int64_t div_round_half_even(int64_t s, int64_t n) { return (int64_t)nearbyint((double)s / (double)n); }

int epi_mean_from_sums(const epi_params* p, int64_t n_total, const int64_t* sums, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    if (!p || !sums || !S || !I || !R || n_total < 1) return fail(EPI_EINVAL, "epi_mean_from_sums: bad argument");
    const int H = p->H, D1 = p->D + 1;
    *S = div_round_half_even(sums[0], n_total);                       // updater.jl:15
    int64_t tot = 0;
    for (int i = 0; i < H; i++) { I[i] = div_round_half_even(sums[2 + i], n_total); tot += I[i]; }   // :16
    *R = p->N - (*S + tot);                                           // :17
    if (T) for (int j = 0; j < D1 * H; j++) T[j] = div_round_half_even(sums[2 + H + j], n_total);     // :19
    if (*R < 0) return fail(EPI_EINVAL, "epi_mean_from_sums: mean R = %lld < 0 (src/updater.jl:18 asserts)", (long long)*R);
    return EPI_OK;
}

int epi_belief_mean(epi_ctx* c, epi_belief* b, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    int64_t sums[2 + EPI_MAX_H + (EPI_MAX_D + 1) * EPI_MAX_H];
    int rc = epi_belief_sums(c, b, sums);
    if (rc) return rc;
    return epi_mean_from_sums(&c->P, b->n, sums, S, I, R, T);
}

// 1-based rank of each quantile: the smallest value with at least ceil(q n) values <= it (rank 1 for q = 0)
static QselRanks qsel_ranks(int64_t n, const double* q, int nq)
{
    QselRanks R;
    R.nq = nq;
    for (int j = 0; j < QSEL_MAX; j++) {
        int64_t r = j < nq ? (int64_t)ceil(q[j] * (double)n) : 1;
        R.rank[j] = r < 1 ? 1 : (r > n ? n : r);
    }
    return R;
}

// bits of the population: an infected total is at most N, so [0, 2^bits) holds every total
static int population_bits(const epi_ctx* c)
{
    int bits = 1;
    while (bits < 31 && (c->P.N >> bits) != 0) bits++;
    return bits;
}

// the passes of a selection over the n totals in dv (its start is already in the state); first_done: the first pass has
// been made (by k_summary16 / k_summary_finish)
static int enqueue_qsel_passes(epi_ctx* c, int64_t n, const int32_t* dv, bool first_done)
{
    unsigned long long* const dh = reinterpret_cast<unsigned long long*>(c->d_qsel);
    QselState* const st = reinterpret_cast<QselState*>(c->d_qsel + QSEL_HIST_BYTES);
    int64_t g = ceil_div(n, (int64_t)QTPB * 8);                        // one trip of eight values per thread
    g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
    // the first window spans at most bits(N) bits, each pass resolves 11
    const int bits = population_bits(c);
    const int passes = bits <= QSEL_LOG_BINS ? 1 : (bits <= 2 * QSEL_LOG_BINS ? 2 : QSEL_MAX_PASSES);
    for (int pass = first_done ? 1 : 0; pass < passes; pass++) {
        k_qsel_pass<<<dim3((unsigned)g), dim3(QTPB), 0, c->stream>>>(n, dv, st, dh);
        k_qsel_narrow<<<1, QTPB, 0, c->stream>>>(st, dh);
    }
    HIPCHECK(hipGetLastError());
    return EPI_OK;
}

// Column sums (added to c->d_sums, which the caller has cleared) and / or order statistics of sum(I) of one belief,
// enqueued; the order statistics stay in the selection state (QselState::lo).  The infected totals go where the log
// weights of an update live (free between updates, 8 n bytes); partial results into the context's small buffer.
static int enqueue_summary(epi_ctx* c, epi_belief* b, bool want_sums, const double* q, int nq)
{
    const int64_t n = b->n;
    int32_t* const dv = reinterpret_cast<int32_t*>(b->lw);
    QselState* const st = reinterpret_cast<QselState*>(c->d_qsel + QSEL_HIST_BYTES);
    const QselRanks R = qsel_ranks(n, q, nq);
    const int32_t* anc = b->has_anc ? b->anc : nullptr;
    static_assert(sizeof(long long) * SUMMARY_SLOT * SUMMARY_MAX_GROUPS <= SMALL_BYTES, "partial results outgrew the context's small buffer");
    bool first_done = false;
    if (want_sums && b->rows == SUMMARY_ROWS) {
        int64_t gs = ceil_div(n, (int64_t)TPB * 8);
        gs = gs < 1 ? 1 : (gs > SUMMARY_MAX_GROUPS ? SUMMARY_MAX_GROUPS : gs);
        long long* const part = reinterpret_cast<long long*>(c->d_small);
        unsigned long long* const dh = reinterpret_cast<unsigned long long*>(c->d_qsel);
        const int W0 = nq > 0 ? population_bits(c) : 0;
        k_summary16<<<dim3((unsigned)gs), dim3(TPB), 0, c->stream>>>(n, b->buf[b->cur], b->stride, anc, dv, part, W0, dh);
        k_summary_finish<<<1, TPB, 0, c->stream>>>((int)gs, part, c->d_sums, R, st, W0, dh);
        first_done = true;
    } else {
        if (want_sums) { int rc = enqueue_sums(c, b); if (rc) return rc; }
        if (nq) {
            int64_t gs = ceil_div(n, (int64_t)TPB * 8);
            gs = gs < 1 ? 1 : (gs > 256 ? 256 : gs);                      // few workgroups: the kernel ends in a ticket
            k_suminf<<<dim3((unsigned)gs), dim3(TPB), 0, c->stream>>>(n, b->buf[b->cur], b->stride, anc, c->P.H, dv,
                                                                    reinterpret_cast<int32_t*>(c->d_small), R, st);
        }
    }
    HIPCHECK(hipGetLastError());
    if (nq) return enqueue_qsel_passes(c, n, dv, first_done);
    return EPI_OK;
}

int epi_belief_suminf_quantiles(epi_ctx* c, epi_belief* b, const double* q, int nq, int64_t* out)
{
    int rc = belief_check(b, "epi_belief_suminf_quantiles");
    if (rc) return rc;
    if (!c || b->ctx != c || !q || !out || nq < 1 || nq > QSEL_MAX) return fail(EPI_EINVAL, "epi_belief_suminf_quantiles: 1 to %d quantiles", QSEL_MAX);
    if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_belief_suminf_quantiles: a phased update is in progress");
    for (int j = 0; j < nq; j++) if (!(q[j] >= 0.0 && q[j] <= 1.0)) return fail(EPI_EINVAL, "epi_belief_suminf_quantiles: q[%d] = %g", j, q[j]);
    HIPCHECK(hipSetDevice(c->device));
    rc = enqueue_summary(c, b, false, q, nq);
    if (rc) return rc;
    HIPCHECK(hipMemcpyAsync(c->h_pin, c->d_qsel + QSEL_HIST_BYTES, sizeof(QselState), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const QselState* h = reinterpret_cast<const QselState*>(c->h_pin);
    if (h->W != 0) return fail(EPI_ESTATE, "epi_belief_suminf_quantiles: selection did not finish (internal)");
    for (int j = 0; j < nq; j++) out[j] = (int64_t)h->lo[j];
    return EPI_OK;
}

// ---- days of the closed loop, enqueued back to back --------------------------------------
// One record per day, written by k_day_pack and read back in one copy after the last day.
struct DayHead {
    double r, lwmax, sumsq;
    unsigned long long w_total;
    int32_t o, S, sumI, R;
    unsigned int flags, pad;
    int32_t quant[QSEL_MAX];
};
static inline size_t day_rec_bytes(int rows) { return (sizeof(DayHead) + sizeof(long long) * (size_t)rows + 63) & ~(size_t)63; }

// collects the day's results into its record and clears what the next day accumulates into (flags, column sums)
__global__ __launch_bounds__(64) void k_day_pack(unsigned char* __restrict__ rec, int rows, DevScalars* __restrict__ scal_b, DevScalars* __restrict__ scal_t,
                                                  const int32_t* __restrict__ truth, int64_t tstride, int H, const int32_t* __restrict__ o_dev,
                                                  const double* __restrict__ r_dev, long long* __restrict__ sums, const QselState* __restrict__ st, int nq)
{
    DayHead* h = reinterpret_cast<DayHead*>(rec);
    long long* tail = reinterpret_cast<long long*>(rec + sizeof(DayHead));
    if (threadIdx.x == 0) {
        h->r = *r_dev; h->o = *o_dev;
        h->lwmax = scal_b->lwmax; h->sumsq = scal_b->sumsq; h->w_total = scal_b->w_total;
        h->flags = scal_b->flags | scal_t->flags; h->pad = nq ? (unsigned int)st->W : 0u;   // pad: bits the selection left undecided (0)
        scal_b->flags = 0u; scal_t->flags = 0u;
        int32_t tot = 0;
        for (int i = 0; i < H; i++) tot += truth[(int64_t)(2 + i) * tstride];
        h->S = truth[0]; h->R = truth[tstride]; h->sumI = tot;
        for (int j = 0; j < QSEL_MAX; j++) h->quant[j] = j < nq ? st->lo[j] : 0;
    }
    for (int r = threadIdx.x; r < rows; r += 64) { tail[r] = sums[r]; sums[r] = 0; }
}

int epi_closed_loop_days(epi_ctx* c, epi_belief* truth, epi_belief* b, int ndays, const double* actions, double a_prev,
                         uint64_t seed_true, uint32_t step_true0, uint64_t seed, uint32_t step0, int nq, const double* q,
                         int64_t* o_out, double* r_out, int64_t* true_sir, double* loglik, double* ess, uint32_t* flags,
                         int64_t* sums, int64_t* quant)
{
    return guarded("epi_closed_loop_days", [&]() -> int {
        int rc = update_checks(c, b, "epi_closed_loop_days");
        if (rc) return rc;
        rc = belief_check(truth, "epi_closed_loop_days");
        if (rc) return rc;
        if (truth->ctx != c || truth->n != 1 || truth->phase != PH_IDLE) return fail(EPI_EINVAL, "epi_closed_loop_days: the true state must be an idle population of one on the same context");
        if (truth == b) return fail(EPI_EINVAL, "epi_closed_loop_days: the true state and the belief must be different populations");
        if (ndays < 1 || !actions) return fail(EPI_EINVAL, "epi_closed_loop_days: ndays = %d", ndays);
        if (nq < 0 || nq > QSEL_MAX || (nq > 0 && (!q || !quant))) return fail(EPI_EINVAL, "epi_closed_loop_days: 0 to %d quantiles", QSEL_MAX);
        for (int j = 0; j < nq; j++) if (!(q[j] >= 0.0 && q[j] <= 1.0)) return fail(EPI_EINVAL, "epi_closed_loop_days: q[%d] = %g", j, q[j]);
        HIPCHECK(hipSetDevice(c->device));
        const int H = c->P.H, D = c->P.D, D1 = D + 1, rows = b->rows;
        const size_t rec = day_rec_bytes(rows), total = rec * (size_t)ndays;
        if (total > c->days_cap) {
            HIPCHECK(hipStreamSynchronize(c->stream));
            if (c->d_days) (void)hipFree(c->d_days);
            c->d_days = nullptr; c->days_cap = 0;
            if (hipMalloc((void**)&c->d_days, total) != hipSuccess) return fail(EPI_ENOMEM, "epi_closed_loop_days: hipMalloc failed");
            c->days_cap = total;
        }
        int32_t* const d_o = reinterpret_cast<int32_t*>(c->d_qsel + QSEL_HIST_BYTES + 128);
        double* const d_r = reinterpret_cast<double*>(c->d_qsel + QSEL_HIST_BYTES + 136);
        const QselState* const st = reinterpret_cast<const QselState*>(c->d_qsel + QSEL_HIST_BYTES);
        HIPCHECK(hipMemsetAsync(c->d_sums, 0, sizeof(long long) * (size_t)rows, c->stream));
        for (int d = 0; d < ndays; d++) {
            // the true system's step (src/POMDPsInterface.jl:55); its observation stays on the device
            StepArgs A;
            memset(&A, 0, sizeof A);
            A.screen = g_screen; A.padded = 1;
            A.n = 1; A.in = truth->buf[truth->cur]; A.in_stride = truth->stride; A.anc = truth->has_anc ? truth->anc : nullptr;
            A.out = truth->buf[truth->cur ^ 1]; A.out_stride = truth->stride;
            A.a = actions[d]; A.a_row_stride = 1;
            A.a_prev = d ? actions[d - 1] : a_prev;
            A.seed = seed_true; A.step = step_true0 + (uint32_t)d; A.slot0 = (uint64_t)truth->slot0; A.scal = truth->scal;
            A.o_out = d_o; A.r_out = d_r;
            HIPCHECK(launch_step<false>(c, A));
            truth->cur ^= 1; truth->has_anc = false;
            // the belief update on that observation (src/POMDPsInterface.jl:56)
            rc = enqueue_update(c, b, own_scratch(b), actions[d], 0, seed, step0 + (uint32_t)d, true, d_o);
            if (rc) return rc;
            // what a plot of the belief history needs of the posterior (src/plots.jl:47)
            rc = enqueue_summary(c, b, sums != nullptr, q, nq);
            if (rc) return rc;
            k_day_pack<<<1, 64, 0, c->stream>>>(c->d_days + rec * (size_t)d, rows, b->scal, truth->scal, truth->buf[truth->cur], truth->stride, H,
                                                d_o, d_r, c->d_sums, st, nq);
            HIPCHECK(hipGetLastError());
        }
        std::vector<unsigned char> hv;
        const unsigned char* h;
        if (total <= PIN_BYTES) {
            HIPCHECK(hipMemcpyAsync(c->h_pin, c->d_days, total, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            h = c->h_pin;
        } else {
            hv.resize(total);
            HIPCHECK(hipMemcpyAsync(hv.data(), c->d_days, total, hipMemcpyDeviceToHost, c->stream));
            HIPCHECK(hipStreamSynchronize(c->stream));
            h = hv.data();
        }
        const double scale = ldexp(1.0, b->frac_bits);
        const int ns = 2 + H + D1 * H;
        for (int d = 0; d < ndays; d++) {
            const DayHead* e = reinterpret_cast<const DayHead*>(h + rec * (size_t)d);
            const long long* tail = reinterpret_cast<const long long*>(h + rec * (size_t)d + sizeof(DayHead));
            if (e->pad != 0u) return fail(EPI_ESTATE, "epi_closed_loop_days: day %d: selection did not finish (an infected total above the population?)", d);
            if (o_out) o_out[d] = e->o;
            if (r_out) r_out[d] = e->r;
            if (true_sir) { true_sir[3 * d] = e->S; true_sir[3 * d + 1] = e->sumI; true_sir[3 * d + 2] = e->R; }
            if (loglik) loglik[d] = e->w_total ? e->lwmax + log((double)e->w_total / scale / (double)b->n_total) - 0.91893853320467274178 : -INFINITY;
            if (ess) ess[d] = e->sumsq > 0.0 ? ((double)e->w_total * (double)e->w_total) / e->sumsq : 0.0;
            if (flags) flags[d] = e->flags | weight_flags(e->w_total, b->n_total);
            if (quant) for (int j = 0; j < nq; j++) quant[(size_t)d * nq + j] = e->quant[j];
            if (sums) {
                int64_t* sd = sums + (size_t)d * ns;
                sd[0] = tail[0]; sd[1] = tail[1];
                for (int i = 0; i < H; i++) sd[2 + i] = tail[2 + i];
                int64_t* ts = sd + 2 + H;
                for (int j = 0; j < D1 * H; j++) ts[j] = 0;
                for (int cidx = 1; cidx < H; cidx++)
                    for (int r = 0; r < D; r++) ts[cidx * D1 + r] = tail[tests_row(H, r, cidx)];
            }
        }
        return EPI_OK;
    });
}

int epi_belief_sample(epi_ctx* c, epi_belief* b, uint64_t seed, uint64_t ctr, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    return guarded("epi_belief_sample", [&]() -> int {
        int rc = belief_check(b, "epi_belief_sample");
        if (rc) return rc;
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_belief_sample: a phased update is in progress");
        HIPCHECK(hipSetDevice(c->device));
        const int H = c->P.H, D = c->P.D, D1 = D + 1;
        Rng g = rng_make(seed, ctr, 0);
        uint32_t w[4];
        rng_block(g, K_INIT, 0x300u, 0, w);
        uint64_t r53 = (((uint64_t)(w[0] >> 5)) << 26) | (uint64_t)(w[1] >> 6);
        int64_t slot = (int64_t)(((__int128)r53 * b->n) >> 53);
        int32_t col = (int32_t)slot;
        HIPCHECK(hipStreamSynchronize(c->stream));
        if (b->has_anc) HIPCHECK(hipMemcpy(&col, b->anc + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<int32_t> h((size_t)b->rows);
        HIPCHECK(hipMemcpy2D(h.data(), sizeof(int32_t), b->buf[b->cur] + col, sizeof(int32_t) * (size_t)b->stride, sizeof(int32_t), (size_t)b->rows, hipMemcpyDeviceToHost));
        if (S) *S = h[0];
        if (R) *R = h[1];
        if (I) for (int i = 0; i < H; i++) I[i] = h[(size_t)(2 + i)];
        if (T) {
            for (int j = 0; j < D1 * H; j++) T[j] = 0;
            for (int cidx = 1; cidx < H; cidx++)
                for (int r = 0; r < D; r++) T[cidx * D1 + r] = h[(size_t)tests_row(H, r, cidx)];
        }
        return EPI_OK;
    });
}

// `count` particles uniformly at random, element j with the stream (seed, ctr0 + j) of epi_belief_sample: one launch
// gathers their rows into a packed buffer, one copy brings them to the host.
__global__ __launch_bounds__(TPB) void k_sample_many(int64_t count, int64_t n, uint64_t seed, uint64_t ctr0, const int32_t* __restrict__ src, int64_t stride,
                                                      const int32_t* __restrict__ anc, int rows, int32_t* __restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= count) return;
    const Rng g = rng_make(seed, ctr0 + (uint64_t)j, 0);
    uint32_t w[4];
    rng_block(g, K_INIT, 0x300u, 0, w);
    const uint64_t r53 = (((uint64_t)(w[0] >> 5)) << 26) | (uint64_t)(w[1] >> 6);
    // floor(r53 n / 2^53) from the 128-bit product
    const uint64_t lo = r53 * (uint64_t)n, hi = __umul64hi(r53, (uint64_t)n);
    const int64_t slot = (int64_t)((hi << 11) | (lo >> 53));
    const int64_t col = anc ? (int64_t)anc[slot] : slot;
    for (int r = 0; r < rows; r++) out[j * rows + r] = src[(int64_t)r * stride + col];
}

int epi_belief_sample_many(epi_ctx* c, epi_belief* b, uint64_t seed, uint64_t ctr0, int64_t count, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    return guarded("epi_belief_sample_many", [&]() -> int {
        int rc = belief_check(b, "epi_belief_sample_many");
        if (rc) return rc;
        if (!c || b->ctx != c) return fail(EPI_EINVAL, "epi_belief_sample_many: the belief belongs to another context");
        if (b->phase != PH_IDLE) return fail(EPI_ESTATE, "epi_belief_sample_many: a phased update is in progress");
        if (count < 1 || count > (1LL << 24)) return fail(EPI_EINVAL, "epi_belief_sample_many: count = %lld not in [1, 2^24]", (long long)count);
        HIPCHECK(hipSetDevice(c->device));
        const int H = c->P.H, D = c->P.D, D1 = D + 1, rows = b->rows;
        DevBuf d;
        if (d.alloc(sizeof(int32_t) * (size_t)count * (size_t)rows) != hipSuccess) return fail(EPI_ENOMEM, "epi_belief_sample_many: hipMalloc failed");
        k_sample_many<<<dim3((unsigned)ceil_div(count, TPB)), dim3(TPB), 0, c->stream>>>(count, b->n, seed, ctr0, b->buf[b->cur], b->stride,
                                                                                         b->has_anc ? b->anc : nullptr, rows, d.as<int32_t>());
        HIPCHECK(hipGetLastError());
        std::vector<int32_t> h((size_t)count * (size_t)rows);
        HIPCHECK(hipMemcpyAsync(h.data(), d.p, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        for (int64_t j = 0; j < count; j++) {
            const int32_t* p = h.data() + (size_t)j * (size_t)rows;
            if (S) S[j] = p[0];
            if (R) R[j] = p[1];
            if (I) for (int i = 0; i < H; i++) I[j * H + i] = p[2 + i];
            if (T) {
                int64_t* t = T + j * (int64_t)D1 * H;
                for (int q = 0; q < D1 * H; q++) t[q] = 0;
                for (int cidx = 1; cidx < H; cidx++)
                    for (int r = 0; r < D; r++) t[cidx * D1 + r] = p[tests_row(H, r, cidx)];
            }
        }
        return EPI_OK;
    });
}

// ---- one population sharded over several devices, driven by one host thread ------------------------
// epi_mctx holds G shards (an epi_ctx and an epi_belief each, shard g = global slots [n g / G, n (g + 1) / G)) and runs
// the phased update over them: the host never waits between the phases of one shard, only once per update for the
// weight totals that the resampling plan needs.  Particles that change shard go device to device
// (hipMemcpyPeerAsync) in packed blocks; the two small exchanges (maximum log weight, then four doubles per shard)
// go either through page-locked host memory or through RCCL (ncclAllReduce / ncclAllGather inside one group call),
// which is looked up at run time so that the library has no link-time dependency on it.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static int rccl_load(RcclApi& R)
{
    R.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!R.handle) R.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!R.handle) return fail(EPI_EINVAL, "epi_mctx_create: RCCL asked for but librccl.so cannot be loaded: %s", dlerror());
    R.CommInitAll = reinterpret_cast<decltype(R.CommInitAll)>(dlsym(R.handle, "ncclCommInitAll"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.handle, "ncclCommDestroy"));
    R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(dlsym(R.handle, "ncclAllReduce"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.handle, "ncclAllGather"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(dlsym(R.handle, "ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(dlsym(R.handle, "ncclGroupEnd"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.handle, "ncclGetErrorString"));
    if (!R.CommInitAll || !R.CommDestroy || !R.AllReduce || !R.AllGather || !R.GroupStart || !R.GroupEnd || !R.GetErrorString)
        return fail(EPI_EINVAL, "epi_mctx_create: librccl.so lacks an entry point this library uses");
    return EPI_OK;
}

// the largest of the shards' local maxima, read where they lie (x[g][0]: shard g's, on its own device)
__global__ void k_max_of(const double* const* x, int G, double* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double mx = -INFINITY;
    for (int g = 0; g < G; g++) {           // system scope: another device wrote it, no cache of this one may answer
        const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(x[g]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const double v = __longlong_as_double((long long)bits);
        mx = v > mx ? v : mx;
    }
    *out = mx;
}

struct epi_mctx {
    int G = 0;
    int64_t n_total = 0;
    epi_params P;
    std::vector<int> device;
    std::vector<epi_ctx*> ctx;
    std::vector<epi_belief*> b;
    std::vector<int64_t> start;             // G + 1 shard boundaries
    std::vector<double*> d_x;               // per shard, on its device: [0] local max, [1] global max, [2..6) totals, [8 .. 8 + 4 G) all totals
    std::vector<int32_t*> d_send, d_recv;   // per shard: packed blocks leaving / arriving (rows x n_local int32 each)
    std::vector<hipEvent_t> ev;             // per shard: its outgoing copies are complete
    // An update normally returns with its copies in flight on the shards' copy streams; the next one steps the slots
    // that were filled locally first, then places the blocks, then steps the rest (epi_pf_predict_part_dev).
    std::vector<hipStream_t> cs;            // per shard: the stream its outgoing copies run on
    std::vector<hipEvent_t> ev_exp;         // per shard: its blocks are packed (main stream)
    std::vector<hipEvent_t> ev_max;         // per shard: its step has left its maximum log weight in d_x[g][0] (main stream)
    std::vector<const double**> d_peer;     // per shard, on its device: the G addresses d_x[0 .. G)
    bool direct_max = false;                // every shard's device can read every other shard's d_x (same device or peer access)
    struct Block { int from, to; int64_t s0, s1; size_t send_off, recv_off; };
    std::vector<Block> late;                // blocks of the last update that are not placed yet
    std::vector<Span> kept;                 // per shard: the columns its own particles filled in the last update
    bool overlap = true;
    int64_t overlapped = 0;                 // updates whose step ran around blocks in flight (on at least one shard)
    double* h_x = nullptr;                  // page-locked: 8 doubles per shard
    bool rccl = false;
    RcclApi api;
    std::vector<ncclComm_t> comm;
    int64_t moved = 0;                      // particles that changed shard in the last update
};

#define NCCLCHECK(m, expr)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(EPI_EHIP, "%s failed: %s", #expr, (m)->api.GetErrorString(r_));   \
    } while (0)

int epi_mctx_destroy(epi_mctx* m)
{
    if (!m) return EPI_OK;
    for (int g = 0; g < (int)m->ctx.size(); g++) {
        if (!m->ctx[g]) continue;
        (void)hipSetDevice(m->device[g]);
        (void)hipStreamSynchronize(m->ctx[g]->stream);
        if (g < (int)m->cs.size() && m->cs[g]) (void)hipStreamSynchronize(m->cs[g]);   // copies into other shards' buffers
    }
    for (size_t g = 0; g < m->comm.size(); g++) if (m->comm[g]) (void)m->api.CommDestroy(m->comm[g]);
    for (int g = 0; g < (int)m->ctx.size(); g++) {
        if (!m->ctx[g]) continue;                               // creation stopped before this shard (its device may not even exist)
        (void)hipSetDevice(m->device[g]);
        if (g < (int)m->ev.size() && m->ev[g]) (void)hipEventDestroy(m->ev[g]);
        if (g < (int)m->ev_exp.size() && m->ev_exp[g]) (void)hipEventDestroy(m->ev_exp[g]);
        if (g < (int)m->ev_max.size() && m->ev_max[g]) (void)hipEventDestroy(m->ev_max[g]);
        if (g < (int)m->d_peer.size() && m->d_peer[g]) (void)hipFree((void*)m->d_peer[g]);
        if (g < (int)m->cs.size() && m->cs[g]) (void)hipStreamDestroy(m->cs[g]);
        if (g < (int)m->d_x.size() && m->d_x[g]) (void)hipFree(m->d_x[g]);
        if (g < (int)m->d_send.size() && m->d_send[g]) (void)hipFree(m->d_send[g]);
        if (g < (int)m->d_recv.size() && m->d_recv[g]) (void)hipFree(m->d_recv[g]);
        if (g < (int)m->b.size() && m->b[g]) (void)epi_belief_destroy(m->b[g]);
        if (m->ctx[g]) (void)epi_ctx_destroy(m->ctx[g]);
    }
    if (m->h_x) (void)hipHostFree(m->h_x);
    if (m->api.handle) (void)dlclose(m->api.handle);
    delete m;
    return EPI_OK;
}

int epi_mctx_create(const int* device_ids, int n_shards, const epi_params* p, int64_t n_total, int use_rccl, epi_mctx** out)
{
    return guarded("epi_mctx_create", [&]() -> int {
        if (!device_ids || !p || !out || n_shards < 1 || n_shards > 64) return fail(EPI_EINVAL, "epi_mctx_create: bad argument (1 to 64 shards)");
        if (n_total < n_shards) return fail(EPI_EINVAL, "epi_mctx_create: fewer particles than shards");
        if (use_rccl)
            for (int g = 0; g < n_shards; g++)
                for (int h = 0; h < g; h++)
                    if (device_ids[g] == device_ids[h]) return fail(EPI_EINVAL, "epi_mctx_create: RCCL needs one device per shard (device %d is named twice)", device_ids[g]);
        epi_mctx* m = new epi_mctx();
        struct Owner { epi_mctx* m; ~Owner() { if (m) epi_mctx_destroy(m); } } owner{m};
        const int G = n_shards;
        m->G = G; m->n_total = n_total; m->P = *p; m->rccl = use_rccl != 0;
        m->device.assign(device_ids, device_ids + G);
        m->ctx.assign(G, nullptr); m->b.assign(G, nullptr); m->d_x.assign(G, nullptr); m->d_send.assign(G, nullptr); m->d_recv.assign(G, nullptr);
        m->ev.assign(G, nullptr); m->ev_exp.assign(G, nullptr); m->cs.assign(G, nullptr); m->kept.assign(G, Span(0, 0));
        m->ev_max.assign(G, nullptr); m->d_peer.assign(G, nullptr);
        bool direct = true;
        m->start.resize(G + 1);
        for (int g = 0; g <= G; g++) m->start[g] = shard_start(n_total, G, g);
        const int rows = state_rows(p->H, p->D);
        for (int g = 0; g < G; g++) {
            int rc = epi_ctx_create(device_ids[g], &m->ctx[g]);
            if (rc) return rc;
            rc = epi_params_set(m->ctx[g], p);
            if (rc) return rc;
            m->ctx[g]->external = true;                         // the phases of an update only enqueue; the host waits where this code says so
            const int64_t nl = m->start[g + 1] - m->start[g];
            rc = epi_belief_create(m->ctx[g], nl, G > 1 ? nl : 0, &m->b[g]);
            if (rc) return rc;
            rc = epi_belief_set_shard(m->b[g], n_total, m->start[g]);
            if (rc) return rc;
            HIPCHECK(hipSetDevice(device_ids[g]));
            HIPCHECK(hipMalloc((void**)&m->d_x[g], sizeof(double) * (size_t)(8 + 4 * G)));
            HIPCHECK(hipMemset(m->d_x[g], 0, sizeof(double) * (size_t)(8 + 4 * G)));
            if (G > 1) {
                HIPCHECK(hipMalloc((void**)&m->d_send[g], sizeof(int32_t) * (size_t)rows * (size_t)nl));
                HIPCHECK(hipMalloc((void**)&m->d_recv[g], sizeof(int32_t) * (size_t)rows * (size_t)nl));
            }
            HIPCHECK(hipEventCreateWithFlags(&m->ev[g], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&m->ev_exp[g], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&m->ev_max[g], hipEventDisableTiming));
            HIPCHECK(hipStreamCreateWithFlags(&m->cs[g], hipStreamNonBlocking));
            for (int h = 0; h < g; h++) {                       // direct copies between the two devices where the hardware allows
                if (device_ids[h] == device_ids[g]) continue;
                int can = 0, can_back = 0;
                bool on = false;
                if (hipDeviceCanAccessPeer(&can, device_ids[g], device_ids[h]) == hipSuccess && can &&
                    hipDeviceCanAccessPeer(&can_back, device_ids[h], device_ids[g]) == hipSuccess && can_back) {
                    (void)hipSetDevice(device_ids[g]);
                    hipError_t e1 = hipDeviceEnablePeerAccess(device_ids[h], 0);
                    (void)hipSetDevice(device_ids[h]);
                    hipError_t e2 = hipDeviceEnablePeerAccess(device_ids[g], 0);
                    (void)hipGetLastError();                    // "already enabled" is not an error here
                    on = (e1 == hipSuccess || e1 == hipErrorPeerAccessAlreadyEnabled) && (e2 == hipSuccess || e2 == hipErrorPeerAccessAlreadyEnabled);
                }
                if (!on) direct = false;
            }
        }
        // where every device reads every other's memory, each shard takes the maximum of the G local maxima itself
        // (k_max_of); its table of their addresses is filled once, here
        m->direct_max = direct;
        if (direct)
            for (int g = 0; g < G; g++) {
                HIPCHECK(hipSetDevice(device_ids[g]));
                HIPCHECK(hipMalloc((void**)&m->d_peer[g], sizeof(double*) * (size_t)G));
                HIPCHECK(hipMemcpy((void*)m->d_peer[g], m->d_x.data(), sizeof(double*) * (size_t)G, hipMemcpyHostToDevice));
            }
        HIPCHECK(hipHostMalloc((void**)&m->h_x, sizeof(double) * 8 * (size_t)G, hipHostMallocDefault));
        if (m->rccl) {
            int rc = rccl_load(m->api);
            if (rc) return rc;
            m->comm.assign(G, nullptr);
            NCCLCHECK(m, m->api.CommInitAll(m->comm.data(), G, device_ids));
        }
        owner.m = nullptr;
        *out = m;
        return EPI_OK;
    });
}

int epi_mctx_shards(const epi_mctx* m) { return m ? m->G : 0; }
int64_t epi_mctx_size(const epi_mctx* m) { return m ? m->n_total : 0; }

int64_t epi_mctx_overlapped_updates(const epi_mctx* m) { return m ? m->overlapped : 0; }

int epi_mctx_set_overlap(epi_mctx* m, int enable)
{
    if (!m) return fail(EPI_EINVAL, "epi_mctx_set_overlap: NULL");
    m->overlap = enable != 0;
    return EPI_OK;
}

// the population is being replaced: blocks in flight are let land and forgotten
static void mctx_drop_late(epi_mctx* m)
{
    if (m->late.empty()) return;
    for (int g = 0; g < m->G; g++) { (void)hipSetDevice(m->device[g]); (void)hipStreamSynchronize(m->cs[g]); }
    m->late.clear();
}

// places the blocks the last update left in flight (each owner's stream waits for the sender's copies)
static int mctx_settle(epi_mctx* m)
{
    for (const epi_mctx::Block& k : m->late) {
        HIPCHECK(hipSetDevice(m->device[k.to]));
        HIPCHECK(hipStreamWaitEvent(m->ctx[k.to]->stream, m->ev[k.from], 0));
        int rc = epi_pf_import_late(m->ctx[k.to], m->b[k.to], k.s0, k.s1, m->d_recv[k.to] + k.recv_off);
        if (rc) return rc;
    }
    m->late.clear();
    return EPI_OK;
}

int epi_mctx_init_normal(epi_mctx* m, double mu, double sigma, uint64_t seed)
{
    if (!m) return fail(EPI_EINVAL, "epi_mctx_init_normal: NULL");
    mctx_drop_late(m);
    for (int g = 0; g < m->G; g++) { int rc = epi_belief_init_normal(m->b[g], mu, sigma, seed); if (rc) return rc; }
    return EPI_OK;
}

int epi_mctx_upload(epi_mctx* m, const int64_t* S, const int64_t* I, const int64_t* R, const int64_t* T)
{
    if (!m || !S || !I || !R) return fail(EPI_EINVAL, "epi_mctx_upload: NULL");
    mctx_drop_late(m);
    const int H = m->P.H, D1 = m->P.D + 1;
    for (int g = 0; g < m->G; g++) {
        const int64_t s0 = m->start[g];
        int rc = epi_belief_upload(m->b[g], S + s0, I + s0 * H, R + s0, T ? T + s0 * D1 * H : nullptr);
        if (rc) return rc;
    }
    return EPI_OK;
}

int epi_mctx_download(epi_mctx* m, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    if (!m) return fail(EPI_EINVAL, "epi_mctx_download: NULL");
    if (int rs = mctx_settle(m)) return rs;
    const int H = m->P.H, D1 = m->P.D + 1;
    for (int g = 0; g < m->G; g++) {
        const int64_t s0 = m->start[g];
        int rc = epi_belief_download(m->b[g], S ? S + s0 : nullptr, I ? I + s0 * H : nullptr, R ? R + s0 : nullptr, T ? T + s0 * D1 * H : nullptr);
        if (rc) return rc;
    }
    return EPI_OK;
}

int epi_mctx_sums(epi_mctx* m, int64_t* sums)
{
    if (!m || !sums) return fail(EPI_EINVAL, "epi_mctx_sums: NULL");
    if (int rs = mctx_settle(m)) return rs;
    const int ns = 2 + m->P.H + (m->P.D + 1) * m->P.H;
    int64_t part[2 + EPI_MAX_H + (EPI_MAX_D + 1) * EPI_MAX_H];
    for (int j = 0; j < ns; j++) sums[j] = 0;
    for (int g = 0; g < m->G; g++) {
        int rc = epi_belief_sums(m->ctx[g], m->b[g], part);
        if (rc) return rc;
        for (int j = 0; j < ns; j++) sums[j] += part[j];
    }
    return EPI_OK;
}

int epi_mctx_mean(epi_mctx* m, int64_t* S, int64_t* I, int64_t* R, int64_t* T)
{
    int64_t sums[2 + EPI_MAX_H + (EPI_MAX_D + 1) * EPI_MAX_H];
    int rc = epi_mctx_sums(m, sums);
    if (rc) return rc;
    return epi_mean_from_sums(&m->P, m->n_total, sums, S, I, R, T);
}

int epi_mpf_update(epi_mctx* m, double a, int64_t o, uint64_t seed, uint32_t step, double* loglik, double* ess, uint32_t* flags, int64_t* moved)
{
    return guarded("epi_mpf_update", [&]() -> int {
        if (!m) return fail(EPI_EINVAL, "epi_mpf_update: NULL");
        const int G = m->G;
        const int rows = state_rows(m->P.H, m->P.D);
        auto dev = [&](int g) { return hipSetDevice(m->device[g]); };
        auto st = [&](int g) { return m->ctx[g]->stream; };
        // 1. fused weight + step of every shard; the local maximum log weight stays on its device.  With blocks of the
        // last update still in flight: first the columns every shard filled from its own particles (whole workgroups),
        // then the blocks, then the rest.
        typedef epi_mctx::Block Block;
        std::vector<Span> inner((size_t)G, Span(0, 0));
        if (!m->late.empty()) {
            for (int g = 0; g < G; g++) {
                const int64_t nl = m->start[g + 1] - m->start[g];
                const int64_t first = ceil_div(m->kept[g].first, CTPB) * CTPB;
                const int64_t last = m->kept[g].second == nl ? nl : m->kept[g].second / CTPB * CTPB;
                if (m->b[g]->phase != PH_PENDING || last - first < (nl / 4 > CTPB ? nl / 4 : CTPB)) continue;   // nothing waits, or little to hide it behind
                int rc = epi_pf_predict_part_dev(m->ctx[g], m->b[g], a, o, seed, step, first, last - first, m->d_x[g]);
                if (rc) return rc;
                inner[g] = Span(first, last);
            }
            for (int g = 0; g < G; g++) if (inner[g].second > inner[g].first) { m->overlapped++; break; }
            if (int rs = mctx_settle(m)) return rs;
        }
        for (int g = 0; g < G; g++) {
            const int64_t nl = m->start[g + 1] - m->start[g];
            int rc = EPI_OK;
            if (inner[g].second == inner[g].first) rc = epi_pf_predict_dev(m->ctx[g], m->b[g], a, o, seed, step, m->d_x[g]);
            else {
                if (inner[g].first > 0) rc = epi_pf_predict_part_dev(m->ctx[g], m->b[g], a, o, seed, step, 0, inner[g].first, m->d_x[g]);
                if (!rc && inner[g].second < nl) rc = epi_pf_predict_part_dev(m->ctx[g], m->b[g], a, o, seed, step, inner[g].second, nl - inner[g].second, m->d_x[g]);
            }
            if (rc) return rc;
        }
        // 2. the global maximum: over RCCL; else read by every shard from the other shards' device memory once their
        // steps are through (no host wait); else (devices without access to each other) through host memory
        if (m->rccl) {
            NCCLCHECK(m, m->api.GroupStart());
            for (int g = 0; g < G; g++) NCCLCHECK(m, m->api.AllReduce(m->d_x[g], m->d_x[g] + 1, 1, ncclDouble, ncclMax, m->comm[g], st(g)));
            NCCLCHECK(m, m->api.GroupEnd());
        } else if (m->direct_max && !(g_node_exchange & 2)) {
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipEventRecord(m->ev_max[g], st(g))); }
            for (int g = 0; g < G; g++) {
                HIPCHECK(dev(g));
                for (int h = 0; h < G; h++) if (h != g) HIPCHECK(hipStreamWaitEvent(st(g), m->ev_max[h], 0));
                k_max_of<<<1, 64, 0, st(g)>>>(m->d_peer[g], G, m->d_x[g] + 1);
                HIPCHECK(hipGetLastError());
            }
        } else {
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipMemcpyAsync(m->h_x + 8 * g, m->d_x[g], sizeof(double), hipMemcpyDeviceToHost, st(g))); }
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipStreamSynchronize(st(g))); }
            double mx = -INFINITY;
            for (int g = 0; g < G; g++) mx = m->h_x[8 * g] > mx ? m->h_x[8 * g] : mx;
            for (int g = 0; g < G; g++) m->h_x[8 * g + 1] = mx;
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipMemcpyAsync(m->d_x[g] + 1, m->h_x + 8 * g + 1, sizeof(double), hipMemcpyHostToDevice, st(g))); }
        }
        // 3. fixed-point weights relative to it, local prefix sums, local totals
        for (int g = 0; g < G; g++) {
            int rc = epi_pf_weigh_dev(m->ctx[g], m->b[g], m->d_x[g] + 1, m->d_x[g] + 2);
            if (rc) return rc;
        }
        // 4. everyone's totals: {weight total, sum of squares, flags, global maximum} per shard -- the update's one wait
        std::vector<double> all((size_t)4 * G);
        if (m->rccl) {
            NCCLCHECK(m, m->api.GroupStart());
            for (int g = 0; g < G; g++) NCCLCHECK(m, m->api.AllGather(m->d_x[g] + 2, m->d_x[g] + 8, 4, ncclDouble, m->comm[g], st(g)));
            NCCLCHECK(m, m->api.GroupEnd());
            HIPCHECK(dev(0));
            std::vector<double> tmp((size_t)4 * G);
            HIPCHECK(hipMemcpyAsync(tmp.data(), m->d_x[0] + 8, sizeof(double) * 4 * (size_t)G, hipMemcpyDeviceToHost, st(0)));
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipStreamSynchronize(st(g))); }
            all = tmp;
        } else {
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipMemcpyAsync(m->h_x + 8 * g + 2, m->d_x[g] + 2, sizeof(double) * 4, hipMemcpyDeviceToHost, st(g))); }
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipStreamSynchronize(st(g))); }
            for (int g = 0; g < G; g++) for (int k = 0; k < 4; k++) all[(size_t)4 * g + k] = m->h_x[8 * g + 2 + k];
        }
        std::vector<uint64_t> w((size_t)G);
        double sumsq = 0.0;
        uint32_t fl = 0;
        unsigned long long W = 0ULL;
        for (int g = 0; g < G; g++) {
            w[g] = (uint64_t)all[(size_t)4 * g]; W += w[g];
            sumsq += all[(size_t)4 * g + 1];
            fl |= (uint32_t)all[(size_t)4 * g + 2];
        }
        const double lwmax = all[3];
        // 5. the plan: shard g's particles are the ancestors of global slots [lo[g], lo[g + 1])
        std::vector<int64_t> lo((size_t)G + 1);
        int rc = epi_pf_plan(m->n_total, G, w.data(), seed, step, lo.data());
        if (rc) return rc;
        // 6. particles whose slot another shard owns: packed on the source, copied device to device, placed on the owner
        // A shard receives at most its own n_local slots, so its receive buffer holds every arriving block side by
        // side.  A shard that holds most of the weight sends more than n_local slots: blocks go out in pieces of at
        // most n_local slots through the one send buffer, which the next piece's export may overwrite because it
        // follows the previous piece's copy on the same stream.
        // When no shard sends more than it owns (the usual case) the blocks lie side by side in the send buffers too, the
        // copies run on the senders' copy streams, and the update returns with them in flight: the next update (or
        // whatever reads the population next) places the blocks.  Buffers are reused one update later, after the host
        // has waited for that update's weight totals on every main stream, i.e. after every import of this one.
        std::vector<Block> blocks;
        std::vector<size_t> recv_used((size_t)G, 0), send_used((size_t)G, 0);
        bool side_by_side = true;
        m->moved = 0;
        for (int g = 0; g < G; g++)
            for (int h = 0; h < G; h++) {
                if (h == g) continue;
                const int64_t s0 = lo[g] > m->start[h] ? lo[g] : m->start[h], s1 = lo[g + 1] < m->start[h + 1] ? lo[g + 1] : m->start[h + 1];
                const int64_t cap = m->start[g + 1] - m->start[g];
                for (int64_t p0 = s0; p0 < s1; p0 += cap) {
                    const int64_t p1 = p0 + cap < s1 ? p0 + cap : s1;
                    blocks.push_back(Block{g, h, p0, p1, send_used[g], recv_used[h]});
                    recv_used[h] += (size_t)rows * (size_t)(p1 - p0);
                    send_used[g] += (size_t)rows * (size_t)(p1 - p0);
                    if (send_used[g] > (size_t)rows * (size_t)cap) side_by_side = false;
                    m->moved += p1 - p0;
                }
            }
        const bool in_flight = m->overlap && side_by_side && !blocks.empty();
        // the ancestors of the slots that stay come first: that pass also leaves the cumulative weights the exports search
        for (int g = 0; g < G; g++) {
            const int64_t s0 = lo[g] > m->start[g] ? lo[g] : m->start[g], s1 = lo[g + 1] < m->start[g + 1] ? lo[g + 1] : m->start[g + 1];
            m->kept[g] = s1 > s0 ? Span(s0 - m->start[g], s1 - m->start[g]) : Span(0, 0);
            if (s1 > s0) { rc = epi_pf_select_local(m->ctx[g], m->b[g], G, g, w.data(), seed, step, s0, s1); if (rc) return rc; }
        }
        for (const Block& k : blocks) {
            int32_t* const src = m->d_send[k.from] + (in_flight ? k.send_off : 0);
            rc = epi_pf_export(m->ctx[k.from], m->b[k.from], G, k.from, w.data(), seed, step, k.s0, k.s1, src);
            if (rc) return rc;
            if (in_flight) continue;
            const size_t bytes = sizeof(int32_t) * (size_t)rows * (size_t)(k.s1 - k.s0);
            HIPCHECK(dev(k.from));
            if (m->device[k.from] == m->device[k.to] && !(g_node_exchange & 1))
                HIPCHECK(hipMemcpyAsync(m->d_recv[k.to] + k.recv_off, src, bytes, hipMemcpyDeviceToDevice, st(k.from)));
            else
                HIPCHECK(hipMemcpyPeerAsync(m->d_recv[k.to] + k.recv_off, m->device[k.to], src, m->device[k.from], bytes, st(k.from)));
        }
        if (in_flight) {
            for (int g = 0; g < G; g++) {
                if (!send_used[g]) continue;
                HIPCHECK(dev(g));
                HIPCHECK(hipEventRecord(m->ev_exp[g], st(g)));
                HIPCHECK(hipStreamWaitEvent(m->cs[g], m->ev_exp[g], 0));
            }
            for (const Block& k : blocks) {
                const size_t bytes = sizeof(int32_t) * (size_t)rows * (size_t)(k.s1 - k.s0);
                HIPCHECK(dev(k.from));
                if (m->device[k.from] == m->device[k.to] && !(g_node_exchange & 1))
                    HIPCHECK(hipMemcpyAsync(m->d_recv[k.to] + k.recv_off, m->d_send[k.from] + k.send_off, bytes, hipMemcpyDeviceToDevice, m->cs[k.from]));
                else
                    HIPCHECK(hipMemcpyPeerAsync(m->d_recv[k.to] + k.recv_off, m->device[k.to], m->d_send[k.from] + k.send_off, m->device[k.from], bytes, m->cs[k.from]));
            }
            for (int g = 0; g < G; g++) { if (!send_used[g]) continue; HIPCHECK(dev(g)); HIPCHECK(hipEventRecord(m->ev[g], m->cs[g])); }
        } else
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipEventRecord(m->ev[g], st(g))); }
        if (in_flight) {
            for (int g = 0; g < G; g++) { rc = epi_pf_finish(m->ctx[g], m->b[g], nullptr); if (rc) return rc; }
            m->late = blocks;
        } else {
            for (const Block& k : blocks) {
                HIPCHECK(dev(k.to));
                HIPCHECK(hipStreamWaitEvent(st(k.to), m->ev[k.from], 0));
                rc = epi_pf_import(m->ctx[k.to], m->b[k.to], k.s0, k.s1, m->d_recv[k.to] + k.recv_off);
                if (rc) return rc;
            }
            for (int g = 0; g < G; g++) { rc = epi_pf_finish(m->ctx[g], m->b[g], nullptr); if (rc) return rc; }
            // the next update's exports overwrite the send buffers: their copies must have been consumed, which the
            // receivers' imports (ordered after the events above) guarantee once every stream has reached this point
            for (int g = 0; g < G; g++) { HIPCHECK(dev(g)); HIPCHECK(hipStreamSynchronize(st(g))); }
        }
        fl |= weight_flags(W, m->n_total);
        const int F = epi_pf_frac_bits(m->n_total);
        if (flags) *flags = fl;
        if (loglik) *loglik = W ? lwmax + log((double)W / ldexp(1.0, F) / (double)m->n_total) - 0.91893853320467274178 : -INFINITY;
        if (ess) *ess = sumsq > 0.0 ? ((double)W * (double)W) / sumsq : 0.0;
        if (moved) *moved = m->moved;
        return EPI_OK;
    });
}

} // extern "C"
