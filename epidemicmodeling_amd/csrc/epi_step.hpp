// One InfectionSim step of one particle, streamed bin by bin over the
// struct-of-arrays device state.
//
// Reference semantics (WhiffleFish/EpidemicModeling):
//   src/simulate.jl:53-61   SymptomaticIsolation
//   src/simulate.jl:73-82   PositiveTests
//   src/simulate.jl:99-129  UpdateIsolations
//   src/simulate.jl:137-151 SimStep
//   src/simulate.jl:37-45   incident_infections, src/init.jl:71-78 RVsum
//   src/simulate.jl:153-158 reward
//   src/POMDPsInterface.jl:11-23 observation (moments from the pre-step state)
//
// Device state layout: int32 rows of `stride` elements, one element per
// particle (struct of arrays):
//   row 0            S
//   row 1            R
//   rows 2 .. 2+H-1  I[age]
//   rows 2+H ..      pending tests that carry across steps, only for
//                    test_delay D >= 1: reference Tests[r, c] for r = 1..D,
//                    c = 2..H at row 2 + H + (r-1)(H-1) + (c-2).
// Row D+1 and column 1 of the reference Tests matrix are always zero between
// steps (simulate.jl:125-126), so they are not stored.
//
// The step is evaluated per age bin with no per-thread arrays, so H and D are
// run-time values.  Floating-point sums whose order matters for the integer
// results (the infection sum of simulate.jl:38-41, the moment sums of
// POMDPsInterface.jl:19-20) are accumulated in the reference's index order.
#pragma once
#include "epi_rng.hpp"
#include "../../include/epi.h"

namespace epi {

__host__ __device__ __forceinline__ int state_rows(int H, int D) { return 2 + H + D * (H - 1); }
__host__ __device__ __forceinline__ int tests_row(int H, int r0, int c0) { return 2 + H + r0 * (H - 1) + (c0 - 1); }  // r0 in [0,D), c0 in [1,H)

// Parameters as the kernels read them (device memory): the caller's struct plus constants that
// depend on the probabilities only, for both binomial draws of a bin (0: isolation, 1: positive test).
// Everything one binomial draw of one age bin reads from the parameters, in one 64-byte record: the kernels fetch it
// with a single wide scalar load (one wait) instead of a chain of dependent loads from the separate tables below.
struct alignas(64) BinConsts {
    double lpq;                 // s_log(pp / (1 - pp))
    double lnq, sr;             // the inversion sampler's constants (as in the tables below)
    uint32_t ppx;               // the rejection sampler's per-probability constants (btrs_bin)
    float pq, a0x2;
    uint32_t nthr;
    int32_t kind, flip;
    float ppf, hqp, c3u, c3d;   // for the table-free bounds of the acceptance test (btrs_side)
    __host__ __device__ __forceinline__ BtrsBin btrs() const { return BtrsBin{ppx, pq, a0x2, lpq}; }
    __host__ __device__ __forceinline__ BtrsSide side() const { return BtrsSide{ppf, hqp, c3u, c3d}; }
};
static_assert(sizeof(BinConsts) == 64, "one 64-byte record per (draw, bin)");

struct DevParams {
    epi_params P;
    double lnq[2][EPI_MAX_H];   // s_log(1 - pp), pp = min(p, 1 - p)
    double sr[2][EPI_MAX_H];    // pp / (1 - pp)
    double lpq[2][EPI_MAX_H];   // s_log(pp / (1 - pp))
    double pp[2][EPI_MAX_H];    // the folded probability min(p, 1 - p)
    // binom_classify per bin: kind 0 = the draw is 0 (p <= 0), 1 = the draw is n (p >= 1), 2 = sampled, by
    // inversion for 0 < n < nthr and by rejection from nthr on; nthr = the smallest n with fl(n pp) >= 10
    // (2^31 if there is none below 2^31)
    uint32_t nthr[2][EPI_MAX_H];
    int32_t kind[2][EPI_MAX_H];
    int32_t flip[2][EPI_MAX_H]; // p > 1/2: the draw is n - Binomial(n, 1 - p)
    LfTable lf;                 // tabulated log factorials (device memory; null on the host)
    BinConsts bc[2][EPI_MAX_H]; // the same constants per (draw, bin); filled by pack_bin_consts once lf is set
    ZigTable zig;               // strips of the normal generator
};

inline void pack_bin_consts(DevParams& D)
{
    for (int j = 0; j < 2; j++)
        for (int i = 0; i < EPI_MAX_H; i++) {
            BinConsts& b = D.bc[j][i];
            const BtrsBin B = btrs_bin(D.pp[j][i]);
            b.lpq = B.lpq; b.lnq = D.lnq[j][i]; b.sr = D.sr[j][i];
            b.ppx = B.ppx; b.pq = B.pq; b.a0x2 = B.a0x2;
            const BtrsSide S = btrs_side(D.pp[j][i]);
            b.ppf = S.ppf; b.hqp = S.hqp; b.c3u = S.c3u; b.c3d = S.c3d;
            b.nthr = D.nthr[j][i]; b.kind = D.kind[j][i]; b.flip = D.flip[j][i];
        }
}

// binom_classify from one record (identical decisions)
__host__ __device__ __forceinline__ int binom_classify_rec(const BinConsts& b, int32_t n, int32_t& result)
{
    result = 0;
    if (n <= 0 || b.kind == 0) return BINOM_DONE;
    if (b.kind == 1) { result = n; return BINOM_DONE; }
    return (uint32_t)n < b.nthr ? BINOM_INVERSION : BINOM_REJECTION;
}

__host__ __device__ inline void derive_params(DevParams& D)
{
    for (int j = 0; j < 2; j++)
        for (int i = 0; i < EPI_MAX_H; i++) {
            const double p = i < D.P.H ? (j ? D.P.p[i] : D.P.q[i]) : 0.0;
            const double pp = p > 0.5 ? 1.0 - p : p;
            const bool ok = pp > 0.0 && pp < 1.0;
            D.lnq[j][i] = ok ? s_log(1.0 - pp) : 0.0;
            D.sr[j][i] = ok ? pp / (1.0 - pp) : 0.0;
            D.lpq[j][i] = ok ? s_log(pp / (1.0 - pp)) : 0.0;
            D.pp[j][i] = pp;
            D.kind[j][i] = !(p > 0.0) ? 0 : (p >= 1.0 ? 1 : 2);
            D.flip[j][i] = p > 0.5 && p < 1.0;                          // only sampled draws are folded
            uint32_t thr = 0x80000000u;
            if (ok && 10.0 / pp < 2147483648.0) {
                int64_t t = (int64_t)(10.0 / pp);
                if (t < 1) t = 1;
                while (t > 1 && !((double)(t - 1) * pp < 10.0)) t--;        // the product is monotone in t: exact search
                while (t < 2147483648LL && (double)t * pp < 10.0) t++;
                thr = (uint32_t)t;
            }
            D.nthr[j][i] = thr;
        }
}

// binom_classify from the per-bin tables (identical decisions)
__host__ __device__ __forceinline__ int binom_classify_tab(const DevParams& D, int j, int bin, int32_t n, int32_t& result)
{
    result = 0;
    const int kind = D.kind[j][bin];
    if (n <= 0 || kind == 0) return BINOM_DONE;
    if (kind == 1) { result = n; return BINOM_DONE; }
    return (uint32_t)n < D.nthr[j][bin] ? BINOM_INVERSION : BINOM_REJECTION;
}

// One binomial draw of a bin (j = 0 isolation, 1 positive test) from its attempt-0 word pair.
__device__ __forceinline__ int32_t bin_binomial(const DevParams& D, int bin, int j, int32_t n, uint32_t fa, uint32_t fb,
                                                const Rng& g, bool screen)
{
    int32_t x;
    const int cls = binom_classify_tab(D, j, bin, n, x);
    if (cls == BINOM_DONE) return x;
    const bool flip = D.flip[j][bin] != 0;
    if (cls == BINOM_INVERSION) x = binom_inversion(n, D.lnq[j][bin], D.sr[j][bin], fa);
    else {
        const BtrsBin B = D.bc[j][bin].btrs();
        const BtrsConsts C = btrs_consts(n, B);
        x = btrs_attempt(C, B.lpq, D.lf, screen, fa, fb);
        if (x < 0) x = btrs_finish(C, B.lpq, D.lf, screen, g, j ? K_POS : K_SYMPT, (uint32_t)bin, 1u);
    }
    return flip ? n - x : x;
}

// Variates from the Philox streams of one (slot, step).
struct PhiloxDraws {
    Rng g;
    bool screen;    // allow the binary32 screens of the rejection samplers (results do not depend on it)
    const DevParams* D;
    mutable uint32_t z_odd;     // words of the odd term of the last gamma pair block
    mutable uint32_t w_odd;
    __device__ __forceinline__ void binomials(int bin, int32_t n_inf, double, int32_t n_tested, double,
                                               int32_t& sympt, int32_t& pos) const
    {
        uint32_t w[4];
        rng_block(g, K_BINOM0, (uint32_t)bin, 0, w);
        // both draws as straight code (no selects on j; the scheduler overlaps the two)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int32_t r = bin_binomial(*D, bin, j, j ? n_tested : n_inf, j ? w[2] : w[0], j ? w[3] : w[1], g, screen);
            if (j) pos = r; else sympt = r;
        }
    }
    // infection terms are drawn in order 0, 1, 2, ...: an even term generates the block of its pair
    __device__ __forceinline__ double gamma(int bin, double shape, double scale) const
    {
        uint32_t wz, wu;
        if ((bin & 1) == 0) {
            uint32_t w[4];
            rng_block(g, K_GAMMA, (uint32_t)bin >> 1, 0, w);
            wz = w[0]; wu = w[2]; z_odd = w[1]; w_odd = w[3];
        } else { wz = z_odd; wu = w_odd; }
        return gamma_draw(shape, scale, wz, wu, &D->zig, g, (uint32_t)bin, screen);
    }
};

// Philox variates that are also written out (rows of n elements per age bin),
// so that another implementation of the step can be driven by the same draws.
struct RecordingDraws {
    PhiloxDraws d; int32_t* sympt; int32_t* pos; double* gam; int64_t n; int64_t idx;
    __device__ __forceinline__ void binomials(int bin, int32_t n_inf, double q, int32_t n_tested, double p,
                                               int32_t& s, int32_t& ps) const
    {
        d.binomials(bin, n_inf, q, n_tested, p, s, ps);
        sympt[(int64_t)bin * n + idx] = s;
        pos[(int64_t)bin * n + idx] = ps;
    }
    __device__ __forceinline__ double gamma(int bin, double shape, double scale) const
    {
        double v = d.gamma(bin, shape, scale);
        gam[(int64_t)bin * n + idx] = v;
        return v;
    }
};

// Variates supplied by the caller: rows of n elements per age bin.
struct InjectedDraws {
    const int32_t* sympt; const int32_t* pos; const double* gam; int64_t n; int64_t idx;
    __device__ __forceinline__ void binomials(int bin, int32_t, double, int32_t, double, int32_t& s, int32_t& p) const
    {
        s = sympt[(int64_t)bin * n + idx];
        p = pos[(int64_t)bin * n + idx];
    }
    __device__ __forceinline__ double gamma(int bin, double, double) const { return gam[(int64_t)bin * n + idx]; }
};

struct StepResult {
    int32_t new_inf;
    int32_t o;          // sum of positive tests
    int32_t sum_inf;    // sum of I after the step
    double mu, var;     // observation moments of the PRE-step state
    uint32_t flags;
};

struct BinOut { int32_t ipost; double mean, sd2; };

// Binary64 quotient a / b of operands whose exponents are ordinary (here: integers below 2^31, or 0 / 0, which gives
// NaN as the division does): the compiler's own expansion (reciprocal estimate, two Newton steps, quotient, residual
// correction) without the two scaling instructions and the fix-up of special operands around it, which do nothing
// for such operands.  Same steps in the same order: the same bits.  (Device code only; the host divides.)
EPI_HD double div_unscaled(double a, double b)
{
#if !defined(__HIP_DEVICE_COMPILE__)
    return a / b;
#else
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e0 = __builtin_fma(-b, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-b, r1, 1.0);
    const double r2 = __builtin_fma(r1, e1, r1);
    const double q0 = a * r2;
    const double res = __builtin_fma(-b, q0, a);
    return __builtin_fma(res, r2, q0);
#endif
}

// Process one age bin (0-based column i) of the isolation update.  Writes the
// shifted pending-test rows of column i+1 and returns I_i after isolation.
// DELAY = false: the build for test_delay = 0, without the pending-test rows
template <bool DELAY = true, class Draws>
__device__ __forceinline__ BinOut process_bin(const epi_params& P, int i, const int32_t* __restrict__ in, int64_t in_stride, int64_t src,
                                              int32_t* __restrict__ out, int64_t out_stride, int64_t dst,
                                              double a, const Draws& draws, int32_t& R, int32_t& o_sum, uint32_t& flags,
                                              int32_t* __restrict__ pos_out, int64_t pos_stride)
{
    const int H = P.H, D = DELAY ? P.D : 0;
    int32_t Ii = in[(int64_t)(2 + i) * in_stride + src];
    int32_t already = 0;
    if (D > 0 && i >= 1)
        for (int r = 0; r < D; r++) already += in[(int64_t)tests_row(H, r, i) * in_stride + src];
    double ntf = __builtin_floor((double)(Ii - already) * a);                  // simulate.jl:78
    int32_t nt = (int32_t)ntf;
    if (nt < 0) flags |= EPI_F_NEG_NT;
    const double pi = P.p[i];
    int32_t sympt, pos;
    draws.binomials(i, Ii, P.q[i], nt, pi, sympt, pos);                         // simulate.jl:58, :79
    BinOut bo;
    bo.mean = ntf * pi;                                                        // POMDPsInterface.jl:19
    double sd = __builtin_sqrt(ntf * pi * (1.0 - pi));
    bo.sd2 = sd * sd;                                                          // POMDPsInterface.jl:20
    double pr = div_unscaled((double)sympt, (double)Ii);                       // simulate.jl:104
    if (pr != pr) pr = 0.0;                                                    // simulate.jl:105
    const double om = 1.0 - pr;
    R += sympt; Ii -= sympt;                                                   // simulate.jl:107-108
    // column i of the tests matrix after :110 and :112; row 0 leaves the pipeline (:114-115),
    // rows 1..D move to rows 0..D-1 of column i+1 (:122), column H wraps to column 1 and is cleared (:125)
    for (int r = 0; r <= D; r++) {
        int32_t tv;
        if (r == D) tv = pos;
        else tv = (i >= 1) ? in[(int64_t)tests_row(H, r, i) * in_stride + src] : 0;
        int32_t ts = (int32_t)__builtin_floor(om * (double)tv);
        if (r == 0) { R += ts; Ii -= ts; }
        else if (i + 1 < H) out[(int64_t)tests_row(H, r - 1, i + 1) * out_stride + dst] = ts;
    }
    if (Ii < 0) flags |= EPI_F_NEG_I;                                          // simulate.jl:117
    o_sum += pos;
    if (pos_out) pos_out[(int64_t)i * pos_stride + dst] = pos;
    bo.ipost = Ii;
    return bo;
}

// Full step of the particle stored at column `src` of `in`, written to column
// `dst` of `out`.
template <bool DELAY = true, class Draws>
__device__ __forceinline__ StepResult particle_step(const epi_params& P, const int32_t* __restrict__ in, int64_t in_stride, int64_t src,
                                                    int32_t* __restrict__ out, int64_t out_stride, int64_t dst,
                                                    double a, const Draws& draws,
                                                    int32_t* __restrict__ pos_out, int64_t pos_stride)
{
    const int H = P.H;
    StepResult res;
    res.flags = 0; res.o = 0;
    int32_t S = in[src];
    int32_t R = in[in_stride + src];
    // The last age bin goes first: its survivors recover (simulate.jl:144) and, through the
    // circular shift of simulate.jl:145, seed term 1 of the infection sum; bin i then feeds term
    // i + 2.  Iteration `it` therefore handles bin (it == 0 ? H-1 : it-1) and infection term it.
    double infsum = 0.0;
    int32_t sum_inf = 0;
    double tm = 0.0, tv = 0.0;
    BinOut last;
    last.ipost = 0; last.mean = 0.0; last.sd2 = 0.0;
#pragma unroll 1
    for (int it = 0; it < H; it++) {
        const int i = it == 0 ? H - 1 : it - 1;
        BinOut bo = process_bin<DELAY>(P, i, in, in_stride, src, out, out_stride, dst, a, draws, R, res.o, res.flags, pos_out, pos_stride);
        if (it == 0) {
            last = bo;
            R += bo.ipost;                                                         // simulate.jl:144
        } else {
            tm += bo.mean; tv += bo.sd2;
            out[(int64_t)(2 + it) * out_stride + dst] = bo.ipost;                  // simulate.jl:145
            sum_inf += bo.ipost;
        }
        infsum += draws.gamma(it, P.k[it] * (double)bo.ipost, P.theta[it]);        // simulate.jl:40, init.jl:71-78
    }
    tm += last.mean; tv += last.sd2;
    int64_t denom = P.N - (int64_t)R;
    int32_t newi = 0;
    if (denom <= 0) res.flags |= EPI_F_BAD_DENOM;
    else {
        double sprop = div_unscaled((double)S, (double)denom);                 // simulate.jl:42
        newi = (int32_t)__builtin_floor(sprop * infsum);                       // simulate.jl:44
    }
    out[(int64_t)2 * out_stride + dst] = newi;                                 // simulate.jl:147
    out[dst] = S - newi;                                                       // simulate.jl:148
    out[out_stride + dst] = R;
    res.new_inf = newi;
    res.sum_inf = sum_inf + newi;
    res.mu = tm; res.var = tv;
    return res;
}

// log weight up to the constant -log(sqrt(2 pi)); a degenerate Normal(mu, 0)
// gives 0 at o == mu and -inf elsewhere.
__device__ __forceinline__ double log_weight(double mu, double var, double od)
{
    if (!(var > 0.0)) return (od == mu) ? 0.0 : -__builtin_inf();
    double dlt = od - mu;
    return -0.5 * (dlt * dlt / var) - 0.5 * s_log(var);
}

__device__ __forceinline__ double reward_of(const epi_params& P, int32_t sum_inf, double a, double a_prev)
{
    double inf_loss = P.inf_loss * (double)sum_inf / (double)P.N;
    double test_loss = P.test_loss * a;
    double rate_loss = P.testrate_loss * __builtin_fabs(a - a_prev);
    return -(inf_loss + test_loss + rate_loss);
}

} // namespace epi
