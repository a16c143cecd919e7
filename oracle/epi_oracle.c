/*
 * epi_oracle.c -- CPU restatement of the InfectionSim particle step and the
 * bootstrap particle-filter update.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker the HIP path is compared against.  It is never
 * linked into, imported by, or called from the shipped library or its Python
 * host package; only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py load it.
 *
 * PARITY UNPINNED: the reference (WhiffleFish/EpidemicModeling) is Julia and
 * cannot be executed where this was written, it ships no golden vectors, and
 * its only runnable test (test/InfectionSimTest.jl:14-19) asserts a bounds
 * property.  The deterministic bookkeeping below follows the reference source
 * line by line and is additionally cross-checked by an independent numpy
 * transliteration (oracle/step_numpy.py); random variates come from the
 * sampler specification written out below (this build's own), checked
 * against closed-form Binomial / Gamma / Normal distributions.
 *
 * Reference lines followed (paths relative to the reference checkout):
 *   src/simulate.jl:37-45    incident_infections      -> incident_infections()
 *   src/simulate.jl:53-61    SymptomaticIsolation     -> step_core() draw 1
 *   src/simulate.jl:73-82    PositiveTests            -> step_core() draw 2
 *   src/simulate.jl:99-129   UpdateIsolations         -> step_core()
 *   src/simulate.jl:137-151  SimStep                  -> step_core()
 *   src/simulate.jl:153-158  reward                   -> reward()
 *   src/simulate.jl:160-166  continuous_gen           -> orc_step_batch()
 *   src/init.jl:71-78        RVsum(::Gamma, N)        -> gamma_draw()
 *   src/POMDPsInterface.jl:11-23 observation          -> obs_moments()
 *   src/updater.jl:5-21      mean                     -> orc_belief_mean()
 *   src/typedef.jl:159-217   State(...) initialisers  -> orc_init_*()
 *   ParticleFilters.jl (external, unpinned; call sites
 *   src/POMDPsInterface.jl:39,56) BootstrapFilter update with the low
 *   variance resampler                                -> orc_pf_*()
 *
 * Host state layout (what a Julia Vector{State} flattens to):
 *   S[n], R[n]            int64
 *   I[n*H]                int64, particle-major (particle p, age i at p*H+i)
 *   T[n*(D+1)*H]          int64, per particle a column-major (D+1) x H matrix
 *                         (row r, column c at p*(D+1)*H + c*(D+1) + r)
 *
 * Floating point: every operation is an IEEE-754 binary64 / binary32
 * operation in the order written; build with -ffp-contract=off.  fma() is
 * used only where written.  No libm transcendental is called on any path
 * that decides an integer result.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_H 64

/* ------------------------------------------------------------------ */
/* parameters                                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t H;          /* infection-age bins (horizon)                  */
    int32_t D;          /* test_delay                                    */
    int64_t N;          /* population                                    */
    double inf_loss, test_loss, testrate_loss;
    double q[ORC_MAX_H];     /* symptomatic isolation probability per age */
    double p[ORC_MAX_H];     /* positive-test probability per age         */
    double k[ORC_MAX_H];     /* Gamma shape per infected individual       */
    double theta[ORC_MAX_H]; /* Gamma scale                               */
} orc_params;

/* error flag bits, OR-ed into *flags */
#define ORC_F_NEG_I      1u   /* I_i went negative (simulate.jl:117 assert)   */
#define ORC_F_NEG_NT     2u   /* tested count negative                        */
#define ORC_F_BAD_DENOM  4u   /* N - R <= 0 in incident_infections            */
#define ORC_F_ZERO_W     8u   /* all weights zero in the filter update        */
#define ORC_F_COARSE_W 16u

/* ------------------------------------------------------------------ */
/* bit casts                                                           */
/* ------------------------------------------------------------------ */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* ------------------------------------------------------------------ */
/* Philox4x32 (Salmon et al., SC'11), counter c[4], key k[2].  The    */
/* sampler streams use 7 rounds (the authors' smallest count that      */
/* passes BigCrush); the 10-round form exists for the published        */
/* known-answer vectors.                                               */
/* ------------------------------------------------------------------ */
#define PHILOX_ROUNDS 7
static void philox4x32(const uint32_t c[4], const uint32_t k[2], int rounds, uint32_t out[4])
{
    uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], k0 = k[0], k1 = k[1];
    for (int r = 0; r < rounds; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox4x32(const uint32_t c[4], const uint32_t k[2], int rounds, uint32_t out[4])
{
    philox4x32(c, k, rounds, out);
}

/* Stream addressing.  One Philox block is addressed by
 *   key     = (seed lo, seed hi)
 *   counter = (particle lo, particle hi, step, tag)
 *   tag     = kind << 24 | bin << 8 | blk
 * kinds: */
#define K_BINOM0  0u  /* words 0,1: sympt attempt 0 ; words 2,3: pos attempt 0      */
#define K_SYMPT   1u  /* blk j: sympt attempts 1+2j (words 0,1) and 2+2j (words 2,3) */
#define K_POS     2u  /* same for pos                                              */
#define K_GAMMA   3u  /* bin = pair j: attempt 0 of infection term 2j (words 0, 2), of term 2j+1 (words 1, 3) */
#define K_GRETRY  7u  /* bin = term, blk b: its attempts 1+2b (words 0, 2) and 2+2b (words 1, 3) */
#define K_BOOST   4u  /* word 0: U for the shape<1 boost                            */
#define K_RESAMP  5u  /* particle = 0: words 0,1 -> 53-bit systematic offset        */
#define K_INIT    6u  /* initial-state generators                                  */

typedef struct { uint32_t key[2]; uint32_t c0, c1, c2; } rng_t;

static inline rng_t rng_make(uint64_t seed, uint64_t particle, uint32_t step)
{
    rng_t g;
    g.key[0] = (uint32_t)seed; g.key[1] = (uint32_t)(seed >> 32);
    g.c0 = (uint32_t)particle; g.c1 = (uint32_t)(particle >> 32); g.c2 = step;
    return g;
}

static inline void rng_block(const rng_t *g, uint32_t kind, uint32_t bin, uint32_t blk, uint32_t w[4])
{
    uint32_t c[4] = { g->c0, g->c1, g->c2, (kind << 24) | (bin << 8) | blk };
    philox4x32(c, g->key, PHILOX_ROUNDS, w);
}

/* uniform in (0,1) from one word: (w + 1/2) 2^-32, exact in binary64 */
static inline double u01(uint32_t w) { return ((double)w + 0.5) * 0x1p-32; }

/* ------------------------------------------------------------------ */
/* elementary functions (specification: the operation sequence below)  */
/* ------------------------------------------------------------------ */
static double s_log(double x) /* x > 0, finite, normal */
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ix = d2u(x);
    int e = (int)(ix >> 52) - 1023;
    double m = u2d((ix & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double hfsq = (0.5 * f) * f;
    double dk = (double)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

static double s_exp(double x) /* x <= ~700; returns 0 below -708 */
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    if (!(x >= -708.0)) return 0.0;          /* also NaN, -inf */
    if (x > 709.0) x = 709.0;
    double kd = floor(fma(x, 1.4426950408889634, 0.5));
    double r = fma(-kd, ln2_hi, x);
    r = fma(-kd, ln2_lo, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kd;
    return p * u2d((uint64_t)(k + 1023) << 52);
}

static float s_logf(float x) /* x in (0, 1], normal */
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fmaf(w, Lg4, Lg2);
    float t2 = z * fmaf(w, Lg3, Lg1);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    float dk = (float)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* Box-Muller in binary32: two words -> two standard normals. */
static void box_muller(uint32_t w0, uint32_t w1, float *z0, float *z1)
{
    float u1 = fmaf((float)w0, 0x1p-32f, 0x1p-33f);          /* (0, 1] */
    float r = sqrtf(-2.0f * s_logf(u1));
    uint32_t quad = w1 >> 30;
    int32_t ti = (int32_t)(w1 & 0x3fffffffu) - (1 << 29);
    float y = ((float)ti + 0.5f) * 0x1p-30f;                 /* [-1/2, 1/2] */
    float ph = y * 1.57079637f;
    float x2 = ph * ph;
    float sp = ph * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-06f, -0.0001984127f), 0.008333334f), -0.16666667f), 1.0f);
    float cp = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -2.755732e-07f, 2.4801588e-05f), -0.0013888889f), 0.041666668f), -0.5f), 1.0f);
    float h = r * 0.70710677f;
    float a = cp * h, b = sp * h;
    float c = a - b, s = a + b;       /* r cos(pi/4+ph), r sin(pi/4+ph) */
    switch (quad) {
    case 0: *z0 = c;  *z1 = s;  break;
    case 1: *z0 = -s; *z1 = c;  break;
    case 2: *z0 = -c; *z1 = -s; break;
    default: *z0 = s; *z1 = -c; break;
    }
}

double orc_log(double x) { return s_log(x); }
double orc_exp(double x) { return s_exp(x); }
float orc_logf(float x) { return s_logf(x); }
void orc_box_muller(uint32_t w0, uint32_t w1, float *z0, float *z1) { box_muller(w0, w1, z0, z1); }

/* ------------------------------------------------------------------ */
/* Binomial(n, p)                                                      */
/*   n*min(p,1-p) < 10 : sequential inversion from 0 (one uniform)     */
/*   otherwise         : transformed rejection with squeeze            */
/*                       (Hoermann 1993, algorithm BTRS)               */
/* attempt a of a draw takes the word pair pair_fn(a).                 */
/* ------------------------------------------------------------------ */
static const double fc_tab[10] = {
    0.08106146679532726, 0.04134069595540929, 0.02767792568499834,
    0.02079067210376509, 0.01664469118982119, 0.01387612882307075,
    0.01189670994589177, 0.01041126526197209, 0.009255462182712733,
    0.008330563433362871 };

static inline double stirling_tail(double k)
{
    if (k <= 9.0) return fc_tab[(int)k];
    double r = 1.0 / (k + 1.0);
    double r2 = r * r;
    return (0.08333333333333333 - (0.002777777777777778 - 0.0007936507936507937 * r2) * r2) * r;
}

/* log(k!) for an integer k >= 0 held in a double: Stirling's formula with the tail above, left to right */
static inline double lfact(double k)
{
    return ((k + 0.5) * s_log(k + 1.0) - (k + 1.0)) + 0.91893853320467274178 + stirling_tail(k);
}

typedef struct {
    const rng_t *g;
    uint32_t bin;
    uint32_t retry_kind;   /* K_SYMPT or K_POS */
    uint32_t first[2];     /* attempt-0 word pair (from the K_BINOM0 block) */
    uint32_t cache_blk;    /* cached retry block index, 0xffffffff = none */
    uint32_t cache[4];
} binom_src;

static void binom_pair(binom_src *s, uint32_t attempt, uint32_t *wa, uint32_t *wb)
{
    if (attempt == 0) { *wa = s->first[0]; *wb = s->first[1]; return; }
    uint32_t blk = (attempt - 1) >> 1, half = (attempt - 1) & 1;
    if (s->cache_blk != blk) { rng_block(s->g, s->retry_kind, s->bin, blk, s->cache); s->cache_blk = blk; }
    *wa = s->cache[2 * half]; *wb = s->cache[2 * half + 1];
}

#define BINOM_MAX_ATTEMPTS 512u

static float recip_tab[65];
__attribute__((constructor)) static void recip_init(void)
{
    for (int x = 1; x <= 64; x++) recip_tab[x] = 1.0f / (float)x;
}

/* exp(x) in binary32, -88 < x <= 0 */
static float s_expf(float x)
{
    float kd = floorf(fmaf(x, 1.44269504f, 0.5f));
    float r = fmaf(-kd, 6.93145752e-01f, x);
    r = fmaf(-kd, 1.42860677e-06f, r);
    float p = 1.38888889e-03f;
    p = fmaf(p, r, 8.33333333e-03f);
    p = fmaf(p, r, 4.16666667e-02f);
    p = fmaf(p, r, 1.66666667e-01f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return u2f(f2u(p) + ((uint32_t)(int32_t)kd << 23));
}

float orc_expf(float x) { return s_expf(x); }

static int64_t binomial_draw(int64_t n, double p, binom_src *src)
{
    if (n <= 0 || !(p > 0.0)) return 0;
    if (p >= 1.0) return n;
    int flip = p > 0.5;
    double pp = flip ? 1.0 - p : p;
    double qq = 1.0 - pp;
    double nd = (double)n;
    int64_t x;
    if (nd * pp < 10.0) {
        uint32_t wa, wb;
        binom_pair(src, 0, &wa, &wb);
        /* sequential inversion in binary32: pmf(x) = pmf(x-1) s (n+1-x)/x, with 1/x tabulated up to 64; the search
         * also ends where the terms have fallen below half an ulp of what is left of u (no progress possible) */
        float nf = (float)(int32_t)n, sf = (float)(pp / qq);
        float a1 = nf + 1.0f;
        float r = s_expf(nf * (float)s_log(qq));
        float u = fmaf((float)(wa >> 8), 0x1p-24f, 0x1p-25f);
        x = 0;
        while (u > r && x < n) {
            u -= r;
            x += 1;
            float xf = (float)(int32_t)x;
            float t = x <= 64 ? recip_tab[x] : 1.0f / xf;
            r *= sf * ((a1 - xf) * t);
            if (r < u * 0x1p-25f) break;
        }
    } else {
        /* Hat in binary32 (Hoermann's constants carry 3 digits); the centre c = n pp + 1/2 and the mode
         * m = floor((n + 1) pp) exactly, from pp in 0.32 fixed point; the fraction of c to 23 bits. */
        uint32_t ppx = (uint32_t)(pp * 4294967296.0 + 0.5);
        float pqf = (float)(pp * qq);
        float a0x2 = (float)(2.0 * (-0.0873 + 0.01 * pp));
        double lpq = s_log(pp / qq);
        float spq = sqrtf((float)(int32_t)n * pqf);
        float b = fmaf(2.53f, spq, 1.15f);
        float a2 = fmaf(0.0496f, b, a0x2);                       /* 2 a */
        uint64_t prod = (uint64_t)(uint32_t)n * ppx + 0x80000000ULL;
        int64_t ci = (int64_t)(prod >> 32);
        float cfr = u2f(0x3f800000u | ((uint32_t)prod >> 9)) - 1.0f;
        int64_t mi = (int64_t)(((uint64_t)((uint32_t)n + 1u) * ppx) >> 32);
        double m = (double)mi;
        x = mi;                                 /* returned only if every attempt fails */
        for (uint32_t att = 0; att < BINOM_MAX_ATTEMPTS; att++) {
            uint32_t wa, wb;
            binom_pair(src, att, &wa, &wb);
            float u = (float)(int32_t)wa * 0x1p-32f;             /* [-1/2, 1/2] */
            float us = fmaxf(0.5f - fabsf(u), 0x1p-25f);
            float v = (float)wb * 0x1p-32f;                      /* [0, 1] */
            double xf = floor(((double)a2 / (double)us + (double)b) * (double)u + (double)cfr);
            if (xf < -1073741824.0) xf = -1073741824.0;
            if (xf > 1073741824.0) xf = 1073741824.0;
            int64_t k = ci + (int64_t)xf;
            if (k < 0 || k > n) continue;
            if (us >= 0.07f && (0.92f - v) * b >= 4.2f) { x = k; break; }
            if (!(v > 0.0f)) { x = k; break; }
            /* log(v alpha / (a/us^2 + b)) <= log f(k) - log f(m), the pmf ratio from log factorials */
            double alpha = (2.83 + 5.1 / (double)b) * (double)spq;
            double lv = s_log((double)v * alpha / (0.5 * (double)a2 / ((double)us * (double)us) + (double)b));
            double kd = (double)k;
            double ub = ((lfact(m) + lfact(nd - m)) - (lfact(kd) + lfact(nd - kd))) + (kd - m) * lpq;
            if (lv <= ub) { x = k; break; }
        }
    }
    return flip ? n - x : x;
}

/* ------------------------------------------------------------------ */
/* Gamma(shape, scale): Marsaglia-Tsang (2000); shape < 1 boosted by   */
/* U^(1/shape).  shape < 2^-30 returns exactly 0 (RVsum's k*N <= 0     */
/* branch, init.jl:73-77, and shapes whose samples underflow).         */
/* ------------------------------------------------------------------ */
#define GAMMA_MAX_ATTEMPTS 512u
#define GAMMA_MIN_SHAPE 0x1p-30

/* Standard normal by the ziggurat of Marsaglia & Tsang (2000) with 256 strips of equal area v: strip k >= 1 is
 * [0, x_k] x [f(x_k), f(x_{k+1})], x_1 = r, x_256 = 0, f(x) = exp(-x^2/2); strip 0 is the base [0, r] x [0, f(r)]
 * plus the tail, nominal width v / f(r).  A 32-bit word gives strip (bits 0-7), sign (bit 8), position j (23 bits). */
#define ZIG_N 256
#define K_ZIG 8u
#define ZIG_MAX_STEPS 64u
static uint32_t zig_kn[ZIG_N];
static float zig_wn[ZIG_N];
static double zig_fn[ZIG_N + 1];
static const double zig_r = 3.6541528853610088, zig_v = 4.92867323399e-3;

__attribute__((constructor)) static void zig_init(void)
{
    double x[ZIG_N + 1];
    x[1] = zig_r;
    for (int k = 1; k < ZIG_N - 1; k++) x[k + 1] = sqrt(-2.0 * s_log(zig_v / x[k] + s_exp(-0.5 * x[k] * x[k])));
    x[ZIG_N] = 0.0;
    zig_fn[0] = 0.0;
    for (int k = 1; k <= ZIG_N; k++) zig_fn[k] = k == ZIG_N ? 1.0 : s_exp(-0.5 * x[k] * x[k]);
    for (int k = 0; k < ZIG_N; k++) {
        double X = k == 0 ? zig_v / zig_fn[1] : x[k];
        zig_wn[k] = (float)(X * 0x1p-23);
        zig_kn[k] = (uint32_t)((x[k + 1] / (double)zig_wn[k]) * (1.0 - 0x1p-22));
    }
}

void orc_zig_table(uint32_t *kn, float *wn, double *fn)
{
    memcpy(kn, zig_kn, sizeof zig_kn); memcpy(wn, zig_wn, sizeof zig_wn); memcpy(fn, zig_fn, sizeof zig_fn);
}

/* the normal of attempt `att` of infection term `term` from its candidate word w; the wedges and the tail take
 * further words from the blocks 0, 1, ... of the stream (K_ZIG, att << 6 | term) */
static float zig_normal(const rng_t *g, uint32_t term, uint32_t att, uint32_t w)
{
    int tail = 0;
    uint32_t sign = 0;
    float x = 0.f;
    for (uint32_t l = 0; l < ZIG_MAX_STEPS; l++) {
        if (!tail) {
            uint32_t k = w & 255u, j = w >> 9;
            sign = (w & 256u) << 23;
            x = (float)j * zig_wn[k];
            if (j < zig_kn[k]) break;                      /* inside the strip's rectangle under the curve */
            if (k == 0) tail = 1;
            else {
                uint32_t q[4];
                rng_block(g, K_ZIG, (att << 6) | term, l, q);
                double xd = (double)x;
                double y = zig_fn[k] + u01(q[0]) * (zig_fn[k + 1] - zig_fn[k]);
                if (y < s_exp(-0.5 * xd * xd)) break;
                w = q[2];
                continue;
            }
        }
        uint32_t q[4];
        rng_block(g, K_ZIG, (att << 6) | term, l, q);
        double xt = -s_log(u01(q[0])) / zig_r, yt = -s_log(u01(q[1]));
        if (yt + yt > xt * xt) { x = (float)(zig_r + xt); break; }
    }
    return u2f(f2u(x) | sign);
}

float orc_zig_normal(uint64_t seed, uint64_t particle, uint32_t step, uint32_t term, uint32_t att, uint32_t w)
{
    rng_t g = rng_make(seed, particle, step);
    return zig_normal(&g, term, att, w);
}

/* c >= 1/sqrt(9 d), 1 <= 9 d c^2 <= 1 + 1.4e-5, in binary32: two Newton steps for the reciprocal square root   */
/* from the bit-shift seed (the iterates stay below the root) and a factor 1 + 6e-6.  The rejection is exact for */
/* every such c: the normal envelope dominates d (log v - v + 1) as soon as 9 d c^2 >= 1, and d v is             */
/* Gamma(d + 1/3) whatever c; the squeeze below carries the allowance for the excess.                            */
static inline float gamma_c(float d)
{
    const float x9 = 9.0f * d, h = 4.5f * d;
    float y = u2f(0x5f3759dfu - (f2u(x9) >> 1));
    y = y * (1.5f - h * (y * y));
    y = y * (1.5f - h * (y * y));
    return y * 1.000006f;
}

float orc_gamma_c(float d) { return gamma_c(d); }

/* Lower bound 1 - A of the acceptance probability exp(x^2/2 + d (1 - v + log v)) for any c with 9 d c^2 = 1 + 2 e,
 * 0 <= e <= 7e-6: the exponent is -e x^2 - 3 d int_0^s r^3/(1+r) dr with s = c x, and the integral is at most
 * (s^4/4)(1 + 2|s|) for |s| <= 1/2, so A = 7e-6 x^2 + (c^2 x^4 / 12)(1 + 1.4e-5)(1 + 2|s|); Marsaglia and Tsang's
 * 0.0331 x^4 (with the allowance for e) where that is the smaller. */
static inline float gamma_squeeze(float c, float xf)
{
    float x2 = xf * xf, x4 = x2 * x2;
    float as = fabsf(c * xf);
    float a_mt = fmaf(0.03311f, x4, 1.4e-5f * x2);
    float a_d = fmaf(((c * c) * x4) * 0.08334f, fmaf(2.0f, as, 1.0f), 7e-6f * x2);
    float a = (as <= 0.5f && a_d < a_mt) ? a_d : a_mt;
    return 0.9999997f - a;
}

float orc_gamma_squeeze(float c, float x) { return gamma_squeeze(c, x); }

static double gamma_draw(double shape, double scale, const rng_t *g, uint32_t bin)
{
    if (!(shape >= GAMMA_MIN_SHAPE)) return 0.0;
    int boost = shape < 1.0;
    double al = boost ? shape + 1.0 : shape;
    double d = al - 0.3333333333333333;
    float c = gamma_c((float)d);
    double v = 1.0;
    uint32_t w[4];
    for (uint32_t att = 0; att < GAMMA_MAX_ATTEMPTS; att++) {
        /* attempt 0 shares the block of the pair bin/2 (half bin&1); attempt a >= 1 takes half (a-1)&1 of
         * block (a-1)/2 of the term's own retry stream; word `half` -> the normal, word 2 + half -> the uniform */
        uint32_t half = att == 0 ? (bin & 1u) : ((att - 1u) & 1u);
        if (att == 0) rng_block(g, K_GAMMA, bin >> 1, 0, w);
        else rng_block(g, K_GRETRY, bin, (att - 1u) >> 1, w);
        float xf = zig_normal(g, bin, att, w[half]);
        float uf = (float)w[2 + half] * 0x1p-32f;
        double x = (double)xf;
        double t = fma((double)c, x, 1.0);
        if (!(t > 0.0)) continue;
        double vv = t * t * t;
        if (uf < gamma_squeeze(c, xf)) { v = vv; break; }
        if (!(uf > 0.0f)) { v = vv; break; }
        if (s_log((double)uf) < 0.5 * (x * x) + d * (1.0 - vv + s_log(vv))) { v = vv; break; }
    }
    double gval = d * v;
    if (boost) {
        uint32_t wb[4];
        rng_block(g, K_BOOST, bin, 0, wb);
        gval = gval * s_exp(s_log(u01(wb[0])) / shape);
    }
    return gval * scale;
}

/* single-draw entry points for distribution tests */
int64_t orc_binomial(uint64_t seed, uint64_t particle, uint32_t step, uint32_t bin, int which, int64_t n, double p)
{
    rng_t g = rng_make(seed, particle, step);
    uint32_t w[4];
    rng_block(&g, K_BINOM0, bin, 0, w);
    binom_src s = { &g, bin, which ? K_POS : K_SYMPT, { w[which ? 2 : 0], w[which ? 3 : 1] }, 0xffffffffu, {0, 0, 0, 0} };
    return binomial_draw(n, p, &s);
}

double orc_gamma(uint64_t seed, uint64_t particle, uint32_t step, uint32_t bin, double shape, double scale)
{
    rng_t g = rng_make(seed, particle, step);
    return gamma_draw(shape, scale, &g, bin);
}

void orc_binomial_many(uint64_t seed, uint64_t first, int64_t count, uint32_t step, uint32_t bin, int which,
                       int64_t n, double p, int64_t *out)
{
    for (int64_t i = 0; i < count; i++) out[i] = orc_binomial(seed, first + (uint64_t)i, step, bin, which, n, p);
}

void orc_gamma_many(uint64_t seed, uint64_t first, int64_t count, uint32_t step, uint32_t bin,
                    double shape, double scale, double *out)
{
    for (int64_t i = 0; i < count; i++) out[i] = orc_gamma(seed, first + (uint64_t)i, step, bin, shape, scale);
}

/* ------------------------------------------------------------------ */
/* one SimStep on one particle (simulate.jl:99-151).                   */
/* Variates either injected (sympt/pos/gam non-NULL) or drawn.         */
/* Value semantics: inputs are not modified.                           */
/* ------------------------------------------------------------------ */
typedef struct {
    int64_t S, R;
    int64_t I[ORC_MAX_H];
    int64_t T[(8) * ORC_MAX_H];     /* (D+1) x H column-major, D <= 7 */
} pstate;

#define ORC_MAX_D 7

static inline double dfloor_to_i(double x) { return floor(x); }

static void step_core(const orc_params *P, const pstate *s, double a,
                      const int64_t *inj_sympt, const int64_t *inj_pos, const double *inj_gam,
                      const rng_t *g,
                      pstate *sp, int64_t *new_inf, int64_t *pos_out, uint32_t *flags)
{
    const int H = P->H, D = P->D, D1 = P->D + 1;
    int64_t I[ORC_MAX_H], sympt[ORC_MAX_H], pos[ORC_MAX_H];
    int64_t T[(ORC_MAX_D + 1) * ORC_MAX_H];
    double sprop[ORC_MAX_H];
    int64_t R = s->R, S = s->S;
    memcpy(I, s->I, sizeof(int64_t) * (size_t)H);
    memcpy(T, s->T, sizeof(int64_t) * (size_t)(D1 * H));

    /* SymptomaticIsolation (simulate.jl:53-61) and PositiveTests (:73-82):
     * both read I before any update. */
    for (int i = 0; i < H; i++) {
        int64_t already = 0;
        for (int r = 0; r < D1; r++) already += T[i * D1 + r];
        double ntf = floor((double)(I[i] - already) * a);          /* simulate.jl:78 */
        int64_t nt = (int64_t)ntf;
        if (nt < 0) *flags |= ORC_F_NEG_NT;
        if (inj_sympt) {
            sympt[i] = inj_sympt[i];
            pos[i] = inj_pos[i];
        } else {
            uint32_t w[4];
            rng_block(g, K_BINOM0, (uint32_t)i, 0, w);
            binom_src bs = { g, (uint32_t)i, K_SYMPT, { w[0], w[1] }, 0xffffffffu, {0, 0, 0, 0} };
            sympt[i] = binomial_draw(I[i], P->q[i], &bs);          /* simulate.jl:58 */
            binom_src bp = { g, (uint32_t)i, K_POS, { w[2], w[3] }, 0xffffffffu, {0, 0, 0, 0} };
            pos[i] = binomial_draw(nt, P->p[i], &bp);              /* simulate.jl:79 */
        }
    }
    /* UpdateIsolations (simulate.jl:104-126) */
    for (int i = 0; i < H; i++) {
        double pr = (double)sympt[i] / (double)I[i];               /* :104 */
        if (pr != pr) pr = 0.0;                                    /* :105 NaN => 0 */
        sprop[i] = pr;
    }
    for (int i = 0; i < H; i++) { R += sympt[i]; I[i] -= sympt[i]; }   /* :107-108 */
    for (int i = 0; i < H; i++) T[i * D1 + D] = pos[i];                /* :110 */
    for (int i = 0; i < H; i++)
        for (int r = 0; r < D1; r++)
            T[i * D1 + r] = (int64_t)floor((1.0 - sprop[i]) * (double)T[i * D1 + r]);   /* :112 */
    for (int i = 0; i < H; i++) { R += T[i * D1 + 0]; I[i] -= T[i * D1 + 0]; }         /* :114-115 */
    for (int i = 0; i < H; i++) if (I[i] < 0) *flags |= ORC_F_NEG_I;                   /* :117 */
    /* circshift(tests, (-1, 1)) then clear column 1 and the last row (:122-126) */
    {
        int64_t T2[(ORC_MAX_D + 1) * ORC_MAX_H];
        for (int c = 0; c < H; c++)
            for (int r = 0; r < D1; r++) {
                int sr = (r + 1) % D1, sc = (c - 1 + H) % H;
                T2[c * D1 + r] = T[sc * D1 + sr];
            }
        for (int r = 0; r < D1; r++) T2[0 * D1 + r] = 0;
        for (int c = 0; c < H; c++) T2[c * D1 + D] = 0;
        memcpy(T, T2, sizeof(int64_t) * (size_t)(D1 * H));
    }
    /* SimStep tail (simulate.jl:144-149) */
    R += I[H - 1];
    {
        int64_t last = I[H - 1];
        for (int i = H - 1; i > 0; i--) I[i] = I[i - 1];
        I[0] = last;
    }
    /* incident_infections (simulate.jl:37-45) over the shifted I */
    double infsum = 0.0;
    for (int i = 0; i < H; i++) {
        double gi;
        if (inj_gam) gi = inj_gam[i];
        else gi = gamma_draw(P->k[i] * (double)I[i], P->theta[i], g, (uint32_t)i);   /* init.jl:71-78 */
        infsum = (i == 0) ? gi : infsum + gi;     /* Int 0 + Float64 is exact */
    }
    int64_t denom = P->N - R;
    int64_t newi;
    if (denom <= 0) { *flags |= ORC_F_BAD_DENOM; newi = 0; }
    else {
        double sprop_s = (double)S / (double)denom;                /* :42 */
        newi = (int64_t)floor(sprop_s * infsum);                   /* :44 */
    }
    I[0] = newi;                                                    /* :147 */
    S -= newi;                                                      /* :148 */
    sp->S = S; sp->R = R;
    memcpy(sp->I, I, sizeof(int64_t) * (size_t)H);
    memcpy(sp->T, T, sizeof(int64_t) * (size_t)(D1 * H));
    *new_inf = newi;
    for (int i = 0; i < H; i++) pos_out[i] = pos[i];
}

/* reward (simulate.jl:153-158) */
static double reward(const orc_params *P, const pstate *sp, double a, double a_prev)
{
    int64_t sumI = 0;
    for (int i = 0; i < P->H; i++) sumI += sp->I[i];
    double inf_loss = P->inf_loss * (double)sumI / (double)P->N;
    double test_loss = P->test_loss * a;
    double rate_loss = P->testrate_loss * fabs(a - a_prev);
    return -(inf_loss + test_loss + rate_loss);
}

/* observation model moments from the PRE-step state (POMDPsInterface.jl:11-23,
 * with `state` read as `s`) */
static void obs_moments(const orc_params *P, const pstate *s, double a, double *mu, double *var)
{
    const int H = P->H, D1 = P->D + 1;
    double tm = 0.0, tv = 0.0;
    for (int i = 0; i < H; i++) {
        int64_t already = 0;
        for (int r = 0; r < D1; r++) already += s->T[i * D1 + r];
        double nt = floor((double)(s->I[i] - already) * a);
        double pi = P->p[i];
        tm += nt * pi;                       /* mean(Binomial(n,p)) */
        double sd = sqrt(nt * pi * (1.0 - pi));   /* std(Binomial) */
        tv += sd * sd;                       /* std(dist)^2 */
    }
    *mu = tm; *var = tv;
}

/* log weight up to the additive constant -log(sqrt(2 pi)):
 *   var > 0 : -(o-mu)^2/(2 var) - log(var)/2
 *   var == 0: 0 if o == mu else -inf   (a degenerate Normal(mu,0)) */
static double log_weight(const orc_params *P, const pstate *s, double a, int64_t o)
{
    double mu, var;
    obs_moments(P, s, a, &mu, &var);
    double od = (double)o;
    if (!(var > 0.0)) return (od == mu) ? 0.0 : -INFINITY;
    double dlt = od - mu;
    return -0.5 * (dlt * dlt / var) - 0.5 * s_log(var);
}

/* ------------------------------------------------------------------ */
/* host-layout load / store                                            */
/* ------------------------------------------------------------------ */
static void load_state(const orc_params *P, int64_t p, const int64_t *S, const int64_t *I, const int64_t *R,
                       const int64_t *T, pstate *s)
{
    const int H = P->H, D1 = P->D + 1;
    s->S = S[p]; s->R = R[p];
    memcpy(s->I, I + p * H, sizeof(int64_t) * (size_t)H);
    if (T) memcpy(s->T, T + p * D1 * H, sizeof(int64_t) * (size_t)(D1 * H));
    else memset(s->T, 0, sizeof(int64_t) * (size_t)(D1 * H));
}

static void store_state(const orc_params *P, int64_t p, const pstate *s, int64_t *S, int64_t *I, int64_t *R, int64_t *T)
{
    const int H = P->H, D1 = P->D + 1;
    S[p] = s->S; R[p] = s->R;
    memcpy(I + p * H, s->I, sizeof(int64_t) * (size_t)H);
    if (T) memcpy(T + p * D1 * H, s->T, sizeof(int64_t) * (size_t)(D1 * H));
}

/* ------------------------------------------------------------------ */
/* batch entry points                                                  */
/* ------------------------------------------------------------------ */

/* SimStep with supplied variates.  sympt,pos: int64[n*H]; gam: double[n*H]. */
uint32_t orc_step_injected(const orc_params *P, int64_t n,
                           const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T,
                           const double *a, int a_per_particle,
                           const int64_t *sympt, const int64_t *pos, const double *gam,
                           int64_t *S2, int64_t *I2, int64_t *R2, int64_t *T2,
                           int64_t *new_inf, int64_t *pos_out)
{
    uint32_t flags = 0;
    const int H = P->H;
    for (int64_t p = 0; p < n; p++) {
        pstate s, sp;
        int64_t ni, po[ORC_MAX_H];
        load_state(P, p, S, I, R, T, &s);
        step_core(P, &s, a_per_particle ? a[p] : a[0], sympt + p * H, pos + p * H, gam + p * H, NULL, &sp, &ni, po, &flags);
        store_state(P, p, &sp, S2, I2, R2, T2);
        if (new_inf) new_inf[p] = ni;
        if (pos_out) memcpy(pos_out + p * H, po, sizeof(int64_t) * (size_t)H);
    }
    return flags;
}

/* continuous_gen over a batch (simulate.jl:160-166): sp, o = sum(pos), r.
 * Particle p draws from the stream of global index particle0 + p. */
uint32_t orc_step_batch(const orc_params *P, int64_t n,
                        const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T,
                        const double *a, int a_per_particle, const double *a_prev, int a_prev_per_particle,
                        uint64_t seed, uint32_t step, uint64_t particle0,
                        int64_t *S2, int64_t *I2, int64_t *R2, int64_t *T2,
                        int64_t *o_out, double *r_out, int64_t *new_inf_out)
{
    uint32_t flags = 0;
    #pragma omp parallel for schedule(static) reduction(|:flags)
    for (int64_t p = 0; p < n; p++) {
        pstate s, sp;
        int64_t ni, po[ORC_MAX_H];
        uint32_t fl = 0;
        load_state(P, p, S, I, R, T, &s);
        rng_t g = rng_make(seed, particle0 + (uint64_t)p, step);
        double ap = a_per_particle ? a[p] : a[0];
        step_core(P, &s, ap, NULL, NULL, NULL, &g, &sp, &ni, po, &fl);
        store_state(P, p, &sp, S2, I2, R2, T2);
        if (o_out) { int64_t o = 0; for (int i = 0; i < P->H; i++) o += po[i]; o_out[p] = o; }
        if (r_out) r_out[p] = reward(P, &sp, ap, a_prev ? (a_prev_per_particle ? a_prev[p] : a_prev[0]) : 0.0);
        if (new_inf_out) new_inf_out[p] = ni;
        flags |= fl;
    }
    return flags;
}

/* observation log-weights of a batch of pre-step states */
void orc_log_weights(const orc_params *P, int64_t n,
                     const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T,
                     double a, int64_t o, double *lw, double *mu_out, double *var_out)
{
    #pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; p++) {
        pstate s;
        load_state(P, p, S, I, R, T, &s);
        lw[p] = log_weight(P, &s, a, o);
        if (mu_out) obs_moments(P, &s, a, mu_out + p, var_out + p);
    }
}

/* ------------------------------------------------------------------ */
/* systematic ("low variance") resampling in fixed point               */
/*   F      = 53 - ceil(log2(n_total))  (so that n_total * 2^F <= 2^53) */
/*   wq_i   = floor(exp(lw_i - lwmax) * 2^F)        (uint64)           */
/*   C_i    = inclusive prefix sum of wq             (exact, any order) */
/*   step   = fl(W / n_total),  W = C_{n-1}                            */
/*   u      = 53-bit uniform in [0,1) from block (K_RESAMP)            */
/*   T_m    = fl(fl(m + u) * step)                                     */
/*   anc(m) = #{ i : C_i <= floor(T_m) }, clamped to n-1               */
/* ------------------------------------------------------------------ */
int orc_frac_bits(int64_t n_total)
{
    int lg = 0;
    while (((int64_t)1 << lg) < n_total) lg++;
    int F = 53 - lg;
    if (F > 40) F = 40;
    return F;
}

double orc_resample_u(uint64_t seed, uint32_t step)
{
    rng_t g = rng_make(seed, 0, step);
    uint32_t w[4];
    rng_block(&g, K_RESAMP, 0, 0, w);
    uint64_t hi = w[0] >> 5, lo = w[1] >> 6;            /* 27 + 26 bits */
    return (double)((hi << 26) | lo) * 0x1p-53;
}

void orc_quantize(int64_t n, const double *lw, double lwmax, int F, uint64_t *wq)
{
    double scale = u2d((uint64_t)(F + 1023) << 52);
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) wq[i] = (uint64_t)floor(s_exp(lw[i] - lwmax) * scale);
}

/* inclusive prefix sum with a starting offset; returns the last value */
uint64_t orc_cumsum(int64_t n, const uint64_t *wq, uint64_t offset, uint64_t *C)
{
    uint64_t acc = offset;
    for (int64_t i = 0; i < n; i++) { acc += wq[i]; C[i] = acc; }
    return acc;
}

/* ancestors of global output slots [m0, m0+count) against the GLOBAL cumulative
 * array C[0..n_total) ; anc values are global indices */
void orc_ancestors(int64_t n_total, const uint64_t *C, uint64_t W, double u,
                   int64_t m0, int64_t count, int64_t *anc)
{
    double stepw = (double)W / (double)n_total;
    #pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < count; j++) {
        double Tm = ((double)(m0 + j) + u) * stepw;
        uint64_t tq = (uint64_t)floor(Tm);
        int64_t lo = 0, hi = n_total;            /* first index with C > tq */
        while (lo < hi) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (C[mid] > tq) hi = mid; else lo = mid + 1;
        }
        if (lo > n_total - 1) lo = n_total - 1;
        anc[j] = lo;
    }
}

/* The reference's own sequential two-pointer walk (ParticleFilters.jl
 * LowVarianceResampler) on the same fixed-point quantities, used to check
 * orc_ancestors(): first i with C_i > floor(T_m). */
void orc_ancestors_walk(int64_t n_total, const uint64_t *C, uint64_t W, double u, int64_t *anc)
{
    double stepw = (double)W / (double)n_total;
    int64_t i = 0;
    for (int64_t m = 0; m < n_total; m++) {
        double Tm = ((double)m + u) * stepw;
        uint64_t tq = (uint64_t)floor(Tm);
        while (C[i] <= tq && i < n_total - 1) i++;
        anc[m] = i;
    }
}

/* Complete single-shard update:  b (n states), a, o  ->  b' (n states).
 * Slot m of b' = SimStep(s_anc) ... see orc_pf_update_lazy for the order
 * actually used by the device path:
 *   1. lw_m   = log weight of PRE-step state s_m          (m = 0..n-1)
 *   2. sp_m   = SimStep(s_m) with the stream of (global slot m, step)
 *   3. resample sp by w: b'_m = sp_anc(m)
 * which is ParticleFilters' predict! / reweight! / resample with the weight
 * taken from the pre-step state. */
uint32_t orc_pf_update(const orc_params *P, int64_t n,
                       const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T,
                       double a, int64_t o, uint64_t seed, uint32_t step,
                       int64_t *S2, int64_t *I2, int64_t *R2, int64_t *T2,
                       int64_t *anc_out, double *lw_out, double *lwmax_out, uint64_t *W_out, double *ess_out)
{
    const int H = P->H, D1 = P->D + 1;
    uint32_t flags = 0;
    double *lw = (double *)malloc(sizeof(double) * (size_t)n);
    uint64_t *wq = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    int64_t *anc = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *pS = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *pR = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *pI = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n * H));
    int64_t *pT = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n * D1 * H));
    orc_log_weights(P, n, S, I, R, T, a, o, lw, NULL, NULL);
    flags |= orc_step_batch(P, n, S, I, R, T, &a, 0, NULL, 0, seed, step, 0, pS, pI, pR, pT, NULL, NULL, NULL);
    double lwmax = -INFINITY;
    for (int64_t i = 0; i < n; i++) if (lw[i] > lwmax) lwmax = lw[i];
    if (lwmax == -INFINITY) {
        /* all weights zero: keep the propagated particles as they are */
        flags |= ORC_F_ZERO_W;
        for (int64_t i = 0; i < n; i++) anc[i] = i;
        if (W_out) *W_out = 0;
        if (ess_out) *ess_out = 0.0;
    } else {
        int F = orc_frac_bits(n);
        orc_quantize(n, lw, lwmax, F, wq);
        double s2 = 0.0;
        for (int64_t i = 0; i < n; i++) { double w = (double)wq[i]; s2 += w * w; }
        uint64_t W = orc_cumsum(n, wq, 0, wq);
        double u = orc_resample_u(seed, step);
        orc_ancestors(n, wq, W, u, 0, n, anc);
        if ((double)n > 1e-3 * (double)W) flags |= ORC_F_COARSE_W;   /* truncation bound n / W of the fixed-point weights */
        if (W_out) *W_out = W;
        if (ess_out) *ess_out = ((double)W * (double)W) / s2;
    }
    for (int64_t m = 0; m < n; m++) {
        int64_t p = anc[m];
        S2[m] = pS[p]; R2[m] = pR[p];
        memcpy(I2 + m * H, pI + p * H, sizeof(int64_t) * (size_t)H);
        if (T2) memcpy(T2 + m * D1 * H, pT + p * D1 * H, sizeof(int64_t) * (size_t)(D1 * H));
    }
    if (anc_out) memcpy(anc_out, anc, sizeof(int64_t) * (size_t)n);
    if (lw_out) memcpy(lw_out, lw, sizeof(double) * (size_t)n);
    if (lwmax_out) *lwmax_out = lwmax;
    free(lw); free(wq); free(anc); free(pS); free(pR); free(pI); free(pT);
    return flags;
}

/* belief mean (updater.jl:5-21): integer sums, round half to even, R by complement.
 * Returns 0, or 1 if avgR < 0 (the reference asserts). */
int orc_belief_mean(const orc_params *P, int64_t n,
                    const int64_t *S, const int64_t *I, const int64_t *T,
                    int64_t *avgS, int64_t *avgI, int64_t *avgR, int64_t *avgT)
{
    const int H = P->H, D1 = P->D + 1;
    int64_t sumS = 0, sumI[ORC_MAX_H] = {0}, sumT[(ORC_MAX_D + 1) * ORC_MAX_H] = {0};
    for (int64_t p = 0; p < n; p++) {
        sumS += S[p];
        for (int i = 0; i < H; i++) sumI[i] += I[p * H + i];
        if (T) for (int j = 0; j < D1 * H; j++) sumT[j] += T[p * D1 * H + j];
    }
    double nd = (double)n;
    *avgS = (int64_t)nearbyint((double)sumS / nd);
    int64_t tot = 0;
    for (int i = 0; i < H; i++) { avgI[i] = (int64_t)nearbyint((double)sumI[i] / nd); tot += avgI[i]; }
    *avgR = P->N - (*avgS + tot);
    if (avgT) for (int j = 0; j < D1 * H; j++) avgT[j] = (int64_t)nearbyint((double)sumT[j] / nd);
    return *avgR < 0;
}

/* ------------------------------------------------------------------ */
/* initial states                                                      */
/* ------------------------------------------------------------------ */

/* State(Normal(mu, sigma), params) (typedef.jl:159-181): each I0_i is a draw
 * of Normal(mu, sigma) truncated to [0, inf) (rejection on Box-Muller
 * normals), rounded half to even; S0 = N - sum(I0), R0 = 0, Tests = 0.
 * Stream: kind K_INIT, bin i, block j: attempt 2j takes the first normal of
 * words (0,1), attempt 2j+1 the first normal of words (2,3). */
void orc_init_normal(const orc_params *P, int64_t n, double mu, double sigma, uint64_t seed, uint64_t particle0,
                     int64_t *S, int64_t *I, int64_t *R, int64_t *T)
{
    const int H = P->H, D1 = P->D + 1;
    for (int64_t p = 0; p < n; p++) {
        rng_t g = rng_make(seed, particle0 + (uint64_t)p, 0);
        int64_t tot = 0;
        for (int i = 0; i < H; i++) {
            double x = 0.0;
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t att = 0; att < 512u; att++) {
                if ((att & 1u) == 0) rng_block(&g, K_INIT, (uint32_t)i, att >> 1, w);
                float z0, z1;
                if ((att & 1u) == 0) box_muller(w[0], w[1], &z0, &z1); else box_muller(w[2], w[3], &z0, &z1);
                double cand = mu + sigma * (double)z0;
                if (cand >= 0.0) { x = cand; break; }
            }
            int64_t v = (int64_t)nearbyint(x);
            I[p * H + i] = v; tot += v;
        }
        S[p] = P->N - tot; R[p] = 0;
        if (T) memset(T + p * D1 * H, 0, sizeof(int64_t) * (size_t)(D1 * H));
    }
}

/* State(params) (typedef.jl:196-217): Bayesian-bootstrap split of N into
 * (S, inf, R) and of inf into I[1..H]; the rounding leftover goes to the
 * largest compartment (S, I[end] or R).  Uniform spacings are generated as
 * normalised exponentials e_j = -log(U_j) (same law as sorted-uniform gaps).
 * Stream: kind K_INIT, bin 0x100 + j/4 word j%4 for the 3-way split,
 * bin 0x200 + i/4 word i%4 for the H-way split. */
void orc_init_simplex(const orc_params *P, int64_t n, uint64_t seed, uint64_t particle0,
                      int64_t *S, int64_t *I, int64_t *R, int64_t *T)
{
    const int H = P->H, D1 = P->D + 1;
    for (int64_t p = 0; p < n; p++) {
        rng_t g = rng_make(seed, particle0 + (uint64_t)p, 0);
        uint32_t w[4];
        rng_block(&g, K_INIT, 0x100u, 0, w);
        double e0 = -s_log(u01(w[0])), e1 = -s_log(u01(w[1])), e2 = -s_log(u01(w[2]));
        double es = (e0 + e1) + e2;
        double Nd = (double)P->N;
        int64_t s0 = (int64_t)nearbyint(Nd * (e0 / es));
        int64_t inf = (int64_t)nearbyint(Nd * (e1 / es));
        int64_t r0 = (int64_t)nearbyint(Nd * (e2 / es));
        double ee[ORC_MAX_H], esum = 0.0;
        for (int i = 0; i < H; i++) {
            if ((i & 3) == 0) rng_block(&g, K_INIT, 0x200u + (uint32_t)(i >> 2), 0, w);
            ee[i] = -s_log(u01(w[i & 3]));
            esum = (i == 0) ? ee[i] : esum + ee[i];
        }
        int64_t tot = 0;
        for (int i = 0; i < H; i++) {
            int64_t v = (int64_t)nearbyint((double)inf * (ee[i] / esum));
            I[p * H + i] = v; tot += v;
        }
        int64_t left = P->N - (s0 + tot + r0);
        if (s0 >= tot && s0 >= r0) s0 += left;              /* argmax, first wins ties */
        else if (tot >= r0) I[p * H + H - 1] += left;
        else r0 += left;
        S[p] = s0; R[p] = r0;
        if (T) memset(T + p * D1 * H, 0, sizeof(int64_t) * (size_t)(D1 * H));
    }
}

/* ------------------------------------------------------------------ */
/* open-loop rollout (simulate.jl:175-196): history of S, sum(I), R     */
/* BEFORE each day's step, sum(pos) and reward of the day.             */
/* hist: int64[n][T][4] = S, sumI, R, sum(pos) ; rew: double[n][T]      */
/* actions: double[n_or_1][T_or_1]                                     */
/* ------------------------------------------------------------------ */
uint32_t orc_rollout(const orc_params *P, int64_t n, int Tdays,
                     const int64_t *S, const int64_t *I, const int64_t *R, const int64_t *T,
                     const double *actions, int act_per_particle, int act_per_day, double a_prev0,
                     uint64_t seed, uint32_t step0, uint64_t particle0,
                     int64_t *hist, double *rew,
                     int64_t *S2, int64_t *I2, int64_t *R2, int64_t *T2)
{
    uint32_t flags = 0;
    #pragma omp parallel for schedule(static) reduction(|:flags)
    for (int64_t p = 0; p < n; p++) {
        pstate s, sp;
        uint32_t fl = 0;
        load_state(P, p, S, I, R, T, &s);
        double ap = a_prev0;
        for (int t = 0; t < Tdays; t++) {
            int64_t ni, po[ORC_MAX_H], sumI = 0, o = 0;
            double a = actions[(act_per_particle ? p * (act_per_day ? Tdays : 1) : 0) + (act_per_day ? t : 0)];
            for (int i = 0; i < P->H; i++) sumI += s.I[i];
            rng_t g = rng_make(seed, particle0 + (uint64_t)p, step0 + (uint32_t)t);
            step_core(P, &s, a, NULL, NULL, NULL, &g, &sp, &ni, po, &fl);
            for (int i = 0; i < P->H; i++) o += po[i];
            if (hist) {
                int64_t *h = hist + (p * Tdays + t) * 4;
                h[0] = s.S; h[1] = sumI; h[2] = s.R; h[3] = o;
            }
            if (rew) rew[p * Tdays + t] = reward(P, &sp, a, ap);
            ap = a;
            s = sp;
        }
        if (S2) store_state(P, p, &s, S2, I2, R2, T2);
        flags |= fl;
    }
    return flags;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int t)
{
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}
