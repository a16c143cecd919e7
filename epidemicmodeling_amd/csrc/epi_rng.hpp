// Counter-based random variates for the gfx950 particle kernels.
//
// Every variate is a pure function of (seed, global particle slot, step,
// stream tag, attempt): a Philox4x32-7 block addressed by
//   key     = (seed lo, seed hi)
//   counter = (slot lo, slot hi, step, kind << 24 | bin << 8 | blk)
// so results do not depend on launch geometry, sharding or execution order.
//
// The elementary functions below are written as explicit sequences of IEEE-754
// operations (compile with -ffp-contract=off; fma only where written) so that
// a host restatement of the same sequence reproduces every accept/reject
// decision bit for bit.  No device-library transcendental is used.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace epi {

// stream kinds (the tag's top byte)
constexpr uint32_t K_BINOM0 = 0u;  // words 0,1: isolation draw attempt 0; words 2,3: positive-test draw attempt 0
constexpr uint32_t K_SYMPT  = 1u;  // blk j: isolation attempts 1+2j (words 0,1) and 2+2j (words 2,3)
constexpr uint32_t K_POS    = 2u;  // same for the positive-test draw
constexpr uint32_t K_GAMMA  = 3u;  // bin = pair j: attempt 0 of infection term 2j (words 0, 2: normal, uniform) and of term 2j+1 (words 1, 3)
constexpr uint32_t K_GRETRY = 7u;  // bin = term, blk b: its attempts 1+2b (words 0, 2) and 2+2b (words 1, 3)
constexpr uint32_t K_BOOST  = 4u;  // word 0: uniform of the shape<1 boost
constexpr uint32_t K_RESAMP = 5u;  // slot 0: words 0,1 -> 53-bit systematic-resampling offset
constexpr uint32_t K_INIT   = 6u;  // initial-state generators
// (8: K_ZIG, the slow path of the normal generator, below)

#define EPI_HD __host__ __device__ __forceinline__
// rarely executed pieces are kept out of line on the device, where inlining them costs the callers registers
#if defined(__HIP_DEVICE_COMPILE__)
#define EPI_HD_OUTLINE __host__ __device__ __noinline__
#else
#define EPI_HD_OUTLINE __host__ __device__ inline
#endif

EPI_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
EPI_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
EPI_HD uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
EPI_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

struct Rng {
    uint32_t k0, k1, c0, c1, c2;
};

EPI_HD Rng rng_make(uint64_t seed, uint64_t slot, uint32_t step)
{
    Rng g;
    g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32);
    g.c0 = (uint32_t)slot; g.c1 = (uint32_t)(slot >> 32); g.c2 = step;
    return g;
}

// a ^ b ^ c.  gfx950 has a three-operand bitwise instruction (truth table 0x96 = parity), which the compiler does
// not select for a chain of two exclusive-ors by itself.
EPI_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

// Philox4x32 (Salmon, Moraes, Dror, Shaw, SC'11) with 7 rounds: the round count the authors report as the smallest
// that passes BigCrush (their default of 10 adds a safety margin; three rounds are 30 % of the generator's cost, and
// the generator is a fifth of the step kernel's).  The round function is checked with the published 7- and 10-round
// known-answer vectors.
EPI_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t (&w)[4])
{
#ifndef EPI_PHILOX_ROUNDS
#define EPI_PHILOX_ROUNDS 7
#endif
#pragma unroll
    for (int r = 0; r < EPI_PHILOX_ROUNDS; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // round 0 mixes wave-uniform words (step, stream tag, key): plain operators let the compiler keep those
        // parts in scalar registers
        uint32_t n0 = r == 0 ? (uint32_t)(p1 >> 32) ^ c1 ^ k0 : xor3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = r == 0 ? (uint32_t)(p0 >> 32) ^ c3 ^ k1 : xor3((uint32_t)(p0 >> 32), c3, k1);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}

EPI_HD void rng_block(const Rng& g, uint32_t kind, uint32_t bin, uint32_t blk, uint32_t (&w)[4])
{
    philox4x32(g.c0, g.c1, g.c2, (kind << 24) | (bin << 8) | blk, g.k0, g.k1, w);
}

// uniform in (0,1): (w + 1/2) 2^-32, exact in binary64
EPI_HD double u01(uint32_t w) { return ((double)w + 0.5) * 0x1p-32; }
EPI_HD double u01_centred(uint32_t w) { return u01(w) - 0.5; }

// ---- elementary functions ---------------------------------------------------

// Correctly rounded binary32 square root and quotient (the values sqrtf and "/" give on the host).  The device
// forms are the compiler's own expansions (hardware estimate, accurate to one ulp, plus the rounding correction)
// without the operand scaling and the special-case fix-up that only subnormal, huge or non-finite operands need:
// every call site keeps its operands between 2^-30 and 2^32 (or exactly zero, which both forms map to zero).
EPI_HD float sqrt_rn(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_sqrtf(x);
    const float lo = u2f(f2u(r) - 1u), hi = u2f(f2u(r) + 1u);
    const float el = __builtin_fmaf(-lo, r, x), eh = __builtin_fmaf(-hi, r, x);
    float y = el <= 0.0f ? lo : r;
    y = eh > 0.0f ? hi : y;
    return y;
#else
    return __builtin_sqrtf(x);
#endif
}

EPI_HD float div_rn(float n, float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return q;
#else
    return n / d;
#endif
}

// natural log, x > 0 finite normal; argument reduction to [sqrt(1/2), sqrt(2))
// and the degree-14 series in s = f/(2+f) of the classic fdlibm scheme.
EPI_HD double s_log(double x)
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
    double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double hfsq = (0.5 * f) * f;
    double dk = (double)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) for x <= ~700; exactly 0 below -708 (and for NaN / -inf)
EPI_HD double s_exp(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    if (!(x >= -708.0)) return 0.0;
    if (x > 709.0) x = 709.0;
    double kd = __builtin_floor(__builtin_fma(x, 1.4426950408889634, 0.5));
    double r = __builtin_fma(-kd, ln2_hi, x);
    r = __builtin_fma(-kd, ln2_lo, r);
    double p = 1.6059043836821613e-10;
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 0.0001984126984126984);
    p = __builtin_fma(p, r, 0.001388888888888889);
    p = __builtin_fma(p, r, 0.008333333333333333);
    p = __builtin_fma(p, r, 0.041666666666666664);
    p = __builtin_fma(p, r, 0.16666666666666666);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    int k = (int)kd;
    return p * u2d((uint64_t)(k + 1023) << 52);
}

// binary32 log for x in (0, 1]
EPI_HD float s_logf(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = div_rn(f, 2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * __builtin_fmaf(w, Lg4, Lg2);
    float t2 = z * __builtin_fmaf(w, Lg3, Lg1);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    float dk = (float)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// Box-Muller in binary32: two words -> two standard normals.  The angle is
// (quadrant + 1/2) pi/2 + ph with |ph| <= pi/4, so sin/cos need only short
// polynomials and one rotation by 45 degrees.
EPI_HD void box_muller(uint32_t w0, uint32_t w1, float& z0, float& z1)
{
    float u1 = __builtin_fmaf((float)w0, 0x1p-32f, 0x1p-33f);   // (0, 1]
    float r = sqrt_rn(-2.0f * s_logf(u1));
    uint32_t quad = w1 >> 30;
    int32_t ti = (int32_t)(w1 & 0x3fffffffu) - (1 << 29);
    float y = ((float)ti + 0.5f) * 0x1p-30f;                    // [-1/2, 1/2]
    float ph = y * 1.57079637f;
    float x2 = ph * ph;
    float sp = ph * __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, 2.7557319e-06f, -0.0001984127f), 0.008333334f), -0.16666667f), 1.0f);
    float cp = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, -2.755732e-07f, 2.4801588e-05f), -0.0013888889f), 0.041666668f), -0.5f), 1.0f);
    float h = r * 0.70710677f;
    float a = cp * h, b = sp * h;
    float c = a - b, s = a + b;
    // rotate (c, s) by quad * 90 degrees
    float zc = (quad & 1u) ? s : c;
    float zs = (quad & 1u) ? c : s;
    if (quad == 1u || quad == 2u) zc = -zc;
    if (quad >= 2u) zs = -zs;
    z0 = zc; z1 = zs;
}

// ---- Binomial(n, p) -----------------------------------------------------------
// n * min(p, 1-p) < 10 : sequential inversion from 0 (one uniform)
// otherwise            : transformed rejection with squeeze (Hoermann 1993, BTRS)

EPI_HD double stirling_tail(double k)
{
    // log(k!) - [ (k + 1/2) log(k+1) - (k+1) + log(sqrt(2 pi)) ]
    if (k <= 9.0) {
        int i = (int)k;
        double t = 0.08106146679532726;
        t = (i == 1) ? 0.04134069595540929 : t;
        t = (i == 2) ? 0.02767792568499834 : t;
        t = (i == 3) ? 0.02079067210376509 : t;
        t = (i == 4) ? 0.01664469118982119 : t;
        t = (i == 5) ? 0.01387612882307075 : t;
        t = (i == 6) ? 0.01189670994589177 : t;
        t = (i == 7) ? 0.01041126526197209 : t;
        t = (i == 8) ? 0.009255462182712733 : t;
        t = (i == 9) ? 0.008330563433362871 : t;
        return t;
    }
    double r = 1.0 / (k + 1.0);
    double r2 = r * r;
    return (0.08333333333333333 - (0.002777777777777778 - 0.0007936507936507937 * r2) * r2) * r;
}

constexpr uint32_t BINOM_MAX_ATTEMPTS = 512u;

// ---- single-precision screens (device only) ---------------------------------------
// The accept/reject decisions of the two rejection samplers are DEFINED by the
// binary64 expressions in binomial_draw / gamma_draw below.  On the device most
// of the logarithms are avoided: the quantity is first estimated in binary32 with
// the hardware reciprocal / log2 approximations, and the estimate decides whenever
// it is further from the threshold than a guard band many times its error bound.
// Only draws inside the band fall through to the defining binary64 expression, so
// results are unchanged bit for bit.
#if defined(__HIP_DEVICE_COMPILE__)
#define EPI_HAVE_SCREEN 1
__device__ __forceinline__ float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float f_ln(float x) { return __builtin_amdgcn_logf(x) * 0.693147181f; }
#endif

// log(k!) for an integer k >= 0 held in a double: Stirling's formula with the tail above, evaluated
// left to right as written.  (Absolute error a few ulp of the value; differences of four of them
// are what the acceptance test below uses.)
EPI_HD double lfact(double k)
{
    return ((k + 0.5) * s_log(k + 1.0) - (k + 1.0)) + 0.91893853320467274178 + stirling_tail(k);
}

// The same values tabulated for k < n (device memory, filled by evaluating lfact on the device), so that
// the acceptance test costs four loads instead of four logarithms; larger k are computed.
struct LfTable {
    const double* t;
    int32_t n;
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ double lfact_far(double k) { return lfact(k); }
__device__ __forceinline__ double lfact_of(const LfTable& T, double k)
{
    if (k < (double)T.n) return T.t[(int32_t)k];
    return lfact_far(k);
}
#else
static inline double lfact_of(const LfTable&, double k) { return lfact(k); }
#endif

// A Binomial(n, p) draw is split into pieces so that a kernel can either loop over attempts
// per thread (binomial_draw) or hand failed attempts to other threads:
//   binom_classify : trivial result / inversion / rejection, and the folded probability
//   binom_inversion: n min(p,1-p) < 10, one uniform, always completes
//   btrs_bin       : per-probability constants of the rejection sampler
//   btrs_consts    : its per-draw constants
//   btrs_attempt   : one attempt from one word pair: accepted k, or -1
enum { BINOM_DONE = 0, BINOM_INVERSION = 1, BINOM_REJECTION = 2 };

EPI_HD int binom_classify(int32_t n, double p, int32_t& result, double& pp, bool& flip)
{
    result = 0; pp = 0.0; flip = false;
    if (n <= 0 || !(p > 0.0)) return BINOM_DONE;
    if (p >= 1.0) { result = n; return BINOM_DONE; }
    flip = p > 0.5;
    pp = flip ? 1.0 - p : p;
    return ((double)n * pp < 10.0) ? BINOM_INVERSION : BINOM_REJECTION;
}

// exp(x) in binary32 for -88 < x <= 0: x = k ln 2 + r, degree-6 polynomial in r, scaled by 2^k (relative error 2e-7)
EPI_HD float s_expf(float x)
{
    const float kd = __builtin_floorf(__builtin_fmaf(x, 1.44269504f, 0.5f));
    float r = __builtin_fmaf(-kd, 6.93145752e-01f, x);
    r = __builtin_fmaf(-kd, 1.42860677e-06f, r);
    float p = 1.38888889e-03f;
    p = __builtin_fmaf(p, r, 8.33333333e-03f);
    p = __builtin_fmaf(p, r, 4.16666667e-02f);
    p = __builtin_fmaf(p, r, 1.66666667e-01f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return u2f(f2u(p) + ((uint32_t)(int32_t)kd << 23));
}

// 1/x for the inversion recurrence (the correctly rounded quotients: a table up to 71, beyond that a division)
#define EPI_R(x) (1.0f / (float)(x))
#define EPI_R8(b) EPI_R(b), EPI_R(b + 1), EPI_R(b + 2), EPI_R(b + 3), EPI_R(b + 4), EPI_R(b + 5), EPI_R(b + 6), EPI_R(b + 7)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#else
static
#endif
const float EPI_RECIP[72] = { 0.0f, EPI_R(1), EPI_R(2), EPI_R(3), EPI_R(4), EPI_R(5), EPI_R(6), EPI_R(7),
    EPI_R8(8), EPI_R8(16), EPI_R8(24), EPI_R8(32), EPI_R8(40), EPI_R8(48), EPI_R8(56), EPI_R8(64) };
constexpr int EPI_RECIP_N = 72;
#ifdef EPI_PHASE_STAMPS
__device__ unsigned long long g_phase[32];   // measurement builds: phase ticks and queue counters of the cooperative step kernel
#endif

// Sequential inversion from 0 in binary32 (n pp < 10: at most a few dozen terms matter):
//   u = fl(w >> 8) 2^-24 + 2^-25 in (0, 1],  r_0 = exp(fl(n) lnq),  r_x = r_(x-1) (s ((fl(n) + 1 - x) (1 / x))),
//   the result is the first x with u <= r_x after u -= r_x for the x before it, or n.
// lnq = fl(s_log(1 - pp)), s = fl(pp / (1 - pp)): per-probability constants, tabulated per age bin by the kernels.
// Far in the tail the terms fall below half an ulp of what is left of u and the subtraction stops making progress
// (this happens when u exceeds the sum of the rounded terms, with probability of the order 1e-6): the search ends
// there, at the x where r_x < u 2^-25.
EPI_HD int32_t binom_inversion(int32_t n, double lnq, double s, uint32_t wa)
{
    const float nf = (float)n, sf = (float)s;
    const float a1 = nf + 1.0f;
    float r = s_expf(nf * (float)lnq);
    float u = __builtin_fmaf((float)(wa >> 8), 0x1p-24f, 0x1p-25f);
    int32_t res = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // the same recurrence with a wave-uniform counter, eight terms per trip: the reciprocals of a trip arrive in one
    // wide scalar load (a load per term costs the loop its latency every time), and the wave leaves the loop at the
    // first trip boundary after its last lane has finished
    bool live = u > r && 0 < n;
#pragma unroll 1
    for (int32_t xb = 1; __any(live); xb += 8) {
        float t8[8];
#ifdef EPI_PHASE_STAMPS
        { const unsigned long long ml = __ballot(live); if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(true)) - 1) { atomicAdd(&g_phase[20], 1ULL); atomicAdd(&g_phase[22], (unsigned long long)__popcll(ml)); } }
#endif
        if (xb + 7 < EPI_RECIP_N) {
#pragma unroll
            for (int k = 0; k < 8; k++) t8[k] = EPI_RECIP[xb + k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) t8[k] = div_rn(1.0f, (float)(xb + k));
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int32_t x = xb + k;
            const float xf = (float)x;
            if (live) {
                u -= r;
                res = x;
                r *= sf * ((a1 - xf) * t8[k]);
                live = u > r && x < n && !(r < u * 0x1p-25f);
            }
        }
    }
#else
    while (u > r && res < n) {
        u -= r;
        res += 1;
        const float xf = (float)res;
        const float t = 1.0f / xf;
        r *= sf * ((a1 - xf) * t);
        if (r < u * 0x1p-25f) break;
    }
#endif
    return res;
}

// ---- the rejection sampler (Hoermann 1993, algorithm BTRS), binary32 hat -------------------------------
// Per (draw, age bin), functions of the folded probability pp = min(p, 1 - p) alone (tabulated by the kernels):
//   ppx  = round(pp 2^32)                      the probability in 0.32 fixed point
//   pq   = (float)(pp (1 - pp))
//   a0x2 = (float)(2 (-0.0873 + 0.01 pp))
//   lpq  = s_log(pp / (1 - pp))                binary64
// Per draw (n >= the inversion threshold), every operation an IEEE binary32 operation in the order written:
//   spq = sqrt(fl(n) pq),  b = fma(2.53, spq, 1.15),  2a = fma(0.0496, b, a0x2)
//   c   = n pp + 1/2 and the mode m = floor((n + 1) pp) from the 64-bit products n ppx + 2^31 and (n + 1) ppx:
//         exact integer parts; the fraction of c keeps its leading 23 bits
// One attempt from the word pair (wa, wb):
//   u  = fl((int32) wa) 2^-32 in [-1/2, 1/2],  us = max(1/2 - |u|, 2^-25),  v = fl(wb) 2^-32 in [0, 1]
//   k  = floor(c) + floor((2a / us + b) u + frac(c)), the inner expression in binary64 from these binary32 values
//   k outside [0, n]: rejected
//   us >= 0.07 and (0.92 - v) b >= 4.2 (binary32; Hoermann's v <= v_r without the division): accepted
//   otherwise accepted iff  log(v alpha / (a / us^2 + b)) <= log f(k) - log f(m)   in binary64, with
//   alpha = (2.83 + 5.1 / b) spq, the right side from the log factorials as above, v = 0 accepted.
// The hat only has to dominate: its constants carry three digits in the source and tolerate binary32.  The target
// ratio f(k) / f(m) is evaluated in binary64 for the exact p, so the accepted k follow Binomial(n, p) up to the
// resolution of the 32-bit uniform after its conversion to binary32 (relative error of a single probability of
// the order b 10^-7, of either sign from one k to the next).
struct BtrsBin {
    uint32_t ppx;
    float pq, a0x2;
    double lpq;
};

EPI_HD BtrsBin btrs_bin(double pp)
{
    BtrsBin B;
    B.ppx = (uint32_t)(pp * 4294967296.0 + 0.5);            // pp <= 1/2: fits
    B.pq = (float)(pp * (1.0 - pp));
    B.a0x2 = (float)(2.0 * (-0.0873 + 0.01 * pp));
    B.lpq = (pp > 0.0 && pp < 1.0) ? s_log(pp / (1.0 - pp)) : 0.0;
    return B;
}

// constants of the table-free bounds of the acceptance test (btrs_bounds), per probability
struct BtrsSide {
    float ppf;          // pp
    float hqp;          // (1 - 2 pp) / 2
    float c3u, c3d;     // (2 P^3 + Q^3) / 3 for (P, Q) = (pp, 1 - pp) above the mode and (1 - pp, pp) below it, rounded up
};
EPI_HD BtrsSide btrs_side(double pp)
{
    const double qq = 1.0 - pp;
    BtrsSide S;
    S.ppf = (float)pp;
    S.hqp = (float)(0.5 * (qq - pp));
    S.c3u = (float)((2.0 * pp * pp * pp + qq * qq * qq) / 3.0 * 1.000001);
    S.c3d = (float)((2.0 * qq * qq * qq + pp * pp * pp) / 3.0 * 1.000001);
    return S;
}

struct BtrsConsts {
    float npq, spq, b, a2, cfr;
    int32_t ci, m, n;
    float hf, rV;       // btrs_consts_quick only: 1/2 - frac((n + 1) pp) and 1 / ((n + 1) pp (1 - pp))
};

EPI_HD BtrsConsts btrs_consts(int32_t n, const BtrsBin& B)
{
    BtrsConsts C;
    C.n = n;
    C.npq = (float)n * B.pq;
    C.spq = sqrt_rn(C.npq);
    C.b = __builtin_fmaf(2.53f, C.spq, 1.15f);
    C.a2 = __builtin_fmaf(0.0496f, C.b, B.a0x2);
    const uint64_t prod = (uint64_t)(uint32_t)n * B.ppx + 0x80000000ULL;
    C.ci = (int32_t)(uint32_t)(prod >> 32);
    C.cfr = u2f(0x3f800000u | ((uint32_t)prod >> 9)) - 1.0f;
    C.m = (int32_t)(uint32_t)(((uint64_t)((uint32_t)n + 1u) * B.ppx) >> 32);
    return C;
}

EPI_HD float btrs_u(uint32_t wa) { return (float)(int32_t)wa * 0x1p-32f; }
EPI_HD float btrs_us(float u) { return __builtin_fmaxf(0.5f - __builtin_fabsf(u), 0x1p-25f); }
EPI_HD float btrs_v(uint32_t wb) { return (float)wb * 0x1p-32f; }

// floor of the defining candidate expression, limited to +-2^30 (far outside [0, n] - floor(c) either way)
EPI_HD_OUTLINE int32_t btrs_kfloor_exact(float a2, float b, float cfr, float u, float us)
{
    double x = __builtin_floor(((double)a2 / (double)us + (double)b) * (double)u + (double)cfr);
    x = x < -1073741824.0 ? -1073741824.0 : x;
    x = x > 1073741824.0 ? 1073741824.0 : x;
    return (int32_t)x;
}

// left side of the acceptance test
EPI_HD_OUTLINE double btrs_lv(float v, float us, float a2, float b, float spq)
{
    if (!(v > 0.0f)) return -__builtin_inf();
    const double alpha = (2.83 + 5.1 / (double)b) * (double)spq;
    return s_log((double)v * alpha / (0.5 * (double)a2 / ((double)us * (double)us) + (double)b));
}

// right side for a k in [0, n]
EPI_HD double btrs_ub(const BtrsConsts& C, int32_t k, double lpq, const LfTable& lf)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (C.n < lf.n) {                        // every argument is inside the table: four loads, no branches
        const double* __restrict__ t = lf.t;
        return ((t[C.m] + t[C.n - C.m]) - (t[k] + t[C.n - k])) + (double)(k - C.m) * lpq;
    }
#endif
    return ((lfact_of(lf, (double)C.m) + lfact_of(lf, (double)(C.n - C.m))) - (lfact_of(lf, (double)k) + lfact_of(lf, (double)(C.n - k)))) + (double)(k - C.m) * lpq;
}

#ifdef EPI_HAVE_SCREEN
// Binary32 estimates (device only).  The candidate's floor is trusted when the binary32 value of the expression is
// further from an integer than its error bound: the hardware reciprocal is good to one ulp and two fused
// multiply-adds follow, so |estimate - exact| < 2.4e-7 (|x| + b); the band is four times that.
__device__ __forceinline__ bool btrs_kfloor_screen(const BtrsConsts& C, float u, float us, float& r_us, int32_t& kf)
{
    r_us = f_rcp(us);
    const float x = __builtin_fmaf(__builtin_fmaf(C.a2, r_us, C.b), u, C.cfr);
    const float fl = __builtin_floorf(x);
    const float fr = x - fl;
    const float dl = __builtin_fmaf(__builtin_fabsf(x) + C.b, 1e-6f, 1e-6f);
    kf = (int32_t)__builtin_amdgcn_fmed3f(fl, -1073741824.0f, 1073741824.0f);
    return fr > dl && fr < 1.0f - dl;
}
// Estimate of btrs_lv: argument good to 8 roundings and two hardware reciprocals (1e-6), the hardware log2 to one ulp
// of its result, so |estimate - exact| < 1e-6 + 3e-7 |lv|; the band is ten times that.
constexpr float BTRS_GUARD_REL = 3e-6f, BTRS_GUARD_ABS = 1e-5f;
__device__ __forceinline__ float btrs_lv_screen(const BtrsConsts& C, float v, float r_us)
{
    const float den = C.b * __builtin_fmaf(0.5f * C.a2, r_us * r_us, C.b);          // b (a / us^2 + b)
    const float num = (v * C.spq) * __builtin_fmaf(2.83f, C.b, 5.1f);               // v alpha b
    return f_ln(num * f_rcp(den));
}
#endif

enum { BTRS_ACCEPT = 0, BTRS_OPEN = 1, BTRS_REJECT = 2 };
#ifdef EPI_HAVE_SCREEN
// Bounds on the right side of the acceptance test, L = log f(k) - log f(m), that need no table.  With x* = (n + 1) pp,
// m = floor(x*), F = x* - m and V = (n + 1) pp (1 - pp), the ratio of neighbouring probabilities is
//   f(m + i) / f(m + i - 1) = (V - s P) / (V + s Q),  s = i - F, (P, Q) = (pp, 1 - pp)                 above the mode,
//   f(m - i) / f(m - i + 1) = (V - s P) / (V + s Q),  s = i - (1 - F), (P, Q) = (1 - pp, pp)          below it,
// so that for j = |k - m|, from log(1 - a) <= -a - a^2/2, log(1 + b) >= b - b^2/2 and the cubic remainders of the two series,
//   U0 - R <= L <= U0,   U0 = -S1 / V + (Q - P) S2 / (2 V^2),   R = (P^3 / (1 - j P / V) + Q^3) S3 / (3 V^3),
//   S1 = sum s = j ((j + 1)/2 - F'),  S2 = sum s^2,  S3 = sum s^3 <= (j (j + 1) / 2)^2          (F' = F or 1 - F).
// Here S2 is taken as j^2 (j/3 + 1/2 - F'), which leaves out a term of at most j/6: e2 = 0.084 j / V^2 on either side;
// j P / V < 1/2 is required (R then uses 2 P^3 + Q^3).  At j = 2 sigma the band R is about 1 / sigma^2: a few draws in
// a thousand stay undecided, where the classical bounds of this kind (BTPE's squeeze, band ~ 1 / sigma) leave one in eight.
// The estimate lvf of the left side decides when it clears a bound by more than the errors of both sides: its own band,
// 2e-6 |U0| for the hardware reciprocal and the roundings, and j / V times the uncertainty of F (pp in 0.32 fixed
// point times n, plus the 23 bits kept).  ACCEPT / REJECT are then the outcome of the defining test; OPEN leaves it to that test.
__device__ __forceinline__ int btrs_bounds(const BtrsConsts& C, const BtrsSide& S, int32_t k, float lvf)
{
    const int32_t d = k - C.m;
    const bool below = d < 0;
    const float j = (float)(below ? -d : d);
    const float hF = below ? -C.hf : C.hf;                       // 1/2 - F'
    const float cq = below ? -S.hqp : S.hqp;                     // (Q - P) / 2
    const float c3 = below ? S.c3d : S.c3u;
    const float P = below ? 1.0f - S.ppf : S.ppf;
    const float rV = C.rV, rV2 = rV * rV;
    const float S1 = j * __builtin_fmaf(0.5f, j, hF);
    const float S2 = (j * j) * __builtin_fmaf(0.33333334f, j, hF);
    const float U0 = -rV * __builtin_fmaf(-(cq * rV), S2, S1);
    const float jp1 = j + 1.0f, q2 = jp1 * jp1;
    const float e2 = (0.084f * j) * rV2;
    const float R = __builtin_fmaf((c3 * (0.25f * q2 * q2)) * rV2, rV, e2);
    const float gf = (j * rV) * __builtin_fmaf((float)C.n, 1.2e-10f, 2e-7f);
    const float g = __builtin_fmaf(__builtin_fabsf(lvf), BTRS_GUARD_REL, __builtin_fmaf(__builtin_fabsf(U0), 2e-6f, BTRS_GUARD_ABS + gf)) + e2;
    const bool valid = (j * P) * rV < 0.5f;
    const bool acc = lvf + g <= U0 - R, rej = lvf - g > U0;
    return !valid ? BTRS_OPEN : (acc ? BTRS_ACCEPT : (rej ? BTRS_REJECT : BTRS_OPEN));
}
#endif

// One attempt: the accepted k (>= 0) or -1.  screen: the device may decide by its binary32 estimates.
EPI_HD int32_t btrs_attempt(const BtrsConsts& C, double lpq, const LfTable& lf, bool screen, uint32_t wa, uint32_t wb)
{
    const float u = btrs_u(wa), us = btrs_us(u), v = btrs_v(wb);
    int32_t kf;
#ifdef EPI_HAVE_SCREEN
    float r_us;
    const bool sure = btrs_kfloor_screen(C, u, us, r_us, kf);
    if (!(screen && sure)) kf = btrs_kfloor_exact(C.a2, C.b, C.cfr, u, us);
#else
    (void)screen;
    kf = btrs_kfloor_exact(C.a2, C.b, C.cfr, u, us);
#endif
    const int32_t k = C.ci + kf;
    if ((uint32_t)k > (uint32_t)C.n) return -1;
    if (us >= 0.07f && (0.92f - v) * C.b >= 4.2f) return k;
#ifdef EPI_HAVE_SCREEN
    float lvf = 0.f;
    if (screen) lvf = btrs_lv_screen(C, v, r_us);
#endif
    const double ub = btrs_ub(C, k, lpq, lf);
#ifdef EPI_HAVE_SCREEN
    if (screen) {
        const double diff = (double)lvf - ub;
        const double guard = (double)__builtin_fmaf(__builtin_fabsf(lvf), BTRS_GUARD_REL, BTRS_GUARD_ABS);
        if (diff < -guard) return k;
        if (diff > guard) return -1;
    }
#endif
    return btrs_lv(v, us, C.a2, C.b, C.spq) <= ub ? k : -1;
}

#if defined(EPI_HAVE_SCREEN)
// One attempt in the order that suits a pass in which most lanes need the acceptance test, for n inside the log-factorial
// table (n < lf.n): the four loads go out as soon as the candidate is known (an out-of-range candidate reads entry 0 and
// is rejected afterwards) and the squeeze and the estimate of the left side are computed while they are in flight.
// Same decisions as btrs_attempt.
__device__ __forceinline__ int32_t btrs_attempt_tab(const BtrsConsts& C, double lpq, const double* __restrict__ lf_t, bool screen, uint32_t wa, uint32_t wb)
{
    const float u = btrs_u(wa), us = btrs_us(u), v = btrs_v(wb);
    float r_us; int32_t kf;
    const bool sure = btrs_kfloor_screen(C, u, us, r_us, kf);
    if (!(screen && sure)) kf = btrs_kfloor_exact(C.a2, C.b, C.cfr, u, us);
    const int32_t k = C.ci + kf;
    const bool inr = (uint32_t)k <= (uint32_t)C.n;
    const uint32_t ki = inr ? (uint32_t)k : 0u, ni = (uint32_t)C.n, mi = (uint32_t)C.m;
    // global address space and unsigned 32-bit indices: base in scalar registers + 32-bit offset (the table holds 2^20 doubles)
    typedef const char __attribute__((address_space(1)))* GlobalBytes;
    typedef const double __attribute__((address_space(1)))* GlobalDoubles;
    const GlobalBytes t = (GlobalBytes)lf_t;
    const double t0 = *(GlobalDoubles)(t + (mi << 3)), t1 = *(GlobalDoubles)(t + ((ni - mi) << 3));
    const double t2 = *(GlobalDoubles)(t + (ki << 3)), t3 = *(GlobalDoubles)(t + ((ni - ki) << 3));
    __builtin_amdgcn_sched_barrier(0);                                          // nothing below moves above the loads
    const bool squeeze = us >= 0.07f && (0.92f - v) * C.b >= 4.2f;
    const float lvf = btrs_lv_screen(C, v, r_us);
    const float grel = screen ? BTRS_GUARD_REL : __builtin_inff(), gabs = screen ? BTRS_GUARD_ABS : __builtin_inff();
    const double guard = (double)__builtin_fmaf(__builtin_fabsf(lvf), grel, gabs);
    const double ub = ((t0 + t1) - (t2 + t3)) + (double)(k - C.m) * lpq;
    const double diff = (double)lvf - ub;
    const bool acc = squeeze || diff < -guard;
    int32_t res = (inr && acc) ? k : -1;
    if (inr && !acc && !(diff > guard)) res = btrs_lv(v, us, C.a2, C.b, C.spq) <= ub ? k : -1;
    return res;
}
#endif

#if defined(EPI_HAVE_SCREEN)
// The per-draw constants for btrs_quick only: the square root is the hardware's (good to one ulp, where btrs_consts
// takes the correctly rounded one), so b, 2a and everything derived from them may differ from the defining values by a
// few parts in 10^7.  btrs_quick's decisions all carry bands that cover this.
__device__ __forceinline__ BtrsConsts btrs_consts_quick(int32_t n, const BtrsBin& B)
{
    BtrsConsts C;
    C.n = n;
    C.npq = (float)n * B.pq;
    C.spq = __builtin_amdgcn_sqrtf(C.npq);
    C.b = __builtin_fmaf(2.53f, C.spq, 1.15f);
    C.a2 = __builtin_fmaf(0.0496f, C.b, B.a0x2);
    const uint64_t prod = (uint64_t)(uint32_t)n * B.ppx + 0x80000000ULL;
    C.ci = (int32_t)(uint32_t)(prod >> 32);
    C.cfr = u2f(0x3f800000u | ((uint32_t)prod >> 9)) - 1.0f;
    const uint64_t pm = (uint64_t)((uint32_t)n + 1u) * B.ppx;
    C.m = (int32_t)(uint32_t)(pm >> 32);
    C.hf = 1.5f - u2f(0x3f800000u | ((uint32_t)pm >> 9));       // 1/2 - frac((n + 1) pp), the fraction to 23 bits
    C.rV = f_rcp(C.npq + B.pq);
    return C;
}

// An attempt without the table: ACCEPT (k valid), REJECT, or OPEN: btrs_attempt on the same words gives the outcome.
// (C from btrs_consts_quick.  The squeeze is claimed only with a margin of 1e-6 on its threshold; the candidate's band
// and the bands of the bounds are wide enough for the constants' few parts in 10^7 as they stand.)
__device__ __forceinline__ int btrs_quick(const BtrsConsts& C, const BtrsSide& S, uint32_t wa, uint32_t wb, int32_t& k)
{
    const float u = btrs_u(wa), us = btrs_us(u), v = btrs_v(wb);
    float r_us; int32_t kf;
    const bool sure = btrs_kfloor_screen(C, u, us, r_us, kf);
    k = C.ci + kf;
    const bool inr = (uint32_t)k <= (uint32_t)C.n;
    const bool squeeze = us >= 0.07f && (0.92f - v) * C.b >= 4.200005f;
#ifdef EPI_PHASE_STAMPS
    {
        const unsigned long long mm = __ballot(sure && inr && !squeeze), all = __ballot(true);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)all) - 1) { atomicAdd(&g_phase[6], (unsigned long long)__popcll(mm)); atomicAdd(&g_phase[7], (unsigned long long)__popcll(all)); }
    }
#endif
    const int bd = btrs_bounds(C, S, k, btrs_lv_screen(C, v, r_us));
    return !sure ? BTRS_OPEN : (!inr ? BTRS_REJECT : (squeeze ? BTRS_ACCEPT : bd));
}
#else
// (the host pass over the kernel bodies sees a declaration; only device code calls it)
__device__ int btrs_quick(const BtrsConsts& C, const BtrsSide& S, uint32_t wa, uint32_t wb, int32_t& k);
__device__ BtrsConsts btrs_consts_quick(int32_t n, const BtrsBin& B);
__device__ int32_t btrs_attempt_tab(const BtrsConsts& C, double lpq, const double* __restrict__ lf_t, bool screen, uint32_t wa, uint32_t wb);
#endif

// Word pair of attempt `att` (>= 1) of a retry stream: block (att-1)/2, words (0,1) or (2,3).
EPI_HD void retry_pair(const Rng& g, uint32_t retry_kind, uint32_t bin, uint32_t att, uint32_t& wa, uint32_t& wb)
{
    uint32_t w[4];
    rng_block(g, retry_kind, bin, (att - 1u) >> 1, w);
    const bool half = ((att - 1u) & 1u) != 0u;
    wa = half ? w[2] : w[0];
    wb = half ? w[3] : w[1];
}

// remaining attempts of a rejection draw, from attempt `att` (>= 1) on
EPI_HD int32_t btrs_finish(const BtrsConsts& C, double lpq, const LfTable& lf, bool screen, const Rng& g, uint32_t kind, uint32_t bin, uint32_t att)
{
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    bool have = false;
#pragma unroll 1
    for (; att < BINOM_MAX_ATTEMPTS; att++) {
        const bool half = ((att - 1u) & 1u) != 0u;
        if (!have || !half) { rng_block(g, kind, bin, (att - 1u) >> 1, w); have = true; }
        const int32_t k = btrs_attempt(C, lpq, lf, screen, half ? w[2] : w[0], half ? w[3] : w[1]);
        if (k >= 0) return k;
    }
    return C.m;                              // only if every attempt fails
}

// fa, fb: the attempt-0 word pair; attempts a >= 1 come from blocks of retry_kind.
// screen: allow the binary32 screens (device only; never changes the result).
EPI_HD int32_t binomial_draw(int32_t n, double p, uint32_t fa, uint32_t fb,
                             const Rng& g, uint32_t retry_kind, uint32_t bin, const LfTable& lf, bool screen = true)
{
    int32_t x; double pp; bool flip;
    const int cls = binom_classify(n, p, x, pp, flip);
    if (cls == BINOM_DONE) return x;
    if (cls == BINOM_INVERSION) x = binom_inversion(n, s_log(1.0 - pp), pp / (1.0 - pp), fa);
    else {
        const BtrsBin B = btrs_bin(pp);
        const BtrsConsts C = btrs_consts(n, B);
        x = btrs_attempt(C, B.lpq, lf, screen, fa, fb);
        if (x < 0) x = btrs_finish(C, B.lpq, lf, screen, g, retry_kind, bin, 1u);
    }
    return flip ? n - x : x;
}

// ---- standard normal: ziggurat (Marsaglia & Tsang 2000), 256 strips of equal area ----------------------------
// Strip k = 1..255 is the rectangle [0, x_k] x [f(x_k), f(x_{k+1})], f(x) = exp(-x^2 / 2), x_1 = r > x_2 > ... >
// x_256 = 0; strip 0 is [0, r] x [0, f(r)] plus the tail beyond r, with nominal width X_0 = area / f(r).  One 32-bit
// word gives the strip (bits 0-7), the sign (bit 8) and a 23-bit position j: x = fl(j) wn[k], wn[k] = X_k 2^-23.
// j < kn[k] (x inside the part of the strip that lies wholly under the curve): x is returned.  That is all the
// fast path does: one 8-byte table entry, a conversion, a multiplication and a comparison, for 98.9 % of the words.
// Otherwise (zig_slow): a wedge point is accepted iff f(x_k) + U (f(x_{k+1}) - f(x_k)) < f(x); the tail is
// Marsaglia's: x = -log(U1) / r, accepted iff -2 log(U2) > x^2, giving r + x; a rejected wedge point starts over
// with a fresh word, a rejected tail point repeats the tail step.  Step l of this takes Philox block l of the
// stream (K_ZIG, attempt << 6 | term): words 0, 1 -> U, U1 / U2 (32-bit binary64 uniforms), word 2 -> the next
// candidate word.  The slow path is binary64 throughout (s_exp, s_log as above).
constexpr int ZIG_N = 256;
constexpr uint32_t K_ZIG = 8u;
constexpr uint32_t ZIG_MAX_STEPS = 64u;
struct ZigEntry { uint32_t kn; float wn; };
struct ZigTable {
    ZigEntry e[ZIG_N];
    double fn[ZIG_N + 1];        // f(x_k), k = 1..256 (fn[0] unused)
    double r;
};

// (host) the table, from the strip area and r of the published 256-strip construction
inline void zig_build(ZigTable& T)
{
    const double r = 3.6541528853610088, v = 4.92867323399e-3;
    double x[ZIG_N + 1];
    x[1] = r;
    for (int k = 1; k < ZIG_N - 1; k++) x[k + 1] = __builtin_sqrt(-2.0 * s_log(v / x[k] + s_exp(-0.5 * x[k] * x[k])));
    x[ZIG_N] = 0.0;
    T.r = r;
    T.fn[0] = 0.0;
    for (int k = 1; k <= ZIG_N; k++) T.fn[k] = k == ZIG_N ? 1.0 : s_exp(-0.5 * x[k] * x[k]);
    for (int k = 0; k < ZIG_N; k++) {
        const double X = k == 0 ? v / T.fn[1] : x[k];
        const double inner = x[k + 1];                       // k = 0: r
        T.e[k].wn = (float)(X * 0x1p-23);
        // every position below kn lies under the curve whatever the rounding of wn and of the product
        T.e[k].kn = (uint32_t)((inner / (double)T.e[k].wn) * (1.0 - 0x1p-22));
    }
}

EPI_HD bool zig_fast(uint32_t w, const ZigEntry e, float& x)
{
    const uint32_t j = w >> 9;
    const float xa = (float)j * e.wn;
    x = u2f(f2u(xa) | ((w & 256u) << 23));
    return j < e.kn;
}

// everything else; term_att = attempt << 6 | infection term
EPI_HD_OUTLINE float zig_slow(const ZigTable* __restrict__ T, const Rng g, uint32_t term_att, uint32_t w)
{
    bool tail = false;
    uint32_t sign = 0u;
    float x = 0.f;
#pragma unroll 1
    for (uint32_t l = 0; l < ZIG_MAX_STEPS; l++) {
        uint32_t q[4];
        rng_block(g, K_ZIG, term_att, l, q);
        if (!tail) {
            const uint32_t k = w & 255u, j = w >> 9;
            sign = (w & 256u) << 23;
            x = (float)j * T->e[k].wn;
            if (j < T->e[k].kn) break;
            if (k == 0u) tail = true;
            else {
                const double xd = (double)x;
                const double y = T->fn[k] + u01(q[0]) * (T->fn[k + 1] - T->fn[k]);
                if (y < s_exp(-0.5 * xd * xd)) break;
                w = q[2];
                continue;
            }
        }
        const double xt = -s_log(u01(q[0])) / T->r, yt = -s_log(u01(q[1]));
        if (yt + yt > xt * xt) { x = (float)(T->r + xt); break; }
    }
    return u2f(f2u(x) | sign);
}

EPI_HD float zig_normal(const ZigTable* __restrict__ T, const Rng& g, uint32_t term_att, uint32_t w)
{
    float x;
    if (zig_fast(w, T->e[w & 255u], x)) return x;
    return zig_slow(T, g, term_att, w);
}

// ---- Gamma(shape, scale) --------------------------------------------------------
// Marsaglia-Tsang (2000); shape < 1 boosted by U^(1/shape); shape < 2^-30 -> 0.
//   gamma_consts  : per-draw constants
//   gamma_attempt : one attempt from one normal and one word: true if accepted
//   gamma_finish  : boost and scale of an accepted attempt
constexpr uint32_t GAMMA_MAX_ATTEMPTS = 512u;

struct GammaConsts {
    double shape, d;
    float c;
    bool boost, screen;
};

EPI_HD bool gamma_is_zero(double shape) { return !(shape >= 0x1p-30); }

// c >= 1 / sqrt(9 d) with 1 <= 9 d c^2 <= 1 + 1.4e-5, in binary32: two Newton steps for the reciprocal square root
// from the bit-shift seed (relative error 3.4e-2 -> 1.8e-3 -> 4.8e-6, the iterates below the root) and a factor
// 1 + 6e-6.  Marsaglia-Tsang's rejection is exact for every c >= 1 / sqrt(9 d): with v = (1 + c x)^3,
// h(x) = d (log v - v + 1) <= -(9 d c^2) x^2 / 2 <= -x^2 / 2, so the normal envelope dominates, and d v has the
// Gamma(d + 1/3) density whatever c is; c = 1 / sqrt(9 d) only maximises the acceptance rate.  The squeeze has to
// allow for the excess: with 9 d c^2 = 1 + 2 e the acceptance probability is exp(-e x^2) times the one at
// x' = x sqrt(1 + 2 e), which is at least 1 - 0.0331 x'^4, hence >= 1 - 0.03311 x^4 - 1.4e-5 x^2.
EPI_HD float gamma_c(float d)
{
    const float x9 = 9.0f * d, h = 4.5f * d;
    float y = u2f(0x5f3759dfu - (f2u(x9) >> 1));
    y = y * (1.5f - h * (y * y));
    y = y * (1.5f - h * (y * y));
    return y * 1.000006f;
}

EPI_HD GammaConsts gamma_consts(double shape, bool screen)
{
    GammaConsts G;
    G.shape = shape;
    G.boost = shape < 1.0;
    const double al = G.boost ? shape + 1.0 : shape;
    G.d = al - 0.3333333333333333;
    G.c = gamma_c((float)G.d);
    G.screen = screen;
    return G;
}

EPI_HD float gamma_u(uint32_t wu) { return (float)wu * 0x1p-32f; }

// The squeeze: true = accepted without a logarithm (v valid).  t = 1 + c x <= 0 is never accepted.
// With s = c x and 9 d c^2 = 1 + 2 e the exponent of the acceptance probability is
//   x^2/2 + d (1 - v + log v) = -e x^2 + 3 d (log(1 + s) - s + s^2/2 - s^3/3) = -e x^2 - 3 d int_0^s r^3 / (1 + r) dr,
// and the integral is at most s^4 / 4 for s >= 0 and s^4 / (4 (1 - |s|)) <= (s^4 / 4)(1 + 2 |s|) for -1/2 <= s < 0, so
// the probability is at least 1 - A,  A = 7e-6 x^2 + (c^2 x^4 / 12)(1 + 1.4e-5)(1 + 2 |s|)   (3 d c^2 / 4 = (1 + 2 e) / 12).
// For the large shapes of this model (d in the thousands) A is of the order 1e-5: the logarithms are almost never
// needed.  Marsaglia and Tsang's own 1 - 0.0331 x^4 (here with the allowance for e) serves where it is the larger.
EPI_HD float gamma_squeeze(float c, float xf)
{
    const float x2 = xf * xf, x4 = x2 * x2;
    const float as = __builtin_fabsf(c * xf);
    const float a_mt = __builtin_fmaf(0.03311f, x4, 1.4e-5f * x2);
    const float a_d = __builtin_fmaf(((c * c) * x4) * 0.08334f, __builtin_fmaf(2.0f, as, 1.0f), 7e-6f * x2);
    const float a = (as <= 0.5f && a_d < a_mt) ? a_d : a_mt;
    return 0.9999997f - a;
}

EPI_HD bool gamma_fast(const GammaConsts& G, float xf, uint32_t wu, double& v)
{
    const double t = __builtin_fma((double)G.c, (double)xf, 1.0);
    v = t * t * t;
    return t > 0.0 && gamma_u(wu) < gamma_squeeze(G.c, xf);
}

#ifdef EPI_HAVE_SCREEN
// binary32 estimate of log u - [x^2/2 + d (1 - v + log v)] (accept iff < 0) for t = c x > -0.9, and the sum
// `mag` of the magnitudes of its terms.  The estimate takes 9 d c^2 = 1, i.e. x^2 / 2 = 4.5 d t^2, which the
// constants above satisfy to 1.4e-5 (an error below 7e-6 x^2, covered by the x^2 term of the band); otherwise the
// error is below 1.5e-6 mag + 1e-6 by rounding analysis.
constexpr float GAMMA_GUARD_REL = 5e-6f, GAMMA_GUARD_ABS = 1e-4f, GAMMA_GUARD_X2 = 2e-5f;
__device__ __forceinline__ float gamma_screen(float tf, float df, float uf, float& mag)
{
    const float t2 = tf * tf;
    float q = 0.1f;
    q = __builtin_fmaf(q, tf, -0.111111111f);
    q = __builtin_fmaf(q, tf, 0.125f);
    q = __builtin_fmaf(q, tf, -0.142857143f);
    q = __builtin_fmaf(q, tf, 0.166666667f);
    q = __builtin_fmaf(q, tf, -0.2f);
    q = __builtin_fmaf(q, tf, 0.25f);
    const float ser = -(t2 * t2) * q;                                 // log1p(t) - t + t^2/2 - t^3/3
    const float l1p = f_ln(1.0f + tf);
    const float big = l1p - tf + 0.5f * t2 - 0.333333333f * t2 * tf;
    const float magb = __builtin_fabsf(l1p) + __builtin_fabsf(tf) + 0.5f * t2 + 0.333333333f * __builtin_fabsf(t2 * tf);
    const bool small = __builtin_fabsf(tf) <= 0.125f;
    const float rhs = 3.0f * df * (small ? ser : big);
    const float lu = f_ln(uf);
    mag = __builtin_fabsf(lu) + 3.0f * df * (small ? __builtin_fabsf(ser) : magb);
    return lu - rhs;
}
#endif

EPI_HD bool gamma_attempt(const GammaConsts& G, float xf, uint32_t wu, double& v)
{
    double vv;
    const bool fast = gamma_fast(G, xf, wu, vv);
    v = vv;
    if (fast) return true;
    const double x = (double)xf;
    const double t = __builtin_fma((double)G.c, x, 1.0);
    if (!(t > 0.0)) return false;
    const float uf = gamma_u(wu);
    if (!(uf > 0.0f)) return true;
    const double x2 = x * x;
#ifdef EPI_HAVE_SCREEN
    if (G.screen) {
        const float tf = G.c * xf;
        if (tf > -0.9f) {
            float mag;
            const float diff = gamma_screen(tf, (float)G.d, uf, mag);
            const float guard = __builtin_fmaf(xf * xf, GAMMA_GUARD_X2, __builtin_fmaf(mag, GAMMA_GUARD_REL, GAMMA_GUARD_ABS));
            if (diff < -guard) return true;
            if (diff > guard) return false;
        }
    }
#endif
    return s_log((double)uf) < 0.5 * x2 + G.d * (1.0 - vv + s_log(vv));
}

// U^(1/shape) of the boost for shape < 1 (out of line on the device, for the same reason as binom_inversion)
// (the generator by value: a reference would put the caller's copy on the stack, 20 bytes per lane stored by every step)
EPI_HD_OUTLINE double gamma_boost(double shape, const Rng g, uint32_t bin)
{
    uint32_t wb[4];
    rng_block(g, K_BOOST, bin, 0, wb);
    return s_exp(s_log(u01(wb[0])) / shape);
}

EPI_HD double gamma_finish(const GammaConsts& G, double v, double scale, const Rng& g, uint32_t bin)
{
    double gval = G.d * v;
    if (G.boost) gval = gval * gamma_boost(G.shape, g, bin);
    return gval * scale;
}

// words of attempt 0 of infection term `bin` (shared block of the pair bin/2): candidate word of the normal, word of the uniform
EPI_HD void gamma_first(const Rng& g, uint32_t bin, uint32_t& wz, uint32_t& wu)
{
    uint32_t w[4];
    rng_block(g, K_GAMMA, bin >> 1, 0, w);
    wz = (bin & 1u) ? w[1] : w[0];
    wu = (bin & 1u) ? w[3] : w[2];
}

// words of attempt att >= 1
EPI_HD void gamma_retry(const Rng& g, uint32_t bin, uint32_t att, uint32_t& wz, uint32_t& wu)
{
    uint32_t w[4];
    rng_block(g, K_GRETRY, bin, (att - 1u) >> 1, w);
    const bool half = ((att - 1u) & 1u) != 0u;
    wz = half ? w[1] : w[0];
    wu = half ? w[3] : w[2];
}

// remaining attempts of a gamma draw from attempt `att` (>= 1) on, then boost and scale
EPI_HD double gamma_finish_from(const GammaConsts& G, double scale, const ZigTable* __restrict__ Z, const Rng& g, uint32_t bin, uint32_t att)
{
    double v = 1.0, vv;                        // v = 1 only if every attempt fails
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    bool have = false;
#pragma unroll 1
    for (; att < GAMMA_MAX_ATTEMPTS; att++) {
        const bool half = ((att - 1u) & 1u) != 0u;
        if (!have || !half) { rng_block(g, K_GRETRY, bin, (att - 1u) >> 1, w); have = true; }
        const float xf = zig_normal(Z, g, (att << 6) | bin, half ? w[1] : w[0]);
        if (gamma_attempt(G, xf, half ? w[3] : w[2], vv)) { v = vv; break; }
    }
    return gamma_finish(G, v, scale, g, bin);
}

// wz, wu: the attempt-0 words (gamma_first)
EPI_HD double gamma_draw(double shape, double scale, uint32_t wz, uint32_t wu, const ZigTable* __restrict__ Z, const Rng& g, uint32_t bin, bool screen = true)
{
    if (gamma_is_zero(shape)) return 0.0;
    const GammaConsts G = gamma_consts(shape, screen);
    double vv;
    if (gamma_attempt(G, zig_normal(Z, g, bin, wz), wu, vv)) return gamma_finish(G, vv, scale, g, bin);
    return gamma_finish_from(G, scale, Z, g, bin, 1u);
}

// 53-bit uniform in [0,1) of the systematic resampler for (seed, step)
EPI_HD double resample_offset(uint64_t seed, uint32_t step)
{
    Rng g = rng_make(seed, 0, step);
    uint32_t w[4];
    rng_block(g, K_RESAMP, 0, 0, w);
    uint64_t hi = w[0] >> 5, lo = w[1] >> 6;
    return (double)((hi << 26) | lo) * 0x1p-53;
}

} // namespace epi
