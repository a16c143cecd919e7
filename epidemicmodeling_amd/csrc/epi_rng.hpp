// Counter-based random variates for the gfx950 particle kernels.
//
// Every variate is a pure function of (seed, global particle slot, step,
// stream tag, attempt): a Philox4x32-10 block addressed by
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
constexpr uint32_t K_GAMMA  = 3u;  // bin = pair j: attempt 0 of infection term 2j (normal 0, word 2) and of term 2j+1 (normal 1, word 3)
constexpr uint32_t K_GRETRY = 7u;  // bin = term, blk b: its attempts 1+2b (normal 0, word 2) and 2+2b (normal 1, word 3)
constexpr uint32_t K_BOOST  = 4u;  // word 0: uniform of the shape<1 boost
constexpr uint32_t K_RESAMP = 5u;  // slot 0: words 0,1 -> 53-bit systematic-resampling offset
constexpr uint32_t K_INIT   = 6u;  // initial-state generators

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

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).
EPI_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t (&w)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
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
    philox4x32_10(g.c0, g.c1, g.c2, (kind << 24) | (bin << 8) | blk, g.k0, g.k1, w);
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

// The defining acceptance test of the transformed-rejection sampler (Hoermann 1993, step 3):
//   log(v alpha / (a/us^2 + b))  <=  log f(k) - log f(m)
//                                 =  [lfact(m) + lfact(n-m)] - [lfact(k) + lfact(n-k)] + (k - m) log(p/q).
// btrs_lv is the left side (out of line on the device: it runs for a few attempts in a hundred thousand).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__
#else
static inline
#endif
double btrs_lv(double v, double us, double a, double b, double alpha)
{
    return s_log(v * alpha / (a / (us * us) + b));
}

// A Binomial(n, p) draw is split into pieces so that a kernel can either loop over attempts
// per thread (binomial_draw) or hand failed attempts to other threads:
//   binom_classify : trivial result / inversion / rejection, and the folded probability
//   binom_inversion: n min(p,1-p) < 10, one uniform, always completes
//   btrs_consts    : per-draw constants of the rejection sampler
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

// 1/x for the inversion recurrence (x <= 64; beyond that a division)
#define EPI_R(x) (1.0 / (double)(x))
#define EPI_R8(b) EPI_R(b), EPI_R(b + 1), EPI_R(b + 2), EPI_R(b + 3), EPI_R(b + 4), EPI_R(b + 5), EPI_R(b + 6), EPI_R(b + 7)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#else
static
#endif
const double EPI_RECIP[72] = { 0.0, EPI_R(1), EPI_R(2), EPI_R(3), EPI_R(4), EPI_R(5), EPI_R(6), EPI_R(7),
    EPI_R8(8), EPI_R8(16), EPI_R8(24), EPI_R8(32), EPI_R8(40), EPI_R8(48), EPI_R8(56), EPI_R8(64) };

// lnq = s_log(1 - pp), s = pp / (1 - pp): per-probability constants, tabulated per age bin by the kernels.
// (Out of line on the device: inlined into a loop over bins, the constants of its exponential are hoisted
// into vector registers that the loop's common path then lacks.)
EPI_HD_OUTLINE int32_t binom_inversion(int32_t n, double lnq, double s, uint32_t wa)
{
    const double nd = (double)n;
    const double a1 = nd + 1.0;
    double r = s_exp(nd * lnq);
    double u = u01(wa);
    int32_t res = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // the same recurrence with a wave-uniform counter, so that 1/x is a scalar table read
    bool live = u > r && 0 < n;
#pragma unroll 1
    for (int32_t x = 1; __any(live); x++) {
        const double xd = (double)x;
        const double t = x <= 64 ? EPI_RECIP[x] : 1.0 / xd;
        if (live) {
            u -= r;
            res = x;
            r *= s * ((a1 - xd) * t);
            live = u > r && x < n;
        }
    }
#else
    while (u > r && res < n) {
        u -= r;
        res += 1;
        const double xd = (double)res;
        const double t = res <= 64 ? EPI_RECIP[res] : 1.0 / xd;
        r *= s * ((a1 - xd) * t);
    }
#endif
    return res;
}

struct BtrsConsts {
    double nd, pp, a, b, c, vr, alpha, m, lpq, spq;
    LfTable lf;
#ifdef EPI_HAVE_SCREEN
    double cfl;                     // floor(c), and below its fraction: the k estimate works on small numbers
    float af, bf, alf, cfr;
    bool scr;
#endif
};

// The constants come in two parts, so that a kernel can start the table loads of the acceptance test between them:
// btrs_consts_k fills what the candidate k needs, btrs_consts_rest the squeeze bound and the hat's area (one binary32
// quotient).  lpq = s_log(pp / (1 - pp)) (a per-probability constant, tabulated per age bin by the kernels).
EPI_HD void btrs_consts_k(BtrsConsts& C, int32_t n, double pp, double lpq, const LfTable& lf, bool screen)
{
    C.nd = (double)n; C.pp = pp; C.lpq = lpq; C.lf = lf;
    // hat parameters (Hoermann's 3-digit constants tolerate binary32 sqrt / reciprocal)
    C.spq = (double)sqrt_rn((float)(C.nd * pp * (1.0 - pp)));
    C.b = 1.15 + 2.53 * C.spq;
    C.a = -0.0873 + 0.0248 * C.b + 0.01 * pp;
    C.c = C.nd * pp + 0.5;
    C.m = __builtin_floor((C.nd + 1.0) * pp);
#ifdef EPI_HAVE_SCREEN
    C.scr = screen;
    C.af = (float)C.a; C.bf = (float)C.b;
    C.cfl = __builtin_floor(C.c);
    C.cfr = (float)(C.c - C.cfl);
#else
    (void)screen;
#endif
}

EPI_HD void btrs_consts_rest(BtrsConsts& C)
{
    const double ib = (double)div_rn(1.0f, (float)C.b);
    C.vr = 0.92 - 4.2 * ib;
    C.alpha = (2.83 + 5.1 * ib) * C.spq;
#ifdef EPI_HAVE_SCREEN
    C.alf = (float)C.alpha;
#endif
}

EPI_HD BtrsConsts btrs_consts(int32_t n, double pp, double lpq, const LfTable& lf, bool screen)
{
    BtrsConsts C;
    btrs_consts_k(C, n, pp, lpq, lf, screen);
    btrs_consts_rest(C);
    return C;
}

#ifdef EPI_HAVE_SCREEN
// Guard band of the binary32 estimate of btrs_lv: its argument carries a relative error below 12 roundings
// (7e-7), the hardware log2 one ulp of its result, so |estimate - exact| < 1e-6 + 3e-7 |lv|; the band is ten
// times that (tools/ubench.hip measures the actual worst case).
constexpr float BTRS_GUARD_REL = 3e-6f, BTRS_GUARD_ABS = 1e-5f;
__device__ __forceinline__ float btrs_lv_screen(const BtrsConsts& C, double us, double v)
{
    const float usf = (float)us;
    return f_ln((float)v * C.alf * f_rcp(__builtin_fmaf(C.af, f_rcp(usf * usf), C.bf)));
}
#endif

// right side of the acceptance test for an in-range k
EPI_HD double btrs_ub(const BtrsConsts& C, double kd)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (C.nd < (double)C.lf.n) {            // every argument is inside the table: four loads, no branches
        const double* __restrict__ t = C.lf.t;
        const int32_t ni = (int32_t)C.nd, mi = (int32_t)C.m, ki = (int32_t)kd;
        return ((t[mi] + t[ni - mi]) - (t[ki] + t[ni - ki])) + (kd - C.m) * C.lpq;
    }
#endif
    return ((lfact_of(C.lf, C.m) + lfact_of(C.lf, C.nd - C.m)) - (lfact_of(C.lf, kd) + lfact_of(C.lf, C.nd - kd))) + (kd - C.m) * C.lpq;
}

// k = floor((2a/us + b) u + c) of one attempt
EPI_HD double btrs_k(const BtrsConsts& C, double u, double us)
{
    double kd = 0.0;
    bool have_k = false;
#ifdef EPI_HAVE_SCREEN
    if (C.scr) {
        // k = floor((2a/us + b) u + c) estimated in binary32; trusted when the value is
        // further from an integer than many times its error bound
        const float t = __builtin_fmaf(__builtin_fmaf(2.0f * C.af, f_rcp((float)us), C.bf), (float)u, C.cfr);
        const float tf = __builtin_floorf(t);
        const float fr = t - tf;
        const float dl = __builtin_fmaf(__builtin_fabsf(t) + C.bf, 4e-6f, 1e-3f);
        if (fr > dl && fr < 1.0f - dl) { kd = C.cfl + (double)tf; have_k = true; }
    }
#endif
    if (!have_k) kd = __builtin_floor((2.0 * C.a / us + C.b) * u + C.c);
    return kd;
}

// one attempt: returns the accepted k (>= 0) or -1
EPI_HD int32_t btrs_attempt(const BtrsConsts& C, uint32_t wa, uint32_t wb)
{
    const double u = u01_centred(wa);
    const double v = u01(wb);
    const double us = 0.5 - __builtin_fabs(u);
    const double kd = btrs_k(C, u, us);
    if (kd < 0.0 || kd > C.nd) return -1;
    if (us >= 0.07 && v <= C.vr) return (int32_t)kd;
#if defined(EPI_HAVE_SCREEN)
    // the table loads of the right side are issued first and the binary32 estimate of the left side, which does not
    // depend on them, is computed while they are in flight
    if (C.nd < (double)C.lf.n) {
        const double* __restrict__ t = C.lf.t;
        const int32_t ni = (int32_t)C.nd, mi = (int32_t)C.m, ki = (int32_t)kd;
        const double t0 = t[mi], t1 = t[ni - mi], t2 = t[ki], t3 = t[ni - ki];
        const float lvf = btrs_lv_screen(C, us, v);
        // with the screens switched off the band is infinite: every attempt takes the defining expression
        const double guard = C.scr ? (double)__builtin_fmaf(__builtin_fabsf(lvf), BTRS_GUARD_REL, BTRS_GUARD_ABS) : __builtin_inf();
        const double lvd = (double)lvf, lin = (kd - C.m) * C.lpq;
        const double ub = ((t0 + t1) - (t2 + t3)) + lin;
        const double diff = lvd - ub;
        if (diff < -guard) return (int32_t)kd;
        if (diff > guard) return -1;
        return btrs_lv(v, us, C.a, C.b, C.alpha) <= ub ? (int32_t)kd : -1;
    }
#endif
    const double ub = btrs_ub(C, kd);
#ifdef EPI_HAVE_SCREEN
    if (C.scr) {
        const float lvf = btrs_lv_screen(C, us, v);
        const double diff = (double)lvf - ub;
        const double guard = (double)__builtin_fmaf(__builtin_fabsf(lvf), BTRS_GUARD_REL, BTRS_GUARD_ABS);
        if (diff < -guard) return (int32_t)kd;
        if (diff > guard) return -1;
    }
#endif
    return btrs_lv(v, us, C.a, C.b, C.alpha) <= ub ? (int32_t)kd : -1;
}

#if defined(EPI_HAVE_SCREEN)
// The constants and one attempt in the order that suits the device.  When every argument of the acceptance test lies
// inside the log-factorial table, every lane runs one straight sequence: a wave executes the acceptance test of nearly
// every draw anyway (some lane is outside the squeeze region), and without the branches the four table loads can be
// issued as soon as k is known; the second part of the constants, the squeeze test and the binary32 estimate of the
// left side are computed while the loads are in flight.  Same operations on the same values as btrs_consts +
// btrs_attempt, hence the same result; C is complete on return (for btrs_finish).
__device__ __forceinline__ int32_t btrs_try(BtrsConsts& C, int32_t n, double pp, double lpq, const LfTable& lf, bool screen,
                                            uint32_t wa, uint32_t wb)
{
    btrs_consts_k(C, n, pp, lpq, lf, screen);
    if (C.nd < (double)lf.n) {
        const double u = u01_centred(wa);
        const double us = 0.5 - __builtin_fabs(u);
        const double kd = btrs_k(C, u, us);
        const bool inr = !(kd < 0.0 || kd > C.nd);
        // global address space and unsigned 32-bit indices: base in scalar registers + 32-bit offset, no 64-bit address arithmetic
        typedef const char __attribute__((address_space(1)))* GlobalBytes;
        typedef const double __attribute__((address_space(1)))* GlobalDoubles;
        const GlobalBytes t = (GlobalBytes)lf.t;
        const uint32_t ni = (uint32_t)n, mi = (uint32_t)(int32_t)C.m;
        const int32_t ki = inr ? (int32_t)kd : 0;                                   // an out-of-range k is rejected below
        // (the table holds 2^20 doubles: byte offsets fit 32 bits)
        const double t0 = *(GlobalDoubles)(t + (mi << 3)), t1 = *(GlobalDoubles)(t + ((ni - mi) << 3));
        const double t2 = *(GlobalDoubles)(t + ((uint32_t)ki << 3)), t3 = *(GlobalDoubles)(t + ((ni - (uint32_t)ki) << 3));
        __builtin_amdgcn_sched_barrier(0);                                          // nothing below moves above the loads
        btrs_consts_rest(C);
        const double v = u01(wb);
        const bool squeeze = us >= 0.07 && v <= C.vr;
        const float lvf = btrs_lv_screen(C, us, v);
        // with the screens switched off the band is infinite (or NaN, which no comparison passes): every attempt outside the
        // squeeze takes the defining expression.  `screen` is wave-uniform, so the two constants are chosen by scalar selects.
        const float grel = screen ? BTRS_GUARD_REL : __builtin_inff(), gabs = screen ? BTRS_GUARD_ABS : __builtin_inff();
        const double guard = (double)__builtin_fmaf(__builtin_fabsf(lvf), grel, gabs);
        const double lvd = (double)lvf, lin = (kd - C.m) * C.lpq;
        const double ub = ((t0 + t1) - (t2 + t3)) + lin;
        const double diff = lvd - ub;
        const bool acc = squeeze || diff < -guard;
        int32_t res = (inr && acc) ? ki : -1;
        if (inr && !acc && !(diff > guard)) res = btrs_lv(v, us, C.a, C.b, C.alpha) <= ub ? ki : -1;
        return res;
    }
    btrs_consts_rest(C);
    return btrs_attempt(C, wa, wb);
}
#else
// (the host pass over the kernel bodies, and a host restatement, see the plain composition)
EPI_HD int32_t btrs_try(BtrsConsts& C, int32_t n, double pp, double lpq, const LfTable& lf, bool screen, uint32_t wa, uint32_t wb)
{
    C = btrs_consts(n, pp, lpq, lf, screen);
    return btrs_attempt(C, wa, wb);
}
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
EPI_HD int32_t btrs_finish(const BtrsConsts& C, const Rng& g, uint32_t kind, uint32_t bin, uint32_t att)
{
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    bool have = false;
#pragma unroll 1
    for (; att < BINOM_MAX_ATTEMPTS; att++) {
        const bool half = ((att - 1u) & 1u) != 0u;
        if (!have || !half) { rng_block(g, kind, bin, (att - 1u) >> 1, w); have = true; }
        const int32_t k = btrs_attempt(C, half ? w[2] : w[0], half ? w[3] : w[1]);
        if (k >= 0) return k;
    }
    return (int32_t)C.m;                     // only if every attempt fails
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
        const BtrsConsts C = btrs_consts(n, pp, s_log(pp / (1.0 - pp)), lf, screen);
        x = btrs_attempt(C, fa, fb);
        if (x < 0) x = btrs_finish(C, g, retry_kind, bin, 1u);
    }
    return flip ? n - x : x;
}

// ---- Gamma(shape, scale) --------------------------------------------------------
// Marsaglia-Tsang (2000); shape < 1 boosted by U^(1/shape); shape < 2^-30 -> 0.
//   gamma_consts  : per-draw constants
//   gamma_attempt : one attempt from one normal and one word: true if accepted
//   gamma_finish  : boost and scale of an accepted attempt
constexpr uint32_t GAMMA_MAX_ATTEMPTS = 512u;

struct GammaConsts {
    double shape, d, c;
    bool boost, screen;
};

EPI_HD bool gamma_is_zero(double shape) { return !(shape >= 0x1p-30); }

// c >= 1 / sqrt(9 d), and within 3e-10 of it.  Marsaglia-Tsang's rejection is exact for every such c: with
// v = (1 + c x)^3, h(x) = d (log v - v + 1) <= -(9 d c^2) x^2 / 2 <= -x^2 / 2, so the normal envelope dominates, and
// d v has the Gamma(d + 1/3) density whatever c is.  c = 1 / sqrt(9 d) only maximises the acceptance rate, so three
// Newton steps for the reciprocal square root (plain multiplies and subtractions, from the bit-shift seed; the
// iterates approach the root from below) and a final factor 1 + 2^-32 replace the square root and the division.
// The squeeze u < 1 - 0.0331 x^4 then overstates the acceptance probability by at most 2.4e-10 x^2, the
// granularity of the 32-bit uniform u itself.
EPI_HD double gamma_c(double d)
{
    const double h = 4.5 * d;                                  // (9 d) / 2
    double y = u2d(0x5FE6EB50C7B537A9ULL - (d2u(9.0 * d) >> 1));
    y = y * (1.5 - h * (y * y));
    y = y * (1.5 - h * (y * y));
    y = y * (1.5 - h * (y * y));
    return y * 1.00000000023283064365386962890625;             // 1 + 2^-32
}

EPI_HD GammaConsts gamma_consts(double shape, bool screen)
{
    GammaConsts G;
    G.shape = shape;
    G.boost = shape < 1.0;
    const double al = G.boost ? shape + 1.0 : shape;
    G.d = al - 0.3333333333333333;
    G.c = gamma_c(G.d);
    G.screen = screen;
    return G;
}

#ifdef EPI_HAVE_SCREEN
// binary32 estimate of log u - [x^2/2 + d (1 - v + log v)] (accept iff < 0) for t = c x > -0.9, and the sum
// `mag` of the magnitudes of its terms; estimate error < 1.5e-6 mag + 1e-6 by the same rounding analysis
// (measured over 2e8 undecided attempts, shapes 1 to 1e8: never above 2e-7 mag + 5e-6).
constexpr float GAMMA_GUARD_REL = 5e-6f, GAMMA_GUARD_ABS = 1e-4f;
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
    const double x = (double)xf;
    const double u = u01(wu);
    const double t = 1.0 + G.c * x;
    if (!(t > 0.0)) return false;
    const double vv = t * t * t;
    const double x2 = x * x;
    v = vv;
    if (u < 1.0 - 0.0331 * (x2 * x2)) return true;
#ifdef EPI_HAVE_SCREEN
    if (G.screen) {
        // x^2/2 + d (1 - v + log v) = 3 d (log1p(t) - t + t^2/2 - t^3/3), t = c x  (since 9 d c^2 = 1)
        const float tf = (float)G.c * xf;
        if (tf > -0.9f) {
            float mag;
            const float diff = gamma_screen(tf, (float)G.d, (float)u, mag);
            const float guard = __builtin_fmaf(mag, GAMMA_GUARD_REL, GAMMA_GUARD_ABS);
            if (diff < -guard) return true;
            if (diff > guard) return false;
        }
    }
#endif
    return s_log(u) < 0.5 * x2 + G.d * (1.0 - vv + s_log(vv));
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

// attempt-0 normal and word of infection term `bin` (shared block of the pair bin/2)
EPI_HD void gamma_first(const Rng& g, uint32_t bin, float& zf, uint32_t& wu)
{
    uint32_t w[4];
    float z0, z1;
    rng_block(g, K_GAMMA, bin >> 1, 0, w);
    box_muller(w[0], w[1], z0, z1);
    zf = (bin & 1u) ? z1 : z0;
    wu = (bin & 1u) ? w[3] : w[2];
}

// normal and word of attempt att >= 1
EPI_HD void gamma_retry(const Rng& g, uint32_t bin, uint32_t att, float& zf, uint32_t& wu)
{
    uint32_t w[4];
    float z0, z1;
    rng_block(g, K_GRETRY, bin, (att - 1u) >> 1, w);
    box_muller(w[0], w[1], z0, z1);
    const bool half = ((att - 1u) & 1u) != 0u;
    zf = half ? z1 : z0;
    wu = half ? w[3] : w[2];
}

// remaining attempts of a gamma draw from attempt `att` (>= 1) on, then boost and scale
EPI_HD double gamma_finish_from(const GammaConsts& G, double scale, const Rng& g, uint32_t bin, uint32_t att)
{
    double v = 1.0, vv;                        // v = 1 only if every attempt fails
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    float z0 = 0.f, z1 = 0.f;
    bool have = false;
#pragma unroll 1
    for (; att < GAMMA_MAX_ATTEMPTS; att++) {
        const bool half = ((att - 1u) & 1u) != 0u;
        if (!have || !half) { rng_block(g, K_GRETRY, bin, (att - 1u) >> 1, w); box_muller(w[0], w[1], z0, z1); have = true; }
        if (gamma_attempt(G, half ? z1 : z0, half ? w[3] : w[2], vv)) { v = vv; break; }
    }
    return gamma_finish(G, v, scale, g, bin);
}

// zf, wu: the attempt-0 normal and word (gamma_first)
EPI_HD double gamma_draw(double shape, double scale, float zf, uint32_t wu, const Rng& g, uint32_t bin, bool screen = true)
{
    if (gamma_is_zero(shape)) return 0.0;
    const GammaConsts G = gamma_consts(shape, screen);
    double vv;
    if (gamma_attempt(G, zf, wu, vv)) return gamma_finish(G, vv, scale, g, bin);
    return gamma_finish_from(G, scale, g, bin, 1u);
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
