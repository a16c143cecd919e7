// Micro-benchmarks of the sampler building blocks on gfx950 (developer tool, not part of the library).
// Prints ns per draw per lane and the equivalent number of full-rate lane-instruction slots
// (256 CUs x 4 SIMD x 32 lanes x clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../epidemicmodeling_amd/csrc/epi_rng.hpp"
using namespace epi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int REP = 16;

__global__ __launch_bounds__(256) void k_philox(uint64_t seed, uint32_t* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Rng g = rng_make(seed, i, 0);
    uint32_t acc = 0;
    for (int r = 0; r < REP; r++) { uint32_t w[4]; rng_block(g, 0, r, 0, w); acc ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_bm(uint64_t seed, float* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Rng g = rng_make(seed, i, 0);
    float acc = 0;
    for (int r = 0; r < REP; r++) { uint32_t w[4]; rng_block(g, 0, r, 0, w); float z0, z1; box_muller(w[0], w[1], z0, z1); acc += z0 + z1; }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_binom(uint64_t seed, const int32_t* nn, double p, int32_t* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Rng g = rng_make(seed, i, 0);
    int32_t acc = 0, nv = nn[i];
    for (int r = 0; r < REP; r++) {
        uint32_t w[4]; rng_block(g, K_BINOM0, r, 0, w);
        acc += binomial_draw(nv, p, w[0], w[1], g, K_SYMPT, r);
    }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_gamma(uint64_t seed, const double* sh, double* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Rng g = rng_make(seed, i, 0);
    double acc = 0, s = sh[i];
    for (int r = 0; r < REP; r++) { float zf; uint32_t wu; gamma_first(g, r, zf, wu); acc += gamma_draw(s, 0.1, zf, wu, g, r); }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_log(const double* x, double* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = x[i], acc = 0;
    for (int r = 0; r < REP; r++) { acc += s_log(v); v += 1.0; }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_div(const double* x, double* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = x[i], acc = 1.0;
    for (int r = 0; r < REP; r++) { acc = 3.0 / (acc + v); }
    out[i] = acc;
}

template <class F> double timeit(F f, int iters = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main()
{
    const int64_t n = 1 << 24;
    const double lanes = 256.0 * 4 * 32, clk = 2.4e9;
    void *out, *in;
    CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&in, n * 8));
    dim3 grid((unsigned)(n / 256)), blk(256);
    auto report = [&](const char* name, double ms) {
        double ns = ms * 1e6 / ((double)n * REP);
        printf("%-34s %8.3f ms  %8.4f ns/draw/chip  %9.1f lane-slots/draw\n", name, ms, ns, ns * 1e-9 * lanes * clk);
    };
    report("philox block", timeit([&] { k_philox<<<grid, blk>>>(1, (uint32_t*)out, n); }));
    report("philox + box-muller pair", timeit([&] { k_bm<<<grid, blk>>>(1, (float*)out, n); }));
    std::vector<double> hx(n); for (int64_t i = 0; i < n; i++) hx[i] = 1.0 + (i % 1000) * 0.37;
    CK(hipMemcpy(in, hx.data(), n * 8, hipMemcpyHostToDevice));
    report("s_log (fp64)", timeit([&] { k_log<<<grid, blk>>>((double*)in, (double*)out, n); }));
    report("fp64 divide (dependent)", timeit([&] { k_div<<<grid, blk>>>((double*)in, (double*)out, n); }));
    std::vector<int32_t> hn(n);
    struct { const char* name; int base, spread; double p; } cases[] = {
        {"binom n~2000 p=.56 (BTRS)", 1000, 2000, 0.56}, {"binom n~10000 p=.05 (BTRS)", 5000, 10000, 0.05},
        {"binom n~10000 p=1.7e-6 (inv)", 5000, 10000, 1.7e-6}, {"binom n~100 p=.05 (inv)", 50, 100, 0.05},
        {"binom n=2000 fixed p=.56", 2000, 1, 0.56}};
    for (auto& c : cases) {
        for (int64_t i = 0; i < n; i++) hn[i] = c.base + (int)((i * 2654435761u) % (unsigned)c.spread);
        CK(hipMemcpy(in, hn.data(), n * 4, hipMemcpyHostToDevice));
        report(c.name, timeit([&] { k_binom<<<grid, blk>>>(2, (int32_t*)in, c.p, (int32_t*)out, n); }));
    }
    for (double base : {0.5, 50.0, 5000.0}) {
        for (int64_t i = 0; i < n; i++) hx[i] = base * (1.0 + (i % 97) * 0.01);
        CK(hipMemcpy(in, hx.data(), n * 8, hipMemcpyHostToDevice));
        char nm[64]; snprintf(nm, sizeof nm, "gamma shape~%g", base);
        report(nm, timeit([&] { k_gamma<<<grid, blk>>>(3, (double*)in, (double*)out, n); }));
    }
    return 0;
}
