/*
 * epi_testing.h -- entry points that exist only in libepi_gfx950_testing.so, the build of the library made with
 * -DEPI_TESTING for its own test suite.  They switch process-wide implementation choices that never change a result
 * (the tests that use them show exactly that); the product library has neither the switches nor the alternative
 * kernels.  No counterpart in the reference.
 */
#ifndef EPI_TESTING_H
#define EPI_TESTING_H

#include "epi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The rejection samplers screen most accept/reject tests in binary32 before
 * evaluating the defining binary64 expression; results are identical either
 * way.  enable = 0 forces the binary64 expression everywhere (testing aid,
 * process-wide). */
int epi_debug_screen(int enable);
/* The samplers take binary32 square roots and quotients whose operands are known to lie between 2^-30 and
 * 2^32; on the device they use the hardware estimate plus the rounding correction, without the scaling and
 * special-case steps of the general expansion.  This self-check evaluates both forms on every binary32 value in
 * [first, last] (mode 0: sqrt(x); 1: 1/x; 2: f/(2+f) with f = x-1, the quotient inside the binary32 logarithm)
 * and returns the number of values where the bits differ (must be 0).  No counterpart in the reference. */
int epi_debug_f32_forms(epi_ctx *ctx, int mode, float first, float last, uint64_t *mismatches);
/* Two implementations of the Philox-driven step kernel give identical results: the cooperative one
 * hands rejected sampler attempts to idle threads of the workgroup, the other loops per thread.
 * cooperative = 1 (default) is the cooperative kernel with one thread per particle up to H = 17, two up to H = 28 and
 * four beyond (which keeps the LDS footprint of a workgroup that of a narrow state), 2 forces the cooperative kernel
 * with one thread per particle, 0 the per-thread one.  Testing aid, process-wide. */
int epi_debug_step_impl(int cooperative);
/* The cooperative step kernel settles most binomial rejection draws in its first phase without the log-factorial
 * table (squeeze, and bounds on log f(k) - log f(m) from the series of the ratio of neighbouring probabilities); what
 * it settles must be what the defining acceptance test gives.  This runs blocks * 256 threads, each 64 attempts at a
 * random n in [n_lo, n_lo + n_span) and the folded probability pp, through both and returns in out5: attempts, settled
 * as accepted, settled as rejected, left open, settled differently from the defining test (must be 0).  No counterpart
 * in the reference. */
int epi_debug_btrs_quick(epi_ctx *ctx, uint64_t seed, int32_t n_lo, uint32_t n_span, double pp, int blocks, uint64_t *out5);
/* epi_mpf_update moves particle blocks between shards with hipMemcpyAsync when both shards share a device and with
 * hipMemcpyPeerAsync otherwise, and finds the global maximum log weight by letting every shard read the other shards'
 * device memory when the devices have access to each other, through host memory otherwise.  A box with one GPU only
 * ever takes the first branch of each.  mode bit 0: peer copies even between shards of one device; bit 1: the maximum
 * through host memory even where devices see each other.  Results are identical in every mode.  Testing aid,
 * process-wide.  No counterpart in the reference. */
int epi_debug_node_exchange(int mode);

#ifdef __cplusplus
}
#endif
#endif /* EPI_TESTING_H */
