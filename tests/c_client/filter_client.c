/* A host written in plain C99 against include/epi.h only: the stand-in for a
 * Julia `ccall` host.  It builds a model, a belief of n particles, runs `updates`
 * filter updates with the given observations and prints the diagnostics and the
 * belief mean, one "key value..." line each, which the test compares with the
 * same run made through the Python binding.
 *
 *   filter_client <n> <seed> <action> <o_1> ... <o_k>
 *   filter_client shards=<G> <n> <seed> <action> <o_1> ... <o_k>     the same population as G shards (epi_mctx), all on device 0
 *
 * Parameters arrive on stdin as text: H D N inf_loss test_loss testrate_loss,
 * then H values each of q, p, k, theta (hex floats are accepted by strtod). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <inttypes.h>
#include "epi.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != EPI_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #call, rc_, epi_last_error()); return 2; } } while (0)

static int read_double(double *x)
{
    char tok[128];
    if (scanf("%127s", tok) != 1) return 0;
    *x = strtod(tok, NULL);
    return 1;
}

/* the sharded form: one population over G shards behind epi_mctx (here all on device 0) */
static int run_sharded(int G, int64_t n, uint64_t seed, double action, const epi_params *P, int n_obs, char **obs)
{
    int dev[64];
    epi_mctx *m = NULL;
    if (G < 1 || G > 64) { fprintf(stderr, "bad shard count\n"); return 1; }
    for (int g = 0; g < G; g++) dev[g] = 0;
    CHECK(epi_mctx_create(dev, G, P, n, 0, &m));
    CHECK(epi_mctx_init_normal(m, 10000.0, 5000.0, seed));
    printf("rows %d size %" PRId64 "\n", epi_state_rows(P->H, P->D), epi_mctx_size(m));
    for (int t = 0; t < n_obs; t++) {
        double loglik = 0.0, ess = 0.0;
        uint32_t flags = 0;
        int64_t moved = 0;
        CHECK(epi_mpf_update(m, action, strtoll(obs[t], NULL, 10), seed, (uint32_t)t, &loglik, &ess, &flags, &moved));
        printf("update %d %a %a %u\n", t, loglik, ess, flags);
    }
    int64_t mS = 0, mR = 0, mI[EPI_MAX_H], mT[(EPI_MAX_D + 1) * EPI_MAX_H];
    CHECK(epi_mctx_mean(m, &mS, mI, &mR, mT));
    printf("mean %" PRId64 " %" PRId64, mS, mR);
    for (int i = 0; i < P->H; i++) printf(" %" PRId64, mI[i]);
    printf("\n");
    CHECK(epi_mctx_destroy(m));
    return 0;
}

int main(int argc, char **argv)
{
    int shards = 0;
    if (argc > 1 && strncmp(argv[1], "shards=", 7) == 0) { shards = atoi(argv[1] + 7); argv++; argc--; }
    if (argc < 5) { fprintf(stderr, "usage: filter_client n seed action o_1 [o_2 ...]\n"); return 1; }
    const int64_t n = strtoll(argv[1], NULL, 10);
    const uint64_t seed = strtoull(argv[2], NULL, 10);
    const double action = strtod(argv[3], NULL);

    epi_params P;
    memset(&P, 0, sizeof P);
    double v[6];
    for (int j = 0; j < 6; j++) if (!read_double(&v[j])) { fprintf(stderr, "short parameter input\n"); return 1; }
    P.H = (int32_t)v[0]; P.D = (int32_t)v[1]; P.N = (int64_t)v[2];
    P.inf_loss = v[3]; P.test_loss = v[4]; P.testrate_loss = v[5];
    if (P.H < 1 || P.H > EPI_MAX_H) { fprintf(stderr, "bad H\n"); return 1; }
    double *tab[4] = { P.q, P.p, P.k, P.theta };
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < P.H; i++)
            if (!read_double(&tab[t][i])) { fprintf(stderr, "short parameter input\n"); return 1; }

    if (shards) return run_sharded(shards, n, seed, action, &P, argc - 4, argv + 4);

    epi_ctx *ctx = NULL;
    epi_belief *b = NULL;
    CHECK(epi_ctx_create(0, &ctx));
    CHECK(epi_params_set(ctx, &P));
    CHECK(epi_belief_create(ctx, n, 0, &b));
    CHECK(epi_belief_init_normal(b, 10000.0, 5000.0, seed));
    printf("rows %d size %" PRId64 "\n", epi_state_rows(P.H, P.D), epi_belief_size(b));

    for (int t = 0; t + 4 < argc; t++) {
        const int64_t o = strtoll(argv[4 + t], NULL, 10);
        double loglik = 0.0, ess = 0.0;
        uint32_t flags = 0;
        CHECK(epi_pf_update(ctx, b, action, o, seed, (uint32_t)t, &loglik, &ess, &flags));
        printf("update %d %a %a %u\n", t, loglik, ess, flags);
    }

    int64_t mS = 0, mR = 0, mI[EPI_MAX_H], mT[(EPI_MAX_D + 1) * EPI_MAX_H];
    CHECK(epi_belief_mean(ctx, b, &mS, mI, &mR, mT));
    printf("mean %" PRId64 " %" PRId64, mS, mR);
    for (int i = 0; i < P.H; i++) printf(" %" PRId64, mI[i]);
    printf("\n");

    const double q[3] = { 0.05, 0.5, 0.95 };
    int64_t qs[3];
    CHECK(epi_belief_suminf_quantiles(ctx, b, q, 3, qs));
    printf("quantiles %" PRId64 " %" PRId64 " %" PRId64 "\n", qs[0], qs[1], qs[2]);

    /* an argument the library must reject, with its message */
    if (epi_belief_create(ctx, -1, 0, &b) != EPI_EINVAL || epi_last_error()[0] == '\0') { fprintf(stderr, "no EINVAL\n"); return 3; }

    CHECK(epi_belief_destroy(b));
    CHECK(epi_ctx_destroy(ctx));
    return 0;
}
