/* san_driver.c -- exercises the oracle (the CPU restatement, TEST INFRASTRUCTURE ONLY) under AddressSanitizer and
 * UndefinedBehaviorSanitizer: `make -C oracle sanitize` builds oracle/_san/san_driver with -fsanitize=address,undefined
 * and tests/test_sanitizers.py runs it.  Every entry point the tests use is called once on small ragged shapes, both
 * element types, both movers, both walks of the stream, one and several threads; results of the two walks must agree. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stretch_oracle.h"

static int failures = 0;
#define CHECK(c)                                                       \
    do                                                                 \
    {                                                                  \
        if (!(c))                                                      \
        {                                                              \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);         \
            ++failures;                                                \
        }                                                              \
    } while (0)

static void run_case(int dtype, int W, int D, int calc, int mover)
{
    const size_t es = dtype == SO_F64 ? 8 : 4;
    double prm64[3] = {1.0, 100.0, 0.05};
    float prm32[3] = {1.0f, 100.0f, 0.05f};
    void* dense = NULL;
    so_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.dtype = dtype;
    cfg.num_walkers = W;
    cfg.num_params = D;
    cfg.calc_id = calc;
    cfg.mover = mover;
    cfg.seed = 5;
    if (calc == SO_CALC_ROSENBROCK)
    {
        cfg.calc_params = dtype == SO_F64 ? (void*)prm64 : (void*)prm32;
        cfg.calc_params_len = 3;
    }
    if (calc == SO_CALC_DENSE_GAUSSIAN)
    {
        dense = calloc((size_t)D * D, es);
        for (int i = 0; i < D; ++i)
        {
            if (dtype == SO_F64)
                ((double*)dense)[(size_t)i * D + i] = 1.0 + 0.1 * i;
            else
                ((float*)dense)[(size_t)i * D + i] = 1.0f + 0.1f * i;
        }
        cfg.calc_params = dense;
        cfg.calc_params_len = D * D;
    }
    if (calc == SO_CALC_SKEWED_GAUSSIAN_2D)
    {
        cfg.calc_params = dtype == SO_F64 ? (void*)prm64 + 16 : (void*)prm32 + 8;  /* 0.05 as epsilon */
        cfg.calc_params_len = 1;
    }
    void* pos = malloc((size_t)W * D * es);
    void* logp = malloc((size_t)W * es);
    so_init_positions(dtype, (int64_t)W * D, 3, pos);
    for (int w = 0; w < W; ++w) CHECK(so_calc_logp(&cfg, (char*)pos + (size_t)w * D * es, (char*)logp + (size_t)w * es) == 0);
    const int n_saved = 3, interval = 2;
    void* chain_a = malloc((size_t)n_saved * W * D * es);
    void* chain_b = malloc((size_t)n_saved * W * D * es);
    uint32_t acc_a[6], acc_b[6];
    so_sampler* s = NULL;
    CHECK(so_create(&cfg, &s) == 0 && s);
    CHECK(so_set_state(s, pos, logp) == 0);
    CHECK(so_run(s, n_saved, interval, chain_a, acc_a, SO_MODE_SEQUENTIAL, 1) == 0);
    if (mover == SO_MOVER_STRETCH)
    {
        CHECK(so_set_state(s, pos, logp) == 0);
        CHECK(so_run(s, n_saved, interval, chain_b, acc_b, SO_MODE_COUNTER, 3) == 0);
        CHECK(memcmp(chain_a, chain_b, (size_t)n_saved * W * D * es) == 0);
        CHECK(memcmp(acc_a, acc_b, sizeof acc_a) == 0);
        /* sharded half-steps (the multi-GPU driver's CPU stand-in) */
        CHECK(so_set_state(s, pos, logp) == 0);
        uint32_t got = 0, part = 0;
        for (int color = 0; color < 2; ++color)
        {
            const int n = W / 2, cut = n / 3;
            CHECK(so_half_step_shard(s, color, 0, cut, &part) == 0);
            got += part;
            CHECK(so_half_step_shard(s, color, cut, n - cut, &part) == 0);
            got += part;
            CHECK(so_half_step_commit(s) == 0);
        }
        CHECK(so_half_steps_done(s) == 2);
        (void)got;
    }
    uint32_t* nacc = malloc(sizeof(uint32_t) * (size_t)W);
    CHECK(so_get_state(s, pos, logp, nacc) == 0);
    (void)so_near_ties(s);
    (void)so_redraws(s);
    {
        uint64_t hs = 0;
        uint32_t w = 0;
        int32_t acc = 0;
        double lu = 0, dl = 0;
        (void)so_last_near_tie(s, &hs, &w, &acc, &lu, &dl); /* (-1: there has been none) */
        (void)so_seek(s, so_half_steps_done(s) / 2);
    }
    so_destroy(s);
    /* analysis restatements over the stored steps */
    void* mean = malloc((size_t)D * es);
    void* cov = malloc((size_t)D * D * es);
    void* corr = malloc((size_t)D * D * es);
    CHECK(so_chain_covariance(dtype, chain_a, n_saved, W, D, 1, mean, cov, corr) == 0);
    void* times = malloc((size_t)D * es);
    void* fn = malloc((size_t)D * n_saved * es);
    CHECK(so_autocorr_times(dtype, chain_a, n_saved, W, D, 4, 0, times, fn) == 0);
    CHECK(so_autocorr_times(dtype, chain_a, n_saved, W, D, 4, 1, times, NULL) == 0);
    free(mean), free(cov), free(corr), free(times), free(fn), free(nacc), free(chain_a), free(chain_b), free(pos), free(logp), free(dense);
}

int main(void)
{
    for (int dtype = 0; dtype < 2; ++dtype)
        for (int mover = 0; mover < 2; ++mover)
        {
            run_case(dtype, 14, 3, SO_CALC_ISO_GAUSSIAN, mover);
            run_case(dtype, 100, 7, SO_CALC_ISO_GAUSSIAN, mover);
            run_case(dtype, 96, 16, SO_CALC_DENSE_GAUSSIAN, mover);
            run_case(dtype, 80, 33, SO_CALC_ROSENBROCK, mover);
            run_case(dtype, 64, 2, SO_CALC_SKEWED_GAUSSIAN_2D, mover);
        }
    /* series transforms: lengths around powers of two */
    for (int n = 2; n <= 130; n += (n < 10 ? 1 : 31))
    {
        double* x = malloc(sizeof(double) * (size_t)n);
        float* y = malloc(sizeof(float) * (size_t)n);
        double avg = 0;
        for (int i = 0; i < n; ++i) x[i] = (double)((i * 37) % 11) - 5.0, y[i] = (float)x[i], avg += x[i];
        avg /= n;
        CHECK(so_norm_autocov(SO_F64, x, avg, n) == 0);
        CHECK(so_norm_autocov(SO_F32, y, avg, n) == 0);
        free(x), free(y);
    }
    {
        /* the AcTime chain generator and the pcg primitives */
        const double off[2] = {0.0, 1.0}, phi[2] = {0.5, 0.9}, var[2] = {1.0, 2.0};
        double* chain = malloc(sizeof(double) * 9 * 6 * 2);
        CHECK(so_ar1_test_chain(0, 6, 2, 8, off, phi, var, chain) == 0);
        free(chain);
        so_pcg64 g, h;
        so_pcg64_seed(&g, 42, 54);
        h = g;
        for (int i = 0; i < 1000; ++i) (void)so_pcg64_next(&g);
        so_pcg64_advance(&h, 0, 1000);
        CHECK(so_pcg64_next(&g) == so_pcg64_next(&h));
        uint64_t m[2], p[2];
        so_pcg64_jump_coeffs(g.inc_hi, g.inc_lo, 0, 12345, m, p);
        (void)so_canonical_f64(~0ULL);
        (void)so_canonical_f32(~0ULL);
    }
    if (!failures) printf("san_driver OK\n");
    return failures ? 1 : 0;
}
