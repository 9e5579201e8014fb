/*
 * stretch_oracle.c -- CPU restatement of the reference's stretch-move ensemble step.
 * TEST INFRASTRUCTURE ONLY (see stretch_oracle.h).  Build: oracle/Makefile (-O2 -ffp-contract=off).
 *
 * Every function names the reference lines it follows (paths relative to /root/reference).
 */
#define _GNU_SOURCE /* sincos, sincosf */
#include "stretch_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ---- pcg64 = setseq_xsl_rr_128_64 (imneme/pcg-cpp, call site MCMCpp/Utility/MultiSampler.h:120) -- */

/* default 128-bit LCG multiplier of pcg-cpp: 2549297995355413924 * 2^64 + 4865540595714422341 */
static u128 pcg_mult(void) { return ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL; }

static u128 mk128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | lo; }

/* engine(state, stream): inc = (stream << 1) | 1; state = (seed + inc) * M + inc
 * (MultiSampler.h:54 -> pcg engine::seed -> engine ctor -> bump) */
void so_pcg64_seed(so_pcg64* g, uint64_t seed, uint64_t stream)
{
    u128 inc = (((u128)stream) << 1) | 1u;
    u128 st = ((u128)seed + inc) * pcg_mult() + inc;
    g->state_hi = (uint64_t)(st >> 64);
    g->state_lo = (uint64_t)st;
    g->inc_hi = (uint64_t)(inc >> 64);
    g->inc_lo = (uint64_t)inc;
}

/* XSL-RR 128 -> 64 output of the state AFTER the LCG step (output_previous == false for 128-bit
 * state): rot = s >> 122, x = hi ^ lo, rotr64(x, rot) */
static uint64_t pcg_output(u128 s)
{
    uint64_t x = (uint64_t)(s >> 64) ^ (uint64_t)s;
    unsigned rot = (unsigned)(s >> 122);
    return (x >> rot) | (x << ((-rot) & 63u));
}

uint64_t so_pcg64_next(so_pcg64* g)
{
    u128 s = mk128(g->state_hi, g->state_lo) * pcg_mult() + mk128(g->inc_hi, g->inc_lo);
    g->state_hi = (uint64_t)(s >> 64);
    g->state_lo = (uint64_t)s;
    return pcg_output(s);
}

/* Brown, "Random number generation with arbitrary strides": coefficients of `delta` LCG steps */
static void jump_coeffs(u128 inc, u128 delta, u128* mult_out, u128* plus_out)
{
    u128 cur_mult = pcg_mult(), cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1u) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1u) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    *mult_out = acc_mult;
    *plus_out = acc_plus;
}

void so_pcg64_jump_coeffs(uint64_t inc_hi, uint64_t inc_lo, uint64_t delta_hi, uint64_t delta_lo,
                          uint64_t mult_out[2], uint64_t plus_out[2])
{
    u128 m, p;
    jump_coeffs(mk128(inc_hi, inc_lo), mk128(delta_hi, delta_lo), &m, &p);
    mult_out[0] = (uint64_t)(m >> 64);
    mult_out[1] = (uint64_t)m;
    plus_out[0] = (uint64_t)(p >> 64);
    plus_out[1] = (uint64_t)p;
}

void so_pcg64_advance(so_pcg64* g, uint64_t delta_hi, uint64_t delta_lo)
{
    u128 m, p;
    jump_coeffs(mk128(g->inc_hi, g->inc_lo), mk128(delta_hi, delta_lo), &m, &p);
    u128 s = m * mk128(g->state_hi, g->state_lo) + p;
    g->state_hi = (uint64_t)(s >> 64);
    g->state_lo = (uint64_t)s;
}

/* libstdc++ 11 generate_canonical<double,53>(pcg64) (bits/random.tcc:3345-3380): one 64-bit draw,
 * u = double(r) [round to nearest] / 2^64, clamped below 1 */
double so_canonical_f64(uint64_t r)
{
    double u = (double)r / 18446744073709551616.0;
    if (u >= 1.0) u = nextafter(1.0, 0.0);
    return u;
}

float so_canonical_f32(uint64_t r)
{
    float u = (float)r / 18446744073709551616.0f;
    if (u >= 1.0f) u = nextafterf(1.0f, 0.0f);
    return u;
}

/* ---- initial positions (this repo's synthetic workload) --------------------------- */

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

void so_init_positions(int32_t dtype, int64_t count, uint64_t salt, void* out)
{
    for (int64_t k = 0; k < count; ++k) {
        uint64_t h = splitmix64((uint64_t)k + salt * 0x632BE59BD9B4E019ULL);
        double u = (double)(h >> 11) * (1.0 / 9007199254740992.0); /* top 53 bits * 2^-53 */
        double x = 4.0 * u - 2.0;
        if (dtype == SO_F64)
            ((double*)out)[k] = x;
        else
            ((float*)out)[k] = (float)x;
    }
}

/* ---- sampler ------------------------------------------------------------------------------------ */

struct so_sampler {
    so_config cfg;
    void* params;      /* private copy of calc_params */
    void* pos;         /* [W][D] */
    void* logp;        /* [W] */
    uint32_t* n_accept; /* [W] accepted proposals since set_state (initial placement not counted) */
    so_pcg64 eng0;     /* engine right after seeding: draw 0 is its next output */
    uint64_t half_steps;
    uint64_t near_ties;
    uint64_t redraws;
    /* the most recent near tie (so_last_near_tie): which decision it was */
    uint64_t tie_half_step;
    uint32_t tie_walker;
    int32_t tie_accepted;
    double tie_ln_u, tie_delta;
};

#define REAL double
#define SFX(n) n##_f64
#define LOG(x) log(x)
#define SQRT(x) sqrt(x)
#define SINCOS(x, s, c) sincos(x, s, c)
#define FMA(a, b, c) fma(a, b, c)
#define FABS(x) fabs(x)
#define CANON(r) so_canonical_f64(r)
#define TIE_EPS 1e-12
#include "stretch_oracle_typed.inc"
#undef REAL
#undef SFX
#undef LOG
#undef SQRT
#undef SINCOS
#undef FMA
#undef FABS
#undef CANON
#undef TIE_EPS

#define REAL float
#define SFX(n) n##_f32
#define LOG(x) logf(x)
#define SQRT(x) sqrtf(x)
#define SINCOS(x, s, c) sincosf(x, s, c)
#define FMA(a, b, c) fmaf(a, b, c)
#define FABS(x) fabsf(x)
#define CANON(r) so_canonical_f32(r)
#define TIE_EPS 6e-7f
#include "stretch_oracle_typed.inc"
#undef REAL
#undef SFX
#undef LOG
#undef SQRT
#undef SINCOS
#undef FMA
#undef FABS
#undef CANON
#undef TIE_EPS

static size_t elem_size(int dtype) { return dtype == SO_F64 ? sizeof(double) : sizeof(float); }

static int check_cfg(const so_config* c)
{
    if (!c) return -1;
    if (c->dtype != SO_F64 && c->dtype != SO_F32) return -2;
    if (c->num_params < 1) return -3;
    /* EnsembleSampler.h:207-208: even walker count, more than 2*D walkers */
    if (c->num_walkers < 2 || (c->num_walkers & 1) || c->num_walkers <= 2 * c->num_params) return -4;
    switch (c->calc_id) {
    case SO_CALC_ISO_GAUSSIAN: break;
    case SO_CALC_DENSE_GAUSSIAN:
        if (!c->calc_params || c->calc_params_len != c->num_params * c->num_params) return -5;
        break;
    case SO_CALC_ROSENBROCK:
        if (!c->calc_params || c->calc_params_len != 3) return -5;
        break;
    case SO_CALC_SKEWED_GAUSSIAN_2D:
        if (!c->calc_params || c->calc_params_len != 1 || c->num_params != 2) return -5;
        break;
    default: return -6;
    }
    return 0;
}

int so_create(const so_config* cfg, so_sampler** out)
{
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!out) return -1;
    so_sampler* s = (so_sampler*)calloc(1, sizeof(*s));
    if (!s) return -7;
    s->cfg = *cfg;
    size_t es = elem_size(cfg->dtype);
    size_t W = (size_t)cfg->num_walkers, D = (size_t)cfg->num_params;
    s->params = NULL;
    if (cfg->calc_params_len > 0) {
        s->params = malloc(es * (size_t)cfg->calc_params_len);
        if (!s->params) { free(s); return -7; }
        memcpy(s->params, cfg->calc_params, es * (size_t)cfg->calc_params_len);
    }
    s->cfg.calc_params = s->params;
    s->pos = calloc(W * D, es);
    s->logp = calloc(W, es);
    s->n_accept = (uint32_t*)calloc(W, sizeof(uint32_t));
    if (!s->pos || !s->logp || !s->n_accept) { so_destroy(s); return -7; }
    so_pcg64_seed(&s->eng0, cfg->seed, cfg->stream);
    *out = s;
    return 0;
}

void so_destroy(so_sampler* s)
{
    if (!s) return;
    free(s->params);
    free(s->pos);
    free(s->logp);
    free(s->n_accept);
    free(s);
}

/* EnsembleSampler::setInitialWalkerPos (EnsembleSampler.h:220-230): copies positions and aux
 * values; the reference additionally counts the placement as one accepted step per walker
 * (Walker.h:76,168) -- that offset is applied by callers, n_accept here counts proposals only. */
int so_set_state(so_sampler* s, const void* positions, const void* logp)
{
    if (!s || !positions || !logp) return -1;
    size_t es = elem_size(s->cfg.dtype);
    size_t W = (size_t)s->cfg.num_walkers, D = (size_t)s->cfg.num_params;
    memcpy(s->pos, positions, W * D * es);
    memcpy(s->logp, logp, W * es);
    memset(s->n_accept, 0, W * sizeof(uint32_t));
    s->half_steps = 0;
    s->near_ties = 0;
    s->redraws = 0;
    return 0;
}

int so_run(so_sampler* s, int64_t n_saved, int32_t interval, void* chain_out,
           uint32_t* accepted_per_step, int32_t mode, int32_t threads)
{
    if (!s || n_saved < 0 || interval < 1) return -1;
    if (mode != SO_MODE_SEQUENTIAL && mode != SO_MODE_COUNTER) return -2;
    if (threads < 1) threads = 1;
    if (threads > 1 && mode != SO_MODE_COUNTER) return -3;
    if (s->cfg.mover == SO_MOVER_DIFFERENTIAL_EVOLUTION && mode != SO_MODE_SEQUENTIAL) return -4;
    if (s->cfg.dtype == SO_F64)
        return run_f64(s, n_saved, interval, (double*)chain_out, accepted_per_step, mode, threads);
    return run_f32(s, n_saved, interval, (float*)chain_out, accepted_per_step, mode, threads);
}

int so_half_step_shard(so_sampler* s, int32_t color, int32_t begin, int32_t count, uint32_t* accepted)
{
    if (!s || (color != 0 && color != 1)) return -1;
    if ((int)(s->half_steps & 1) != color) return -2; /* red, black, red, ... */
    if (s->cfg.mover != SO_MOVER_STRETCH) return -4;
    if (begin < 0 || count < 0 || begin + count > s->cfg.num_walkers / 2) return -3;
    uint32_t a = s->cfg.dtype == SO_F64 ? half_step_shard_f64(s, color, begin, count)
                                        : half_step_shard_f32(s, color, begin, count);
    if (accepted) *accepted = a;
    return 0;
}

int so_half_step_commit(so_sampler* s)
{
    if (!s) return -1;
    s->half_steps += 1;
    return 0;
}

void* so_positions_ptr(so_sampler* s) { return s ? s->pos : NULL; }
void* so_logp_ptr(so_sampler* s) { return s ? s->logp : NULL; }

int so_get_state(so_sampler* s, void* positions, void* logp, uint32_t* n_accept)
{
    if (!s) return -1;
    size_t es = elem_size(s->cfg.dtype);
    size_t W = (size_t)s->cfg.num_walkers, D = (size_t)s->cfg.num_params;
    if (positions) memcpy(positions, s->pos, W * D * es);
    if (logp) memcpy(logp, s->logp, W * es);
    if (n_accept) memcpy(n_accept, s->n_accept, W * sizeof(uint32_t));
    return 0;
}

uint64_t so_half_steps_done(const so_sampler* s) { return s ? s->half_steps : 0; }
uint64_t so_near_ties(const so_sampler* s) { return s ? s->near_ties : 0; }

int so_seek(so_sampler* s, uint64_t ensemble_steps_done)
{
    if (!s || s->cfg.mover != SO_MOVER_STRETCH) return -1; /* (differential evolution: the position depends on the draws thrown away) */
    s->half_steps = 2 * ensemble_steps_done;
    return 0;
}

int so_last_near_tie(const so_sampler* s, uint64_t* half_step, uint32_t* walker, int32_t* accepted, double* ln_u, double* delta)
{
    if (!s || s->near_ties == 0) return -1;
    if (half_step) *half_step = s->tie_half_step;
    if (walker) *walker = s->tie_walker;
    if (accepted) *accepted = s->tie_accepted;
    if (ln_u) *ln_u = s->tie_ln_u;
    if (delta) *delta = s->tie_delta;
    return 0;
}
uint64_t so_redraws(const so_sampler* s) { return s ? s->redraws : 0; }

int so_chain_covariance(int32_t dtype, const void* steps, int64_t n_steps, int32_t walkers, int32_t dims, int32_t slice,
                        void* mean, void* cov, void* corr)
{
    if (!steps || !cov || !corr || n_steps < 1 || walkers < 1 || dims < 1 || slice < 1) return -1;
    if (dtype == SO_F64)
        chain_covariance_f64((const double*)steps, n_steps, walkers, dims, slice, (double*)mean, (double*)cov, (double*)corr);
    else if (dtype == SO_F32)
        chain_covariance_f32((const float*)steps, n_steps, walkers, dims, slice, (float*)mean, (float*)cov, (float*)corr);
    else
        return -1;
    return 0;
}

/* std::normal_distribution<double>(0, 1) over pcg64 as libstdc++ implements it (bits/random.tcc:1802-1835, GCC 11): the
 * polar method, two canonical draws per attempt, the second variate kept for the next call. */
typedef struct { so_pcg64 eng; int have; double saved; } normal_src;
static double normal_next(normal_src* g)
{
    double ret;
    if (g->have) {
        g->have = 0;
        ret = g->saved;
    } else {
        double x, y, r2;
        do {
            x = 2.0 * so_canonical_f64(so_pcg64_next(&g->eng)) - 1.0;
            y = 2.0 * so_canonical_f64(so_pcg64_next(&g->eng)) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = sqrt(-2 * log(r2) / r2);
        g->saved = x * mult;
        g->have = 1;
        ret = y * mult;
    }
    return ret * 1.0 + 0.0;
}

/* The chain of the reference's AutoCorrCalc known-answer test (test/sequential/AcTime/src/main.cpp:23-52): W walkers of
 * D independent AR(1) parameters moved by Mover::AutoRegressiveMove (Movers/Diagnostic/AutoRegressiveMove.h:78-105).
 * Initial points come from the test's own mover object, whose engine is pcg64(0) on the DEFAULT stream
 * (AutoRegressiveMove.h:55, MultiSampler.h:42), parameter by parameter; the steps from the sampler's copy of it, reseeded
 * to (run_number, stream 0) (EnsembleSampler.h:217), walkers in order 0..W-1 (red set then black set,
 * EnsembleSampler.h:342-354), D normal variates each.  chain: [n_steps + 1][W][D], step 0 = the initial points. */
int so_ar1_test_chain(int32_t run_number, int32_t W, int32_t D, int32_t n_steps, const double* offsets, const double* phis,
                      const double* vars, double* chain)
{
    if (!offsets || !phis || !vars || !chain || W < 1 || D < 1 || n_steps < 0) return -1;
    double* sd = (double*)malloc(sizeof(double) * (size_t)D);
    if (!sd) return -7;
    for (int i = 0; i < D; ++i) sd[i] = sqrt(vars[i]) * sqrt(1.0 - phis[i] * phis[i]);
    normal_src init;
    memset(&init, 0, sizeof init);
    { /* pcg64 engine(0) without a stream argument: the 128-bit default increment */
        const u128 inc = mk128(6364136223846793005ULL, 1442695040888963407ULL);
        const u128 st = ((u128)0 + inc) * pcg_mult() + inc;
        init.eng.state_hi = (uint64_t)(st >> 64);
        init.eng.state_lo = (uint64_t)st;
        init.eng.inc_hi = (uint64_t)(inc >> 64);
        init.eng.inc_lo = (uint64_t)inc;
    }
    for (int i = 0; i < D; ++i) {
        const double total = sd[i] / sqrt(1.0 - phis[i] * phis[i]);
        for (int j = 0; j < W; ++j) chain[(size_t)j * D + i] = total * normal_next(&init);
    }
    normal_src run;
    memset(&run, 0, sizeof run);
    so_pcg64_seed(&run.eng, (uint64_t)(int64_t)run_number, 0);
    for (int t = 1; t <= n_steps; ++t) {
        const double* prev = chain + (size_t)(t - 1) * W * D;
        double* next = chain + (size_t)t * W * D;
        for (int j = 0; j < W * D; ++j) {
            const int i = j % D;
            next[j] = (offsets[i] + (phis[i] * prev[j])) + (sd[i] * normal_next(&run));
        }
    }
    free(sd);
    return 0;
}

int so_norm_autocov(int32_t dtype, void* chain, double avg, int32_t n)
{
    if (!chain || n < 2 || (dtype != SO_F64 && dtype != SO_F32)) return -1;
    if (dtype == SO_F64) {
        const int lg = fft_log2_f64(n), fft = 1 << lg;
        double* tw = (double*)malloc(sizeof(double) * (size_t)fft);
        double* work = (double*)malloc(sizeof(double) * 4 * (size_t)fft);
        fft_twiddles_f64(lg, tw);
        norm_autocov_f64((double*)chain, avg, n, lg, tw, work);
        free(tw);
        free(work);
    } else {
        const int lg = fft_log2_f32(n), fft = 1 << lg;
        float* tw = (float*)malloc(sizeof(float) * (size_t)fft);
        float* work = (float*)malloc(sizeof(float) * 4 * (size_t)fft);
        fft_twiddles_f32(lg, tw);
        norm_autocov_f32((float*)chain, (float)avg, n, lg, tw, work);
        free(tw);
        free(work);
    }
    return 0;
}

int so_autocorr_times(int32_t dtype, const void* steps, int32_t n_steps, int32_t walkers, int32_t dims, int32_t window_scaling,
                      int32_t emulate_defect, void* out, void* functions)
{
    if (!steps || !out || n_steps < 2 || walkers < 1 || dims < 1) return -1;
    if (dtype == SO_F64)
        autocorr_times_f64((const double*)steps, n_steps, walkers, dims, window_scaling, emulate_defect, (double*)out, (double*)functions);
    else if (dtype == SO_F32)
        autocorr_times_f32((const float*)steps, n_steps, walkers, dims, window_scaling, emulate_defect, (float*)out, (float*)functions);
    else
        return -1;
    return 0;
}

int so_calc_logp(const so_config* cfg, const void* x, void* out)
{
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!x || !out) return -1;
    if (cfg->dtype == SO_F64)
        *(double*)out = calc_logp_f64(cfg->calc_id, cfg->num_params, (const double*)cfg->calc_params,
                                      (const double*)x);
    else
        *(float*)out = calc_logp_f32(cfg->calc_id, cfg->num_params, (const float*)cfg->calc_params,
                                     (const float*)x);
    return 0;
}
