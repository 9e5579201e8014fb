/*
 * stretch_oracle.h -- CPU restatement of the reference's stretch-move ensemble step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only as
 * the checker / reported baseline.  The shipped sampler (libmcmcpp_hip.so) never links or calls it.
 *
 * What it restates (paths relative to /root/reference):
 *   MCMCpp/EnsembleSampler.h:284-310,341-360   runMCMC / performStep  (red half, then black half)
 *   MCMCpp/Movers/StretchMove.h:100-123        updateWalker           (partner, z, proposal, accept)
 *   MCMCpp/Utility/MultiSampler.h:54,60,86,99  the three random draws per update
 *   MCMCpp/Utility/GwDistribution.h:45-58      z = ((sqrt(a)-1/sqrt(a))*u + 1/sqrt(a))^2, a = 2
 *   MCMCpp/Walker/Walker.h:105,162-179         accept / stay bookkeeping, optional chain store
 *   MCMCpp/Chain/ChainBlock.h:125-131          chain cell layout  step*W*D + walker*D + p
 *   MCMCpp/Analysis/CovarianceMatrix.h:154-257 covariance / correlation of the stored samples (next row f2)
 *   imneme/pcg-cpp (un-vendored submodule, version unpinned by .gitmodules) pcg64 =
 *     setseq_xsl_rr_128_64: restated from the published algorithm (128-bit LCG, XSL-RR output,
 *     Brown's O(log n) jump-ahead); libstdc++ 11 generate_canonical / exponential_distribution
 *     (bits/random.tcc, bits/random.h) for the u64 -> real transforms.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement bit-for-bit against
 * fixtures in tests/golden/ produced by the reference itself (compiled from /root/reference by
 * oracle/Makefile into oracle/_ref/, run by tests/golden/make_golden.py), and the pcg64 stream
 * against numpy's independent PCG64 implementation.
 */
#ifndef STRETCH_ORACLE_H
#define STRETCH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element type of positions / log-probabilities (the reference's ParamType) */
enum { SO_F64 = 0, SO_F32 = 1 };

/* built-in log-posterior Calculators (canonical operation order documented in DESIGN.md) */
enum {
    SO_CALC_ISO_GAUSSIAN = 0,   /* -1/2 sum x^2                       params: none                  */
    SO_CALC_DENSE_GAUSSIAN = 1, /* -1/2 x^T P x                       params: P[D*D] row-major      */
    SO_CALC_ROSENBROCK = 2,     /* -c sum b(x_{i+1}-x_i^2)^2+(a-x_i)^2 params: a, b, c               */
    SO_CALC_SKEWED_GAUSSIAN_2D = 3 /* reference test target, D == 2   params: epsilon               */
};

/* how the random stream is walked */
enum {
    SO_MODE_SEQUENTIAL = 0, /* one engine advanced draw by draw, exactly as the reference does      */
    SO_MODE_COUNTER = 1     /* every walker jumps to draw 3*(h*n+i); embarrassingly parallel        */
};

#define SO_MOVER_STRETCH 0
#define SO_MOVER_DIFFERENTIAL_EVOLUTION 1

typedef struct so_config {
    int32_t dtype;       /* SO_F64 / SO_F32 */
    int32_t num_walkers; /* W, even, > 2*D (EnsembleSampler.h:207-208) */
    int32_t num_params;  /* D */
    int32_t calc_id;     /* SO_CALC_* */
    const void* calc_params; /* array of dtype elements, may be NULL when the calculator has none */
    int32_t calc_params_len; /* number of elements in calc_params */
    int32_t gw_alpha_num;    /* stretch scale a = num/den of GwDistribution<T, num, den>; 0/0 = 2/1 */
    int32_t gw_alpha_den;
    int32_t mover;           /* SO_MOVER_*: 0 = StretchMove, 1 = DifferentialEvolution (sequential mode only) */
    uint64_t seed;   /* EnsembleSampler ctor randSeed (sign-extended int), MultiSampler.h:54 */
    uint64_t stream; /* 0 for EnsembleSampler (EnsembleSampler.h:217) */
} so_config;

typedef struct so_sampler so_sampler;

/* lifecycle; every function returns 0 on success, negative on argument errors */
int so_create(const so_config* cfg, so_sampler** out);
void so_destroy(so_sampler* s);

/* positions[W*D] walker-major, logp[W]; resets counters and rewinds the random stream to draw 0 */
int so_set_state(so_sampler* s, const void* positions, const void* logp);

/* Run n_saved*interval ensemble steps; the last step of every `interval` is written to chain_out
 * (n_saved*W*D elements, may be NULL).  accepted_per_step (n_saved*interval entries, may be NULL)
 * receives the number of accepted proposals of every executed ensemble step.  threads > 1 is only
 * legal with SO_MODE_COUNTER (static walker partition, OpenMP). */
int so_run(so_sampler* s, int64_t n_saved, int32_t interval, void* chain_out,
           uint32_t* accepted_per_step, int32_t mode, int32_t threads);

/* One half-step (color 0 = red, 1 = black) restricted to walkers [begin, begin+count) of that colour,
 * counter-addressed draws; after every shard of the colour has been stepped (by this sampler or by
 * replicas that exchange the updated rows) call so_half_step_commit once to advance the stream.
 * Used by the CPU tests of the multi-GPU driver. */
int so_half_step_shard(so_sampler* s, int32_t color, int32_t begin, int32_t count, uint32_t* accepted);
int so_half_step_commit(so_sampler* s);
/* the sampler's own position / log-posterior arrays (W*D and W elements), for zero-copy views */
void* so_positions_ptr(so_sampler* s);
void* so_logp_ptr(so_sampler* s);

/* any pointer may be NULL */
int so_get_state(so_sampler* s, void* positions, void* logp, uint32_t* n_accept);

uint64_t so_half_steps_done(const so_sampler* s);
/* decisions whose margin |lnU - delta| was within a few ulp: a different libm `log` could flip them */
uint64_t so_near_ties(const so_sampler* s);
/* the most recent of them: the half-step (counted from set_state / so_seek's origin), the walker (0 .. W-1), the decision
 * taken and the two sides of the comparison; -1 when there has been none */
int so_last_near_tie(const so_sampler* s, uint64_t* half_step, uint32_t* walker, int32_t* accepted, double* ln_u, double* delta);
/* StretchMove only: position the random stream as if `ensemble_steps_done` steps had been executed since set_state
 * (what mcmcpp_hip_seek does on the device): a checkpointed ensemble can be stepped on from any step */
int so_seek(so_sampler* s, uint64_t ensemble_steps_done);
/* bounded_rand rejections (only possible when W/2 is not a power of two) */
uint64_t so_redraws(const so_sampler* s);

/* evaluate a calculator on one D-vector (used to build initial logp arrays in tests and bench) */
int so_calc_logp(const so_config* cfg, const void* x, void* out);

/* --- next row f2: Analysis::CovarianceMatrix (MCMCpp/Analysis/CovarianceMatrix.h:154-224) over stored steps
 * [n_steps][walkers][dims], every `slice`-th step; cov and corr are [dims][dims], mean [dims] (may be NULL) */
int so_chain_covariance(int32_t dtype, const void* steps, int64_t n_steps, int32_t walkers, int32_t dims, int32_t slice,
                        void* mean, void* cov, void* corr);

/* Analysis::Detail::AutoCov::calcNormAutoCov (Analysis/Detail/AutoCov.h:146-155): chain[n] in, normalised circular
 * autocovariance out; and Analysis::AutoCorrCalc::calcAutoCorrTimes with all walkers (Analysis/AutoCorrCalc.h:151-207)
 * over steps[n][walkers][dims] -> out[dims].  emulate_defect: 1 restates the reference to the letter (its transferWalker
 * adds each series onto the previous walker's result), 0 is the documented computation (see the .inc). */
int so_norm_autocov(int32_t dtype, void* chain, double avg, int32_t n);
int so_autocorr_times(int32_t dtype, const void* steps, int32_t n_steps, int32_t walkers, int32_t dims, int32_t window_scaling,
                      int32_t emulate_defect, void* out, void* functions /* NULL or [dims][n_steps]: the averaged functions */);

/* The chain of the reference's AutoCorrCalc known-answer test (test/sequential/AcTime/src/main.cpp): AR(1) walkers moved
 * by Mover::AutoRegressiveMove with libstdc++'s normal_distribution over pcg64.  chain: [n_steps + 1][W][D]. */
int so_ar1_test_chain(int32_t run_number, int32_t W, int32_t D, int32_t n_steps, const double* offsets, const double* phis,
                      const double* vars, double* chain);

/* --- pcg64 (setseq_xsl_rr_128_64) primitives, exposed for known-answer tests ------------------- */
typedef struct so_pcg64 {
    uint64_t state_hi, state_lo;
    uint64_t inc_hi, inc_lo;
} so_pcg64;

void so_pcg64_seed(so_pcg64* g, uint64_t seed, uint64_t stream);
uint64_t so_pcg64_next(so_pcg64* g);
void so_pcg64_advance(so_pcg64* g, uint64_t delta_hi, uint64_t delta_lo);
/* affine map of `delta` LCG steps: state' = mult*state + plus (mod 2^128) */
void so_pcg64_jump_coeffs(uint64_t inc_hi, uint64_t inc_lo, uint64_t delta_hi, uint64_t delta_lo,
                          uint64_t mult_out[2] /*hi,lo*/, uint64_t plus_out[2] /*hi,lo*/);
double so_canonical_f64(uint64_t r);
float so_canonical_f32(uint64_t r);

/* splitmix64-hash initial positions, uniform in [-2, 2): x[k] = 4*u(k) - 2 */
void so_init_positions(int32_t dtype, int64_t count, uint64_t salt, void* out);

#ifdef __cplusplus
}
#endif
#endif /* STRETCH_ORACLE_H */
