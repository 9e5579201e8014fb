/*
 * mcmcpp_hip.h -- C ABI of libmcmcpp_hip.so, the MI355X (gfx950) implementation of the reference's
 * stretch-move ensemble step.
 *
 * The reference (jmatta1/MCMCpp) is a header-only C++ template library and has no FFI of its own; the
 * boundary below is what its sampler facade would bind for this path.  Each entry point names the
 * reference interface it stands in for (paths relative to /root/reference).  The header-only host
 * facade in include/MCMCpp/ (same class names and template parameters as the reference) is the only
 * intended caller; INTEGRATION.md shows the binding.
 *
 * Conventions: plain C types only; every function returns MCMCPP_HIP_OK (0) or a positive error code
 * and never throws; pointers are caller-owned HOST memory unless the name says `device`; a handle is
 * used from one host thread at a time (the facade keeps the reference's sampler mutex,
 * ParallelEnsembleSampler.h:184-210); the handle owns all device buffers, streams and graphs.
 *
 * Data layout (same as the reference's setInitialWalkerPos arguments and Chain cells,
 * EnsembleSampler.h:220-230, Chain/ChainBlock.h:125-131):
 *   positions[w*D + p]   walker-major, parameter-contiguous; walkers [0, W/2) are the red half,
 *                        [W/2, W) the black half (EnsembleSampler.h:211-215)
 *   logp[w]              the walker's auxiliary value = log-posterior of its position
 *   chain[s*W*D + w*D + p]  stored step s
 */
#ifndef MCMCPP_HIP_H
#define MCMCPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMCPP_HIP_ABI_VERSION 2

/* return codes */
enum {
    MCMCPP_HIP_OK = 0,
    MCMCPP_HIP_E_ARG = 1,         /* bad argument (the reference asserts: EnsembleSampler.h:207-208,327,335) */
    MCMCPP_HIP_E_HIP = 2,         /* a HIP runtime call failed; see mcmcpp_hip_last_error */
    MCMCPP_HIP_E_NO_DEVICE = 3,   /* no gfx950 device / device ordinal out of range */
    MCMCPP_HIP_E_UNSUPPORTED = 4, /* configuration outside what the kernels were built for */
    MCMCPP_HIP_E_STATE = 5,       /* run/get before set_state */
    MCMCPP_HIP_E_NOMEM = 6,
    MCMCPP_HIP_E_COMM = 7         /* the collective library (RCCL) is missing or one of its calls failed */
};

/* ParamType of the reference templates */
enum { MCMCPP_HIP_F64 = 0, MCMCPP_HIP_F32 = 1 };

/* device Calculators (reference concept: ParamType calcLogPostProb(ParamType*),
 * Utility/UserOjbectsTest.h:144-145); host twins live in include/MCMCpp/Device/Calculators.h */
enum {
    MCMCPP_HIP_CALC_ISO_GAUSSIAN = 0,      /* params: none                    */
    MCMCPP_HIP_CALC_DENSE_GAUSSIAN = 1,    /* params: P[D*D] row-major        */
    MCMCPP_HIP_CALC_ROSENBROCK = 2,        /* params: a, b, c                 */
    MCMCPP_HIP_CALC_SKEWED_GAUSSIAN_2D = 3 /* params: epsilon; requires D == 2 */
};

/* calculator ids from here on belong to user plug-ins (mcmcpp_hip_register_calculator) */
#define MCMCPP_HIP_CALC_USER_BASE 1000

typedef struct mcmcpp_hip_sampler mcmcpp_hip_sampler;

/* Stands in for the constructor arguments of EnsembleSampler (EnsembleSampler.h:66-67,199-218),
 * ParallelEnsembleSampler (ParallelEnsembleSampler.h:119-120) and StretchMove (Movers/StretchMove.h:62-66). */
typedef struct mcmcpp_hip_config {
    uint32_t struct_size;    /* sizeof(mcmcpp_hip_config) */
    int32_t dtype;           /* MCMCPP_HIP_F64 / MCMCPP_HIP_F32 */
    int32_t num_walkers;     /* W: even and > 2*D */
    int32_t num_params;      /* D: 1..1024 */
    int32_t calc_id;         /* MCMCPP_HIP_CALC_* */
    int32_t calc_params_len; /* number of dtype elements behind calc_params */
    const void* calc_params; /* host pointer, copied */
    uint64_t seed;           /* randSeed, sign-extended to 64 bit (MultiSampler.h:54) */
    uint64_t stream;         /* pcg stream; EnsembleSampler uses 0 (EnsembleSampler.h:217) */
    int32_t device;          /* HIP device ordinal, -1 = the current device */
    /* Walkers of each half updated by this handle: [shard_begin, shard_begin + shard_count) of W/2.
     * shard_count == 0 means the whole half.  A sharded handle still holds the full ensemble (the
     * complementary half must be readable); the caller exchanges the updated rows between handles
     * after every half-step (see mcmcpp_hip_half_step_async). */
    int32_t shard_begin;
    int32_t shard_count;
    int32_t graph_steps;     /* ensemble steps per hipGraph replay; 0 = library default, -1 = no graphs */
    /* stretch scale a = gw_alpha_num / gw_alpha_den of GwDistribution<ParamType, Num, Denom>
     * (Utility/GwDistribution.h:45-55; StretchMove's default is 2/1); 0/0 selects 2/1 */
    int32_t gw_alpha_num;
    int32_t gw_alpha_den;
    void* device_positions;  /* optional caller-owned DEVICE buffer of W*D elements used in place of an
                                internal one (e.g. memory registered with a collective library) */
    void* hip_stream;        /* hipStream_t to launch on when MCMCPP_HIP_FLAG_CALLER_STREAM is set; otherwise ignored and the
                                handle owns a private non-blocking stream.  NULL (the legacy default stream) is accepted,
                                but HIP cannot capture graphs on it: such a handle steps with plain launches */
    uint32_t flags;          /* MCMCPP_HIP_FLAG_* */
    uint32_t mover;          /* MCMCPP_HIP_MOVER_*: which Mover::updateWalker the kernels implement; 0 = StretchMove */
    /* One ensemble split over comm_world GPUs (one handle per GPU; stands in for ParallelEnsembleSampler's threadCount
     * workers, ParallelEnsembleSampler.h:119-120,285-291, and the red/black controller's two barriers per step,
     * Threading/RedBlkCtrlerSpinLock.h:240-322).  comm_world >= 1 makes this handle rank comm_rank of an RCCL
     * communicator (created from comm_id, or taken from comm) and lets mcmcpp_hip_run step the split ensemble: the
     * handle's shard must be the rank's slice, [comm_rank * (W/2) / comm_world, + (W/2) / comm_world) -- shard_count = 0
     * selects exactly that -- and W/2 must divide by comm_world.  Every rank calls set_state / run / get_state with the
     * same arguments; the exchanges (one ncclAllGather per ensemble step, see mcmcpp_hip_last_run_exchange) are enqueued on
     * the launch stream by run itself.  If the preparation of a run fails on one rank, every rank's run returns an error
     * and none has launched anything (the ranks agree on a status word first).  0 = no communicator. */
    int32_t comm_world;
    int32_t comm_rank;
    const void* comm_id;     /* MCMCPP_HIP_COMM_ID_BYTES bytes made by mcmcpp_hip_comm_unique_id on one rank and handed to
                                all ranks by the caller's own means (file, MPI, torch.distributed ...), or NULL */
    void* comm;              /* or: an existing ncclComm_t of comm_world ranks on this handle's device (borrowed) */
    /* Independent ensembles ("chains") stepped by the same kernel launches -- what several EnsembleSampler objects with
     * seeds randSeed, randSeed + 1, ... would compute one after the other (BASELINE config 4 on one GPU).  Chain k is an
     * ensemble of num_walkers walkers of its own, seeded with seed + k on the same stream.  0 or 1: one ensemble.  With
     * K = num_chains > 1 every array argument gains a leading chain dimension: set_state positions[K][W][D], logp[K][W];
     * run chain_out[K][n_saved][W][D] (chain k's stored steps are contiguous, like that sampler's own Chain),
     * accepted_per_step[K][steps]; get_state likewise.  Counters are summed over the chains.  Whole ensembles on one
     * device (no shards, no communicator); at most 16 chains. */
    int32_t num_chains;
    int32_t reserved0;
} mcmcpp_hip_config;

#define MCMCPP_HIP_COMM_ID_BYTES 128
/* ncclGetUniqueId: the rendezvous token of a new communicator (call on one rank). */
int mcmcpp_hip_comm_unique_id(void* id_out);

/* Movers (reference MCMCpp/Movers/).  DIFFERENTIAL_EVOLUTION = Mover::DifferentialEvolution
 * (Movers/DifferentialEvolution.h:80-112) with the gamma and jitter bounds its constructors set -- the samplers work on
 * a copy of the user's mover whose copy constructor re-initialises both, so overrideGamma / overrideRandBounds never
 * reach a sampler in the reference either.  Same stream, same draws, same chain as the reference, bit for bit
 * (tests/test_diffevo.py); one whole ensemble per handle: run / set_state / get_state / counters / calc_logp are
 * supported, seek, shards and half_step_async are not (the stream position depends on the draws thrown away so far). */
#define MCMCPP_HIP_MOVER_STRETCH 0u
#define MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION 1u

/* launch every kernel and copy on the caller's stream (config.hip_stream), so that the caller's own work on
 * that stream -- e.g. RCCL collectives issued through torch.distributed -- is ordered with the half-steps */
#define MCMCPP_HIP_FLAG_CALLER_STREAM 1u

/* User-compiled device Calculators (the reference takes the Calculator as a template argument of the Mover,
 * Movers/StretchMove.h:42-43; here it is a device functor built with hipcc against
 * mcmcpp_amd/csrc/mcmcpp_hip_plugin.hpp).  table_f64 / table_f32 are what the plug-in's
 * mcmcpp_hip_plugin_<name>_f64() / _f32() return (either may be NULL); calc_id >= MCMCPP_HIP_CALC_USER_BASE;
 * params_len >= 0 makes mcmcpp_hip_create insist on exactly that many parameters, -1 accepts any.
 * Registering an id again replaces it (handles created earlier keep what they were created with). */
int mcmcpp_hip_register_calculator(int32_t calc_id, const void* table_f64, const void* table_f32, int32_t params_len);

/* EnsembleSampler::EnsembleSampler / ~EnsembleSampler */
int mcmcpp_hip_create(const mcmcpp_hip_config* cfg, mcmcpp_hip_sampler** out);
void mcmcpp_hip_destroy(mcmcpp_hip_sampler* h);

/* message of the last failure on this handle (h == NULL: last failure of mcmcpp_hip_create on this thread) */
const char* mcmcpp_hip_last_error(const mcmcpp_hip_sampler* h);

/* EnsembleSampler::setInitialWalkerPos (EnsembleSampler.h:220-230): uploads positions[W*D] and
 * auxValues[W], zeroes the per-walker counters and rewinds the random stream to its first draw. */
int mcmcpp_hip_set_state(mcmcpp_hip_sampler* h, const void* positions, const void* logp);

/* EnsembleSampler::runMCMC (EnsembleSampler.h:284-310) with the slicing mode of setSlicingMode folded in:
 * executes n_saved*interval ensemble steps (red half-step then black half-step each,
 * EnsembleSampler.h:341-354); the last step of every `interval` is a stored step.
 *   chain_out          n_saved*W*D elements receiving the stored steps, or NULL to store nothing
 *   accepted_per_step  n_saved*interval counters (accepted proposals of each executed ensemble step), or NULL
 * Synchronous: returns when the results are in host memory. */
int mcmcpp_hip_run(mcmcpp_hip_sampler* h, int64_t n_saved, int32_t interval, void* chain_out,
                   uint32_t* accepted_per_step);

/* The same run on a worker thread owned by the handle, so that the caller can look at stored steps while the device keeps
 * stepping -- what a PostStepAction needs (EnsembleSampler.h:356-359: called after every ensemble step with the chain as it
 * stands).  run_async returns at once; wait_stored(count) returns when the first `count` stored steps of this run are
 * complete in chain_out (or with the run's error code if it ended before that); run_wait joins the worker and returns what
 * mcmcpp_hip_run would have returned.  Between run_async and run_wait the handle accepts only wait_stored. */
int mcmcpp_hip_run_async(mcmcpp_hip_sampler* h, int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step);
int mcmcpp_hip_wait_stored(mcmcpp_hip_sampler* h, int64_t count);
int mcmcpp_hip_run_wait(mcmcpp_hip_sampler* h);

/* Pinned (page-locked, device-visible) host memory for Chain blocks (stands in for the allocation of
 * Chain/ChainBlock.h:115-131).  When chain_out of mcmcpp_hip_run lies in such memory the step launches forward stored
 * steps straight into it: no staging ring on the host, no host copy.  NULL when the allocation fails. */
void* mcmcpp_hip_host_alloc(uint64_t bytes);
void mcmcpp_hip_host_free(void* p);

/* Current walker state (what Walker::getCurrState / getCurrAuxData / getAcceptedProposals expose,
 * Walker/Walker.h:111-122).  n_accept[w] counts accepted proposals since set_state or reset_counters;
 * the reference additionally counts the initial placement (Walker.h:76,168) -- the facade adds it.
 * Counts stay below 2^31 (the reference's counters are 32-bit ints as well).  Any pointer may be NULL. */
int mcmcpp_hip_get_state(mcmcpp_hip_sampler* h, void* positions, void* logp, uint32_t* n_accept);

/* Checkpoint / resume.  The random stream has no per-walker state: (seed, stream, ensemble steps done) address
 * every draw, so a checkpoint is get_state's arrays plus one integer.  After set_state (which rewinds to the
 * first draw) this repositions the stream as if `ensemble_steps_done` steps had been executed, without touching
 * walkers or counters.  (The reference cannot do this: its engine state is private to the Mover.) */
int mcmcpp_hip_seek(mcmcpp_hip_sampler* h, uint64_t ensemble_steps_done);

/* EnsembleSampler::reset (EnsembleSampler.h:312-322): zero the counters, keep positions and stream. */
int mcmcpp_hip_reset_counters(mcmcpp_hip_sampler* h);

/* getAcceptedSteps / getTotalSteps material (EnsembleSampler.h:260-282), plus two parity diagnostics:
 *   accepted        sum of n_accept over the walkers this handle updates
 *   ensemble_steps  ensemble steps executed since set_state / reset_counters
 *   near_ties       accept decisions whose margin was within a few ulp (could flip under another libm log)
 *   redraws         pcg bounded_rand rejections met (only possible when W/2 is not a power of two; the
 *                   reference would have consumed one more draw there, so trajectories part)
 * Any pointer may be NULL. */
int mcmcpp_hip_get_counters(mcmcpp_hip_sampler* h, uint64_t* accepted, uint64_t* ensemble_steps,
                            uint64_t* near_ties, uint64_t* redraws);

/* Calculator::calcLogPostProb evaluated by the device functor for `count` D-vectors (host in, host out). */
int mcmcpp_hip_calc_logp(mcmcpp_hip_sampler* h, const void* positions, int64_t count, void* logp_out);

/* ---- measurement ---------------------------------------------------------------------------------- */

/* GPU time of the last mcmcpp_hip_run between HIP events recorded on the launch stream around the
 * step launches (excludes uploads / downloads), and the number of step-kernel launches it covers: one per
 * ensemble step for ensembles small enough to be stepped by the full-step kernel, two (red, black) otherwise. */
int mcmcpp_hip_last_run_timing(mcmcpp_hip_sampler* h, double* gpu_ms, int64_t* step_launches);

/* Host-side cost of the last mcmcpp_hip_run: the time the calling thread spent enqueueing the step launches (and, for a
 * split ensemble, the exchanges), the wall time of the whole call, and -- split ensembles only -- the GPU time of one
 * ensemble step's exchange, averaged over a sample of steps bracketed by HIP events (0 when none was sampled).
 * Any pointer may be NULL. */
int mcmcpp_hip_last_run_host_timing(mcmcpp_hip_sampler* h, double* enqueue_ms, double* wall_ms, double* exchange_us_per_step);

/* Split ensembles: what the exchanges of the last mcmcpp_hip_run moved.  Ranks of a communicator of more than one rank
 * exchange only the rows that moved (index, log-posterior, row, packed into blocks of `block_slots` walkers per rank and
 * all-gathered once per ensemble step -- once per half-step for slices too large for the one-launch-per-step kernels);
 * the bound is learned from the run, and a chunk of steps in which some rank moved more walkers than a block holds is
 * REPEATED from a snapshot with blocks that hold a whole slice, so the chain never depends on the bound.
 *   bytes_per_step   bytes this rank received per ensemble step (all-gathering whole slices: (G-1)/G of the ensemble)
 *   repeated_chunks  chunks of steps that were rolled back and repeated
 *   block_slots      walkers per block at the end of the run (0: whole slices were all-gathered)
 * Any pointer may be NULL. */
int mcmcpp_hip_last_run_exchange(mcmcpp_hip_sampler* h, double* bytes_per_step, int64_t* repeated_chunks, int64_t* block_slots);

/* ---- multi-GPU single ensemble (one handle per GPU) -------------------------------- */

/* (A handle with a communicator needs none of these: mcmcpp_hip_run drives the split ensemble itself.  They remain for
 * callers that bring their own exchange.)
 * Enqueue ONE half-step (color 0 = red, 1 = black) for this handle's shard on its stream and return
 * without waiting.  The caller then exchanges the updated rows (device_positions + offset, see
 * mcmcpp_hip_shard_span) with the other handles -- e.g. an in-place RCCL all-gather on the same
 * stream -- before enqueueing the next half-step.  save_slot >= 0 also writes the shard's rows of
 * that stored step into the device chain buffer given to mcmcpp_hip_bind_device_chain. */
int mcmcpp_hip_half_step_async(mcmcpp_hip_sampler* h, int32_t color, int64_t save_slot);
int mcmcpp_hip_bind_device_chain(mcmcpp_hip_sampler* h, void* device_chain, int64_t slots);
/* device pointer of the ensemble positions (W*D elements) and the element span [offset, offset+count)
 * this handle rewrites during a half-step of `color` */
void* mcmcpp_hip_device_positions(mcmcpp_hip_sampler* h);
int mcmcpp_hip_shard_span(mcmcpp_hip_sampler* h, int32_t color, int64_t* offset_elems, int64_t* count_elems);
int mcmcpp_hip_synchronize(mcmcpp_hip_sampler* h);

/* ---- chain analysis on the device -------------------------------------------- */

/* Analysis::CovarianceMatrix (reference MCMCpp/Analysis/CovarianceMatrix.h:154-257): mean, covariance and
 * correlation matrix of stored chain steps.  The sums  s_i = sum x_i,  S_ij = sum x_i x_j  over all samples are a
 * rank-N update X^T X and run on the matrix cores (fp64 accumulation for both element types); the slicing the
 * reference's calculateCovar(start, end, sliceInterval) does while iterating is expressed by step_stride.
 *   create      dtype MCMCPP_HIP_F64 / _F32 = the chain's ParamType; device -1 = current
 *   add_steps   n_steps stored steps of num_walkers*num_params elements each (host memory), consecutive ones
 *               step_stride steps apart; may be called repeatedly (chain blocks, several chains)
 *   finish      CovarianceMatrix::finalizeMatrix: mean[D], cov[D*D], corr[D*D] in the chain's element type (any may
 *               be NULL); *num_points = samples used.  The accumulator keeps its sums: more steps may follow.
 * The reference's sequential Kahan sums cannot be kept by a parallel sum: results agree with it to rounding
 * (tests/test_moments.py states the tolerance), not bit for bit. */
typedef struct mcmcpp_hip_moments mcmcpp_hip_moments;
int mcmcpp_hip_moments_create(int32_t dtype, int32_t device, int32_t num_walkers, int32_t num_params, mcmcpp_hip_moments** out);
void mcmcpp_hip_moments_destroy(mcmcpp_hip_moments* m);
int mcmcpp_hip_moments_reset(mcmcpp_hip_moments* m);
int mcmcpp_hip_moments_add_steps(mcmcpp_hip_moments* m, const void* steps, int64_t n_steps, int64_t step_stride);
/* the same for n_steps contiguous stored steps that already live in device memory (e.g. a device chain bound with
 * mcmcpp_hip_bind_device_chain): no upload; returns when the sums are updated */
int mcmcpp_hip_moments_add_device_steps(mcmcpp_hip_moments* m, const void* device_steps, int64_t n_steps);
int mcmcpp_hip_moments_finish(mcmcpp_hip_moments* m, int64_t* num_points, void* mean, void* cov, void* corr);
const char* mcmcpp_hip_moments_last_error(const mcmcpp_hip_moments* m);

/* Analysis::AutoCorrCalc::calcAutoCorrTimes (reference MCMCpp/Analysis/AutoCorrCalc.h:151-207, with
 * Analysis/Detail/AutoCov.h:146-322): the integrated autocorrelation time of every parameter.  Per (walker, parameter)
 * series: Kahan average, circular autocovariance by two radix-2 FFTs over the next power of two >= n_steps, divided by
 * lag 0; per parameter: the walkers' functions Kahan-summed and divided by their number, then the windowed sum
 * (window_scaling = the reference's setAutoCorrScaleFactor value, default 4).  One workgroup per series, transforms in LDS.
 *   steps          n_steps pointers to stored steps of num_walkers*num_params elements (host memory), oldest first
 *   walkers_to_use 0 = all; otherwise that many walkers, evenly spaced over the ensemble (the reference draws a
 *                  subset seeded from std::random_device, i.e. irreproducibly)
 *   times          [num_params]: the time, or minus the final sum where the window never closed (as the reference)
 *   functions      NULL, or [num_params][n_steps]: the averaged autocovariance functions
 * Every operation is the reference's own expression in the reference's order (twiddle factors from the host's libm):
 * results are bit-identical to the oracle's restatement, which is pinned bit for bit to the reference's AutoCov.  Not
 * reproduced: the reference's transferWalker ADDS each series onto what its scratch array holds (the previous
 * walker's function; uninitialised memory for the first), see INTEGRATION.md 4b. */
int mcmcpp_hip_autocorr_times(int32_t dtype, int32_t device, const void* const* steps, int64_t n_steps, int32_t num_walkers, int32_t num_params,
                              int32_t walkers_to_use, int32_t window_scaling, void* times, void* functions);
/* the same for a chain that already lives in device memory: n_steps contiguous stored steps ([n_steps][W][D], e.g. a device
 * chain bound with mcmcpp_hip_bind_device_chain); no upload */
int mcmcpp_hip_autocorr_times_device(int32_t dtype, int32_t device, const void* device_steps, int64_t n_steps, int32_t num_walkers,
                                     int32_t num_params, int32_t walkers_to_use, int32_t window_scaling, void* times, void* functions);
const char* mcmcpp_hip_autocorr_last_error(void);

int mcmcpp_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MCMCPP_HIP_H */
