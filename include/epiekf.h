/*
 * epiekf.h -- C ABI of libepiekf.so: the MI355X (gfx950) ensemble engine for the
 * reference's per-region EKF/EKS hot path.
 *
 * Drop-in boundary.  The reference's interface for this path is a family of
 * MATLAB functions with one signature (Tools/SIAlphaModelEKF.m:1,
 * Tools/SIAlphaModelEKFOptControlled.m:1, Tools/SIAlphaModelBackwardEKF.m:1,
 * Tools/SIAlphaModelBackwardEKFOptControlled.m:1,
 * Tools/NewCaseEKFEstimatorWithOptimalNPI.m:1):
 *
 *   [u_opt, u_opt_smooth, S_MINUS, S_PLUS, S_SMOOTH, P_MINUS, P_PLUS, P_SMOOTH,
 *    K_GAIN, innovations, rho] = F(u, x, params, s_init, Ps_init, s_final,
 *    Ps_final, w_bar, v_bar, Q_w, R_v, beta, gamma, inv_monitor_len, order)
 *
 * A MEX gateway / ctypes stub binds exactly the entry points below (see
 * INTEGRATION.md).  One call runs B independent filter chains ("chains" =
 * region x cost-factor x Monte-Carlo member); B = 1 with the identity series
 * maps reproduces one reference call, and with B = 1 every array below has
 * exactly MATLAB's column-major memory layout (u: n_npi x T, S: m x T,
 * P: m x m x T), so a gateway can pass mxGetPr() pointers straight through.
 *
 * Memory layout (batched, "SoA"): time-major, then row, then chain:
 *   x         [T][Sx]            observations; NaN = missing (GenericEKF.m:122)
 *   u         [T][n_npi][Su]     controls; NaN = "choose optimally" (OptControlled.m:49-58)
 *   R_series  [T][Sx]            R_v given as 1xT vector (fixed_R = false, GenericEKF.m:82-85)
 *   R_scalar  [B]                R_v given as a scalar   (fixed_R = true,  GenericEKF.m:79-81)
 *   prm       [EPI_PRM_COUNT][B] params struct + v_bar/beta/gamma (epiekf_layout.h)
 *   s_init    [m][B]   Ps_init [m*m][B]   s_final [m][B]   Ps_final [m*m][B]   Q [m*m][B]
 *   outputs   S_* [T][m][B], P_* [T][m*m][B] (element e = row + m*col), K_GAIN [T][m][B],
 *             u_opt* [T][n_npi][B], innovations/rho [T][B]
 * Chain c reads column x_series[c] of x / R_series and column u_series[c] of u
 * (NULL map = identity, then Sx resp. Su must equal B): the Pareto sweep's 250
 * cost factors of one region share that region's series
 * (Tools/TrainPredictPrescribeNPI.m:421-460).
 *
 * Ownership / threading / errors: the caller owns every buffer; the library
 * never frees or retains caller memory; every entry point may be called from
 * several threads at once; all device work of a *_device call is enqueued on
 * the caller's stream (stages that are off the critical path run on a helper
 * stream that is forked from and joined back into the caller's stream, so the
 * call behaves like work on that one stream, also under stream capture: a
 * helper stream that joined a caller's capture is not handed to any other call
 * afterwards -- it is retired until epi_host_pool_release()).
 * What the library keeps between calls, all of it freed by
 * epi_host_pool_release(): the CU count of each device it has seen, idle
 * helper streams (one per concurrent call and device, with their events), and
 * for the *_host entry points a pool of contexts per device (stream, device
 * arena, pinned staging buffer) and one worker thread per device used by the
 * *_multi calls.  Return value 0 or a negative epi_status; `err` (256 bytes,
 * may be NULL) receives the reference's own error() text for the four
 * reference errors.
 */
#ifndef EPIEKF_H
#define EPIEKF_H

#include <stddef.h>
#include <stdint.h>
#include "epiekf_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EPIEKF_ABI_VERSION 6

/* which reference function the chain runs */
typedef enum epi_model {
    EPI_MODEL_SIA3 = 0,          /* Tools/SIAlphaModelEKF.m                      (m = 3) */
    EPI_MODEL_SIA6 = 1,          /* Tools/SIAlphaModelEKFOptControlled.m         (m = 6) */
    EPI_MODEL_SIA3_BWD = 2,      /* Tools/SIAlphaModelBackwardEKF.m              (m = 3) */
    EPI_MODEL_SIA6_BWD = 3,      /* Tools/SIAlphaModelBackwardEKFOptControlled.m (m = 6) */
    EPI_MODEL_NEWCASE6 = 4,      /* Tools/NewCaseEKFEstimatorWithOptimalNPI.m    (m = 6) */
    EPI_MODEL_NEWCASE6_CODEGEN = 5 /* MatlabCodeGenerator/NewCaseEKFEstimatorWithOptimalNPI.m */
} epi_model;

typedef enum epi_obs_type {      /* params.obs_type, Tools/SIAlphaModelEKF.m:52-58 */
    EPI_OBS_NEWCASES = 0,
    EPI_OBS_TOTALCASES = 1
} epi_obs_type;

typedef enum epi_status {
    EPI_OK = 0,
    EPI_ERR_UNDEFINED_ORDER = -1, /* 'Undefined order'  GenericExtendedKalmanFilter.m:111,151 */
    EPI_ERR_Q_MISMATCH = -2,      /* 'Process noise covariance noise mismatch'      :75 */
    EPI_ERR_R_MISMATCH = -3,      /* 'Observation noise covariance noise mismatch'  :90 */
    EPI_ERR_OBS_TYPE = -4,        /* 'unknown observation type'  SIAlphaModelEKF.m:57,87 */
    EPI_ERR_BAD_ARG = -5,         /* NULL / size / range error in the descriptor */
    EPI_ERR_WORKSPACE = -6,       /* workspace too small */
    EPI_ERR_HIP = -7,             /* HIP runtime failure (no device, launch error, ...) */
    EPI_ERR_UNSUPPORTED = -8
} epi_status;

/* output selection bits, in the order of the reference's output list */
typedef enum epi_out {
    EPI_OUT_U_OPT = 1 << 0,
    EPI_OUT_U_OPT_SMOOTH = 1 << 1,
    EPI_OUT_S_MINUS = 1 << 2,
    EPI_OUT_S_PLUS = 1 << 3,
    EPI_OUT_S_SMOOTH = 1 << 4,
    EPI_OUT_P_MINUS = 1 << 5,
    EPI_OUT_P_PLUS = 1 << 6,
    EPI_OUT_P_SMOOTH = 1 << 7,
    EPI_OUT_K_GAIN = 1 << 8,
    EPI_OUT_INNOVATIONS = 1 << 9,
    EPI_OUT_RHO = 1 << 10,
    EPI_OUT_ALL = (1 << 11) - 1
} epi_out;

typedef struct epi_batch_desc {
    int32_t abi_version;  /* EPIEKF_ABI_VERSION */
    int32_t model;        /* epi_model */
    int32_t B;            /* chains */
    int32_t T;            /* time samples  = size(x, 2) */
    int32_t Sx, Su;       /* distinct observation / control series */
    int32_t n_npi;        /* size(u, 1), 1..EPI_MAX_NPI */
    int32_t L;            /* inv_monitor_len */
    int32_t order;        /* 1 or 2 (2 is accepted: all SI-alpha Hessian callbacks are zero,
                             Tools/SIAlphaModelEKF.m:92-109) */
    int32_t obs_type;     /* epi_obs_type */
    int32_t r_mode;       /* 0: R_scalar[B] (scalar R_v, adaptive when beta != 1); 1: R_series */
    int32_t q_mode;       /* 0: fixed per-chain m x m Q_w, Q [m*m][B] (the only form the reference's callers use);
                             1: time-varying, Q [T][m*m][B] = Q(:,:,k) of filter step k (GenericEKF.m:63-73; a
                                length-T vector Q_w is q(k)*eye(m)); generic models only, dense kernels */
    uint32_t out_mask;    /* epi_out bits: which outputs are written */
    int32_t phase;        /* 0: forward EKF then backward EKS (one reference call).  For per-kernel timing a
                             caller may enqueue the stages one by one, in order, on the same buffers:
                             1 = forward kernel; 2 = smoother (3 then 4); 3 = pinv kernel; 4 = backward recursion */
    int32_t path_hint;    /* kernels of the generic (symmetrising) models: 0 = decide on the device (both the
                             symmetric-packed and the dense variant are enqueued, one returns at once);
                             1 = epi_ekf_precheck_device() said the batch qualifies for the symmetric-packed
                             kernels (Ps_init bit-wise symmetric, Q_w diagonal): enqueue only those;
                             2 = dense kernels only */
    int32_t time_pipe;    /* a full call (phase 0) of a generic model on the packed kernels (path_hint = 1, R_v a per-day
                             series, T >= 128) may run "pipelined in time": the forward kernel in four time segments (50, 35,
                             12, 3 % of the days), each followed by the eks_pinv grid of its days on a helper stream, so that
                             only the last days' pinv stands between the forward pass and the smoother.  0 = the library
                             decides (on when the batch leaves a quarter of the SIMDs idle -- the shards of the sweep on 2,
                             4, 8 GPUs), 1 = on, -1 = off.  Results are bit-identical either way. */
    int32_t lane_block;   /* layout of the OUTPUT arrays (and of the workspace) of epi_ekf_run_device.  0 or >= B: the
                             classic [T][rows][B].  blk in 1..B-1 (8 recommended): chain-blocked,
                             element (t, row, c) at ((t*nblk + c/blk)*rows + row)*blk + c%blk, nblk = ceil(B/blk) --
                             the rows of blk neighbouring chains form one contiguous block, which is what HBM wants
                             to see from ~100 concurrent stores per wave; arrays are then sized for
                             nblk*blk chains and one-row arrays (innovations, rho, pinv_rank) are [T][nblk*blk].
                             Inputs are never blocked.  epi_ekf_run_host accepts the classic layout only. */
    int32_t shape;        /* how the 6-state generic models are mapped to lanes (epi_shape): 0 = decide by batch size,
                             1 = one lane per chain (ekf_fwd_sym / eks_bwd_sym: least total work, what a batch that fills
                             the chip wants), 2 = four lanes per chain (ekf_fwd_quad / eks_bwd_quad: every 6 x 6 matrix as
                             a 2 x 2 grid of 3 x 3 blocks over a DPP quad; a ~2x shorter per-day instruction stream and
                             4x the wavefronts, what a batch that does NOT fill the chip wants -- DESIGN.md 4),
                             3 = one WAVEFRONT per chain (ekf_fwd_wave / eks_bwd_wave: lane e = i + 6 j owns element (i, j) of
                             every 6 x 6 matrix, operands exchanged through LDS; the shortest per-day latency, for batches of
                             at most one chain per SIMD -- the reference's own one-call-per-cost-weight loop; needs R_v as a
                             per-day series, else falls back to 2), 4 = SIX lanes per chain, ten chains per wavefront
                             (ekf_fwd_hex / eks_bwd_hex: lane j of a chain owns column j of every 6 x 6 matrix; the stored
                             covariances are symmetric bit for bit, so every product needs one transpose through LDS and the
                             Jacobian's zeros are skipped -- about half the quad shape's instructions per day; same conditions
                             as 3, except that a scalar R_v falls back to 2).  Auto: 3 up to 640 chains (up to 1 024 with a
                             scalar R_v, whose monitor the wave shape runs inline), 4 up to 20 480 (the 9 375-chain shard of the
                             headline sweep on one of 8 GPUs: 2.6 ms against 3.2 with 2 and 4.9 with 1), 2 up to 16 384 where 4
                             cannot run, then 1.
                             The 3-state generic models know 1 and 3 (there: SEVEN chains per wavefront, nine lanes each,
                             ekf_fwd_wave3 / eks_bwd_wave3; auto: 3 up to 2 048 chains).  Results are bit-identical in all
                             shapes.  NewCaseEKFEstimatorWithOptimalNPI knows 1 (the dense kernels) and 3 (auto: 3 up to 1 024 chains). */
    int32_t storage;      /* element type of the OUTPUT arrays: 0 = fp64 (the reference's), 1 = fp32 storage with fp64
                             register arithmetic (BASELINE config 5): every selected output is the fp64 result rounded
                             once to fp32; the forward quantities the smoother reads back stay fp64 in the workspace.
                             epi_ekf_run_device only. */
    int32_t exact_nonfinite; /* What happens to chains whose covariance overflows (status bit 0: the non-finite guard of
                             GenericEKF.m:211 fired).  The packed, quad and hex kernels skip products with structural zeros,
                             which is exact only for finite operands: after an overflow they may carry a finite number where
                             MATLAB has NaN.  0 (the default, ABI 5) = the reference's behaviour whenever the call runs the
                             smoother: the marked chains are run a second time by the dense kernels, in place, so that their
                             Inf / NaN pattern is the dense evaluation's -- the reference's, and the C oracle's -- at every
                             day; 1 = the same, and the smoother is run to find the chains even when no smoothed output is
                             selected; -1 = off, the outputs are what the fast kernels leave (`status` still tells which
                             chains).  Generic models, full call (phase 0), fixed Q_w, fp64 storage.  Cost when no chain is
                             marked: six small launches that return at once (~25 us; until ABI 4 the second pass's pinv grid
                             dispatched every (tile, step) workgroup -- 0.15 ms at 75 000 x 520 -- now it is 8 tiles wide and
                             walks the list of marked chains); (2 B + 1 + B) more int32 of workspace.  The *_host entry points
                             reach the same result without device-side launches: epi_ekf_run_host[_multi] look at the status
                             words that come back with the outputs and enqueue the second pass only when a chain is marked. */
    int32_t placement_tries; /* HOST-pointer entry points only (ABI 6; the device-pointer entry points ignore it: their caller owns
                             the allocation and can compare allocations with epi_ekf_time_stages_device).  Where the allocator
                             puts the ~14 arrays a pass streams concurrently changes the time of the forward kernel and of the
                             smoother by 5-15 % -- a property of the allocation, invisible to a caller who hands over host
                             arrays.  N > 1: when the call has to allocate a NEW device arena, its own kernels are timed on up to
                             N (<= EPI_PLACEMENT_MAX_TRIES) candidate arenas and the fastest is kept, with the pooled context,
                             for the calls that follow (epi_host_pool_release frees it); epi_outputs.placement receives the
                             report.  0 / 1 = off.  Cost: once per arena, ~N + 1 times the call's device time. */
    /* TEST HOOKS (ABI 6; until ABI 5 an environment variable read on every call).  Both are 0 in production: a default-
       constructed descriptor never sets them, nothing else in the library reads process-global state.  They only choose
       code paths that the batch size otherwise chooses, so that small tests reach them; results are identical for every value. */
    int32_t test_window;  /* > 0: the kernels that keep their buffer descriptors fixed over an addressing window of days (the hex
                             shape, ekf_hex.hpp; the one-lane kernels of ekf_lane6.hpp) use windows of at most that many days
                             (>= 2) instead of as many as fit 2 GiB.  It can only SHORTEN the window. */
    int32_t test_flags;   /* bit 0: a full call in the hex shape takes the reverse-time pipeline (pinv grids of the earlier days
                             beside the smoother's first launches) whatever the batch size and day count, which otherwise
                             engages beyond 768 hex wavefronts and 128 days only.
                             bit 1: the innovation monitor's scan kernel (ekf_monitor) replays rho whatever the batch size; below
                             65 537 chains the scan-free grid (ekf_monitor_par) otherwise does.
                             bit 2: a one-lane batch is cut into two chain ranges in the middle of its waves -- forward kernel,
                             pinv grid (the first range's beside the second forward launch) and, on the fixed-descriptor
                             smoother, the smoother with the monitor between its launches -- as batches of more waves than SIMDs
                             are cut after their first round of resident waves. */
} epi_batch_desc;

typedef enum epi_shape { EPI_SHAPE_AUTO = 0, EPI_SHAPE_LANE = 1, EPI_SHAPE_QUAD = 2, EPI_SHAPE_WAVE = 3, EPI_SHAPE_HEX = 4 } epi_shape;

typedef struct epi_inputs {
    const int32_t *x_series; /* [B] or NULL */
    const int32_t *u_series; /* [B] or NULL */
    const double *x, *u, *R_series, *R_scalar, *prm;
    const double *s_init, *Ps_init, *s_final, *Ps_final, *Q;
} epi_inputs;

#define EPI_PLACEMENT_MAX_TRIES 8
typedef struct epi_placement_report {
    int32_t tries;        /* candidate arenas timed; 0 = none (placement_tries <= 1, or the call reused a pooled arena) */
    int32_t chosen;       /* the one kept */
    float ms[EPI_PLACEMENT_MAX_TRIES];   /* device time of the call's kernels on each candidate */
} epi_placement_report;

typedef struct epi_outputs {
    double *u_opt, *u_opt_smooth;
    double *S_MINUS, *S_PLUS, *S_SMOOTH;
    double *P_MINUS, *P_PLUS, *P_SMOOTH;
    double *K_GAIN, *innovations, *rho;
    /* extras (not reference outputs; may be NULL) */
    int32_t *pinv_rank;   /* [T][B] rank kept by pinv at smoother step k (-1: not executed / guard) */
    int32_t *status;      /* [B] per-chain flags: bit0 non-finite P_MINUS guard hit (GenericEKF.m:211),
                             bit1 Jacobi sweep cap reached, bits 8.. minimum pinv rank seen */
    epi_placement_report *placement;   /* host-pointer entry points: what placement_tries > 1 did (may be NULL) */
} epi_outputs;

/* ---- EKF / EKS ---------------------------------------------------------- */
int epi_model_dim(int model);                                   /* 3, 6 or -1 */
int epi_ekf_validate(const epi_batch_desc *d, char *err);       /* descriptor checks only, no GPU */
size_t epi_ekf_workspace_bytes(const epi_batch_desc *d);        /* device scratch a run needs */
/* Synchronous: (forward + monitor, pinv grid, smoother) milliseconds -- ms[3] -- of this call on THESE device arrays, enqueued
 * stage by stage between HIP events on `stream`, averaged over as many rounds as fill `min_ms` of device time after one untimed
 * round.  For comparing ALLOCATIONS: the same arrays give the same times run after run, another allocation of the same arrays
 * may be 5-15 % slower (where its physical pages fall); a caller that will run many passes times one, allocates again while
 * holding the first, and keeps the faster (what placement_tries does for the host-pointer entry points). */
int epi_ekf_time_stages_device(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, void *workspace,
                               size_t workspace_bytes, void *stream, double min_ms, double *ms, char *err);
/* Synchronous: inspects Ps_init / Q (device pointers) and reports in *fast_ok whether path_hint = 1 is valid. */
int epi_ekf_precheck_device(const epi_batch_desc *d, const epi_inputs *in, void *stream, int *fast_ok, char *err);

/* The lane_block that matches the way epi_ekf_run_device will launch this batch on the current device: the number of
 * chains one wavefront handles (64, or fewer when the launch is split into equally full rounds; 16 where the 6-state
 * models run four lanes per chain, see `shape`).  With it every
 * wavefront's loads and stores of a step are one contiguous piece per array -- the fastest of the blocked layouts
 * (DESIGN.md 3).  Returns 0 for an invalid descriptor. */
int epi_ekf_preferred_lane_block(const epi_batch_desc *d);

/* All pointers in `in`/`out`/`workspace` are DEVICE pointers on the current HIP
 * device; `stream` is a hipStream_t (NULL = default stream).  Asynchronous:
 * returns after enqueueing.  Outputs not selected in out_mask may be NULL. */
int epi_ekf_run_device(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out,
                       void *workspace, size_t workspace_bytes, void *stream, char *err);

/* Same call on HOST pointers (what a MEX gateway calls): copies in, runs, copies the selected outputs back and
 * synchronises.  Replaces one call of Tools/SIAlphaModelEKF.m:1 (B = 1) or a whole loop of them
 * (Tools/TrainPredictPrescribeNPI.m:421-460, B = 250 cost weights).  Classic layout only (lane_block = 0).
 * Device memory, a pinned staging buffer and a stream come from a per-device pool of contexts that lives as long as the
 * library (no hipMalloc / hipFree per call after the first): a call whose inputs + outputs fit the staging buffer
 * (64 MiB) moves them with ONE host-to-device and ONE device-to-host copy.  Thread-safe. */
int epi_ekf_run_host(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out,
                     int device, char *err);

/* The same on SEVERAL GPUs of the node: the B chains are cut into n_devices contiguous blocks
 * (block r = chains [r * ceil(B / n_devices), ...), a shorter or empty last block), one host thread per block uploads
 * its chains, runs them on device_ids[r] (NULL: devices 0 .. n_devices-1) and writes the out_mask-selected outputs
 * straight into the caller's arrays -- chains are independent, so there is no exchange between the devices; this is the
 * loop over regions / cost weights of Tools/TrainPredictPrescribeNPI.m:93,421 spread over the GPUs.  A device may be
 * named more than once (two blocks then share it).  Returns the first error of any block. */
int epi_ekf_run_host_multi(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out,
                           int n_devices, const int *device_ids, char *err);

/* Frees everything the library keeps between calls: pooled host contexts (device arenas, pinned buffers, streams), idle
 * helper streams and their events, and the *_multi worker threads.  Optional: call before unloading the library.  Not to be
 * called while another thread is inside a library call. */
void epi_host_pool_release(void);

/* ---- forward simulators and cost (Tools/SIalpha_Controlled.m, SEIRP.m, NPICost.m) ---- */
typedef struct epi_sim_desc {
    int32_t abi_version;
    int32_t B;        /* chains */
    int32_t K;        /* steps */
    int32_t Su;       /* distinct control series */
    int32_t n_npi;
    int32_t noise;    /* 0: noise-free; 1: z given [K][3][B] standard normal draws */
    int32_t with_cost;/* 1: also J0/J1 of NPICost over the simulated span */
    int32_t prefix_days; /* epi_sialpha_score_device: days already summed into J0_prefix / J1_prefix */
    int32_t u_block;  /* 0 or >= Su: u is [K][n_npi][Su]; otherwise u is chain-blocked like an output of
                         epi_ekf_run_device with lane_block = u_block (u_opt_smooth fed straight into the scoring) */
} epi_sim_desc;

/* SIalpha_Controlled.m:1-32 batched.  sp [EPI_SIM_PRM_COUNT][B]; u [K][n_npi][Su];
 * outputs s,i,alpha [K][B] (initial sample dropped, :30-32); J0,J1 [B] when with_cost:
 * J0 = mean(s.*i.*alpha), J1 = mean(weights.*u) with weights [n_npi][B] constant over time. */
enum {
    EPI_SIM_S0 = 0, EPI_SIM_I0, EPI_SIM_ALPHA0, EPI_SIM_ALPHA_MIN, EPI_SIM_ALPHA_MAX, EPI_SIM_GAMMA,
    EPI_SIM_B, EPI_SIM_BETA, EPI_SIM_S_STD, EPI_SIM_I_STD, EPI_SIM_ALPHA_STD, EPI_SIM_DT,
    EPI_SIM_A = 12,      /* a(1:12)     */
    EPI_SIM_U_MAX = 24,  /* u_max(1:12) */
    EPI_SIM_W = 36,      /* NPICost weights(1:12), constant over time */
    EPI_SIM_PRM_COUNT = 48
};
int epi_sialpha_sim_device(const epi_sim_desc *d, const int32_t *u_series, const double *u,
                           const double *sp, const double *z, double *s, double *i, double *alpha,
                           double *J0, double *J1, void *stream, char *err);

/* Scenario scoring of the Pareto sweep (Tools/TrainPredictPrescribeNPI.m:481-493): simulate the horizon under the
 * smoothed optimal control from the end-of-history state and return NPICost over [historic days, horizon days].
 * J0_prefix[B] / J1_prefix[B] hold the sequential sums over the `prefix_days` historic days of
 * s.*i.*alpha and of weights(:).*inputs(:) (column-major order); the kernel continues both sums in order, so
 * J0 = mean([newcases_hist, newcases_sim]) and J1 = mean(weights.*[u_hist, u_sim]) exactly as NPICost.m:6-10. */
int epi_sialpha_score_device(const epi_sim_desc *d, const int32_t *u_series, const double *u, const double *sp,
                             const double *z, const double *J0_prefix, const double *J1_prefix, double *s, double *i,
                             double *alpha, double *J0, double *J1, void *stream, char *err);

/* Tools/NPICost.m:1-10 batched over B chains: J0 = mean(newcases), J1 = mean(weights(:).*inputs(:)), both summed
 * sequentially in MATLAB's column-major element order (NPI index fastest, then time).  newcases [T][B];
 * inputs [T][n_npi][Su] with u_series [B] or NULL (identity, Su == B); weights [T][n_npi][B] when
 * weights_per_day != 0, else [n_npi][B] (the same weights every day, as TrainPredictPrescribeNPI.m:485-493 builds
 * them).  J0, J1 [B]. */
int epi_npi_cost_device(int32_t B, int32_t T, int32_t n_npi, int32_t Su, int32_t weights_per_day,
                        const int32_t *u_series, const double *newcases, const double *inputs, const double *weights,
                        double *J0, double *J1, void *stream, char *err);

/* Random-NPI Monte-Carlo scenarios of a region (Tools/TrainPredictPrescribeNPI.m:496-521): n_scen plans on the K
 * forecast days with u(jj,t) = randi([NPI_MINS(jj), NPI_MAXES(jj)]) -- scenarios with 1-based index < n_scen/2 are
 * constant over time (:502), the rest are redrawn every day -- each simulated with SIalpha_Controlled from the
 * end-of-history state and scored with NPICost over [historic days, forecast days] (u = cat(2, IP, u), :515-517).
 * Chain c = scenario * R + region.  sp [EPI_SIM_PRM_COUNT][R] per region (EPI_SIM_U_MAX rows are NPI_MAXES);
 * u_min [n_npi][R] = NPI_MINS; z [K][3][n_scen*R] standard-normal draws when noise != 0, else NULL;
 * J0_prefix/J1_prefix [R]: sequential sums over the prefix_days historic days (as for epi_sialpha_score_device;
 * NULL when prefix_days == 0); u_out [K][n_npi][n_scen*R] or NULL; J0, J1 [n_scen][R].
 * The integer draws come from Philox4x32-10 keyed by (seed_lo, seed_hi) with counter (region, scenario, NPI/4, day):
 * reproducible on any launch geometry and by the CPU oracle; MATLAB's own randi stream is not reproduced. */
typedef struct epi_mc_desc {
    int32_t abi_version;
    int32_t R;          /* regions */
    int32_t n_scen;     /* scenarios per region (500 in the reference) */
    int32_t K;          /* forecast days */
    int32_t n_npi;
    int32_t noise;
    int32_t prefix_days;
    uint32_t seed_lo, seed_hi;
} epi_mc_desc;
int epi_random_npi_mc_device(const epi_mc_desc *d, const double *sp, const double *u_min, const double *z,
                             const double *J0_prefix, const double *J1_prefix, double *u_out, double *J0, double *J1,
                             void *stream, char *err);

/* Pareto-front filter and optimum of the sweep (Tools/TrainPredictPrescribeNPI.m:624-633), per region:
 * on_front(ii) = (sum(J0 < J0(ii) & J1 < J1(ii)) == 0);  [~, I_opt] = min((J0/max(J0)).^2 + (J1/max(J1)).^2).
 * J0, J1 [R][P] (region-major -- the chain order of the sweep); on_front [R][P] (0/1) or NULL; i_opt [R] 0-based or
 * NULL.  P <= 8192 (the points of a region are staged in LDS). */
int epi_pareto_front_device(int32_t R, int32_t P, const double *J0, const double *J1, int32_t *on_front,
                            int32_t *i_opt, void *stream, char *err);

/* ---- the Pareto sweep over the NPI-cost weights as ONE call (Tools/TrainPredictPrescribeNPI.m:421-493, 624-633) ----
 * The reference walks `for ll = 1 : num_pareto_front_points` (:421): SIAlphaModelEKFOptControlled with
 * params.epsilon = human_npi_cost_factor(ll) (:460), SIalpha_Controlled over the horizon under opt_control_input_smooth from
 * the end-of-history state (:481), NPICost over [historic, horizon] (:493); after the loop the non-dominated points and
 * I_opt (:624-633).  epi_sweep_run_device is epi_ekf_run_device (same descriptor, inputs, outputs, workspace; model
 * EPI_MODEL_SIA6, phase ignored = full call, fp64 u_opt_smooth selected) followed by epi_sialpha_score_device on the last
 * T - t_hist days of the u_opt_smooth it wrote and, when on_front / i_opt are given, epi_pareto_front_device -- enqueued so
 * that scoring and filter run BESIDE the smoother's pass over the observed days (the horizon's u_opt_smooth is final after
 * the smoother's first T - 1 - t_hist steps; packed kernels, path_hint = 1; otherwise they follow it).  Results are
 * bit-identical to the three separate calls.
 *   sp [EPI_SIM_PRM_COUNT][B], J0_prefix / J1_prefix [B]: as for epi_sialpha_score_device (EPI_SIM_S0.. = s/i/alpha_historic
 *   (end), prefix_days = t_hist);  J0, J1 [B];  on_front [R][P] / i_opt [R] or NULL (both NULL: no filter -- a shard of the
 *   sweep that does not hold whole regions; then R, P are ignored). */
typedef struct epi_sweep_desc {
    int32_t abi_version;
    int32_t R;        /* regions of this call */
    int32_t P;        /* cost weights per region: chain c = region * P + ll, B == R * P */
    int32_t t_hist;   /* NumNPIdays: observed days; the T - t_hist days after them are the horizon, 1 <= t_hist < T */
} epi_sweep_desc;
int epi_sweep_run_device(const epi_batch_desc *d, const epi_inputs *in, const epi_outputs *out, void *workspace,
                         size_t workspace_bytes, const epi_sweep_desc *sd, const double *sp, const double *J0_prefix,
                         const double *J1_prefix, double *J0, double *J1, int32_t *on_front, int32_t *i_opt, void *stream,
                         char *err);

/* The same from HOST pointers and PER-REGION inputs, for all regions at once and on several GPUs (what a MEX gateway
 * binds: matlab/epiekf_pipeline_mex.cpp; replaces the two loops Tools/TrainPredictPrescribeNPI.m:93 and :421 around the
 * 6-state filter).  The host sends R columns (a few hundred KB), the device expands them to the R * P chains, runs
 * filter -> scoring -> front filter, and returns (J0, J1) per chain, the front, I_opt and the optimum's plan per region --
 * a few MB instead of the 5.6 - 49 GB of per-chain filter outputs.  Regions are cut into n_devices contiguous blocks (whole
 * regions, so the front filter needs no exchange); device_ids NULL = devices 0 .. n_devices-1.
 * Inputs (host, region-minor: a MATLAB R x rows matrix is the [rows][R] array):
 *   x [T][R], u [T][n_npi][R] (NaN over the horizon), R_series [T][R];
 *   prm [EPI_PRM_COUNT][R] (row EPI_PRM_EPSILON is ignored), s_init [6][R], Ps_init [36][R], s_final [6][R],
 *   Ps_final [36][R], Q [36][R];  eps [P] = human_npi_cost_factor;
 *   sp [EPI_SIM_PRM_COUNT][R], J0_prefix / J1_prefix [R]  (scoring inputs, see epi_sialpha_score_device).
 * Outputs (host; any may be NULL):
 *   J0, J1 [R][P];  on_front [R][P];  i_opt [R] (0-based);
 *   u_opt [T][n_npi][R] = u_opt_smooth of chain (r, i_opt[r]);  S_opt [T][6][R] = its S_SMOOTH;
 *   extras: per-chain filter outputs selected by out_mask, classic layout [T][rows][R * P] (pinv_rank / status are not
 *   returned); with out_mask != 0 the filter writes the classic layout and the call pays PCIe for what it selects. */
typedef struct epi_prescribe_desc {
    int32_t abi_version;
    int32_t R, P;            /* regions, cost weights per region */
    int32_t T, t_hist;       /* days incl. the horizon; observed days */
    int32_t n_npi, L, order, obs_type;
    uint32_t out_mask;       /* epi_out bits of the per-chain extras (0 = none) */
    int32_t shape, time_pipe;/* as in epi_batch_desc (0 = let the library decide) */
    int32_t placement_tries; /* as in epi_batch_desc: N > 1 = a new device arena is the fastest of up to N candidates */
} epi_prescribe_desc;
typedef struct epi_prescribe_inputs {
    const double *x, *u, *R_series;
    const double *prm, *s_init, *Ps_init, *s_final, *Ps_final, *Q;
    const double *eps;
    const double *sp, *J0_prefix, *J1_prefix;
} epi_prescribe_inputs;
typedef struct epi_prescribe_outputs {
    double *J0, *J1;
    int32_t *on_front, *i_opt;
    double *u_opt, *S_opt;
    epi_outputs extras;
    epi_placement_report *placement;   /* what placement_tries > 1 did on the first device's block (may be NULL) */
} epi_prescribe_outputs;
int epi_sweep_prescribe_host(const epi_prescribe_desc *d, const epi_prescribe_inputs *in, const epi_prescribe_outputs *out,
                             int n_devices, const int *device_ids, char *err);

/* ---- Tools/Rt_ExpFitEKF.m:1 -- 2-state exponential-fit EKF/EKS over the new-case counts, order 1 or 2 ----
 * [S_MINUS, S_PLUS, P_MINUS, P_PLUS, K_GAIN, S_SMOOTH, P_SMOOTH, innovations, rho] =
 *     Rt_ExpFitEKF(x, s_init, params, w_bar, v_bar, Ps_init, Q_w, R_v, beta, gamma, inv_monitor_len, order)
 * batched over B chains: x [T][Sx] (NaN = missing/forecast day), x_series [B] or NULL (identity, Sx == B),
 * rp [EPI_RT_PRM_COUNT][B] holding every other argument of the signature per chain.  Outputs [T][2][B] (S_*, K_GAIN),
 * [T][4][B] (P_*, column-major 2 x 2), [T][B] (innovations, rho).  S_MINUS, S_PLUS, P_MINUS, P_PLUS are required
 * (the smoother reads them back); the others may be NULL.  `order` other than 1 or 2 returns
 * EPI_ERR_UNDEFINED_ORDER ('Undefined order', Rt_ExpFitEKF.m:46,77).  exp/tanh are evaluated in a fixed operation
 * order (< 1 ulp / a few ulp from libm) that the CPU oracle shares: results are reproducible bit for bit. */
enum {
    EPI_RT_TIME_SCALE = 0, EPI_RT_ALPHA = 1, EPI_RT_SIGMA = 2,   /* params(1:3) */
    EPI_RT_W_BAR = 3,      /* w_bar(1:2) */
    EPI_RT_V_BAR = 5, EPI_RT_R_V = 6, EPI_RT_BETA_EKF = 7, EPI_RT_GAMMA_EKF = 8,
    EPI_RT_S_INIT = 9,     /* s_init(1:2) */
    EPI_RT_PS_INIT = 11,   /* Ps_init(:), column-major */
    EPI_RT_Q_W = 15,       /* Q_w(:), column-major */
    EPI_RT_PRM_COUNT = 19
};
typedef struct epi_rt_desc {
    int32_t abi_version;
    int32_t B, T, Sx;
    int32_t L;       /* inv_monitor_len, 1..106 */
    int32_t order;   /* 1 or 2 */
} epi_rt_desc;
typedef struct epi_rt_outputs {
    double *S_MINUS, *S_PLUS, *P_MINUS, *P_PLUS, *K_GAIN, *S_SMOOTH, *P_SMOOTH, *innovations, *rho;
} epi_rt_outputs;
int epi_rt_expfit_validate(const epi_rt_desc *d, char *err);
/* device pointers, enqueues on `stream` */
int epi_rt_expfit_run_device(const epi_rt_desc *d, const int32_t *x_series, const double *x, const double *rp,
                             const epi_rt_outputs *out, void *stream, char *err);
/* host pointers (what a MEX gateway calls): uploads, runs, downloads, synchronises */
int epi_rt_expfit_run_host(const epi_rt_desc *d, const int32_t *x_series, const double *x, const double *rp,
                           const epi_rt_outputs *out, int device, char *err);

/* ---- per-region preprocessing: data-set columns -> filter inputs (Tools/TrainPredictPrescribeNPI.m:142-198,201-202,240) ----
 * All series are [T][S] (day-major, region-minor) -- the x / R_series layout of epi_inputs; ip / ip_filled are
 * [T][n_npi][S] -- the u layout.  cases (and deaths, optional) are CUMULATIVE confirmed counts with NaN for missing
 * days; population [S].  For every region:
 *   new_refined  = diff([c(1); c]), negatives -> 0, a NaN last day <- last valid day, other NaN -> 0      (:166-178)
 *   new_smoothed = filter(ones(1,W), W, new_refined)                                                        (:173)
 *   zero_lag     = filtfilt(ones(1,W2), W2, new_refined), W2 = round(W/2)                                   (:174)
 *   x_new = new_smoothed / N;  x_total = cumsum(new_smoothed) / N                                           (:175-180)
 *   R_v   = 0.1 * ((zero_lag - new_refined) / N).^2                                                         (:240)
 *   fatality = cumsum(filter(.., deaths part)) ./ cumsum(new_smoothed), NaN -> 0                            (:183-197)
 *   I0    = max(min_cases, mean(first `first_num_days` positive samples of new_smoothed))                   (:201-202)
 *   ip_filled: N/A (NaN) levels take the previous day's level, leading N/A -> 0                            (:142-150)
 * Any output pointer may be NULL.  T must exceed 3*(W2-1) (filtfilt's 'Data length must be larger than ...'
 * error) and be >= 2 (:168 'Insufficient data'); 1 <= W <= 32. */
typedef struct epi_pre_desc {
    int32_t abi_version;
    int32_t S, T, n_npi;
    int32_t W;               /* SmoothingWinLen (7 in the reference) */
    int32_t first_num_days;  /* first_num_days_for_case_estimation */
    double min_cases;
} epi_pre_desc;
typedef struct epi_pre_outputs {
    double *new_refined, *new_smoothed, *zero_lag, *x_new, *x_total, *R_v, *fatality;   /* [T][S] */
    double *I0;                                                                         /* [S] */
    double *ip_filled;                                                                  /* [T][n_npi][S] */
} epi_pre_outputs;
size_t epi_preprocess_workspace_bytes(const epi_pre_desc *d);
int epi_preprocess_device(const epi_pre_desc *d, const double *cases, const double *deaths, const double *population,
                          const double *ip, const epi_pre_outputs *out, void *workspace, size_t workspace_bytes,
                          void *stream, char *err);

/* ---- regression between the EKF rounds (Tools/TrainPredictPrescribeNPI.m:251-276, 'NONNEGATIVELS') ----
 * For every region:  reg_coef_a = lsqnonneg(X, y); reg_coef_b = 0; then the loop :266-276 (at most max_iters = 100
 * passes: coef_temp = lsqnonneg(X, y - reg_coef_b), coef0_temp = mean(y - X*reg_coef_a), accepted while the squared
 * error decreases).  X [D][n][S] = NPI_MAXES - InterventionPlans over the regression window, y [D][S] = the smoothed
 * alpha estimate; outputs a [n][S], b [S], min_err [S] (may be NULL), iters [S] (accepted passes, may be NULL),
 * flag [S] (lsqnonneg exit flag of the first solve: 1, or 0 when its inner loop hit 3n iterations; may be NULL).
 * lsqnonneg = Lawson & Hanson's active-set algorithm with MATLAB's tolerance 10*eps*norm(X,1)*length(X), evaluated on
 * the normal equations with a diagonally pivoted Cholesky for the passive-set solves.  1 <= n <= 12. */
typedef struct epi_nnls_desc {
    int32_t abi_version;
    int32_t S, D, n;
    int32_t max_iters;   /* NONNEGATIVELS_IRERATIONS (100 in the reference) */
} epi_nnls_desc;
int epi_nnls_affine_fit_device(const epi_nnls_desc *d, const double *X, const double *y, double *a, double *b,
                               double *min_err, int32_t *iters, int32_t *flag, void *stream, char *err);

/* Host-pointer forms of the three stages around the filter (same arrays in host memory; staged through device `device`,
 * synchronous): what matlab/epiekf_pipeline_mex.cpp binds for TrainPredictPrescribeNPI.m:142-198 (preprocessing),
 * :251-276 (regression) and :496-521 (random-NPI Monte-Carlo). */
int epi_preprocess_host(const epi_pre_desc *d, const double *cases, const double *deaths, const double *population,
                        const double *ip, const epi_pre_outputs *out, int device, char *err);
int epi_nnls_affine_fit_host(const epi_nnls_desc *d, const double *X, const double *y, double *a, double *b,
                             double *min_err, int32_t *iters, int32_t *flag, int device, char *err);
int epi_random_npi_mc_host(const epi_mc_desc *d, const double *sp, const double *u_min, const double *z,
                           const double *J0_prefix, const double *J1_prefix, double *u_out, double *J0, double *J1,
                           int device, char *err);

/* SEIRP.m:1-32 / SEIRPSaturatedResource.m:1-38 batched: par [K][7][B] per-step parameter arrays in the
 * order alpha_e, alpha_i, kappa, rho, beta, mu, gamma (or constant-in-time: par [1][7][B], par_steps=1);
 * init [5][B] = s0,e0,i0,r0,p0; out [K][5][B].  saturated != 0: sat [6][B] = beta_0,beta_s,mu_0,mu_s,
 * sigma,i_0 and par rows 4,5 (beta, mu) are ignored.  integrator: 0 = explicit Euler (the reference,
 * parity), 1 = classical RK4 (extension; no reference oracle). */
int epi_seirp_sim_device(int32_t B, int32_t K, int32_t par_steps, double dt, int32_t saturated,
                         int32_t integrator, const double *par, const double *init, const double *sat,
                         double *out, void *stream, char *err);

/* Tools/SI_Controlled.m:1-23 batched: 2-state forward Euler with a time-dependent infection rate.  alpha [K-1][Sa]
 * (the loop reads alpha(1 : K-1)) with alpha_series [B] or NULL (identity, Sa == B); prm [3][B] = beta, s0, i0;
 * outputs s, i [K][B], the first sample being the initial condition (:15-16). */
int epi_si_controlled_device(int32_t B, int32_t K, int32_t Sa, double dt, const int32_t *alpha_series, const double *alpha,
                             const double *prm, double *s, double *i, void *stream, char *err);
int epi_si_controlled_host(int32_t B, int32_t K, int32_t Sa, double dt, const int32_t *alpha_series, const double *alpha,
                           const double *prm, double *s, double *i, int device, char *err);

/* testScripts/testSIR01.m:15-36 (BASELINE config 1) batched: the 3-compartment SIR with return flow r -> s, forward Euler
 * without clamps, s(t+1) = (-alpha s i + gamma r) dt + s etc.  prm [6][B] = alpha, beta, gamma, s0, i0, r0 per parameter set;
 * out [K][3][B] (rows s, i, r; the first sample is the initial state, :28-30). */
int epi_sir_sim_device(int32_t B, int32_t K, double dt, const double *prm, double *out, void *stream, char *err);
int epi_sir_sim_host(int32_t B, int32_t K, double dt, const double *prm, double *out, int device, char *err);

/* Host-pointer variants of the three entry points above (same arrays in host memory; the library stages them through
 * device `device` and synchronises): what a MEX gateway for SIalpha_Controlled.m / SEIRP.m / SEIRPSaturatedResource.m /
 * NPICost.m binds (matlab/epiekf_sim_mex.cpp). */
int epi_sialpha_sim_host(const epi_sim_desc *d, const int32_t *u_series, const double *u, const double *sp,
                         const double *z, double *s, double *i, double *alpha, double *J0, double *J1, int device,
                         char *err);
int epi_seirp_sim_host(int32_t B, int32_t K, int32_t par_steps, double dt, int32_t saturated, int32_t integrator,
                       const double *par, const double *init, const double *sat, double *out, int device, char *err);
int epi_npi_cost_host(int32_t B, int32_t T, int32_t n_npi, int32_t Su, int32_t weights_per_day, const int32_t *u_series,
                      const double *newcases, const double *inputs, const double *weights, double *J0, double *J1,
                      int device, char *err);

/* Measurement utility (not part of the reference's interface): copies n doubles src -> dst with the filter
 * kernels' access shape (8 B per lane); used to calibrate the HBM traffic counters on a known byte count. */
int epi_calib_copy_f64_device(const double *src, double *dst, size_t n, void *stream, char *err);

const char *epi_status_string(int status);
int epi_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
