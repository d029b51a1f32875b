/* vof.h - C ABI of the MI355X-native variational optical-flow solver (libvof.so).
 *
 * The reference has no FFI layer: its boundary is the Python function
 *   source/optical_flow.py:715-724  variational_optical_flow(movie, delta_x, delta_t, speed_alpha,
 *                                    remodelling_alpha, smoothing_sigma, initial_v_x, initial_v_y,
 *                                    initial_remodelling, use_direct_solver)
 * The entry points below are what a binding for that function calls (see INTEGRATION.md for the
 * ctypes stub); each one names the reference lines it replaces.  Plain pointers and sizes only,
 * no exceptions cross the ABI, no global state (contrast PETSc.Options(), OF.py:1081-1092).
 *
 * Conventions: images are row-major (N_i, N_j) float64, axis 0 = "x" = i, axis 1 = "y" = j
 * (OF.py:730-733).  Pair k is (frame k, frame k+1) (OF.py:794-795).  Return value 0 = success,
 * negative = error (message via vof_last_error).  Non-convergence is NOT an error: it is reported
 * per pair in vof_pair_stats, like the reference which only prints a warning (OF.py:1135-1138).
 */
#ifndef VOF_H
#define VOF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOF_VERSION 202 /* 0.2.2: vof_pair_stats carries the batch time (0.2.1: vof_params carries its own size and the ABI version) */

typedef struct vof_ctx vof_ctx;

/* Solver parameters.  Defaults (vof_default_params) reproduce the reference's settings.
 * ABI guard: the first two fields are filled in by vof_default_params and checked by every entry point that takes a
 * vof_params (a binding compiled against another layout is rejected with an error instead of being read past its end). */
typedef struct vof_params {
    uint32_t struct_size;      /* sizeof(vof_params) of the header the caller was built with */
    uint32_t abi_version;      /* VOF_VERSION of that header */
    double speed_alpha;        /* OF.py:718  */
    double remodelling_alpha;  /* OF.py:719  */
    double delta_x;            /* OF.py:716; velocities are returned in delta_x/delta_t units (OF.py:1189-1190) */
    double delta_t;            /* OF.py:717  */
    double initial_v_x;        /* OF.py:721,800: initial guess in delta_x/delta_t units */
    double initial_v_y;        /* OF.py:722,801 */
    double initial_remodelling;/* OF.py:723,802 */
    double rtol;               /* OF.py:1120: 1e-6, ||b - A x||_2 <= rtol ||b||_2 (unpreconditioned, OF.py:1126) */
    int32_t max_iterations;    /* OF.py:1120: 1000 BiCGStab iterations */
    int32_t nu_pre;            /* block-GS sweeps before the coarse-grid correction on level 0 (default 2) */
    int32_t nu_post;           /* ... and after (default 2) */
    int32_t reference_quirks;  /* 1 (default): OF.py:698-699 'dy' == 'dx'; OF.py:1205 speed_functional bug */
    int32_t coarse_precision;  /* storage of the Galerkin stencils (preconditioner only): 3 (default) 8-bit float off-diagonal
                                  blocks (units of a power of two per block position) + float32 diagonal block that absorbs their
                                  rounding errors (block row sums kept; 120 bytes per coarse point, +1 % iterations); 2: the same
                                  with bfloat16 off-diagonal blocks (180 bytes, iteration counts of float32); 1: float32 (324
                                  bytes); 0: float64 */
    int32_t vcycle_precision;  /* storage of the V-cycle vectors: 0 float64; 1 float32; 2 auto = float32 for the first 8
                                  iterations, float64 afterwards; 3 (default) = float64 on level 0, float32 on the levels below
                                  for the first 8 iterations.  Krylov vectors, operator products, residuals and the stopping rule are
                                  always FP64; so is the arithmetic of every cycle kernel, except the level-0 smoother of the
                                  float32-vector modes 1 / 2 (k_sweep0p: packed float32 - part of the preconditioner only) */
    int32_t nu_pre_coarse;     /* sweeps on the levels >= 1 (default 1); 0 = same as nu_pre / nu_post */
    int32_t nu_post_coarse;
    int32_t w_cycle_level;     /* l >= 0 (default 1): level l visits level l+1 w_cycle_visits times per cycle (a one-level W-cycle); -1: V-cycle */
    int32_t w_cycle_visits;    /* visits of level w_cycle_level + 1 per cycle (default 3); 0 = 2 */
    int32_t krylov_method;     /* 0: BiCGStab only (the reference's KSP type 'bcgs', OF.py:1081); 1: restarted GMRES only;
                                  2 (default): BiCGStab, then GMRES(gmres_restart) for the pairs that have not met the
                                  stopping rule after fallback_after iterations or broke down.  Same preconditioner,
                                  same stopping rule; `iterations` counts the Krylov steps of both phases */
    int32_t gmres_restart;     /* restart length (default 100, at most 128); the basis takes (restart + 1) float64 vectors per pair in
                                  flight and is capped at half of the free device memory when the fallback first runs */
    int32_t fallback_after;    /* BiCGStab iterations before the fallback (default 25) */
    int32_t warm_start_stride; /* vof_solve_stack_dev / _host (per batch), stacks whose first phase is >= 16 Mpixel of pairs: every stride-th pair is solved first from the
                                  constant initial fields, the others start from the solution of their nearest solved neighbour
                                  (the reference warm-starts pair k from pair k-1, OF.py:803-806).  Default 3; 0 or 1: every pair
                                  starts from the constant initial fields.  Same stopping rule either way */
    int32_t preconditioner;    /* 0: multigrid cycle only; 1: direct - block-tridiagonal LU of the level-0 operator by image rows
                                  (dense 3 n_j x 3 n_j Schur blocks inverted in-house; n_i (3 n_j)^2 doubles per pair in flight), the
                                  reference's SuperLU branch (OF.py:1146-1147) as the preconditioner of the same Krylov iteration;
                                  2 (default): the multigrid cycle (at most 150 Krylov steps where the direct re-solve takes
                                  seconds: images up to ~530 pixels wide), and the pairs it leaves unconverged once more with the
                                  direct preconditioner when its buffers fit into the free device memory */
    int32_t reserved0;         /* keeps the size a multiple of 8; must be 0 */
} vof_params;

/* Per-pair solver report (the reference prints these: OF.py:1131-1154). */
typedef struct vof_pair_stats {
    int32_t iterations;        /* BiCGStab iterations used */
    int32_t converged;         /* 1 iff the INDEPENDENT residual below meets the stopping rule, ||b - A x||^2 <= rtol^2 ||b||^2
                                  (OF.py:1135 solver.is_converged, OF.py:1120,1126), evaluated on the residual recomputed
                                  from x after the solve - never on the solver's recursively updated residual */
    double relative_residual;  /* independent ||A x - b|| / ||b|| after the solve (OF.py:1151) */
    double L1_functional;      /* OF.py:1178-1180 */
    double speed_functional;   /* OF.py:1181-1182 (the true one; the dict-level bug is applied by the caller) */
    double remodelling_functional; /* OF.py:1183 */
    double batch_ms;           /* GPU time (HIP events on the solver's stream) of the batch this pair was solved in: hierarchy
                                  set-up + Krylov iteration + epilogue of batch_pairs pairs advancing together, i.e. this
                                  pair's share is batch_ms / batch_pairs (the reference prints per-pair wall times,
                                  OF.py:1073-1076, 1156-1157); a pair re-solved by a fallback carries the sum of its shares */
    int32_t batch_pairs;       /* pairs in that batch */
    int32_t reserved;          /* 0 */
} vof_pair_stats;

/* Kernel classes for the built-in HIP-event profiler (vof_profile_*). */
enum vof_kernel_id {
    VOF_K_RHS = 0, VOF_K_APPLY0, VOF_K_GS0, VOF_K_GS, VOF_K_RESIDUAL, VOF_K_RESTRICT, VOF_K_PROLONG,
    VOF_K_GALERKIN0, VOF_K_GALERKIN, VOF_K_COARSE_SETUP, VOF_K_COARSE_SOLVE, VOF_K_VECTOR, VOF_K_REDUCE,
    VOF_K_FINALIZE, VOF_K_FUNCTIONALS, VOF_K_COARSE_TAIL, VOF_K_COUNT
};

int vof_version(void);
/* sizeof(vof_params) of the library (a binding asserts that its own struct has this size). */
size_t vof_params_size(void);
/* Fills *p with the defaults.  struct_size = sizeof(vof_params) as the CALLER declares it: if it differs from the
 * library's, nothing is written and -1 is returned (a stale binding can neither be overrun nor half-initialised). */
int vof_default_params(vof_params* p, size_t struct_size);

/* Environment variables read once by vof_create (A/B experiment switches; results are identical, only speed changes):
 *   VOF_SWEEP_GEO=AA|AB|BA|BB  strip geometry of the fused sweep on (level 0, stored levels); default AB
 *   VOF_STREAM_APPLY=0         level-0 operator: simple kernel instead of the LDS-streaming one
 *   VOF_FUSE_RESTRICT=0        level 0: separate residual and restriction kernels
 *   VOF_FUSE_PROLONG=0         level 0: separate prolongation kernel instead of interpolating inside the first post-sweep
 *   VOF_SWEEP0=0               level 0: the generic fused sweep kernel instead of the dedicated k_sweep0
 *   VOF_DIRECT_LU=own|blocked|rocsolver  direct preconditioner: dense inverse of the Schur blocks by the one-workgroup kernel / the
 *                              blocked inverse on the matrix cores / rocSOLVER (default: own up to 192 unknowns per image row,
 *                              blocked beyond; rocSOLVER only when asked for here - an A/B reference, loaded with dlopen)
 *   VOF_ROCSOLVER_LIB=path     rocSOLVER library to load (VOF_DIRECT_LU=rocsolver only)
 *   VOF_FUSE_APPLY=0           the Krylov product after a cycle: separate operator kernel instead of the trailing stage of the
 *                              cycle's last smoothing pass
 *   VOF_SWEEP0M=0|1            level 0, float64 vectors: 0 = the 4-wave kernel k_sweep0; 1 = k_sweep0m with one sweep per pass
 *                              (default: k_sweep0m, two sweeps per pass)
 *   VOF_SWEEP0R=0              level 0, float64 vectors: the LDS-ring pass k_sweep0m instead of the register-resident k_sweep0r
 *   VOF_SWEEP0R_MIN_BLOCKS=n   ... k_sweep0r from n one-wave blocks per launch on (default 512; smaller launches use k_sweep0m)
 *   VOF_FUSE_B=0               the BiCGStab updates s = r - alpha v and p = r + beta (p - omega v) by their stand-alone kernels instead
 *                              of inside the first pre-smoothing pass of the cycle that consumes them (same bits)
 *   VOF_FUSE_RR=0              level 0: the coarse right-hand side R (b - A x) by the stand-alone residual + restriction kernel instead
 *                              of as the trailing stage of the pre-smoothing pass
 *   VOF_SWEEP0P=0              level 0, float32 vectors: k_sweep0 (float64 arithmetic) instead of the packed-float32 k_sweep0p
 *   VOF_PRECOND_QUIRKS=hs      experiment: hierarchy (h) / smoother (s) of the preconditioner with (1) or without (0) the 'dy' == 'dx' quirk
 *   VOF_COARSEST_MAX=3..9      coarsen until max(n_i, n_j) <= this (default 5); changes the hierarchy depth, hence iteration counts
 *   VOF_COARSE_TAIL=0          levels whose whole grid fits one workgroup: one launch per operation instead of the fused
 *                              LDS-resident coarse-tail kernel
 *   VOF_FUSE_RESU=0            stored levels: stand-alone residual + restriction kernels instead of the coarse right-hand side
 *                              from the last sweep's update (k_resrestrict_u)
 *   VOF_SWEEP_ST=0             stored levels, packed stencil formats: the generic k_sweep instead of k_sweep_st
 *   VOF_FOLD_STORED=1          stored levels: coarse-grid correction interpolated inside the first post-sweep
 *   VOF_SKIP_COLOUR0=0         W-cycle revisits: full first pre-smoothing sweep (default: colour 0 is left alone, same bits)
 *   VOF_TRACE=1                direct preconditioner: progress lines on stderr
 * Debug switches (fault attribution; they change timing, never results):
 *   VOF_DEBUG_SYNC=1           the context's stream is synchronised and asked for its error after every launch scope; the
 *                              first failure is reported on stderr and appended to every later error text as
 *                              "scope #n, kernel class 'name', level l, pairs, image size: error"
 *   VOF_DEBUG_SYNC_FILE=path   (with VOF_DEBUG_SYNC) the scope about to be waited for is written to this file first, so that
 *                              a process the driver aborts leaves the name of the launch that was in flight
 *   VOF_DEBUG_CANARY=1         every device buffer of the context is allocated between two 4-KiB guard regions of a known byte
 *                              pattern, checked by vof_debug_check_canaries and vof_destroy (out-of-bounds WRITES are named
 *                              by buffer and offset)
 *   VOF_DEBUG_POISON=1         every device buffer is filled with 0xFF bytes (NaN as floating point) when it is allocated: a read of
 *                              workspace nothing has written shows up as a non-finite result
 *   VOF_DEBUG_ALLOC_LOG=1      base, end, size and name of every device buffer on stderr (maps a faulting address to a buffer)
 *   VOF_TRACE_HOST=1           vof_solve_stack_host: timeline of the host pipeline (page touching, pinning, copies, batches) on stderr */

/* One context = one device = one host thread at a time.  Owns device workspaces for images of
 * (n_i, n_j) and up to max_pairs_in_flight frame pairs solved concurrently (batch dimension).
 * stream: a hipStream_t to launch on, or NULL for a context-owned stream. */
int vof_create(vof_ctx** out, int device_id, int n_i, int n_j, int max_pairs_in_flight, void* stream);
void vof_destroy(vof_ctx* ctx);
const char* vof_last_error(const vof_ctx* ctx); /* ctx may be NULL: error of the last failed vof_create */
size_t vof_workspace_bytes(const vof_ctx* ctx);
/* Device bytes a context for (n_i, n_j, max_pairs_in_flight) allocates (without host-API staging) when it is used with the
 * default storage formats; _for: with the given vof_params.coarse_precision / vcycle_precision (the stencil storage of the
 * stored levels is sized by the format in use: 120 / 180 / 324 / 648 bytes per coarse point; a context re-allocates it,
 * never shrinking, when a call asks for a wider format than any call before). */
size_t vof_query_workspace(int n_i, int n_j, int max_pairs_in_flight);
size_t vof_query_workspace_for(int n_i, int n_j, int max_pairs_in_flight, int coarse_precision, int vcycle_precision);
/* Free / total device memory in bytes; returns 0 on success. */
int vof_device_memory(int device_id, size_t* free_bytes, size_t* total_bytes);
int vof_num_levels(const vof_ctx* ctx);

/* Replaces the frame-pair loop OF.py:791-1186 + epilogue OF.py:1189-1191 for a whole stack.
 * movie: (n_frames, n_i, n_j) float64 (already blurred if blurring is wanted, OF.py:770-773).
 * v_x, v_y, remodelling, speed: (n_frames-1, n_i, n_j) float64, caller allocated; speed may be NULL.
 * stats: n_frames-1 entries, host memory, may be NULL.
 * _host: all array pointers are host memory (copies are staged by the library).
 * _dev : all array pointers are device memory on the context's device (no PCIe traffic). */
int vof_solve_stack_host(vof_ctx* ctx, const double* movie, int n_frames, const vof_params* p,
                         double* v_x, double* v_y, double* remodelling, double* speed,
                         vof_pair_stats* stats);
int vof_solve_stack_dev(vof_ctx* ctx, const double* movie, int n_frames, const vof_params* p,
                        double* v_x, double* v_y, double* remodelling, double* speed,
                        vof_pair_stats* stats);

/* Replaces blur_movie (OF.py:282-306): per-frame Gaussian blur, scipy.ndimage.gaussian_filter semantics
 * (mode='nearest'); weights = the 2*radius+1 normalised taps in host memory (radius = int(4*sigma + 0.5) for the
 * reference's skimage call).  The context fixes the frame size; any number of frames. */
int vof_blur_stack_dev(vof_ctx* ctx, const double* in_dev, double* out_dev, int n_frames, const double* weights, int radius);
int vof_blur_stack_host(vof_ctx* ctx, const double* in_host, double* out_host, int n_frames, const double* weights, int radius);

/* Summary of one (speed_alpha, remodelling_alpha) combination of vary_regularisation (OF.py:1978-1983). */
typedef struct vof_variation_stats {
    double speed_mean, speed_variance;             /* np.mean / np.var of result['speed'] (OF.py:1978-1979) */
    double remodelling_mean, remodelling_variance; /* ... of result['remodelling'] (OF.py:1980-1981) */
    double L1_functional, speed_functional, remodelling_functional; /* sums over the pairs; speed_functional is the true
                                                      alpha * sum |grad u|^2 (the caller applies the OF.py:1205 quirk) */
    double max_relative_residual;                  /* worst pair */
    int32_t converged_last;                        /* flag of the last pair = result['converged'] (OF.py:1202, 1982) */
    int32_t converged_all;                         /* 1 if every pair met the stopping rule */
    int32_t max_iterations_used;                   /* worst pair */
    int32_t reserved;
} vof_variation_stats;

/* Replaces vary_regularisation (OF.py:1918-1998): every (speed_alphas[i], remodelling_alphas[j]) combination is an
 * independent solve of the same movie.  The movie is uploaded (and blurred, if blur_weights != NULL; taps as for
 * vof_blur_stack_*) once, all solves and the mean / variance reductions run on the device, and only the summaries
 * cross PCIe.  base: all other parameters.  out: n_speed_alphas * n_remodelling_alphas entries, row-major [i][j]. */
int vof_vary_regularisation_host(vof_ctx* ctx, const double* movie, int n_frames, const vof_params* base,
                                 const double* speed_alphas, int n_speed_alphas,
                                 const double* remodelling_alphas, int n_remodelling_alphas,
                                 const double* blur_weights, int blur_radius, vof_variation_stats* out);

/* Mean and (population) variance of n device-resident doubles, deterministic two-pass reduction. */
int vof_field_moments_dev(vof_ctx* ctx, const double* field_dev, size_t n, double* mean, double* variance);

/* Replaces the sampling loop of subsample_velocities_for_visualisation (OF.py:1614-1632) for a device-resident
 * field stack (n_fields, n_i, n_j): out[k][a][b] = field[k][a*box + offset][b*box + offset], a < n_i / box,
 * b < n_j / box (the reference uses offset = round(box / 2)).  out_dev: (n_fields, n_i/box, n_j/box) doubles. */
int vof_subsample_dev(vof_ctx* ctx, const double* field_dev, int n_fields, int box, int offset, double* out_dev);

/* Benchmark harness (no counterpart in the reference; SURVEY.md section 8(d) fixes the recipe): n_frames frames of the
 * exactly translating synthetic texture, written to device memory,
 *   I_t(i, j) = clip(0.5 + scale * sum_k a_k cos(2 pi (f_k (i - ox_t) + g_k (j - oy_t)) / period + phi_k), 0, 1).
 * mode_params: host, 4 * n_modes doubles (f, g, a, phi); frame_offsets: host, (ox_t, oy_t) per frame. */
int vof_texture_stack_dev(vof_ctx* ctx, double* out_dev, int n_frames, const double* mode_params, int n_modes,
                          const double* frame_offsets, double period, double scale);

/* Smoother implementation: 1 (default) = fused streaming 4-colour sweep (one launch per sweep),
 * 0 = one launch per colour (the simple reference kernels, kept for A/B tests). */
int vof_set_fused_sweeps(vof_ctx* ctx, int on);

/* Fixed-work kernel benchmark (SURVEY 8(d) "fixed sweep count"): n_sweeps full 4-colour block-GS
 * sweeps of the fine level on n_pairs pairs of a device-resident movie.  Used by scripts/gpu_sweep_micro.py and
 * scripts/gpu_sweep_shape.py (bench.py times the whole solve and reads the per-class profiler instead). */
int vof_bench_sweeps_dev(vof_ctx* ctx, const double* movie, int n_pairs, const vof_params* p, int n_sweeps);

/* Built-in profiler: when enabled, every kernel launch is bracketed by HIP events on the
 * context's stream; totals are accumulated per kernel class and multigrid level. */
int vof_profile_enable(vof_ctx* ctx, int on);
int vof_profile_reset(vof_ctx* ctx);
/* Restrict event recording to one kernel class / level (-1 = any); keeps the timed region light. */
int vof_profile_filter(vof_ctx* ctx, int kernel_id, int level);
/* level < 0: sum over levels.  Outputs: number of launches, total milliseconds. */
int vof_profile_get(vof_ctx* ctx, int kernel_id, int level, int64_t* launches, double* total_ms);
/* Sum over the recorded launches of the number of frame pairs each launch actually processed
 * (converged pairs are skipped by later launches), i.e. the "units" of the roofline figure. */
int vof_profile_get_units(vof_ctx* ctx, int kernel_id, int level, int64_t* pair_launches);
/* Sum over the recorded launches of their algorithmic bytes (bytes per pixel of DESIGN.md section 3 x
 * level pixels x pairs processed); 0 for kernel classes that do not report it. */
int vof_profile_get_bytes(vof_ctx* ctx, int kernel_id, int level, double* algorithmic_bytes);
/* Same sum with the bytes a launch minimally has to MOVE.  Equal to the algorithmic bytes except for the level-0 smoother
 * k_sweep0m, where one pass over the data performs two sweeps (algorithmic: 80 bytes per pixel and sweep performed; moved:
 * 80 per pixel and pass). */
int vof_profile_get_moved(vof_ctx* ctx, int kernel_id, int level, double* moved_bytes);
const char* vof_kernel_name(int kernel_id);

/* ---- debug / test entry points: single building blocks on device memory of the context -------
 * All vectors are interior-grid vectors of level `level`, layout [pair][3][n_i(level)][n_j(level)].
 * vof_debug_setup must be called first (uploads frames, builds the Galerkin hierarchy). */
int vof_debug_setup(vof_ctx* ctx, const double* movie_host, int n_pairs, const vof_params* p);
/* 0: all guard regions intact (or VOF_DEBUG_CANARY off); -5: damaged, vof_last_error names buffer and offset */
int vof_debug_check_canaries(vof_ctx* ctx);
int vof_debug_level_shape(vof_ctx* ctx, int level, int* n_i, int* n_j);
int vof_debug_rhs(vof_ctx* ctx, double* b_host);                                   /* level 0 */
int vof_debug_apply(vof_ctx* ctx, int level, const double* x_host, double* y_host); /* y = A_l x */
int vof_debug_gs(vof_ctx* ctx, int level, double* x_host, const double* b_host, int colour);
/* one full fused 4-colour sweep (reverse: colour order 3,2,1,0; from_zero: ignore x, start from 0) */
int vof_debug_sweep(vof_ctx* ctx, int level, double* x_host, const double* b_host, int reverse, int from_zero);
/* nu full sweeps as the multigrid cycle runs them (on level 0: the passes of k_sweep0m, two sweeps each) */
int vof_debug_smooth(vof_ctx* ctx, int level, double* x_host, const double* b_host, int nu, int reverse, int from_zero);
int vof_debug_restrict(vof_ctx* ctx, int level, const double* fine_host, double* coarse_host);
int vof_debug_prolong_add(vof_ctx* ctx, int level, double* fine_host, const double* coarse_host);
/* stored level >= 1: coarse right-hand side R (b - A x_new) straight after ONE forward sweep x_old -> x_new with right-hand
 * side b (x_old_host NULL: the sweep started from zero), computed from the sweep's update alone (k_resrestrict_u) */
int vof_debug_resrestrict_u(vof_ctx* ctx, int level, const double* x_new_host, const double* x_old_host, double* coarse_host);
int vof_debug_stencil(vof_ctx* ctx, int level, double* c_host); /* [pair][81][n_i][n_j], level >= 1 */
int vof_debug_vcycle(vof_ctx* ctx, const double* r_host, double* e_host);
int vof_debug_coarse_solve(vof_ctx* ctx, const double* r_host, double* e_host);
/* y = M r (one cycle), v = A y, dots[2 k] = (v, r), dots[2 k + 1] = (v, v) of pair k - the Krylov step as the solver runs it;
 * *fused = 1 if the product came out of the cycle's last smoothing pass (k_sweep0m's trailing stage) */
int vof_debug_vcycle_apply(vof_ctx* ctx, const double* r_host, double* y_host, double* v_host, double* dots_host, int* fused);

#ifdef __cplusplus
}
#endif
#endif /* VOF_H */
