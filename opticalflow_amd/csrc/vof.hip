// vof.hip - host side of libvof.so: context, workspace, multigrid hierarchy, BiCGStab driver, C ABI.
//
// Replaces the per-pair body of source/optical_flow.py::variational_optical_flow (OF.py:791-1186:
// scipy.sparse assembly + PETSc KSP bcgs / composite PC) with a batched, matrix-free solve on one
// MI355X: right-preconditioned BiCGStab (the reference's KSP type, OF.py:1081) whose preconditioner is
// one geometric-multigrid V-cycle with a 4-colour 3x3-block Gauss-Seidel smoother and Galerkin coarse
// operators; stopping rule ||b - A x|| <= rtol ||b|| (OF.py:1120,1126).  All frame pairs of a batch
// advance together; per-pair scalars stay on the device.
#include "vof_device.hpp"
#include "vof_sweep0r.hpp"
#include "vof_sweep0p.hpp"
#include "vof_direct.hpp"
#include "../../include/vof.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace vof;

namespace {

constexpr int COARSEST_MAX = 5;      // coarsen until max(n_i, n_j) <= COARSEST_MAX (5: dense inverse of <= 75 unknowns; was 9 = 243 unknowns, whose
                                     // Gauss-Jordan inversion took 4.4 ms per batch - the fused coarse-tail kernel makes the extra level free)
int coarsest_max() {                 // experiment switch VOF_COARSEST_MAX=3..9 (read per call: create and workspace query agree)
    if (const char* e = getenv("VOF_COARSEST_MAX")) { int v = atoi(e); if (v >= 3 && v <= 9) return v; }
    return COARSEST_MAX;
}
constexpr int MAX_PROF_RECS = 32768;
constexpr int AUTO_F64_AFTER = 8;   // vcycle_precision == 2: switch the V-cycle vectors to float64 after this many iterations

struct Level {
    int ni = 0, nj = 0;
    size_t npts = 0;
    void* C = nullptr;   // stored stencil [B][81][npts] (double or float); level 0: only for 1-level grids
    // V-cycle vectors; element type is float or double (ctx->vfloat), allocations are sized for double
    void* x = nullptr;  // levels >= 1
    void* b = nullptr;  // levels >= 1
    void* r = nullptr;  // residual scratch (all levels but the last)
    void* x2 = nullptr; // ping-pong partner of x for the out-of-place fused sweeps
};

struct ProfRec {
    hipEvent_t e0, e1;
    int kid, level, units;
    double bytes;  // algorithmic bytes of this launch (units x bytes per pair; SURVEY 8(d): per sweep / operator application performed)
    double moved;  // minimal bytes the launch has to move (differs when one pass performs two sweeps)
};

}  // namespace

struct vof_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int Ni = 0, Nj = 0, B = 0;
    std::vector<Level> L;
    double *kx = nullptr, *kb = nullptr, *kr = nullptr, *krh = nullptr, *kp = nullptr, *kv = nullptr, *kt = nullptr;
    double* ky = nullptr;   // V-cycle outputs y = M p and z = M s (V-typed: float when vfloat)
    double* kz = nullptr;
    double* b32 = nullptr;  // V-typed copy of the V-cycle right-hand side (p or s) when vfloat
    bool vfloat = false;    // V-cycle vectors stored as float32 (arithmetic stays FP64)
    bool emit64 = false;    // vcycle_precision 3: the level-1 visit in progress hands its result up as float64 (see vcycle_t)
    bool vcoarse32 = false; // vcycle_precision 3: float64 vectors on level 0, float32 on the levels below (the two meet in the fused
                            // residual + restriction kernel and in the post-smoothing pass that interpolates the correction)
    const PairParam* pp = nullptr;   // per-pair (alpha, beta, frame) overrides of the current batch ("virtual pairs") or nullptr
    PairParam* pp_buf = nullptr;     // device storage for them (B entries, lazy)
    // warm start (two-phase solve of a stack): interior solutions of the phase-1 pairs, and per pair of the current
    // batch the index of the saved solution to start from (nullptr: constant initial fields)
    double* warm_x = nullptr;
    size_t warm_cap = 0;
    int* warm_src = nullptr;
    const int* guess_src = nullptr;
    double* partials = nullptr;
    int nblk = 0;
    PairScalars* sc = nullptr;
    int* active = nullptr;
    double* func3 = nullptr;
    double *W = nullptr, *invT = nullptr;
    int nd = 0;
    // host mirrors (pinned)
    int* h_active = nullptr;
    PairScalars* h_sc = nullptr;
    double* h_func3 = nullptr;
    char* h_bounce = nullptr;        // pinned bounce buffer of the debug / test entry points (lazy)
    hipEvent_t ev_batch[2] = {nullptr, nullptr};   // start / end of the batch in flight (vof_pair_stats.batch_ms)
    // staging for the host-pointer API (allocated lazily)
    double* st_movie = nullptr;
    double* st_out[4] = {nullptr, nullptr, nullptr, nullptr};
    double* st_out2[4] = {nullptr, nullptr, nullptr, nullptr};   // second output set (copy / solve overlap of the host API)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_solved[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr}, ev_uploaded[2] = {nullptr, nullptr};
    double* st_movie2 = nullptr;                                 // second frame buffer (upload of the next batch under the solve)
    double *blur_tmp = nullptr, *blur_w = nullptr, *blur_io = nullptr;   // Gaussian blur scratch (lazy)
    double* tex_tab = nullptr;                                           // synthetic-texture tables (lazy)
    size_t tex_cap = 0;
    // GMRES fallback (allocated on first use): basis vectors V_0..V_m (each B * len0), per-pair state, partials, flags
    double* gm_V = nullptr;
    GmresState* gm_state = nullptr;
    double* gm_partials = nullptr;
    int* gm_cycle = nullptr;
    int gm_m = 0;
    long long gmres_pairs = 0;   // pairs handed to the fallback since the context was created
    struct Alloc { void* raw; char* user; size_t bytes; const char* name; int line; };
    std::vector<Alloc> allocs;     // every device buffer of the context (raw != user only with guard regions)
    int c_bytes_per_point = 0;     // stencil storage allocated per point of a stored level (120 float8 .. 648 float64)
    size_t bytes = 0;
    std::string err;
    // state of the last setup
    const double* frames = nullptr;  // device pointer to frame 0 of the current batch
    int npairs = 0;
    vof_params prm;
    int cfmt = 0;   // storage format of the stored stencils of the current hierarchy: 0 double, 1 float, 2 CoefB16 (levels >= 1;
                    // the stored level 0 of a one-level grid is always double)
    bool fused = true;   // fused streaming 4-colour sweeps (false: one launch per colour)
    bool geo_b_fine = false, geo_b_stored = true;   // strip geometry of the fused sweep per level class
    bool fuse_prolong = true;   // level 0: coarse-grid correction interpolated inside the first post-sweep
    bool fuse_restrict = true;  // level 0: residual + restriction in one streaming pass
    bool sweep_st = true;       // stored levels, packed bfloat16 stencils: k_sweep_st (VOF_SWEEP_ST=0: the generic k_sweep)
    bool skip_colour0 = true;   // ... revisits of a W-cycle: the first pre-smoothing sweep leaves colour 0 alone (VOF_SKIP_COLOUR0=0: full sweep)
    bool fold_stored = false;   // ... with the coarse-grid correction interpolated inside the first post-sweep (VOF_FOLD_STORED=1; measured:
                                // the sweep gets slower by what the stand-alone prolongation kernel costs, so that one stays)
    bool fuse_resu = true;      // stored levels: coarse right-hand side from the last pre-smoothing sweep's update (k_resrestrict_u;
                                // VOF_FUSE_RESU=0: stand-alone residual + restriction kernels)
    bool stream_apply = true;   // LDS-streaming level-0 operator kernel with fused reductions (false: simple kernel)
    bool sweep0 = true;         // level 0: dedicated k_sweep0 kernel (VOF_SWEEP0=0: the generic k_sweep<SweepFine, GeoA>)
    // direct preconditioner (block-tridiagonal LU by image rows, vof_direct.hpp); buffers allocated on first use
    bool direct_on = false;          // the current batch is preconditioned by the direct solver instead of the multigrid cycle
    int dir_cap = 0;                 // pairs the direct buffers hold
    double *dir_T = nullptr, *dir_tabs = nullptr, *dir_W = nullptr, *dir_r = nullptr, *dir_y = nullptr, *dir_x = nullptr, *dir_t = nullptr;
    int *dir_ipiv = nullptr, *dir_info = nullptr;
    double *dir_R = nullptr, *dir_C = nullptr, *dir_D = nullptr;   // blocked inverse: row panel, column panel, inverted diagonal tile
    int dir_ld = 0;                  // leading dimension of the dense blocks (m, or m rounded up to the tile size of the blocked inverse)
    void* roc_handle = nullptr;      // rocblas_handle for rocSOLVER
    long long direct_pairs = 0;      // pairs solved with the direct preconditioner since the context was created
    // Krylov product fused into the last smoothing pass of a cycle (k_sweep0m's trailing stage): requested by the Krylov loop
    // before the cycle, consumed by the final level-0 smoothing call if the fused path applies
    bool trail_enabled = true;  // VOF_FUSE_APPLY=0: always the separate operator kernel
    bool trail_set = false, trail_done = false;
    S0Trail trail_req;
    // BiCGStab vector update folded into the cycle's first pre-smoothing pass (k_sweep0r, BF): set by the Krylov loop, consumed by
    // the first level-0 pass from zero of the cycle (sweep_level_t), which resets bf_mode
    // residual + restriction of level 0 as the trailing stage of the pre-smoothing pass (k_sweep0r, TRAIL = 2): requested by
    // vcycle_t around the pre-smoothing of level 0, honoured by sweep_level_t when the pass is the two-sweep pass from zero
    bool fuse_rr = true;        // VOF_FUSE_RR=0: the stand-alone kernel k_stream_resrestrict0
    void* rr_out = nullptr;     // coarse right-hand side to write (nullptr: not requested)
    bool rr_f32 = false, rr_done = false;
    bool fuse_b = true;         // VOF_FUSE_B=0: the stand-alone kernels k_update_s / k_update_p
    int bf_mode = 0;            // 0: none pending; 1: s = r - alpha v (+ (s, s), half-step test); 2: p = r + beta (p_old - omega v)
    S0BSrc bf{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int trail_nblk = 0;         // per-pair partial sums the fused pass wrote
    bool sweep0m = true;        // level 0, float64 vectors, even n_j: k_sweep0m (VOF_SWEEP0M=0: k_sweep0)
    bool sweep0m_pairs = true;  // ... two sweeps per pass (VOF_SWEEP0M=1: one sweep per pass)
    bool sweep0r = true;        // ... the register-resident pass k_sweep0r (VOF_SWEEP0R=0: the LDS-ring pass k_sweep0m)
    long sweep0r_min_blocks = 512;   // ... for launches of at least this many one-wave blocks (VOF_SWEEP0R_MIN_BLOCKS; the tests set 0)
    bool sweep0p = true;        // level 0, float32 cycle vectors (vcycle_precision 1 / 2): the packed-float32 register-resident pass
                                // k_sweep0p, two sweeps per pass (VOF_SWEEP0P=0: the LDS-ring kernel k_sweep0 with float64 arithmetic)
    bool tail_enabled = true;   // fused LDS-resident coarse-tail kernel (VOF_COARSE_TAIL=0: one launch per operation)
    int tail_first = -1;        // first level of the tail (-1: no tail for this grid)
    size_t tail_lds = 0;        // dynamic LDS bytes of k_tail_cycle
    TailArgs tail;              // levels / LDS layout; the operation list is rebuilt when the cycle parameters change
    int tail_key[6] = {-9, -9, -9, -9, -9, -9};   // cycle parameters the operation list was built for
    // profiler
    bool prof = false;
    int prof_kid = -1, prof_level = -1;  // filter (-1 = any)
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> free_events;
    long long prof_dropped = 0;
    double prof_ms[VOF_K_COUNT][16];
    long long prof_n[VOF_K_COUNT][16];
    long long prof_units[VOF_K_COUNT][16];
    double prof_bytes[VOF_K_COUNT][16];
    double prof_moved[VOF_K_COUNT][16];
    int cur_units = 0;  // frame pairs the next launches process (active pairs of the batch)
    // experiment (VOF_PRECOND_QUIRKS=hs, two digits): does the PRECONDITIONER see the reference's 'dy' == 'dx' quirk (OF.py:698-699)
    // in its Galerkin hierarchy (h) / in its level-0 smoother and residual (s)?  The Krylov product always does.
    bool pq_hier = true, pq_smooth = true;
    // fault attribution (see the "debug switches" paragraph of include/vof.h)
    int dbg_sync = 0;            // VOF_DEBUG_SYNC=1: synchronise + check after every launch scope; the first failure names its kernel class
    long long dbg_seq = 0;       // launch scopes checked so far
    std::string dbg_fault;       // the first failure
    int dbg_fd = -1;             // VOF_DEBUG_SYNC_FILE: the scope in flight is written here before it is waited for (survives an abort)
    bool dbg_canary = false;     // VOF_DEBUG_CANARY=1: every device buffer sits between two guard pages of a known pattern
    bool dbg_alloc_log = false;  // VOF_DEBUG_ALLOC_LOG=1: base / size / name of every device buffer on stderr
    bool dbg_poison = false;     // VOF_DEBUG_POISON=1: every new device buffer is filled with 0xFF bytes (NaN as float / double, -1 as int)
};

static std::string g_create_error;

namespace {

// VOF_DEBUG_SYNC: wait for the launches of the scope that has just ended and ask for their error, so that a fault is
// reported against the kernel class and level that caused it instead of at some later synchronising call.
void dbg_sync_check(vof_ctx* c, const char* what, int level) {
    if (!c->dbg_sync) return;
    ++c->dbg_seq;
    char line[256];
    if (c->dbg_fd >= 0) {
        int n = snprintf(line, sizeof line, "in flight: scope #%lld '%s' level %d, %d pairs, grid %dx%d (%d levels)%-40s\n", c->dbg_seq, what, level,
                         c->cur_units, c->Ni, c->Nj, (int)c->L.size(), "");
        if (pwrite(c->dbg_fd, line, (size_t)n, 0) < 0) { /* diagnostics only */ }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipError_t e2 = hipGetLastError();
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess && c->dbg_fault.empty()) {
        snprintf(line, sizeof line, "VOF_DEBUG_SYNC: scope #%lld, kernel class '%s', level %d, %d pairs, image %dx%d: %s", c->dbg_seq, what, level,
                 c->cur_units, c->Ni, c->Nj, hipGetErrorString(e));
        c->dbg_fault = line;
        fprintf(stderr, "%s\n", line);
        fflush(stderr);
    }
}

struct Prof {
    vof_ctx* c;
    bool on;
    int dkid, dlevel;
    ProfRec rec;
    Prof(vof_ctx* c_, int kid, int level, double bytes_per_pair = 0.0, double moved_per_pair = -1.0) : c(c_), on(false), dkid(kid), dlevel(level) {
        if (!c->prof) return;
        if (c->prof_kid >= 0 && kid != c->prof_kid) return;
        if (c->prof_level >= 0 && level != c->prof_level) return;
        if ((int)c->recs.size() >= MAX_PROF_RECS) { c->prof_dropped++; return; }
        if (c->free_events.size() >= 2) {
            rec.e0 = c->free_events.back(); c->free_events.pop_back();
            rec.e1 = c->free_events.back(); c->free_events.pop_back();
        } else {
            if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) return;
        }
        rec.kid = kid; rec.level = level < 0 ? 0 : (level > 15 ? 15 : level);
        rec.units = c->cur_units;
        rec.bytes = bytes_per_pair * c->cur_units;
        rec.moved = (moved_per_pair < 0.0 ? bytes_per_pair : moved_per_pair) * c->cur_units;
        on = true;
        hipEventRecord(rec.e0, c->stream);
    }
    ~Prof() {
        if (on) {
            hipEventRecord(rec.e1, c->stream);
            c->recs.push_back(rec);
        }
        if (c->dbg_sync) dbg_sync_check(c, vof_kernel_name(dkid), dlevel);
    }
};

void prof_collect(vof_ctx* c) {
    if (c->recs.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto& r : c->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            c->prof_ms[r.kid][r.level] += ms;
            c->prof_n[r.kid][r.level] += 1;
            c->prof_units[r.kid][r.level] += r.units;
            c->prof_bytes[r.kid][r.level] += r.bytes;
            c->prof_moved[r.kid][r.level] += r.moved;
        }
        c->free_events.push_back(r.e0);
        c->free_events.push_back(r.e1);
    }
    c->recs.clear();
}

#define HIPCHK(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char buf_[512];                                                                        \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            c->err = buf_;                                                                         \
            if (!c->dbg_fault.empty()) c->err += " [" + c->dbg_fault + "]";                        \
            return -2;                                                                             \
        }                                                                                          \
    } while (0)

constexpr size_t DBG_GUARD = 4096;        // bytes of guard pattern on either side of a buffer (VOF_DEBUG_CANARY=1)
constexpr int DBG_GUARD_BYTE = 0xC5;

template <typename T>
int dev_alloc_named(vof_ctx* c, T** p, size_t n, const char* name, int line) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(n * sizeof(T), 256);
    if (c->dbg_canary) {
        HIPCHK(hipMalloc(&q, bytes + 2 * DBG_GUARD));
        HIPCHK(hipMemset(q, DBG_GUARD_BYTE, DBG_GUARD));
        HIPCHK(hipMemset((char*)q + DBG_GUARD + bytes, DBG_GUARD_BYTE, DBG_GUARD));
        c->allocs.push_back({q, (char*)q + DBG_GUARD, bytes, name, line});
        *p = (T*)((char*)q + DBG_GUARD);
    } else {
        HIPCHK(hipMalloc(&q, bytes));
        c->allocs.push_back({q, (char*)q, bytes, name, line});
        *p = (T*)q;
    }
    c->bytes += bytes;
    if (c->dbg_poison) HIPCHK(hipMemset(*p, 0xFF, bytes));
    if (c->dbg_alloc_log)
        fprintf(stderr, "vof alloc ctx=%p %s (vof.hip:%d) base=%p end=%p bytes=%zu\n", (void*)c, name, line, (void*)*p, (void*)((char*)*p + bytes), bytes);
    return 0;
}
#define dev_alloc(c, p, n) dev_alloc_named(c, p, n, #p, __LINE__)

// Releases one buffer of the context (a buffer that is re-allocated larger); nullptr is fine.
int dev_free(vof_ctx* c, void* user) {
    if (!user) return 0;
    for (size_t i = 0; i < c->allocs.size(); ++i)
        if ((void*)c->allocs[i].user == user) {
            HIPCHK(hipFree(c->allocs[i].raw));
            c->bytes -= c->allocs[i].bytes;
            c->allocs.erase(c->allocs.begin() + (long)i);
            return 0;
        }
    c->err = "internal: dev_free of a pointer the context does not own";
    return -1;
}

constexpr int DEFAULT_COARSE_PRECISION = 3;
inline int vof_params_default_coarse_precision() { return DEFAULT_COARSE_PRECISION; }
// bytes of stencil storage per point of a stored level in format `fmt` (vof_params.coarse_precision)
inline int coef_bytes_per_point(int fmt) { return fmt == 3 ? 30 * 4 : (fmt == 2 ? 45 * 4 : (fmt == 1 ? 81 * 4 : 81 * 8)); }

// Stencil storage of the levels >= 1 for format `fmt`: allocated for the default format by vof_create, re-allocated
// (never shrunk) when a call asks for a wider one.  Round 2 sized it for float64 whatever the format: 648 instead of 120
// bytes per coarse point, 45 % of the workspace never touched.
int ensure_stencil_storage(vof_ctx* c, int fmt) {
    const int need = coef_bytes_per_point(fmt);
    if (need <= c->c_bytes_per_point || c->L.size() < 2) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t l = 1; l < c->L.size(); ++l) {
        Level& lv = c->L[l];
        if (int rc = dev_free(c, lv.C)) return rc;
        lv.C = nullptr;
    }
    for (size_t l = 1; l < c->L.size(); ++l) {
        Level& lv = c->L[l];
        uint32_t* C = nullptr;
        if (int rc = dev_alloc(c, &C, (size_t)c->B * (need / 4) * CLay(lv.ni, lv.nj).plane)) return rc;
        lv.C = C;
    }
    c->c_bytes_per_point = need;
    c->frames = nullptr;   // a hierarchy built earlier is gone: the debug entry points ask for a new vof_debug_setup
    return 0;
}

// Buffers that depend on the parameters of the call (every entry point runs this through check_params).
int ensure_storage(vof_ctx* c) {
    if (int rc = ensure_stencil_storage(c, c->prm.coarse_precision)) return rc;
    if (c->vfloat && !c->b32)   // float32 copy of the cycle's right-hand side (vcycle_precision 1 / 2 only)
        if (int rc = dev_alloc(c, &c->b32, (size_t)c->B * 3 * c->L[0].npts)) return rc;
    return 0;
}

// Guard pages of every buffer of the context against their pattern; the number of damaged buffers, each named in `report`.
int dbg_check_canaries(vof_ctx* c, std::string* report) {
    int bad = 0;
    std::vector<unsigned char> h(2 * DBG_GUARD);
    if (!c->dbg_canary) return 0;
    for (const auto& a : c->allocs) {
        if (hipMemcpy(h.data(), a.raw, DBG_GUARD, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(h.data() + DBG_GUARD, a.user + a.bytes, DBG_GUARD, hipMemcpyDeviceToHost) != hipSuccess) {
            if (report) *report += "guard pages unreadable; ";
            return -1;
        }
        long first_lo = -1, first_hi = -1;
        for (size_t i = 0; i < DBG_GUARD; ++i) if (h[DBG_GUARD - 1 - i] != DBG_GUARD_BYTE) { first_lo = (long)i + 1; break; }   // bytes BELOW the buffer
        for (size_t i = 0; i < DBG_GUARD; ++i) if (h[DBG_GUARD + i] != DBG_GUARD_BYTE) { first_hi = (long)i; break; }            // bytes PAST its end
        if (first_lo >= 0 || first_hi >= 0) {
            ++bad;
            char line[256];
            snprintf(line, sizeof line, "buffer %s (vof.hip:%d, %zu bytes at %p): written %s%ld bytes %s; ", a.name, a.line, a.bytes, (void*)a.user,
                     first_hi >= 0 ? "" : "-", first_hi >= 0 ? first_hi : first_lo, first_hi >= 0 ? "past its end" : "before its start");
            if (report) *report += line;
        }
    }
    return bad;
}

inline dim3 grid2d(int ni, int nj, int z) { return dim3((nj + BX - 1) / BX, (ni + BY - 1) / BY, z); }
inline dim3 grid2d_colour(int ni, int nj, int colour, int z) {
    int cp = colour >> 1, cq = colour & 1;
    int mi = (ni - cp + 1) / 2, mj = (nj - cq + 1) / 2;
    return dim3(std::max(1, (mj + BX - 1) / BX), std::max(1, (mi + BY - 1) / BY), z);
}
const dim3 blk2d(BX, BY, 1);

inline size_t frame_stride(const vof_ctx* c) { return (size_t)c->Ni * c->Nj; }

// ---------------------------------------------------------------- level kernels
// VT = storage type of the V-cycle vectors (float when ctx->vfloat, else double).
#define VDISPATCH(c, ...)                                   \
    do {                                                    \
        if ((c)->vfloat) { using VT = float; __VA_ARGS__; } \
        else { using VT = double; __VA_ARGS__; }            \
    } while (0)

template <typename T> struct TypeTag { typedef T type; };
// CT = storage format of the stored stencil of level l (word type CW)
#define CDISPATCH(c, l, ...)                                                                        \
    do {                                                                                            \
        const int cf_ = ((l) > 0) ? (c)->cfmt : 0;                                                  \
        if (cf_ == 2) { using CT = CoefB16; using CW = uint32_t; __VA_ARGS__; }                     \
        else if (cf_ == 3) { using CT = CoefF8; using CW = uint32_t; __VA_ARGS__; }                 \
        else if (cf_ == 1) { using CT = float; using CW = float; __VA_ARGS__; }                     \
        else { using CT = double; using CW = double; __VA_ARGS__; }                                 \
    } while (0)
inline double coef_bytes(const vof_ctx* c, int l) { const int f = l > 0 ? c->cfmt : 0; return f == 3 ? 30.0 * 4 : (f == 2 ? 45.0 * 4 : (f == 1 ? 81.0 * 4 : 81.0 * 8)); }

// one colour, in place, one launch per colour: the simple reference smoother (double vectors only)
void gs_colour(vof_ctx* c, int l, double* x, const double* b, int colour, int np, const int* active) {
    Level& lv = c->L[l];
    dim3 g = grid2d_colour(lv.ni, lv.nj, colour, np);
    if (l == 0 && c->L.size() > 1) {
        Prof p(c, VOF_K_GS0, 0, 20.0 * lv.npts);
        k_gs0<<<g, blk2d, 0, c->stream>>>(c->frames, frame_stride(c), c->Nj, lv.ni, lv.nj, c->prm.speed_alpha,
                                          c->prm.remodelling_alpha, c->prm.reference_quirks, x, b, colour, active, c->pp);
    } else {
        Prof p(c, VOF_K_GS, l, (coef_bytes(c, l) + 72.0) / 4.0 * lv.npts);
        CDISPATCH(c, l, (k_gs<CT><<<g, blk2d, 0, c->stream>>>((const CW*)lv.C, lv.ni, lv.nj, x, b, colour, active)));
    }
}

// y = A_l x (mode 0) or y = b - A_l x (mode 1); XT/BT/YT storage types (level 0 matrix-free),
// stored levels use one type for all three.
// Band height of the row-streaming kernels: bands of <= 128 rows (balanced, even).  A block marches through its band
// sequentially, so a launch lasts at least (band height / 2 + pipeline depth) steps; when only a few pairs are still
// active (the tail iterations of a batch, or a small stack) the grid does not fill the 256 CUs and shorter bands
// (more, shorter blocks) cut that latency floor.
int pick_band_height(int rows, int nx, int units) {
    const long blocks128 = (long)nx * ((rows + 127) / 128) * std::max(1, units);
    const int cap = blocks128 < 768 ? 32 : (blocks128 < 1536 ? 64 : 128);   // 768 = 3 blocks per CU
    const int nb = (rows + cap - 1) / cap;
    return std::max(2, (((rows + nb - 1) / nb + 1) / 2) * 2);
}

// Geometry of the streaming level-0 operator kernel: 128-column strips, bands of <= 128 rows (even height).
struct ApplyGrid { int TI, nblk; dim3 grid; };
ApplyGrid apply_grid(const vof_ctx* c, int np) {
    const Level& lv = c->L[0];
    int TI = pick_band_height(lv.ni, (lv.nj + AP_OUT - 1) / AP_OUT, c->cur_units);
    ApplyGrid g;
    g.TI = TI;
    g.grid = dim3((lv.nj + AP_OUT - 1) / AP_OUT, (lv.ni + TI - 1) / TI, np);
    g.nblk = g.grid.x * g.grid.y;
    return g;
}

// y = A x (mode 0) or y = b - A x (mode 1) on the matrix-free level 0.  Optional fused reductions into
// c->partials (slot 0: y.dotvec, or y.y when dotvec == nullptr; slot 1: y.y when both are asked for); the
// number of per-pair partials is apply_grid().nblk.
template <typename XT, typename BT, typename YT>
void apply_fine_t(vof_ctx* c, const XT* x, const BT* b, YT* y, int mode, int np, const int* active,
                  const double* dotvec = nullptr, int want_yy = 0, YT* ycopy = nullptr) {
    Level& lv = c->L[0];
    const double bytes = (8.0 + 3.0 * sizeof(XT) + ((y ? 3.0 : 0.0) + (ycopy ? 3.0 : 0.0)) * sizeof(YT) + (mode ? 3.0 * sizeof(BT) : 0.0) +
                          (dotvec ? 24.0 : 0.0)) * lv.npts;
    Prof p(c, VOF_K_APPLY0, 0, bytes);
    if (c->stream_apply) {
        ApplyGrid ag = apply_grid(c, np);
        double* part = (dotvec || want_yy) ? c->partials : nullptr;
        if (mode)
            k_stream_apply0<1, XT, BT, YT><<<ag.grid, AP_THREADS, 0, c->stream>>>(
                c->frames, frame_stride(c), c->Nj, lv.ni, lv.nj, ag.TI, c->prm.speed_alpha, c->prm.remodelling_alpha,
                c->prm.reference_quirks, x, b, y, dotvec, want_yy, part, ag.nblk, active, c->pp, ycopy);
        else
            k_stream_apply0<0, XT, BT, YT><<<ag.grid, AP_THREADS, 0, c->stream>>>(
                c->frames, frame_stride(c), c->Nj, lv.ni, lv.nj, ag.TI, c->prm.speed_alpha, c->prm.remodelling_alpha,
                c->prm.reference_quirks, x, b, y, dotvec, want_yy, part, ag.nblk, active, c->pp, ycopy);
        return;
    }
    if (ycopy) { c->err = "internal: second output without the streaming operator kernel"; return; }
    dim3 g = grid2d(lv.ni, lv.nj, np);
    if (mode)
        k_apply0<1, XT, BT, YT><<<g, blk2d, 0, c->stream>>>(c->frames, frame_stride(c), c->Nj, lv.ni, lv.nj,
                                                            c->prm.speed_alpha, c->prm.remodelling_alpha,
                                                            c->prm.reference_quirks, x, b, y, active, c->pp);
    else
        k_apply0<0, XT, BT, YT><<<g, blk2d, 0, c->stream>>>(c->frames, frame_stride(c), c->Nj, lv.ni, lv.nj,
                                                            c->prm.speed_alpha, c->prm.remodelling_alpha,
                                                            c->prm.reference_quirks, x, b, y, active, c->pp);
}

template <typename VT>
void apply_stored_t(vof_ctx* c, int l, const VT* x, const VT* b, VT* y, int mode, int np, const int* active) {
    Level& lv = c->L[l];
    dim3 g = grid2d(lv.ni, lv.nj, np);
    Prof p(c, VOF_K_RESIDUAL, l, (coef_bytes(c, l) + (mode ? 9.0 : 6.0) * sizeof(VT)) * lv.npts);
    if (mode) CDISPATCH(c, l, (k_apply<CT, 1, VT><<<g, blk2d, 0, c->stream>>>((const CW*)lv.C, lv.ni, lv.nj, x, b, y, active)));
    else CDISPATCH(c, l, (k_apply<CT, 0, VT><<<g, blk2d, 0, c->stream>>>((const CW*)lv.C, lv.ni, lv.nj, x, b, y, active)));
}

// V-cycle internal operator application on level l (all vectors VT)
template <typename VT>
void apply_level_t(vof_ctx* c, int l, const VT* x, const VT* b, VT* y, int mode, int np, const int* active) {
    if (l == 0 && c->L[0].C == nullptr) apply_fine_t<VT, VT, VT>(c, x, b, y, mode, np, active);
    else apply_stored_t<VT>(c, l, x, b, y, mode, np, active);
}

// Krylov-level products on level 0 with FP64 results: out = A y (y V-typed) and out = b - A x (all double).
// When `fuse` is set and the streaming kernel is in use, the reductions (out.dotvec and/or out.out) are fused
// into the operator kernel and the function returns the number of per-pair partials it wrote; otherwise 0
// (the caller then launches k_dot2).
int krylov_apply(vof_ctx* c, const void* y, double* out, int np, const int* active, const double* dotvec = nullptr,
                 int want_yy = 0) {
    if (c->L[0].C) { apply_stored_t<double>(c, 0, (const double*)y, nullptr, out, 0, np, active); return 0; }
    const bool fuse = c->stream_apply && (dotvec || want_yy);
    if (c->vfloat) apply_fine_t<float, double, double>(c, (const float*)y, nullptr, out, 0, np, active, fuse ? dotvec : nullptr, fuse ? want_yy : 0);
    else apply_fine_t<double, double, double>(c, (const double*)y, nullptr, out, 0, np, active, fuse ? dotvec : nullptr, fuse ? want_yy : 0);
    return fuse ? apply_grid(c, np).nblk : 0;
}
// (out == nullptr with want_norm: only the norm is wanted - the streaming kernel then writes nothing; returns 0 if that is
// not possible, and the caller falls back to a residual vector)
// out2 (streaming kernel only, see residual_copy_ok): a second copy of the residual
inline bool residual_copy_ok(const vof_ctx* c) { return c->stream_apply && !c->L[0].C; }
int residual_d(vof_ctx* c, const double* x, const double* b, double* out, int np, const int* active, int want_norm = 0,
               double* out2 = nullptr) {
    if (!out && !(want_norm && c->stream_apply && !c->L[0].C)) return 0;
    if (c->L[0].C) { apply_stored_t<double>(c, 0, x, b, out, 1, np, active); return 0; }
    const bool fuse = c->stream_apply && want_norm;
    apply_fine_t<double, double, double>(c, x, b, out, 1, np, active, nullptr, fuse ? 1 : 0, out2);
    return fuse ? apply_grid(c, np).nblk : 0;
}

template <typename VT>
void restrict_level_t(vof_ctx* c, int l, const VT* fine, VT* coarse, int np, const int* active) {
    Level &f = c->L[l], &k = c->L[l + 1];
    Prof p(c, VOF_K_RESTRICT, l, 3.0 * sizeof(VT) * (f.npts + k.npts));
    k_restrict<VT><<<grid2d(k.ni, k.nj, np), blk2d, 0, c->stream>>>(fine, f.ni, f.nj, coarse, k.ni, k.nj, active);
}

// level 0, matrix-free: coarse right-hand side b_1 = R (b - A x) in one pass (no fine residual in HBM)
template <typename VT, typename CVT = VT>
void resrestrict_fine_t(vof_ctx* c, const VT* x, const VT* b, CVT* bc, int np, const int* active) {
    Level &f = c->L[0], &k = c->L[1];
    int TI = pick_band_height(f.ni, (k.nj + RR_CO - 1) / RR_CO, c->cur_units);
    dim3 g((k.nj + RR_CO - 1) / RR_CO, (k.ni + TI / 2 - 1) / (TI / 2), np);
    Prof p(c, VOF_K_APPLY0, 0, (8.0 + 6.0 * sizeof(VT)) * f.npts + 3.0 * sizeof(CVT) * k.npts);
    k_stream_resrestrict0<VT, VT, CVT><<<g, AP_THREADS, 0, c->stream>>>(
        c->frames, frame_stride(c), c->Nj, f.ni, f.nj, TI, c->prm.speed_alpha, c->prm.remodelling_alpha,
        c->prm.reference_quirks && c->pq_smooth, x, b, bc, k.ni, k.nj, active, c->pp);
}

// stored level l >= 1, straight after ONE forward Gauss-Seidel sweep x_old -> x_new (x_old == nullptr: from zero): the coarse
// right-hand side b_{l+1} = R (b - A x_new) from the sweep's update alone (k_resrestrict_u) - no b, no diagonal blocks, half
// of the off-diagonal coefficients, no residual vector in HBM
inline double resu_coef_bytes(const vof_ctx* c) {   // average per fine point: 8 + 6 + 2 + 0 neighbour blocks over the four colours
    return c->cfmt == 3 ? 12.0 * 4 : (c->cfmt == 2 ? 18.5 * 4 : (c->cfmt == 1 ? 36.0 * 4 : 36.0 * 8));
}
template <typename VT>
void resrestrict_u_t(vof_ctx* c, int l, const VT* x_new, const VT* x_old, VT* bc, int np, const int* active) {
    Level &f = c->L[l], &k = c->L[l + 1];
    const double vs = sizeof(VT);
    // algorithmic: the residual and the restriction it performs (as apply_stored_t + restrict_level_t count them)
    const double algo = (coef_bytes(c, l) + 9.0 * vs) * f.npts + 3.0 * vs * (f.npts + k.npts);
    const double moved = (resu_coef_bytes(c) + (x_old ? 6.0 : 3.0) * vs) * f.npts + 3.0 * vs * k.npts;
    Prof p(c, VOF_K_RESIDUAL, l, algo, moved);
    dim3 g(1, (k.ni + BY - 1) / BY, np);
    CDISPATCH(c, l, {
        if (x_old) k_resrestrict_u<CT, VT, true><<<g, blk2d, 0, c->stream>>>((const CW*)f.C, f.ni, f.nj, x_new, x_old, bc, k.ni, k.nj, active);
        else k_resrestrict_u<CT, VT, false><<<g, blk2d, 0, c->stream>>>((const CW*)f.C, f.ni, f.nj, x_new, x_old, bc, k.ni, k.nj, active);
    });
}

template <typename VT>
void prolong_add_level_t(vof_ctx* c, int l, VT* fine, const VT* coarse, int np, const int* active) {
    Level &f = c->L[l], &k = c->L[l + 1];
    Prof p(c, VOF_K_PROLONG, l, 3.0 * sizeof(VT) * (2 * f.npts + k.npts));
    k_prolong_add<VT><<<grid2d(f.ni, f.nj, np), blk2d, 0, c->stream>>>(fine, f.ni, f.nj, coarse, k.ni, k.nj, active);
}

template <typename VT>
void coarse_solve_t(vof_ctx* c, const VT* r, VT* e, int np, const int* active) {
    Prof p(c, VOF_K_COARSE_SOLVE, (int)c->L.size() - 1);
    k_coarse_solve<VT><<<np, 256, c->nd * sizeof(double), c->stream>>>(c->invT, c->nd, r, e, active);
}

// k_sweep0m (merged colours, 16-byte accesses, up to two sweeps per pass) needs float64 vectors and an even row length
inline bool sweep0m_usable(const vof_ctx* c) {
    return c->sweep0m && c->sweep0 && c->fused && !c->geo_b_fine && !c->vfloat && (c->L[0].nj % 2 == 0) && c->L[0].C == nullptr;
}

// k_sweep0p (float32 cycle vectors: packed float32 arithmetic, two strips per wave, up to two sweeps per pass)
inline bool sweep0p_usable(const vof_ctx* c) {
    return c->sweep0p && c->sweep0r && c->sweep0 && c->fused && !c->geo_b_fine && c->vfloat && (c->L[0].nj % 2 == 0) && c->L[0].C == nullptr &&
           c->prm.reference_quirks && c->pq_smooth;
}

// k_sweep_st: stored levels with packed bfloat16 stencils and the 128-column strip geometry
inline bool sweep_st_usable(const vof_ctx* c, int l) {
    return l > 0 && c->L[l].C != nullptr && c->sweep_st && c->geo_b_stored && c->cfmt >= 2;
}

// Strips and bands of a level-0 pass of k_sweep0m / k_sweep0r (NSW sweeps per pass) and whether the register-resident kernel takes it
struct S0Geo { int nx, ny, TI; bool s0r; };
inline S0Geo s0_geometry(const vof_ctx* c, int rows, int NSW, bool trail) {
    const Level& lv = c->L[0];
    S0Geo g;
    const int out = S0_W - 8 * NSW - (trail ? 4 : 0);
    g.nx = (lv.nj + out - 1) / out;
    g.TI = pick_band_height(rows, g.nx, c->cur_units);
    g.ny = (rows + g.TI - 1) / g.TI;
    g.s0r = c->sweep0r && c->prm.reference_quirks && c->pq_smooth && (long)g.nx * g.ny * std::max(1, c->cur_units) >= c->sweep0r_min_blocks;
    return g;
}

// Will the next cycle start with a two-sweep level-0 pass from zero of k_sweep0r?  Then the vector update that forms the cycle's
// right-hand side is folded into that pass (the conditions mirror vcycle_t -> smooth_level_t -> sweep_level_t).
inline bool fold_b_usable(const vof_ctx* c) {
    return c->fuse_b && !c->direct_on && !c->vfloat && c->L.size() > 1 && c->tail_first != 0 && c->fused && c->prm.nu_pre >= 2 &&
           sweep0m_usable(c) && c->sweep0m_pairs && s0_geometry(c, c->L[0].ni, 2, false).s0r && s0_geometry(c, c->L[0].ni, 2, true).s0r;
}

// One full 4-colour sweep x_in -> x_out (x_in == nullptr: zero initial guess); reverse = colours 3,2,1,0.
// nsweeps = 2 (level 0, k_sweep0m only): two consecutive sweeps in one pass.
template <typename VT>
void sweep_level_t(vof_ctx* c, int l, const VT* x_in, VT* x_out, const VT* b, bool reverse, int np,
                   const int* active, const VT* ecoarse = nullptr, int nsweeps = 1, bool with_trail = false, bool ec32 = false,
                   bool out64 = false, bool skip0 = false) {
    // out64: x_out is written as float64 although VT is float (k_sweep_st only; the caller has checked that it applies)
    // skip0: x_in comes straight from a reverse sweep with the same b - colour 0 needs no update (k_sweep_st; elsewhere ignored)
    // ec32: `ecoarse` really points at float32 data (float64 level 0 above float32 coarse levels; k_sweep0m only)
    Level& lv = c->L[l];
    int po = reverse ? 1 : 0;
    int rows = lv.ni + po;
    if constexpr (std::is_same<VT, double>::value) {
        if (l == 0 && lv.C == nullptr && sweep0m_usable(c)) {
            // k_sweep0m: merged colours, 16-byte accesses, `nsweeps` (1 or 2) sweeps per pass; strips are not shifted by po
            const int NSW = nsweeps >= 2 ? 2 : 1;
            const bool trail = with_trail && x_in != nullptr;
            // the coarse right-hand side R (b - A x_out) as the trailing stage of the two-sweep pass from zero (k_sweep0r only)
            const bool rr = c->rr_out && !x_in && NSW == 2 && !po && !ecoarse && VOF_S0R_BCARRY && s0_geometry(c, rows, NSW, true).s0r;
            const S0Geo geo = s0_geometry(c, rows, NSW, trail || rr);
            const int nx = geo.nx, TI = geo.TI, ny = geo.ny;
            dim3 g((unsigned)nx * ny * np, 1, 1);
            int nci = 0, ncj = 0;
            double ebytes = 0.0;
            if (ecoarse) { nci = c->L[1].ni; ncj = c->L[1].nj; ebytes = (ec32 ? 12.0 : 24.0) * c->L[1].npts; }
            if (c->bf_mode) {   // the pass forms its right-hand side itself (the Krylov loop has checked fold_b_usable)
                if (x_in || NSW != 2 || po || !geo.s0r || ecoarse || trail) { c->err = "folded vector update: the cycle did not start with the expected pass"; c->bf_mode = -1; return; }
                const int mode = c->bf_mode;
                c->bf_mode = 0;
                Fine0 f0{c->frames, frame_stride(c), c->Nj, c->prm.speed_alpha, c->prm.remodelling_alpha, 1, c->pp};
                S0Trail tr{rr ? (double*)c->rr_out : nullptr, nullptr, 0, nullptr};
                const size_t ldsr = rr ? S0R<2, 2>::LDS_TOTAL : S0R<2, 0>::LDS_TOTAL;
                const int kci = c->L[1].ni, kcj = c->L[1].nj;
                {   // I + r(3) + v(3) (+ p_old(3)) in, x(3) + b(3) out (+ the coarse right-hand side)
                    const double cb = rr ? (c->rr_f32 ? 12.0 : 24.0) * c->L[1].npts : 0.0;
                    const double moved = (8.0 + (mode == 2 ? 15.0 : 12.0) * 8.0) * lv.npts + cb;
                    Prof p(c, VOF_K_GS0, 0, moved + 80.0 * lv.npts + (rr ? 56.0 * lv.npts : 0.0), moved);
#define VOF_LAUNCH_BF(TR_, ET_, BF_) k_sweep0r<2, false, true, TR_, ET_, 0, 1, BF_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, 0, nx, ny, np, nullptr, x_out, b, active, nullptr, kci, kcj, tr, 0, 0, c->bf)
                    if (rr && c->rr_f32) { if (mode == 1) VOF_LAUNCH_BF(2, float, 1); else VOF_LAUNCH_BF(2, float, 2); }
                    else if (rr) { if (mode == 1) VOF_LAUNCH_BF(2, double, 1); else VOF_LAUNCH_BF(2, double, 2); }
                    else { if (mode == 1) VOF_LAUNCH_BF(0, double, 1); else VOF_LAUNCH_BF(0, double, 2); }
#undef VOF_LAUNCH_BF
                    if (rr) c->rr_done = true;
                }
                if (mode == 1) {   // (s, s): stopping rule at the half step; pairs done there get their x += alpha y and leave the cycle
                    const size_t len = 3 * lv.npts;
                    { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_S><<<np, 64, 0, c->stream>>>(c->sc, c->partials, nx * ny, c->active, c->prm.rtol, c->prm.max_iterations); }
                    { Prof p(c, VOF_K_VECTOR, 0);
                      k_fix_half<double><<<dim3(64, np), RBLK, 0, c->stream>>>(c->kx, (const double*)c->ky, len, c->sc);
                      k_clear_half<<<(np + 255) / 256, 256, 0, c->stream>>>(c->sc, np); }
                }
                return;
            }
            // bytes the pass moves: I + b(3) + x(3) in, x(3) out (+ coarse e), whatever the number of fused sweeps; algorithmic
            // bytes (80 per sweep performed): the second sweep of a double pass counts as a full sweep
            double moved = (8.0 + (x_in ? 9.0 : 6.0) * 8.0) * lv.npts + ebytes;
            double algo = moved + (NSW - 1) * 80.0 * lv.npts;
            S0Trail tr{nullptr, nullptr, 0, nullptr};
            if (trail) {   // + the operator product v = A x_out with its dot products: algorithmic 56 (+24 for the dot partner)
                tr = c->trail_req;
                const double dv = tr.dotvec ? 24.0 : 0.0;
                algo += (56.0 + dv) * lv.npts;
                moved += (24.0 + dv) * lv.npts;     // only v out and the dot partner in: x_out and the image are in LDS
                c->trail_nblk = nx * ny;
                c->trail_done = true;
            }
            if (rr) {   // + the coarse right-hand side out (x_out, b and the image are in registers / LDS); algorithmic: the 56 B per
                        // pixel the stand-alone residual + restriction kernel reads
                const double cb = (c->rr_f32 ? 12.0 : 24.0) * c->L[1].npts;
                algo += 56.0 * lv.npts + cb;
                moved += cb;
            }
            Prof p(c, VOF_K_GS0, 0, algo, moved);
            Fine0 f0{c->frames, frame_stride(c), c->Nj, c->prm.speed_alpha, c->prm.remodelling_alpha, c->prm.reference_quirks && c->pq_smooth, c->pp};
            const size_t lds = (size_t)(6 * NSW + 2 + (trail ? 4 : 0)) * s0_row_bytes(8) + (ecoarse ? (size_t)9 * (S0_W / 2 + 2) * 8 : 0);
#define VOF_LAUNCH_S0M(NS_)                                                                                                        \
            do {                                                                                                                    \
                if (trail && ecoarse && ec32) k_sweep0m<NS_, true, false, 1, float><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, (const float*)ecoarse, nci, ncj, tr); \
                else if (ecoarse && ec32) k_sweep0m<NS_, true, false, 0, float><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, (const float*)ecoarse, nci, ncj, tr); \
                else if (trail && ecoarse) k_sweep0m<NS_, true, false, 1><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else if (trail) k_sweep0m<NS_, false, false, 1><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else if (ecoarse) k_sweep0m<NS_, true, false, 0><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else if (!x_in) k_sweep0m<NS_, false, true, 0><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else k_sweep0m<NS_, false, false, 0><<<g, 128 * NS_, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
            } while (0)
#define S0R_BF(NS_) ((VOF_S0R_BCARRY && (NS_) == 2) ? 3 : 0)   /* the two-sweep pass from zero reads b once (vof_sweep0r.hpp) */
#define VOF_LAUNCH_S0R(NS_, PO_)                                                                                                   \
            do {                                                                                                                    \
                const size_t ldsr = trail ? S0R<NS_, 1>::LDS_TOTAL : S0R<NS_, 0>::LDS_TOTAL;                                         \
                if (trail && ecoarse && ec32) k_sweep0r<NS_, true, false, 1, float, PO_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, (const float*)ecoarse, nci, ncj, tr); \
                else if (ecoarse && ec32) k_sweep0r<NS_, true, false, 0, float, PO_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, (const float*)ecoarse, nci, ncj, tr); \
                else if (trail && ecoarse) k_sweep0r<NS_, true, false, 1, double, PO_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else if (trail) k_sweep0r<NS_, false, false, 1, double, PO_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else if (ecoarse) k_sweep0r<NS_, true, false, 0, double, PO_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else if (!x_in) k_sweep0r<NS_, false, true, 0, double, PO_, 1, S0R_BF(NS_)><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
                else k_sweep0r<NS_, false, false, 0, double, PO_><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, tr); \
            } while (0)
            // (the register-resident pass is compiled with the reference's derivative quirk built in; one wave per block needs
            // a few waves per SIMD-slot to fill the chip: tiny stacks - 128 x 128 x 8: 14 blocks - stay with the 4-wave LDS pass)
            if (rr) {
                S0Trail trr{(double*)c->rr_out, nullptr, 0, nullptr};
                const size_t ldsr = S0R<2, 2>::LDS_TOTAL;
                if (c->rr_f32) k_sweep0r<2, false, true, 2, float, 0, 1, 3><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, nullptr, c->L[1].ni, c->L[1].nj, trr);
                else k_sweep0r<2, false, true, 2, double, 0, 1, 3><<<g, 64, ldsr, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, nullptr, c->L[1].ni, c->L[1].nj, trr);
                c->rr_done = true;
                return;
            }
            if (geo.s0r) {
                if (NSW == 2) { if (po) VOF_LAUNCH_S0R(2, 1); else VOF_LAUNCH_S0R(2, 0); }
                else { if (po) VOF_LAUNCH_S0R(1, 1); else VOF_LAUNCH_S0R(1, 0); }
            }
            else if (NSW == 2) VOF_LAUNCH_S0M(2); else VOF_LAUNCH_S0M(1);
#undef VOF_LAUNCH_S0R
#undef S0R_BF
#undef VOF_LAUNCH_S0M
            return;
        }
    }
    if constexpr (std::is_same<VT, float>::value) {
        if (l == 0 && lv.C == nullptr && sweep0p_usable(c)) {
            const int NSW = nsweeps >= 2 ? 2 : 1;
            const int out = S0_W - 8 * NSW;
            const int nx = (lv.nj + out - 1) / out, nxp = (nx + 1) / 2;
            const int TI = pick_band_height(rows, nxp, c->cur_units);
            const int ny = (rows + TI - 1) / TI;
            dim3 g((unsigned)nxp * ny * np, 1, 1);
            int nci = 0, ncj = 0;
            double ebytes = 0.0;
            if (ecoarse) { nci = c->L[1].ni; ncj = c->L[1].nj; ebytes = 12.0 * c->L[1].npts; }
            const double moved = (8.0 + (x_in ? 9.0 : 6.0) * 4.0) * lv.npts + ebytes;   // I + b(3) + x(3) in, x(3) out, float32 vectors
            Prof p(c, VOF_K_GS0, 0, moved + (NSW - 1) * 44.0 * lv.npts, moved);
            Fine0 f0{c->frames, frame_stride(c), c->Nj, c->prm.speed_alpha, c->prm.remodelling_alpha, 1, c->pp};
#define VOF_LAUNCH_S0P(NS_, PO_)                                                                                                    \
            do {                                                                                                                    \
                const size_t ldsp = S0R<NS_, 0>::LDS_BYTES;                                                                         \
                if (ecoarse) k_sweep0p<NS_, true, false, PO_><<<g, 64, ldsp, c->stream>>>(f0, lv.ni, lv.nj, TI, nx, nxp, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj); \
                else if (!x_in) k_sweep0p<NS_, false, true, PO_><<<g, 64, ldsp, c->stream>>>(f0, lv.ni, lv.nj, TI, nx, nxp, ny, np, x_in, x_out, b, active, nullptr, 0, 0); \
                else k_sweep0p<NS_, false, false, PO_><<<g, 64, ldsp, c->stream>>>(f0, lv.ni, lv.nj, TI, nx, nxp, ny, np, x_in, x_out, b, active, nullptr, 0, 0); \
            } while (0)
            if (NSW == 2) { if (po) VOF_LAUNCH_S0P(2, 1); else VOF_LAUNCH_S0P(2, 0); }
            else { if (po) VOF_LAUNCH_S0P(1, 1); else VOF_LAUNCH_S0P(1, 0); }
#undef VOF_LAUNCH_S0P
            return;
        }
    }
    const bool geoB = (l > 0) ? c->geo_b_stored : c->geo_b_fine;
    const int out = geoB ? GeoB::OUT : GeoA::OUT, W = geoB ? GeoB::W : GeoA::W, IW = geoB ? GeoB::IW : GeoA::IW;
    const int TI = pick_band_height(rows, (lv.nj + (geoB ? 0 : po) + out - 1) / out, c->cur_units);
    const int nx = (lv.nj + (geoB ? 0 : po) + out - 1) / out, ny = (rows + TI - 1) / TI;
    dim3 g((unsigned)nx * ny * np, 1, 1);
    const double vs = sizeof(VT);
    int nci = 0, ncj = 0;
    double ebytes = 0.0;
    if (ecoarse) { nci = c->L[l + 1].ni; ncj = c->L[l + 1].nj; ebytes = 3.0 * vs * c->L[l + 1].npts; }
    if (l == 0 && lv.C == nullptr) {
        Prof p(c, VOF_K_GS0, 0, (8.0 + (x_in ? 9.0 : 6.0) * vs) * lv.npts + ebytes);   // I + b(3) + x(3) in, x(3) out (+ coarse e)
        SweepFine pol;
        pol.frames = c->frames; pol.frame_stride = frame_stride(c); pol.Nj = c->Nj;
        pol.alpha = c->prm.speed_alpha; pol.beta = c->prm.remodelling_alpha; pol.quirks = c->prm.reference_quirks && c->pq_smooth;
        pol.pp = c->pp;
        size_t lds = (size_t)(SW_RING * 3 * W) * sizeof(VT) + (size_t)(SW_RING * IW) * sizeof(double) +
                     (ecoarse ? (size_t)(3 * 3 * (W / 2 + 2)) * sizeof(VT) : 0);
        if (geoB) k_sweep<SweepFine, GeoB, VT><<<g, GeoB::THREADS, lds, c->stream>>>(pol, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
        else if (c->sweep0) {   // the dedicated level-0 kernel (same geometry, schedule and bits as k_sweep<SweepFine, GeoA>)
            Fine0 f0{pol.frames, pol.frame_stride, pol.Nj, pol.alpha, pol.beta, pol.quirks, pol.pp};
            if (ecoarse) k_sweep0<VT, true, false><<<g, S0_THREADS, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
            else if (!x_in) k_sweep0<VT, false, true><<<g, S0_THREADS, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
            else k_sweep0<VT, false, false><<<g, S0_THREADS, lds, c->stream>>>(f0, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
        }
        else k_sweep<SweepFine, GeoA, VT><<<g, GeoA::THREADS, lds, c->stream>>>(pol, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
    } else {
        Prof p(c, VOF_K_GS, l, (coef_bytes(c, l) + (x_in ? 9.0 : 6.0) * vs) * lv.npts + ebytes);   // C + b(3) + x(3) in, x(3) out (+ coarse e)
        size_t lds = (size_t)(SW_RING * 3 * W) * sizeof(VT);
        if (sweep_st_usable(c, l)) {   // packed stencil formats: the kernel with the decoupled coefficient stream
            const uint32_t* Cw = (const uint32_t*)lv.C;
            auto launch = [&](auto ct_tag) {
            using PCT = typename decltype(ct_tag)::type;
            if (ecoarse) {   // the sweep starts from x_in + P ecoarse (coarse rows through a 3-row LDS ring)
                const size_t lds_e = lds + (size_t)9 * (W / 2 + 2) * sizeof(VT);
                if (out64) k_sweep_st<PCT, VT, double, true><<<g, GeoB::THREADS, lds_e, c->stream>>>(Cw, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, (double*)x_out, b, active, ecoarse, nci, ncj, 0);
                else k_sweep_st<PCT, VT, VT, true><<<g, GeoB::THREADS, lds_e, c->stream>>>(Cw, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj, 0);
            } else {
                const int sk = (skip0 && x_in && !reverse) ? 1 : 0;
                if (out64) k_sweep_st<PCT, VT, double><<<g, GeoB::THREADS, lds, c->stream>>>(Cw, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, (double*)x_out, b, active, nullptr, 0, 0, sk);
                else k_sweep_st<PCT, VT><<<g, GeoB::THREADS, lds, c->stream>>>(Cw, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, nullptr, 0, 0, sk);
            }
            };
            if (c->cfmt == 3) launch(TypeTag<CoefF8>{}); else launch(TypeTag<CoefB16>{});
            return;
        }
        CDISPATCH(c, l, {
            SweepStored<CT> pol; pol.C = (const CW*)lv.C; pol.plane = CLay(lv.ni, lv.nj).plane;
            if (geoB) k_sweep<SweepStored<CT>, GeoB, VT><<<g, GeoB::THREADS, lds, c->stream>>>(pol, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
            else k_sweep<SweepStored<CT>, GeoA, VT><<<g, GeoA::THREADS, lds, c->stream>>>(pol, lv.ni, lv.nj, TI, po, nx, ny, np, x_in, x_out, b, active, ecoarse, nci, ncj);
        });
    }
}

// nu sweeps (from a zero guess if from_zero, else from x); the result is guaranteed to end in `x`.
template <typename VT>
VT* smooth_level_t(vof_ctx* c, int l, VT* x, VT* tmp, const VT* b, int nu, bool from_zero, bool reverse, int np,
                   const int* active, const VT* ecoarse = nullptr, bool allow_swap = false, bool final_smooth = false,
                   bool ec32 = false, bool out64 = false, bool skip0 = false) {
    // Returns the buffer that holds the result: `x`, or `tmp` when allow_swap is set and the last out-of-place sweep
    // ended there (saves a device-to-device copy on the coarse levels).
    // ecoarse: coarse-grid correction still to be added (x += P ecoarse).  On the matrix-free level 0 it is folded
    // into the first sweep (coarse rows streamed through LDS); otherwise the prolongation kernel runs first.
    const size_t bytes = (size_t)np * 3 * c->L[l].npts * sizeof(VT);
    const bool fold = ecoarse && nu > 0 && c->fused && c->fuse_prolong && !from_zero &&
                      ((l == 0 && c->L[0].C == nullptr && !c->geo_b_fine) || (sweep_st_usable(c, l) && c->fold_stored));
    if (ecoarse && !fold) {
        prolong_add_level_t<VT>(c, l, x, ecoarse, np, active);
        ecoarse = nullptr;
    }
    if (nu <= 0) {
        if (from_zero) hipMemsetAsync(x, 0, bytes, c->stream);
        return x;
    }
    if (!c->fused) {   // reference path: one launch per colour, in place (double vectors only)
        if (from_zero) hipMemsetAsync(x, 0, bytes, c->stream);
        for (int s = 0; s < nu; ++s)
            for (int k = 0; k < 4; ++k)
                gs_colour(c, l, (double*)x, (const double*)b, reverse ? 3 - k : k, np, active);
        return x;
    }
    // out-of-place fused sweeps: choose the first destination so that the last pass writes into x.  On level 0 a pass of
    // k_sweep0m performs two sweeps (temporal blocking): nu sweeps = ceil(nu / 2) passes over the data.
    const bool two = l == 0 && ((std::is_same<VT, double>::value && sweep0m_usable(c) && c->sweep0m_pairs) ||
                                (std::is_same<VT, float>::value && sweep0p_usable(c) && c->sweep0m_pairs));
    const int npass = two ? (nu + 1) / 2 : nu;
    const VT* src = from_zero ? nullptr : x;
    VT* dst = (from_zero && (npass % 2 == 1)) ? x : tmp;
    int left = nu;
    for (int s = 0; s < npass; ++s) {
        const int ns = two ? std::min(2, left) : 1;
        // the cycle's very last pass also delivers the Krylov product of its result, if one was requested
        const bool trail = final_smooth && s == npass - 1 && c->trail_set && c->trail_enabled && l == 0 &&
                           std::is_same<VT, double>::value && sweep0m_usable(c) && src != nullptr;
        sweep_level_t<VT>(c, l, src, dst, b, reverse, np, active, s == 0 ? ecoarse : (const VT*)nullptr, ns, trail, ec32,
                          out64 && s == npass - 1, skip0 && s == 0);
        left -= ns;
        src = dst;
        dst = (dst == x) ? tmp : x;
    }
    if (src != x) {
        if (allow_swap) return tmp;
        hipMemcpyAsync(x, src, bytes, hipMemcpyDeviceToDevice, c->stream);
    }
    return x;
}

// ---- coarse tail (k_tail_cycle): levels tail_first .. last in one launch, one workgroup per pair
void tail_emit(const vof_ctx* c, std::vector<unsigned char>* ops, int l, bool from_zero) {
    const vof_params& P = c->prm;
    const int last = (int)c->L.size() - 1, tl = l - c->tail_first;
    auto emit = [&](int code, int arg) { ops[0].push_back((unsigned char)code); ops[1].push_back((unsigned char)tl); ops[2].push_back((unsigned char)arg); };
    if (l == last) { emit(T_COARSE, 0); return; }
    const int nu1 = P.nu_pre_coarse > 0 ? P.nu_pre_coarse : P.nu_pre;
    const int nu2 = P.nu_post_coarse > 0 ? P.nu_post_coarse : P.nu_post;
    emit(T_SMOOTH, (nu1 << 2) | (from_zero ? 1 : 0));
    emit(T_RESTRICT, 0);
    tail_emit(c, ops, l + 1, true);
    if (P.w_cycle_level == l && l + 1 < last) {
        const int visits = P.w_cycle_visits > 0 ? P.w_cycle_visits : 2;
        for (int v = 1; v < visits; ++v) tail_emit(c, ops, l + 1, false);
    }
    emit(T_PROLONG, 0);
    emit(T_SMOOTH, (nu2 << 2) | 2);
}

// (Re)build the operation list for the current cycle parameters; false: the tail cannot be used (too many operations)
bool tail_prepare(vof_ctx* c) {
    if (c->tail_first < 0 || !c->tail_enabled || !c->fused) return false;
    const vof_params& P = c->prm;
    const int key[6] = {P.nu_pre, P.nu_post, P.nu_pre_coarse, P.nu_post_coarse, P.w_cycle_level, P.w_cycle_visits};
    if (memcmp(key, c->tail_key, sizeof key) != 0) {
        std::vector<unsigned char> ops[3];
        tail_emit(c, ops, c->tail_first, true);
        memcpy(c->tail_key, key, sizeof key);
        if (ops[0].size() > (size_t)TAIL_MAX_OPS - 1 || std::max({P.nu_pre, P.nu_post, P.nu_pre_coarse, P.nu_post_coarse}) > 63) { c->tail.n_ops = -1; return false; }
        c->tail.n_ops = (int)ops[0].size();
        for (int i = 0; i < c->tail.n_ops; ++i) { c->tail.op[i] = ops[0][i]; c->tail.op_level[i] = ops[1][i]; c->tail.op_arg[i] = ops[2][i]; }
    }
    return c->tail.n_ops > 0;
}

template <typename VT>
void tail_cycle_t(vof_ctx* c, VT* x, const VT* b, int np, const int* active, bool from_zero) {
    TailArgs A = c->tail;
    const int l0 = c->tail_first, last = (int)c->L.size() - 1;
    for (int l = l0; l <= last; ++l) A.L[l - l0].C = c->L[l].C;
    A.invT = c->invT;
    if (!from_zero) A.op_arg[0] &= ~1;   // the first operation is the pre-smoothing of the top tail level
    const Level& top = c->L[l0];
    Prof p(c, VOF_K_COARSE_TAIL, l0, (from_zero ? 6.0 : 9.0) * sizeof(VT) * top.npts);   // b in, x (in and) out; stencils stay in cache
    CDISPATCH(c, l0, (k_tail_cycle<CT, VT><<<np, TAIL_THREADS, c->tail_lds, c->stream>>>(A, b, x, from_zero ? 1 : 0, active)));
}

// vcycle_precision 3 applies when level 0 runs the kernels in which the two storage types meet: k_stream_resrestrict0 (float64
// in, float32 out) and k_sweep0m with the interpolated correction (float32 in); anything else keeps float64 everywhere
inline bool coarse32_ok(const vof_ctx* c, int nu_post) {
    return c->vcoarse32 && !c->vfloat && c->L.size() > 1 && c->L[0].C == nullptr && sweep0m_usable(c) && c->fuse_prolong &&
           c->fuse_restrict && c->stream_apply && nu_post > 0;
}

// One multigrid cycle on level l for A_l x = b, starting from a zero guess (from_zero) or from the contents of x.
// (x, tmp) are the level's ping-pong buffers.  Returns the buffer holding the result: `x`, or - on the levels >= 1,
// where the caller only reads it - `tmp`.  With prm.w_cycle_level == l the next coarser level is visited twice
// (the second visit continues from the first one's result): a W-cycle restricted to one level.
template <typename VT>
// after_post: x holds the result of a previous visit of this level with the same b, i.e. of its reverse post-smoothing sweep
VT* vcycle_t(vof_ctx* c, int l, VT* x, VT* tmp, const VT* b, int np, const int* active, bool from_zero = true,
             bool after_post = false) {
    int last = (int)c->L.size() - 1;
    if (l == last) { coarse_solve_t<VT>(c, b, x, np, active); return x; }
    if (l == c->tail_first && l > 0 && tail_prepare(c)) { tail_cycle_t<VT>(c, x, b, np, active, from_zero); return x; }
    Level& lv = c->L[l];
    Level& nx = c->L[l + 1];
    const int nu1 = (l > 0 && c->prm.nu_pre_coarse > 0) ? c->prm.nu_pre_coarse : c->prm.nu_pre;
    const int nu2 = (l > 0 && c->prm.nu_post_coarse > 0) ? c->prm.nu_post_coarse : c->prm.nu_post;
    // pre-smoothing.  Level 0: result forced into x (the Krylov loop owns that buffer).  Stored levels: the result may end in
    // the ping-pong partner (the caller only reads the buffer this function returns), so the two just trade names - and the
    // partner then still holds the input of the last sweep, which is all k_resrestrict_u needs besides the result.
    const bool resu = l > 0 && lv.C != nullptr && c->fused && c->fuse_resu && nu1 >= 1;
    bool rr_fused = false;   // level 0: the coarse right-hand side came out of the pre-smoothing pass (k_sweep0r, TRAIL = 2)
    if constexpr (std::is_same<VT, double>::value) {
        if (l == 0 && c->fuse_rr && from_zero && nu1 == 2 && lv.C == nullptr && c->stream_apply && c->fuse_restrict && sweep0m_usable(c) && c->sweep0m_pairs) {
            c->rr_f32 = coarse32_ok(c, nu2);
            c->rr_out = nx.b;
            c->rr_done = false;
        }
    }
    if (l > 0 && c->fused) {
        VT* xr = smooth_level_t<VT>(c, l, x, tmp, b, nu1, from_zero, false, np, active, nullptr, /*allow_swap=*/true, false, false, false,
                                    /*skip0=*/after_post && !from_zero && nu2 >= 1 && c->skip_colour0);
        if (xr != x) std::swap(x, tmp);
    } else {
        smooth_level_t<VT>(c, l, x, tmp, b, nu1, from_zero, false, np, active);
    }
    if (l == 0 && c->rr_out) { rr_fused = c->rr_done; c->rr_out = nullptr; c->rr_done = false; }
    if constexpr (std::is_same<VT, double>::value) {
        if (l == 0 && coarse32_ok(c, nu2)) {
            // float64 vectors on level 0, float32 below: the fused residual + restriction writes the coarse right-hand side as
            // float32, the levels below run in float32, and the post-smoothing pass interpolates the float32 correction
            if (!rr_fused) resrestrict_fine_t<double, float>(c, x, b, (float*)nx.b, np, active);
            float* fx = (float*)nx.x;
            float* ft = (float*)nx.x2;
            // The last visit of level 1 hands its result up as float64 when its last operation is a k_sweep_st sweep (a regular
            // stored level with post-smoothing): 12 more bytes per level-1 point written there, but the pass above then reads
            // the correction as it does in the all-float64 cycle - measured: interpolating from float32 rows costs that pass
            // 6 % (5 ms per step at 255 pairs), widening the stores of the level-1 sweep costs 1 ms
            const int nu2c = c->prm.nu_post_coarse > 0 ? c->prm.nu_post_coarse : c->prm.nu_post;
            const bool can64 = 1 < last && !(c->tail_first == 1 && tail_prepare(c)) && c->L[1].C != nullptr && c->sweep_st &&
                               c->geo_b_stored && c->cfmt >= 2 && nu2c > 0;
            const int visits = (c->prm.w_cycle_level == 0 && 1 < last) ? (c->prm.w_cycle_visits > 0 ? c->prm.w_cycle_visits : 2) : 1;
            c->emit64 = can64 && visits == 1;
            float* fe = vcycle_t<float>(c, 1, fx, ft, (const float*)nx.b, np, active, true);
            for (int v = 1; v < visits; ++v) {
                float* other = (fe == fx) ? ft : fx;
                c->emit64 = can64 && v == visits - 1;
                fe = vcycle_t<float>(c, 1, fe, other, (const float*)nx.b, np, active, false);
            }
            c->emit64 = false;
            return smooth_level_t<double>(c, 0, x, tmp, b, nu2, false, true, np, active, (const double*)fe, /*allow_swap=*/true,
                                          /*final_smooth=*/true, /*ec32=*/!can64);
        }
    }
    if (resu) {
        const VT* x_old = (from_zero && nu1 == 1) ? nullptr : tmp;
        resrestrict_u_t<VT>(c, l, x, x_old, (VT*)nx.b, np, active);
    } else if (l == 0 && lv.C == nullptr && c->stream_apply && c->fuse_restrict) {
        if (!rr_fused) resrestrict_fine_t<VT>(c, x, b, (VT*)nx.b, np, active);
    } else {
        apply_level_t<VT>(c, l, x, b, (VT*)lv.r, 1, np, active);
        restrict_level_t<VT>(c, l, (const VT*)lv.r, (VT*)nx.b, np, active);
    }
    VT* cx = (VT*)nx.x;
    VT* ct = (VT*)nx.x2;
    VT* ec = vcycle_t<VT>(c, l + 1, cx, ct, (const VT*)nx.b, np, active, true);
    if (c->prm.w_cycle_level == l && l + 1 < last) {
        const int visits = c->prm.w_cycle_visits > 0 ? c->prm.w_cycle_visits : 2;
        for (int v = 1; v < visits; ++v) {
            VT* other = (ec == cx) ? ct : cx;
            ec = vcycle_t<VT>(c, l + 1, ec, other, (const VT*)nx.b, np, active, false, /*after_post=*/true);
        }
    }
    // (level 1 under a float64 level 0, vcycle_precision 3: the last post-smoothing sweep writes the result as float64)
    const bool out64 = c->emit64 && l == 1 && std::is_same<VT, float>::value && nu2 > 0;
    return smooth_level_t<VT>(c, l, x, tmp, b, nu2, false, true, np, active, ec, /*allow_swap=*/true, /*final_smooth=*/l == 0,
                              /*ec32=*/false, out64);
}

template <typename VT> int direct_apply_t(vof_ctx* c, VT* z, const VT* r, int np);   // direct preconditioner, below

// One cycle M b -> *xslot (c->ky or c->kz).  The out-of-place sweeps may leave the result in the level-0 ping-pong partner
// instead (an odd number of passes); the two buffers then trade places - a pointer swap instead of a copy of the vector.
void vcycle(vof_ctx* c, double** xslot, const void* b, int np, const int* active) {
    if (c->direct_on) {   // the direct preconditioner takes the place of the cycle (every pair of the batch, active or not)
        VDISPATCH(c, direct_apply_t<VT>(c, (VT*)*xslot, (const VT*)b, np));
        return;
    }
    void* res = nullptr;
    VDISPATCH(c, res = (void*)vcycle_t<VT>(c, 0, (VT*)*xslot, (VT*)c->L[0].x2, (const VT*)b, np, active));
    if (res != (void*)*xslot) {
        c->L[0].x2 = (void*)*xslot;
        *xslot = (double*)res;
    }
}

// Build the Galerkin hierarchy and the coarsest-level dense inverse for the current batch.
int build_hierarchy(vof_ctx* c, int np) {
    const vof_params& P = c->prm;
    c->cfmt = P.coarse_precision;
    int nl = (int)c->L.size();
    for (int l = 0; l + 1 < nl; ++l) {
        Level &f = c->L[l], &k = c->L[l + 1];
        dim3 g = grid2d(k.ni, k.nj, np);
        if (l == 0) {
            Prof p(c, VOF_K_GALERKIN0, 0);
            CDISPATCH(c, 1, (k_galerkin<double, CT, true><<<g, blk2d, 0, c->stream>>>(
                                 c->frames, frame_stride(c), c->Nj, P.speed_alpha, P.remodelling_alpha, P.reference_quirks && c->pq_hier,
                                 nullptr, f.ni, f.nj, (CW*)k.C, k.ni, k.nj, c->pp)));
        } else {
            Prof p(c, VOF_K_GALERKIN, l);
            CDISPATCH(c, 1, (k_galerkin<CT, CT, false><<<g, blk2d, 0, c->stream>>>(
                                 nullptr, 0, 0, 0.0, 0.0, 0, (const CW*)f.C, f.ni, f.nj, (CW*)k.C, k.ni, k.nj, nullptr)));
        }
    }
    return 0;
}

}  // namespace

// The 1-level case needs the fine stencil in stored form.
namespace vof {
__global__ __launch_bounds__(NT) void k_store_fine_stencil(const double* __restrict__ frames, size_t frame_stride,
                                                           int Nj, double alpha, double beta, int quirks, int ni,
                                                           int nj, double* __restrict__ C,
                                                           const PairParam* __restrict__ pp) {
    int q = blockIdx.x * BX + threadIdx.x, p = blockIdx.y * BY + threadIdx.y, pair = blockIdx.z;
    if (p >= ni || q >= nj) return;
    size_t npts = (size_t)ni * nj, idx = (size_t)p * nj + q;
    int fidx = pair;
    if (pp) { alpha = pp[pair].alpha; beta = pp[pair].beta; fidx = pp[pair].frame; }
    PixCoef k = pix_coef(frames + (size_t)fidx * frame_stride, Nj, p, q, quirks);
    const CLay L(ni, nj);
    double* out = C + (size_t)pair * 81 * L.plane + L.idx(p, q);
    (void)idx;
    npts = L.plane;
    for (int oi = -1; oi <= 1; ++oi)
        for (int oj = -1; oj <= 1; ++oj) {
            double blk[9];
            int tp = p + oi, tq = q + oj;
            if (tp < 0 || tp >= ni || tq < 0 || tq >= nj) {
                for (int t = 0; t < 9; ++t) blk[t] = 0.0;
            } else {
                folded_block(k, alpha, beta, p, q, ni, nj, oi, oj, blk);
            }
            for (int t = 0; t < 9; ++t) out[(size_t)(((oi + 1) * 3 + (oj + 1)) * 9 + t) * npts] = blk[t];
        }
}
}  // namespace vof

namespace {

int setup_batch(vof_ctx* c, const double* frames_dev, int np) {
    c->frames = frames_dev;
    c->npairs = np;
    c->cur_units = np;
    int nl = (int)c->L.size();
    if (nl == 1) {
        Level& f = c->L[0];
        Prof p(c, VOF_K_GALERKIN0, 0);
        k_store_fine_stencil<<<grid2d(f.ni, f.nj, np), blk2d, 0, c->stream>>>(
            c->frames, frame_stride(c), c->Nj, c->prm.speed_alpha, c->prm.remodelling_alpha,
            c->prm.reference_quirks, f.ni, f.nj, (double*)f.C, c->pp);
        c->cfmt = 0;
    } else {
        build_hierarchy(c, np);
    }
    Level& last = c->L[nl - 1];
    {
        Prof p(c, VOF_K_COARSE_SETUP, nl - 1);
        CDISPATCH(c, nl - 1, (k_coarse_build<CT><<<np, 256, 0, c->stream>>>((const CW*)last.C, last.ni, last.nj, c->W)));
        dbg_sync_check(c, "coarse_build", nl - 1);
        k_coarse_invert<<<np, 1024, 0, c->stream>>>(c->W, c->nd, c->invT);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

inline dim3 rgrid(const vof_ctx* c, int np) { return dim3(c->nblk, np, 1); }

// Number of active pairs (copies the flags to the host; synchronises the stream).
int count_active(vof_ctx* c, int np) {
    if (hipMemcpyAsync(c->h_active, c->active, np * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    int n = 0;
    for (int k = 0; k < np; ++k) n += c->h_active[k] != 0;
    return n;
}

// Buffers of the GMRES fallback: restart length = min(requested, what fits in half of the free device memory).
int gmres_buffers(vof_ctx* c, int want_m) {
    want_m = std::min(want_m, GM_MAXM);
    if (c->gm_V) return 0;   // allocated once per context
    const size_t vec = (size_t)c->B * 3 * c->L[0].npts * sizeof(double);
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t fixed = (size_t)c->B * (sizeof(GmresState) + (GM_NV + 1) * c->nblk * sizeof(double) + sizeof(int));
    const double budget = 0.5 * (double)free_b - (double)fixed;
    int fit = budget > 0 ? (int)std::min<double>(budget / (double)vec, 1e6) - 1 : 0;
    int m = std::min(want_m, fit);
    if (m < 4) { c->err = "not enough device memory for the GMRES fallback (lower max_pairs_in_flight)"; return -3; }
    if (int rc = dev_alloc(c, &c->gm_V, (size_t)(m + 1) * c->B * 3 * c->L[0].npts)) return rc;
    if (int rc = dev_alloc(c, &c->gm_state, (size_t)c->B)) return rc;
    if (int rc = dev_alloc(c, &c->gm_partials, (size_t)c->B * (GM_NV + 1) * c->nblk)) return rc;
    c->gm_m = m;
    return 0;
}

// Restarted, right-preconditioned GMRES on the pairs that are not converged yet: x = x_0 + M (V_k y), M = one
// multigrid cycle (float64 vectors), restart from the true residual b - A x.  `iterations` keeps counting Krylov steps
// (one cycle application each) on top of the BiCGStab iterations already spent.
int gmres_phase(vof_ctx* c, int np, int* handed_over) {
    const vof_params& P = c->prm;
    hipStream_t s = c->stream;
    const size_t len = 3 * c->L[0].npts;
    const size_t vstride = (size_t)c->B * len;
    if (!c->gm_cycle) {   // flags are needed before the (large) basis is
        if (int rc = dev_alloc(c, &c->gm_cycle, (size_t)c->B)) return rc;
    }
    k_gm_begin<<<(np + 63) / 64, 64, 0, s>>>(c->sc, c->active, c->gm_cycle, np, P.max_iterations);
    dbg_sync_check(c, "gm_begin", 0);
    int nact = count_active(c, np);
    if (nact < 0) { c->err = "stream synchronize failed"; return -2; }
    if (nact == 0) return 0;
    *handed_over = nact;
    if (!c->gm_V) {
        int rc = gmres_buffers(c, P.gmres_restart > 0 ? P.gmres_restart : 100);
        if (rc == -3) { c->err.clear(); return 0; }   // no room: leave the pairs unconverged (reported per pair)
        if (rc) return rc;
    }
    c->gmres_pairs += nact;
    const int m = std::min(c->gm_m, P.gmres_restart > 0 ? P.gmres_restart : 100);
    c->vfloat = false;   // float64 cycle vectors: the basis vectors are the cycle's right-hand sides
    c->vcoarse32 = false;
    double* V = c->gm_V;
    double* w = c->kt;
    const dim3 rg = rgrid(c, np);
    const int coef_c = (int)(offsetof(GmresState, c) / sizeof(double)), coef_y = (int)(offsetof(GmresState, y) / sizeof(double));
    for (;;) {
        c->cur_units = nact;
        int nb = residual_d(c, c->kx, c->kb, V, np, c->gm_cycle, 1);      // V_0 = b - A x and its norm
        if (!nb) { Prof p(c, VOF_K_REDUCE, 0); k_dot2<<<rg, RBLK, 0, s>>>(V, V, nullptr, nullptr, len, c->partials, c->gm_cycle); nb = c->nblk; }
        { Prof p(c, VOF_K_VECTOR, 0);
          k_gm_init<<<np, 64, 0, s>>>(c->gm_state, c->sc, c->partials, nb, c->active, c->gm_cycle, P.max_iterations); }
        nact = count_active(c, np);
        if (nact < 0) { c->err = "stream synchronize failed"; return -2; }
        if (nact == 0) break;
        c->cur_units = nact;
        { Prof p(c, VOF_K_VECTOR, 0, 16.0 * len); k_gm_scale<<<rg, RBLK, 0, s>>>(V, V, len, c->gm_state, c->active); }
        int jdone = 0;
        for (int j = 0; j < m; ++j) {
            const int* act = c->active;
            vcycle(c, &c->ky, V + (size_t)j * vstride, np, act);             // z = M v_j
            krylov_apply(c, c->ky, w, np, act);                              // w = A z
            for (int pass = 0; pass < 2; ++pass) {                           // classical Gram-Schmidt, twice
                for (int i0 = 0; i0 <= j; i0 += GM_NV) {
                    int cnt = std::min(GM_NV, j + 1 - i0);
                    Prof p(c, VOF_K_REDUCE, 0, 8.0 * len * (cnt + 1));
                    k_gm_multidot<<<rg, RBLK, 0, s>>>(V + (size_t)i0 * vstride, vstride, cnt, w, len, c->gm_partials, act);
                    k_gm_hcoef<<<np, 64, 0, s>>>(c->gm_state, c->gm_partials, c->nblk, i0, cnt, pass, act);
                }
                for (int i0 = 0; i0 <= j; i0 += GM_NV) {
                    int cnt = std::min(GM_NV, j + 1 - i0);
                    bool last = pass == 1 && i0 + GM_NV > j;                 // last chunk: write v_{j+1} (unnormalised) + norm
                    Prof p(c, VOF_K_VECTOR, 0, 8.0 * len * (cnt + 2));
                    k_gm_axpy<<<rg, RBLK, 0, s>>>(V + (size_t)i0 * vstride, vstride, i0, cnt, c->gm_state, coef_c, -1.0, w,
                                                  last ? V + (size_t)(j + 1) * vstride : w, len, act, 0,
                                                  last ? c->gm_partials : nullptr);
                }
            }
            { Prof p(c, VOF_K_VECTOR, 0, 16.0 * len);
              k_gm_givens<<<np, 64, 0, s>>>(c->gm_state, c->sc, c->gm_partials, c->nblk, j, c->active, P.max_iterations);
              k_gm_scale<<<rg, RBLK, 0, s>>>(V + (size_t)(j + 1) * vstride, V + (size_t)(j + 1) * vstride, len, c->gm_state, c->active); }
            jdone = j + 1;
            nact = count_active(c, np);
            if (nact < 0) { c->err = "stream synchronize failed"; return -2; }
            if (nact == 0) break;
            c->cur_units = nact;
        }
        // x += M (V_k y) for every pair of this cycle
        { Prof p(c, VOF_K_VECTOR, 0); k_gm_solve_y<<<(np + 63) / 64, 64, 0, s>>>(c->gm_state, c->gm_cycle, np); }
        for (int i0 = 0; i0 < jdone; i0 += GM_NV) {
            int cnt = std::min(GM_NV, jdone - i0);
            Prof p(c, VOF_K_VECTOR, 0, 8.0 * len * (cnt + 2));
            k_gm_axpy<<<rg, RBLK, 0, s>>>(V + (size_t)i0 * vstride, vstride, i0, cnt, c->gm_state, coef_y, 1.0,
                                          i0 ? c->kp : nullptr, c->kp, len, c->gm_cycle, 1, nullptr);
        }
        vcycle(c, &c->ky, c->kp, np, c->gm_cycle);
        { Prof p(c, VOF_K_VECTOR, 0, 24.0 * len); k_gm_xpy<<<rg, RBLK, 0, s>>>(c->kx, c->ky, len, c->gm_cycle); }
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------- direct preconditioner
// rocSOLVER (dense LU + inverse of the m x m Schur blocks) is loaded on first use: the multigrid path has no such dependency.
struct RocSolverApi {
    void* lib = nullptr;
    int (*create_handle)(void**) = nullptr;
    int (*destroy_handle)(void*) = nullptr;
    int (*set_stream)(void*, hipStream_t) = nullptr;
    int (*getrf)(void*, int, int, double*, int, long long, int*, long long, int*, int) = nullptr;
    int (*getri)(void*, int, double*, int, long long, int*, long long, int*, int) = nullptr;
    std::string err;
    // already in the process (loaded by us earlier, or e.g. by PyTorch)?  Then using it costs nothing; a first load of the
    // ~0.9 GB library can take minutes on a machine that has never read it.
    bool resident() {
        if (lib) return true;
        const char* names[] = {"librocsolver.so.0", "librocsolver.so"};
        for (const char* n : names)
            if (void* h = dlopen(n, RTLD_NOLOAD | RTLD_LAZY)) { dlclose(h); return true; }
        return false;
    }
    bool load() {
        if (lib) return true;
        const char* names[] = {getenv("VOF_ROCSOLVER_LIB"), "librocsolver.so.0", "librocsolver.so", "/opt/rocm/lib/librocsolver.so.0"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            const char* e = dlerror();   // (one call: dlerror() clears the message it returns)
            err = std::string("cannot load rocSOLVER: ") + (e ? e : "not found");
            return false;
        }
        create_handle = (int (*)(void**))dlsym(lib, "rocblas_create_handle");
        destroy_handle = (int (*)(void*))dlsym(lib, "rocblas_destroy_handle");
        set_stream = (int (*)(void*, hipStream_t))dlsym(lib, "rocblas_set_stream");
        getrf = (int (*)(void*, int, int, double*, int, long long, int*, long long, int*, int))dlsym(lib, "rocsolver_dgetrf_strided_batched");
        getri = (int (*)(void*, int, double*, int, long long, int*, long long, int*, int))dlsym(lib, "rocsolver_dgetri_strided_batched");
        if (!create_handle || !destroy_handle || !set_stream || !getrf || !getri) {
            err = "rocSOLVER / rocBLAS symbols missing";
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
RocSolverApi g_roc;

// Dense inverse of the Schur blocks, all in-house: the one-workgroup Gauss-Jordan kernel with partial pivoting up to
// DIRECT_OWN_MAX unknowns per image row (images up to 66 pixels wide), the blocked Gauss-Jordan on the FP64 matrix cores
// beyond (vof_direct.hpp).  VOF_DIRECT_LU=own|blocked|rocsolver forces one (rocSOLVER getrf + getri,
// round 2's choice for wide images, is loaded with dlopen only when asked for: an A/B reference, not a product path).
constexpr int DIRECT_OWN_MAX = 192;   // (round 2: 640; the blocked inverse is 6.6 x faster on the reference's 400-combination sweep at 128 x 128: 29.6 -> 4.5 s)
bool direct_uses_rocsolver(const vof_ctx*) {
    const char* e = getenv("VOF_DIRECT_LU");
    return e && e[0] == 'r';
}
bool direct_uses_blocked(const vof_ctx* c) {
    if (const char* e = getenv("VOF_DIRECT_LU")) return e[0] == 'b';
    return 3 * c->L[0].nj > DIRECT_OWN_MAX;
}
int direct_ld(const vof_ctx* c) {
    const int m = 3 * c->L[0].nj;
    return direct_uses_blocked(c) ? ((m + DNB - 1) / DNB) * DNB : m;
}

// The automatic re-solve (preconditioner 2) needs nothing but room for the buffers.
int direct_capacity(vof_ctx* c, int want);
bool direct_ok_for_fallback(vof_ctx* c) { return direct_capacity(c, 1) >= 1; }

// device bytes the direct preconditioner needs per pair in flight
size_t direct_bytes_per_pair(const vof_ctx* c) {
    const size_t ni = c->L[0].ni, nj = c->L[0].nj, m = 3 * nj, ld = (size_t)direct_ld(c);
    const size_t panels = direct_uses_blocked(c) ? (2 * ld + DNB) * DNB : 0;
    return (ni * ld * ld + ld * ld + panels + ni * nj * DIR_TAB + (3 * ni + 1) * m) * sizeof(double) + (m + 1) * sizeof(int);
}

// how many pairs the direct preconditioner can hold (0: it does not fit / is not available)
int direct_capacity(vof_ctx* c, int want) {
    if (c->dir_cap > 0) return c->dir_cap;
    if (c->L.size() < 2 || c->L[0].C != nullptr) return 0;                      // one-level grids are solved directly anyway
    if ((size_t)3 * c->L[0].nj > 8192) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    const double per = (double)direct_bytes_per_pair(c);
    int fit = (int)std::min<double>(0.6 * (double)free_b / per, 1e6);
    return std::max(0, std::min(fit, want));
}

int direct_alloc(vof_ctx* c, int pairs) {
    if (c->dir_cap >= pairs) return 0;
    if (c->dir_cap > 0) { c->err = "direct preconditioner buffers already allocated for a smaller batch"; return -3; }
    const bool trace = getenv("VOF_TRACE") != nullptr;
    const size_t ni = c->L[0].ni, nj = c->L[0].nj, m = 3 * nj, P = (size_t)pairs, ld = (size_t)direct_ld(c);
    c->dir_ld = (int)ld;
    if (direct_uses_rocsolver(c)) {
        // (the library is ~0.9 GB: its first load on a machine can take minutes; the small blocks use the built-in kernel)
        if (trace) { fprintf(stderr, "[vof] direct_alloc: loading rocSOLVER\n"); fflush(stderr); }
        if (!g_roc.load()) { c->err = g_roc.err; return -3; }
        if (trace) { fprintf(stderr, "[vof] direct_alloc: rocSOLVER loaded, allocating for %d pairs\n", pairs); fflush(stderr); }
    }
    if (int rc = dev_alloc(c, &c->dir_T, P * ni * ld * ld)) return rc;
    if (int rc = dev_alloc(c, &c->dir_W, P * ld * ld)) return rc;
    if (direct_uses_blocked(c)) {
        if (int rc = dev_alloc(c, &c->dir_R, P * ld * DNB)) return rc;
        if (int rc = dev_alloc(c, &c->dir_C, P * ld * DNB)) return rc;
        if (int rc = dev_alloc(c, &c->dir_D, P * DNB * DNB)) return rc;
        const int lds = (DNB * DNB + DNB * DNB_LDB) * (int)sizeof(double);
        HIPCHK(hipFuncSetAttribute((const void*)k_dir_bgj_panel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_dir_bgj_update, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    if (int rc = dev_alloc(c, &c->dir_tabs, P * ni * nj * DIR_TAB)) return rc;
    if (int rc = dev_alloc(c, &c->dir_r, P * ni * m)) return rc;
    if (int rc = dev_alloc(c, &c->dir_y, P * ni * m)) return rc;
    if (int rc = dev_alloc(c, &c->dir_x, P * ni * m)) return rc;
    if (int rc = dev_alloc(c, &c->dir_t, P * m)) return rc;
    if (int rc = dev_alloc(c, &c->dir_ipiv, P * m)) return rc;
    if (int rc = dev_alloc(c, &c->dir_info, P)) return rc;
    if (direct_uses_rocsolver(c) && !c->roc_handle) {
        if (trace) { fprintf(stderr, "[vof] direct_alloc: rocblas_create_handle\n"); fflush(stderr); }
        if (g_roc.create_handle(&c->roc_handle) != 0) { c->err = "rocblas_create_handle failed"; return -2; }
        if (g_roc.set_stream(c->roc_handle, c->stream) != 0) { c->err = "rocblas_set_stream failed"; return -2; }
        if (trace) { fprintf(stderr, "[vof] direct_alloc: handle ready\n"); fflush(stderr); }
    }
    c->dir_cap = pairs;
    return 0;
}

// factorisation for the current batch (frames / PairParam table as set up by solve_batch)
int direct_setup(vof_ctx* c, int np) {
    const vof_params& P = c->prm;
    const int ni = c->L[0].ni, nj = c->L[0].nj, m = 3 * nj, ld = c->dir_ld;
    const size_t sT = (size_t)ni * ld * ld, sW = (size_t)ld * ld, sTab = (size_t)ni * nj * DIR_TAB, rowTab = (size_t)nj * DIR_TAB;
    hipStream_t s = c->stream;
    Prof pr(c, VOF_K_COARSE_SETUP, 0);
    HIPCHK(hipMemsetAsync(c->dir_info, 0, (size_t)np * sizeof(int), s));
    k_dir_tables<<<dim3((nj + 255) / 256, ni, np), 256, 0, s>>>(c->frames, frame_stride(c), c->Nj, P.speed_alpha, P.remodelling_alpha,
                                                               P.reference_quirks, ni, nj, c->dir_tabs, c->pp);
    const dim3 gm((m + 255) / 256, m, np), gs((ld + 255) / 256, ld, np);
    const bool trace = getenv("VOF_TRACE") != nullptr;
    const bool blocked = direct_uses_blocked(c);
    const int nt = ld / DNB;
    const size_t bgj_lds = (size_t)(DNB * DNB + DNB * DNB_LDB) * sizeof(double);
    for (int p = 0; p < ni; ++p) {
        if (trace && (p < 2 || p == ni - 1)) { HIPCHK(hipStreamSynchronize(s)); fprintf(stderr, "[vof] direct_setup: row %d of %d (m = %d, %d pairs)\n", p, ni, m, np); fflush(stderr); }
        double* Tp = c->dir_T + (size_t)p * ld * ld;
        if (p > 0) k_dir_W<<<gm, 256, 0, s>>>(Tp - (size_t)ld * ld, sT, c->dir_tabs + (size_t)(p - 1) * rowTab, sTab, nj, c->dir_W, sW, ld);
        k_dir_schur<<<gs, 256, 0, s>>>(c->dir_tabs + (size_t)p * rowTab, sTab, nj, p > 0 ? c->dir_W : nullptr, sW, Tp, sT, ld);
        if (blocked) {
            for (int kt = 0; kt < nt; ++kt) {
                k_dir_bgj_panel<<<dim3(nt, 2, np), 256, bgj_lds, s>>>(Tp, sT, ld, kt, c->dir_R, c->dir_C, c->dir_D, c->dir_info);
                k_dir_bgj_update<<<dim3(nt, nt, np), 256, bgj_lds, s>>>(Tp, sT, ld, kt, c->dir_R, c->dir_C, c->dir_D);
            }
        } else if (direct_uses_rocsolver(c)) {
            if (g_roc.getrf(c->roc_handle, m, m, Tp, m, (long long)sT, c->dir_ipiv, (long long)m, c->dir_info, np) != 0 ||
                g_roc.getri(c->roc_handle, m, Tp, m, (long long)sT, c->dir_ipiv, (long long)m, c->dir_info, np) != 0) {
                c->err = "rocSOLVER getrf / getri failed";
                return -2;
            }
        } else {
            k_dir_invert<<<np, 1024, 2 * (size_t)m * sizeof(double), s>>>(Tp, sT, m, c->dir_ipiv, c->dir_info);
        }
    }
    HIPCHK(hipGetLastError());
    if (!direct_uses_rocsolver(c)) {   // a singular pivot anywhere makes the preconditioner useless for that pair: report it instead of iterating on garbage
        std::vector<int> info((size_t)np);
        HIPCHK(hipMemcpyAsync(info.data(), c->dir_info, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int k = 0; k < np; ++k)
            if (info[k] != 0) { c->err = "direct preconditioner: a Schur block of pair " + std::to_string(k) + " of the batch is singular"; return -2; }
    }
    return 0;
}

// z = A^{-1} r by block forward / backward substitution (r, z: level-0 vectors of the V-cycle type)
template <typename VT>
int direct_apply_t(vof_ctx* c, VT* z, const VT* r, int np) {
    const int ni = c->L[0].ni, nj = c->L[0].nj, m = 3 * nj, ld = c->dir_ld;
    const size_t sT = (size_t)ni * ld * ld, sTab = (size_t)ni * nj * DIR_TAB, rowTab = (size_t)nj * DIR_TAB, sV = (size_t)ni * m;
    hipStream_t s = c->stream;
    Prof pr(c, VOF_K_COARSE_SOLVE, 0);
    const dim3 gp(64, np), gv((m + 255) / 256, np), gg((m + 63) / 64, np);
    k_dir_permute<const VT, true><<<gp, 256, 0, s>>>(r, c->dir_r, ni, nj);
    for (int p = 0; p < ni; ++p) {      // forward: y_p = r_p - L_p T_{p-1} y_{p-1}
        if (p > 0) k_dir_gemv<<<gg, 256, 0, s>>>(c->dir_T + (size_t)(p - 1) * ld * ld, sT, m, c->dir_y + (size_t)(p - 1) * m, sV, c->dir_t, (size_t)m, ld);
        k_dir_rowupdate<<<gv, 256, 0, s>>>(c->dir_tabs + (size_t)p * rowTab, sTab, nj, -1, c->dir_r + (size_t)p * m, sV,
                                          p > 0 ? c->dir_t : nullptr, (size_t)m, c->dir_y + (size_t)p * m, sV);
    }
    for (int p = ni - 1; p >= 0; --p) {  // backward: x_p = T_p (y_p - U_p x_{p+1})
        k_dir_rowupdate<<<gv, 256, 0, s>>>(c->dir_tabs + (size_t)p * rowTab, sTab, nj, +1, c->dir_y + (size_t)p * m, sV,
                                          p + 1 < ni ? c->dir_x + (size_t)(p + 1) * m : nullptr, sV, c->dir_t, (size_t)m);
        k_dir_gemv<<<gg, 256, 0, s>>>(c->dir_T + (size_t)p * ld * ld, sT, m, c->dir_t, (size_t)m, c->dir_x + (size_t)p * m, sV, ld);
    }
    k_dir_permute<VT, false><<<gp, 256, 0, s>>>(z, c->dir_x, ni, nj);
    return 0;
}

int solve_batch(vof_ctx* c, const double* frames_dev, int np, double* vx, double* vy, double* gm, double* speed,
                vof_pair_stats* stats) {
    const vof_params& P = c->prm;
    // preconditioner 2 ("auto"): when the direct re-solve is available, the multigrid attempt is not run to the reference's
    // 1000 iterations (OF.py:1120) - where the cycle works it needs 3 .. 100 Krylov steps, and a pair that
    // has not converged after MG_ATTEMPT_CAP of them is handed to the direct preconditioner, which settles it in one or two
    struct MaxItGuard { vof_params& p; int saved; ~MaxItGuard() { p.max_iterations = saved; } } max_it_guard{c->prm, c->prm.max_iterations};
    constexpr int MG_ATTEMPT_CAP = 150;
    // (images up to 1026 pixels wide - the reference's real-data sizes -: the direct re-solve takes 3.3 s per pair at 514^2, 18 s and
    // 79 GB at 1026^2, the multigrid attempt to 1000 iterations as long again)
    if (P.preconditioner == 2 && !c->direct_on && c->L.size() > 1 && P.max_iterations > MG_ATTEMPT_CAP && 3 * c->L[0].nj <= 3100 &&
        direct_ok_for_fallback(c))
        c->prm.max_iterations = MG_ATTEMPT_CAP;
    if (stats) HIPCHK(hipEventRecord(c->ev_batch[0], c->stream));
    // storage type of the cycle vectors for this batch (an earlier batch may have switched to float64: "auto"
    // precision after 8 iterations, GMRES fallback)
    c->vfloat = (P.vcycle_precision == 1 || P.vcycle_precision == 2) && c->fused && c->L.size() > 1 && !c->direct_on;
    c->vcoarse32 = P.vcycle_precision == 3 && !c->direct_on;
    if (c->direct_on) {   // direct preconditioner: block-tridiagonal LU instead of the Galerkin hierarchy
        if (np > c->dir_cap) { c->err = "batch larger than the direct preconditioner's buffers"; return -1; }
        c->frames = frames_dev;
        c->npairs = np;
        c->cur_units = np;
        if (int rc = direct_setup(c, np)) return rc;
        c->direct_pairs += np;
    } else if (int rc = setup_batch(c, frames_dev, np)) return rc;
    Level& f = c->L[0];
    const size_t len = 3 * f.npts;
    hipStream_t s = c->stream;
    // right-hand side and its norm
    // initial guess (OF.py:799-802: constants, in pixels/frame), right-hand side, initial residual r0 and shadow residual r^ = r0
    double sx = P.delta_t / P.delta_x;
    const bool zero_guess = !c->guess_src && (P.initial_v_x == 0.0 && P.initial_v_y == 0.0 && P.initial_remodelling == 0.0);
    double* rh = c->krh;   // r^: from the zero guess it IS b (read-only from here on) - no copy; a restart switches to the real buffer
    {   // b and the block partial sums of (b, b) in one pass; from the zero guess r0 = b is written along
        Prof p(c, VOF_K_RHS, 0);
        k_rhs_norm<<<rgrid(c, np), RBLK, 0, s>>>(frames_dev, frame_stride(c), c->Nj, f.ni, f.nj, c->kb, zero_guess ? c->kr : nullptr,
                                                c->partials, c->pp);
    }
    { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_BNORM><<<np, 64, 0, s>>>(c->sc, c->partials, c->nblk, c->active, P.rtol, P.max_iterations); }
    int nb0 = c->nblk;   // per-pair partial sums of (r0, r0): from the zero guess they are those of (b, b), still in place
    if (zero_guess) {
        HIPCHK(hipMemsetAsync(c->kx, 0, (size_t)np * len * sizeof(double), s));
        rh = c->kb;
    } else {
        if (c->guess_src) {   // warm start from the solution of a neighbouring, already solved pair (cf. OF.py:803-806)
            Prof p(c, VOF_K_VECTOR, 0, 16.0 * len);
            k_gather_guess<<<rgrid(c, np), RBLK, 0, s>>>(c->kx, c->warm_x, c->guess_src, len, f.npts, P.initial_v_x * sx, P.initial_v_y * sx,
                                                        P.initial_remodelling);
        } else {
            Prof p(c, VOF_K_VECTOR, 0); k_fill<<<dim3(c->nblk, np), 256, 0, s>>>(c->kx, f.npts, P.initial_v_x * sx, P.initial_v_y * sx, P.initial_remodelling);
        }
        // r0 = b - A x0 with the partial sums of (r0, r0) and, where the streaming kernel runs, the copy r^ from the same pass
        const bool two = residual_copy_ok(c);
        nb0 = residual_d(c, c->kx, c->kb, c->kr, np, nullptr, 1, two ? c->krh : nullptr);
        if (!two) HIPCHK(hipMemcpyAsync(c->krh, c->kr, (size_t)np * len * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    // (p and v need no initialisation: the first iteration after a (re)start sets p = r without reading either)
    if (!nb0) {   // the operator kernel in use does not fuse the norm
        Prof p(c, VOF_K_REDUCE, 0); k_dot2<<<rgrid(c, np), RBLK, 0, s>>>(c->kr, c->kr, nullptr, nullptr, len, c->partials, nullptr);
        nb0 = c->nblk;
    }
    { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_R0><<<np, 64, 0, s>>>(c->sc, c->partials, nb0, c->active, P.rtol, P.max_iterations); }

    // krylov_method: 0 = BiCGStab only (the reference's 'bcgs'); 1 = GMRES only; 2 = BiCGStab, and restarted GMRES for
    // the pairs that have not met the tolerance after `fallback_after` iterations (or broke down)
    const int bicg_limit = P.krylov_method == 1 ? 0 : (P.krylov_method == 2 ? std::min(P.max_iterations, P.fallback_after) : P.max_iterations);
    int it_total = 0;   // BiCGStab iterations of this batch (all rounds)
    auto bicg_loop = [&](int limit) -> int {
    bool ran_on_r = false;   // this round's first iteration ran the cycle on r instead of writing p = r
    for (int it = 0; it < limit; ++it, ++it_total) {
        HIPCHK(hipMemcpyAsync(c->h_active, c->active, np * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        int nact = 0;
        for (int k = 0; k < np; ++k) nact += c->h_active[k] != 0;
        if (nact == 0) break;
        // vcycle_precision 2 ("auto"): float32 V-cycle vectors for the first iterations, float64 for stragglers
        // (in the slowly converging regimes float32 storage costs iterations)
        if (it_total == AUTO_F64_AFTER) { if (P.vcycle_precision == 2) c->vfloat = false; c->vcoarse32 = false; }
        c->cur_units = nact;
        const int* act = c->active;
        void* vrhs_p = c->vfloat ? (void*)c->b32 : (void*)c->kp;   // V-cycle right-hand sides (V-typed)
        void* vrhs_s = c->vfloat ? (void*)c->b32 : (void*)c->kr;
        const double vsz = c->vfloat ? 4.0 : 8.0;
        // p = r + beta (p - omega v).  The iteration after a (re)start has p = r: the cycle then runs straight on r (unless it
        // needs a float32 copy of its right-hand side) and no p is written - the next iteration finds that p in r^ = r0.
        const bool on_r = it == 0 && !c->vfloat;
        const bool fold_b = fold_b_usable(c);
        if (on_r) vrhs_p = (void*)c->kr;
        else if (fold_b && it > 0) {
            // folded into the cycle's first pass (k_sweep0r, BF = 2).  The new p goes to the buffer of t, which is dead here,
            // and the two trade names (the bands of the pass overlap: no update in place)
            const double* p_old = (it == 1 && ran_on_r) ? rh : c->kp;
            c->bf = S0BSrc{c->kr, c->kv, p_old, c->kt, c->sc, nullptr};
            c->bf_mode = 2;
            std::swap(c->kp, c->kt);
            vrhs_p = (void*)c->kp;
        } else {
            Prof p(c, VOF_K_VECTOR, 0, 8.0 * len * (it == 0 ? 2 : 4) + (c->vfloat ? 4.0 * len : 0.0));
            const double* p_old = (it == 1 && ran_on_r) ? rh : c->kp;
            VDISPATCH(c, (k_update_p<VT><<<rgrid(c, np), RBLK, 0, s>>>(c->kp, p_old, c->kr, c->kv, len, c->sc, act,
                                                                     c->vfloat ? (VT*)c->b32 : (VT*)nullptr, it == 0 ? 1 : 0)));
        }
        if (it == 0) ran_on_r = on_r;
        // y = M p and v = A y with (r^, v): the product comes out of the cycle's last smoothing pass when that path applies
        c->trail_req = S0Trail{c->kv, rh, 0, c->partials};
        c->trail_set = true; c->trail_done = false;
        vcycle(c, &c->ky, vrhs_p, np, act);
        c->trail_set = false;
        if (c->bf_mode) { if (c->err.empty()) c->err = "folded vector update (p): not consumed by the cycle"; c->bf_mode = 0; return -1; }
        int nb1 = c->trail_done ? c->trail_nblk : krylov_apply(c, c->ky, c->kv, np, act, rh, 0);
        if (!nb1) { Prof p(c, VOF_K_REDUCE, 0, 16.0 * len); k_dot2<<<rgrid(c, np), RBLK, 0, s>>>(rh, c->kv, nullptr, nullptr, len, c->partials, act); nb1 = c->nblk; }
        { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_ALPHA><<<np, 64, 0, s>>>(c->sc, c->partials, nb1, c->active, P.rtol, P.max_iterations); }
        if (fold_b) {
            // s = r - alpha v, (s, s), the half-step test and its x += alpha y: in / right after the first pass of the cycle on s
            // (k_sweep0r, BF = 1).  s goes to the buffer of t (dead until the end of this cycle); r's buffer becomes t's
            c->bf = S0BSrc{c->kr, c->kv, nullptr, c->kt, c->sc, c->partials};
            c->bf_mode = 1;
            std::swap(c->kr, c->kt);
            vrhs_s = (void*)c->kr;
        } else {
        { Prof p(c, VOF_K_VECTOR, 0, 8.0 * len * 3 + (c->vfloat ? 4.0 * len : 0.0));   // s = r - alpha v, (s, s)
          VDISPATCH(c, (k_update_s<VT><<<rgrid(c, np), RBLK, 0, s>>>(c->kr, c->kv, len, c->sc, c->partials, act, c->vfloat ? (VT*)c->b32 : (VT*)nullptr))); }
        { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_S><<<np, 64, 0, s>>>(c->sc, c->partials, c->nblk, c->active, P.rtol, P.max_iterations); }
        { Prof p(c, VOF_K_VECTOR, 0);                                  // pairs done at the half step: x += alpha y
          VDISPATCH(c, (k_fix_half<VT><<<dim3(64, np), RBLK, 0, s>>>(c->kx, (const VT*)c->ky, len, c->sc)));
          k_clear_half<<<(np + 255) / 256, 256, 0, s>>>(c->sc, np); }
        }
        // z = M s and t = A z with (t, s) and (t, t)
        c->trail_req = S0Trail{c->kt, c->kr, 1, c->partials};
        c->trail_set = true; c->trail_done = false;
        vcycle(c, &c->kz, vrhs_s, np, act);
        c->trail_set = false;
        if (c->bf_mode) { if (c->err.empty()) c->err = "folded vector update (s): not consumed by the cycle"; c->bf_mode = 0; return -1; }
        int nb2 = c->trail_done ? c->trail_nblk : krylov_apply(c, c->kz, c->kt, np, act, c->kr, 1);
        if (!nb2) { Prof p(c, VOF_K_REDUCE, 0, 16.0 * len); k_dot2<<<rgrid(c, np), RBLK, 0, s>>>(c->kt, c->kr, c->kt, c->kt, len, c->partials, act); nb2 = c->nblk; }
        { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_OMEGA><<<np, 64, 0, s>>>(c->sc, c->partials, nb2, c->active, P.rtol, P.max_iterations); }
        { Prof p(c, VOF_K_VECTOR, 0, 8.0 * len * 6 + 2.0 * vsz * len);   // x += alpha y + omega z; r = s - omega t; (r,r), (r^,r)
          VDISPATCH(c, (k_update_xr<VT><<<rgrid(c, np), RBLK, 0, s>>>(c->kx, (const VT*)c->ky, (const VT*)c->kz, c->kr, c->kt, rh, len, c->sc, c->partials, act))); }
        { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_R><<<np, 64, 0, s>>>(c->sc, c->partials, c->nblk, c->active, P.rtol, P.max_iterations); }
    }
    return 0;
    };
    if (int rc = bicg_loop(bicg_limit)) return rc;
    // independent residual (OF.py:1150-1151): ||b - A x|| recomputed from x for every pair
    // (keep == false: only the norm, no residual vector - it is needed again only if a pair has to be restarted)
    auto independent_residual = [&](bool keep) -> int {
        c->cur_units = np;
        int nb3 = keep ? 0 : residual_d(c, c->kx, c->kb, nullptr, np, nullptr, 1);
        if (!nb3) nb3 = residual_d(c, c->kx, c->kb, c->kt, np, nullptr, 1);
        if (!nb3) { Prof p(c, VOF_K_REDUCE, 0); k_dot2<<<rgrid(c, np), RBLK, 0, s>>>(c->kt, c->kt, nullptr, nullptr, len, c->partials, nullptr); nb3 = c->nblk; }
        { Prof p(c, VOF_K_VECTOR, 0); k_scalar<S_FINAL><<<np, 64, 0, s>>>(c->sc, c->partials, nb3, c->active, P.rtol, P.max_iterations); }
        HIPCHK(hipGetLastError());
        return 0;
    };
    if (int rc = independent_residual(false)) return rc;
    // The stopping rule is evaluated on the independent residual.  BiCGStab tests its recursively updated residual, which
    // drifts from the true one (by rounding; visibly so near the attainable accuracy): pairs it declared converged whose
    // recomputed residual misses the tolerance are restarted from that residual (r = r^ = b - A x, p = v = 0), which a
    // further iteration or two settles.  What is still open afterwards goes to GMRES (krylov_method 2).
    if (bicg_limit > 0) {
        for (int round = 0; round < 3; ++round) {
            { Prof p(c, VOF_K_VECTOR, 0); k_bicg_restart<<<(np + 63) / 64, 64, 0, s>>>(c->sc, c->active, np, P.max_iterations); }
            int nact = count_active(c, np);
            if (nact < 0) { c->err = "stream synchronize failed"; return -2; }
            if (nact == 0) break;
            if (int rc = independent_residual(true)) return rc;   // the restart needs the residual vector itself (rare)
            c->cur_units = nact;
            rh = c->krh;   // (the restarting pairs get their new r^ written; the others are done and no longer read theirs)
            { Prof p(c, VOF_K_VECTOR, 0, 8.0 * len * 3);
              k_restart_vectors<<<rgrid(c, np), RBLK, 0, s>>>(c->kr, rh, c->kt, len, c->active); }
            if (int rc = bicg_loop(std::min(bicg_limit, 8))) return rc;
            if (int rc = independent_residual(false)) return rc;
        }
    }
    if (P.krylov_method != 0) {
        // GMRES takes the pairs BiCGStab left unconverged, and those whose recursively updated residual met the
        // tolerance while the true one does not (the usual drift of BiCGStab at tight tolerances)
        int handed_over = 0;
        if (int rc = gmres_phase(c, np, &handed_over)) return rc;
        if (handed_over)
            if (int rc = independent_residual(false)) return rc;
    }
    // functionals (OF.py:1167-1183) and epilogue (OF.py:1159-1166, 1189-1191)
    { Prof p(c, VOF_K_FINALIZE, 0);   // one pass over the solution: outputs + functionals
      k_finalize_functionals<<<rgrid(c, np), RBLK, 0, s>>>(frames_dev, frame_stride(c), f.ni, f.nj, P.speed_alpha, P.remodelling_alpha,
                                                           P.reference_quirks, c->kx, P.delta_x / P.delta_t, vx, vy, gm, speed,
                                                           c->partials, c->pp);
      k_sum3<<<np, 64, 0, s>>>(c->partials, c->nblk, c->func3); }
    HIPCHK(hipGetLastError());
    if (stats) {
        HIPCHK(hipMemcpyAsync(c->h_sc, c->sc, np * sizeof(PairScalars), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(c->h_func3, c->func3, np * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(c->ev_batch[1], s));
        HIPCHK(hipStreamSynchronize(s));
        float batch_ms = 0.f;
        HIPCHK(hipEventElapsedTime(&batch_ms, c->ev_batch[0], c->ev_batch[1]));
        for (int k = 0; k < np; ++k) {
            const PairScalars& q = c->h_sc[k];
            stats[k].iterations = q.iterations;
            stats[k].relative_residual = q.bnorm2 > 0 ? std::sqrt(q.rnorm2 / q.bnorm2) : 0.0;
            stats[k].converged = (q.rnorm2 <= q.tol2) ? 1 : 0;   // the rule on the independent residual (NaN: 0)
            stats[k].L1_functional = c->h_func3[3 * k];
            stats[k].speed_functional = c->h_func3[3 * k + 1];
            stats[k].remodelling_functional = c->h_func3[3 * k + 2];
            stats[k].batch_ms = batch_ms;
            stats[k].batch_pairs = np;
            stats[k].reserved = 0;
        }
    }
    return 0;
}

int check_params(vof_ctx* c, const vof_params* p) {
    if (!p) { c->err = "params is NULL"; return -1; }
    if (p->struct_size != sizeof(vof_params) || p->abi_version != VOF_VERSION) {
        char buf[256];
        snprintf(buf, sizeof buf, "vof_params ABI mismatch: caller has struct_size %u / version %u, library has %zu / %d "
                 "(fill the struct with vof_default_params(p, sizeof *p) of the matching include/vof.h)",
                 p->struct_size, p->abi_version, sizeof(vof_params), VOF_VERSION);
        c->err = buf;
        return -4;
    }
    if (!(p->delta_x != 0.0) || !(p->delta_t != 0.0)) { c->err = "delta_x and delta_t must be non-zero"; return -1; }
    if (p->nu_pre < 0 || p->nu_post < 0 || p->nu_pre + p->nu_post == 0) { c->err = "nu_pre + nu_post must be > 0"; return -1; }
    if (p->nu_pre_coarse < 0 || p->nu_post_coarse < 0) { c->err = "nu_*_coarse must be >= 0"; return -1; }
    if (p->w_cycle_level < -1 || p->w_cycle_level > 15) { c->err = "w_cycle_level must be -1 or a level index"; return -1; }
    if (p->w_cycle_visits < 0 || p->w_cycle_visits > 8) { c->err = "w_cycle_visits must be in [0, 8]"; return -1; }
    if (!(p->rtol > 0.0)) { c->err = "rtol must be > 0"; return -1; }
    if (p->coarse_precision < 0 || p->coarse_precision > 3) { c->err = "coarse_precision must be 0, 1, 2 or 3"; return -1; }
    if (p->vcycle_precision < 0 || p->vcycle_precision > 3) { c->err = "vcycle_precision must be 0, 1, 2 or 3"; return -1; }
    if (p->krylov_method < 0 || p->krylov_method > 2) { c->err = "krylov_method must be 0, 1 or 2"; return -1; }
    if (p->gmres_restart < 0 || p->gmres_restart > GM_MAXM) { c->err = "gmres_restart must be in [0, 128]"; return -1; }
    if (p->fallback_after < 0) { c->err = "fallback_after must be >= 0"; return -1; }
    if (p->warm_start_stride < 0) { c->err = "warm_start_stride must be >= 0"; return -1; }
    if (p->preconditioner < 0 || p->preconditioner > 2) { c->err = "preconditioner must be 0, 1 or 2"; return -1; }
    c->prm = *p;
    // float32 V-cycle vectors need the fused sweeps and a multi-level hierarchy
    c->vfloat = (p->vcycle_precision == 1 || p->vcycle_precision == 2) && c->fused && c->L.size() > 1;
    c->vcoarse32 = false;   // set per batch (solve_batch); the debug entry points run every level in one storage type
    return ensure_storage(c);
}

}  // namespace

// ============================================================================================ C ABI
extern "C" {

int vof_version(void) { return VOF_VERSION; }

size_t vof_params_size(void) { return sizeof(vof_params); }

int vof_default_params(vof_params* p, size_t struct_size) {
    if (!p || struct_size != sizeof(vof_params)) return -1;   // a binding built against another layout: write nothing
    memset(p, 0, sizeof *p);
    p->struct_size = (uint32_t)sizeof(vof_params);
    p->abi_version = VOF_VERSION;
    p->speed_alpha = 1.0;          // OF.py:718
    p->remodelling_alpha = 1000.0; // OF.py:719
    p->delta_x = 1.0;
    p->delta_t = 1.0;
    p->rtol = 1e-6;                // OF.py:1120
    p->max_iterations = 1000;      // OF.py:1120
    p->nu_pre = 2;                 // (2,2) sweeps on level 0 ...
    p->nu_post = 2;
    p->nu_pre_coarse = 1;          // ... (1,1) on the stored-stencil levels (measured best time to solution)
    p->nu_post_coarse = 1;
    p->w_cycle_level = 1;          // level 1 visits level 2 several times per cycle (one-level W-cycle) ...
    p->w_cycle_visits = 3;         // ... three times: 5.35 -> 3.4 BiCGStab iterations on the benchmark workload
    p->reference_quirks = 1;
    p->coarse_precision = 3;       // Galerkin stencils (preconditioner only): 8-bit float off-diagonal blocks in units of a power
                                   // of two per block position, float32 diagonal block that keeps the block row sums: 120 B per
                                   // point, +1 % iterations against float32 / bfloat16 (2) on the benchmark, same counts in the
                                   // other regimes (scripts/gpu_regimes3.py)
    p->vcycle_precision = 3;       // float64 V-cycle vectors on level 0, float32 below (0: float64 everywhere, 1: float32 everywhere)
    p->krylov_method = 2;          // BiCGStab (the reference's 'bcgs'); stragglers are finished by restarted GMRES
    p->gmres_restart = 100;        // capped by the free device memory: (restart + 1) vectors per pair in flight
    p->warm_start_stride = 3;      // vof_solve_stack_dev: every 3rd pair first, the others start from their solved neighbour
    p->fallback_after = 25;        // BiCGStab iterations before the fallback (the benchmark regimes need 3-17)
    p->preconditioner = 2;         // multigrid cycle; pairs it leaves unconverged are re-solved with the direct preconditioner if it fits
    return 0;
}

const char* vof_last_error(const vof_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
size_t vof_workspace_bytes(const vof_ctx* ctx) { return ctx ? ctx->bytes : 0; }
int vof_num_levels(const vof_ctx* ctx) { return ctx ? (int)ctx->L.size() : 0; }

size_t vof_query_workspace(int n_i, int n_j, int B) { return vof_query_workspace_for(n_i, n_j, B, DEFAULT_COARSE_PRECISION, 3); }

size_t vof_query_workspace_for(int n_i, int n_j, int B, int coarse_precision, int vcycle_precision) {
    if (n_i < 4 || n_j < 4 || B < 1 || coarse_precision < 0 || coarse_precision > 3) return 0;
    size_t ni = n_i - 2, nj = n_j - 2, total = 0, b = (size_t)B;
    std::vector<std::pair<size_t, size_t>> lv{{ni, nj}};
    while (std::max(lv.back().first, lv.back().second) > (size_t)coarsest_max())
        lv.push_back({(lv.back().first + 1) / 2, (lv.back().second + 1) / 2});
    size_t words = 0;   // 32-bit words of stencil storage
    total += (9 + ((vcycle_precision == 1 || vcycle_precision == 2) && lv.size() > 1 ? 1 : 0)) * b * 3 * ni * nj;
    for (size_t l = 0; l < lv.size(); ++l) {
        size_t npts = lv[l].first * lv[l].second;
        if (l + 1 < lv.size()) total += (l > 0 ? 2 : 1) * b * 3 * npts;
        if (l > 0) total += 2 * b * 3 * npts;
        const size_t plane = CLay((int)lv[l].first, (int)lv[l].second).plane;
        if (lv.size() == 1) total += b * 81 * plane;
        else if (l > 0) words += b * (size_t)(coef_bytes_per_point(coarse_precision) / 4) * plane;
    }
    size_t nd = 3 * lv.back().first * lv.back().second;
    total += b * nd * 2 * nd + b * nd * nd;
    return total * sizeof(double) + words * 4 + (size_t)B * 3 * 256 * sizeof(double) + 4096;
}

int vof_device_memory(int device_id, size_t* free_bytes, size_t* total_bytes) {
    size_t f = 0, t = 0;
    if (hipSetDevice(device_id) != hipSuccess) return -1;
    if (hipMemGetInfo(&f, &t) != hipSuccess) return -2;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

const char* vof_kernel_name(int k) {
    static const char* names[VOF_K_COUNT] = {"rhs", "apply0", "gs0", "gs", "residual", "restrict", "prolong",
                                             "galerkin0", "galerkin", "coarse_setup", "coarse_solve", "vector",
                                             "reduce", "finalize", "functionals", "coarse_tail"};
    return (k >= 0 && k < VOF_K_COUNT) ? names[k] : "?";
}

void vof_destroy(vof_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) {
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && (c->dbg_sync || c->dbg_canary))
            fprintf(stderr, "vof_destroy: the context's stream ends with an error: %s\n", hipGetErrorString(e));
    }
    if (c->dbg_canary) {
        std::string rep;
        if (dbg_check_canaries(c, &rep) != 0) { fprintf(stderr, "vof_destroy: VOF_DEBUG_CANARY: %s\n", rep.c_str()); fflush(stderr); }
    }
    if (c->dbg_fd >= 0) close(c->dbg_fd);
    prof_collect(c);
    for (auto e : c->free_events) hipEventDestroy(e);
    for (const auto& a : c->allocs) hipFree(a.raw);
    if (c->h_active) hipHostFree(c->h_active);
    if (c->h_sc) hipHostFree(c->h_sc);
    if (c->h_func3) hipHostFree(c->h_func3);
    if (c->h_bounce) hipHostFree(c->h_bounce);
    for (int i = 0; i < 2; ++i) if (c->ev_batch[i]) hipEventDestroy(c->ev_batch[i]);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_solved[i]) hipEventDestroy(c->ev_solved[i]);
        if (c->ev_copied[i]) hipEventDestroy(c->ev_copied[i]);
        if (c->ev_uploaded[i]) hipEventDestroy(c->ev_uploaded[i]);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static int create_impl(vof_ctx* c, int device_id, int n_i, int n_j, int B, void* stream) {
    if (n_i < 4 || n_j < 4) { c->err = "image must be at least 4x4 (the mirror boundary rows need N >= 4)"; return -1; }
    if (B < 1) { c->err = "max_pairs_in_flight must be >= 1"; return -1; }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) { c->err = "no such HIP device"; return -1; }
    HIPCHK(hipSetDevice(device_id));
    c->device = device_id;
    c->Ni = n_i; c->Nj = n_j; c->B = B;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIPCHK(hipStreamCreate(&c->stream)); c->own_stream = true; }
    memset(c->prof_ms, 0, sizeof c->prof_ms);
    memset(c->prof_n, 0, sizeof c->prof_n);
    memset(c->prof_units, 0, sizeof c->prof_units);
    memset(c->prof_bytes, 0, sizeof c->prof_bytes);
    memset(c->prof_moved, 0, sizeof c->prof_moved);
    vof_default_params(&c->prm, sizeof c->prm);
    if (const char* e = getenv("VOF_DEBUG_SYNC")) c->dbg_sync = atoi(e);
    if (const char* e = getenv("VOF_DEBUG_CANARY")) c->dbg_canary = e[0] != '0';
    if (const char* e = getenv("VOF_DEBUG_ALLOC_LOG")) c->dbg_alloc_log = e[0] != '0';
    if (const char* e = getenv("VOF_DEBUG_POISON")) c->dbg_poison = e[0] != '0';
    if (c->dbg_sync)
        if (const char* e = getenv("VOF_DEBUG_SYNC_FILE")) c->dbg_fd = open(e, O_WRONLY | O_CREAT, 0644);
    if (const char* e = getenv("VOF_PRECOND_QUIRKS")) { c->pq_hier = e[0] != '0'; c->pq_smooth = e[0] && e[1] != '0'; }
    if (!c->pq_smooth) c->trail_enabled = false;   // the fused Krylov product shares the smoother's operator
    if (const char* e = getenv("VOF_SWEEP_GEO")) {   // experiment switch: "AA", "AB" (default), "BA", "BB" = fine,stored
        c->geo_b_fine = e[0] == 'B';
        c->geo_b_stored = e[0] && e[1] == 'B';
    }
    if (const char* e = getenv("VOF_STREAM_APPLY")) c->stream_apply = e[0] != '0';
    if (const char* e = getenv("VOF_FUSE_RESTRICT")) c->fuse_restrict = e[0] != '0';
    if (const char* e = getenv("VOF_FUSE_PROLONG")) c->fuse_prolong = e[0] != '0';
    if (const char* e = getenv("VOF_FUSE_RESU")) c->fuse_resu = e[0] != '0';
    if (const char* e = getenv("VOF_SWEEP_ST")) c->sweep_st = e[0] != '0';
    if (const char* e = getenv("VOF_FOLD_STORED")) c->fold_stored = e[0] != '0';
    if (const char* e = getenv("VOF_SKIP_COLOUR0")) c->skip_colour0 = e[0] != '0';
    if (const char* e = getenv("VOF_COARSE_TAIL")) c->tail_enabled = e[0] != '0';
    if (const char* e = getenv("VOF_SWEEP0")) c->sweep0 = e[0] != '0';
    if (const char* e = getenv("VOF_FUSE_APPLY")) c->trail_enabled = e[0] != '0';
    if (const char* e = getenv("VOF_SWEEP0R")) c->sweep0r = e[0] != '0';
    if (const char* e = getenv("VOF_SWEEP0P")) c->sweep0p = e[0] != '0';
    if (const char* e = getenv("VOF_SWEEP0R_MIN_BLOCKS")) c->sweep0r_min_blocks = atol(e);
    if (const char* e = getenv("VOF_FUSE_B")) c->fuse_b = e[0] != '0';
    if (const char* e = getenv("VOF_FUSE_RR")) c->fuse_rr = e[0] != '0';
    if (const char* e = getenv("VOF_SWEEP0M")) { c->sweep0m = e[0] != '0'; c->sweep0m_pairs = e[0] != '0' && e[0] != '1'; }
    // level shapes
    Level l0; l0.ni = n_i - 2; l0.nj = n_j - 2; l0.npts = (size_t)l0.ni * l0.nj;
    c->L.push_back(l0);
    while (std::max(c->L.back().ni, c->L.back().nj) > coarsest_max()) {
        Level k; k.ni = (c->L.back().ni + 1) / 2; k.nj = (c->L.back().nj + 1) / 2; k.npts = (size_t)k.ni * k.nj;
        c->L.push_back(k);
    }
    int nl = (int)c->L.size();
    if (nl > 16) { c->err = "too many levels"; return -1; }
    size_t len0 = 3 * l0.npts;
    // (b32, the float32 copy of the cycle's right-hand side, exists only while vcycle_precision 1 / 2 is in use: ensure_storage)
    for (double** v : {&c->kx, &c->kb, &c->kr, &c->krh, &c->kp, &c->kv, &c->kt, &c->ky, &c->kz})
        if (int rc = dev_alloc(c, v, (size_t)B * len0)) return rc;
    for (int l = 0; l < nl; ++l) {
        Level& lv = c->L[l];
        auto valloc = [&](void** q) { double* t = nullptr; int rc = dev_alloc(c, &t, (size_t)B * 3 * lv.npts); *q = t; return rc; };
        // residual scratch: level 0 needs it only with the stand-alone residual + restriction kernels (experiment switches)
        if (l + 1 < nl && (l > 0 || !(c->stream_apply && c->fuse_restrict))) if (int rc = valloc(&lv.r)) return rc;
        if (l + 1 < nl) if (int rc = valloc(&lv.x2)) return rc;
        if (l > 0) {
            if (int rc = valloc(&lv.x)) return rc;
            if (int rc = valloc(&lv.b)) return rc;
        }
        if (nl == 1) {   // a one-level grid keeps its fine stencil as float64
            double* C = nullptr;
            if (int rc = dev_alloc(c, &C, (size_t)B * 81 * CLay(lv.ni, lv.nj).plane)) return rc;
            lv.C = C;
        }
    }
    // stored stencils of the levels >= 1: sized for the default format; a call that asks for a wider one re-allocates them
    if (int rc = ensure_stencil_storage(c, vof_params_default_coarse_precision())) return rc;
    c->nd = 3 * (int)c->L.back().npts;
    {   // coarse tail: the deepest run of levels that fit one workgroup (and its LDS)
        int first = -1;
        for (int l = 1; l < nl; ++l) {
            const CLay lay(c->L[l].ni, c->L[l].nj);
            if (c->L[l].npts <= (size_t)TAIL_MAX_PTS && lay.sub <= (size_t)TAIL_MAX_SUB) { first = l; break; }
        }
        if (first >= 0) first = std::max(first, nl - TAIL_MAX_LEVELS);
        if (first >= 1 && first < nl - 1) {
            memset(&c->tail, 0, sizeof c->tail);
            int off = 0;
            for (int l = first; l < nl; ++l) {
                TailLevel& t = c->tail.L[l - first];
                const CLay lay(c->L[l].ni, c->L[l].nj);
                t.ni = c->L[l].ni; t.nj = c->L[l].nj; t.hj = lay.hj; t.sub = (int)lay.sub; t.plane = lay.plane;
                t.xo = off; off += 3 * (t.ni + 2) * (t.nj + 2);
                t.bo = off; off += 3 * t.ni * t.nj;
            }
            c->tail.ro = off; off += 3 * (int)c->L[first].npts;
            c->tail.nl = nl - first;
            c->tail.nd = c->nd;
            c->tail.n_ops = 0;
            c->tail_lds = (size_t)off * sizeof(double);
            if (c->tail_lds <= (size_t)150 * 1024) {
                c->tail_first = first;
                const int lds = (int)c->tail_lds;
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<CoefB16, double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<CoefB16, float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<CoefF8, double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<CoefF8, float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<float, double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<double, double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                HIPCHK(hipFuncSetAttribute((const void*)k_tail_cycle<double, float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            }
        }
    }
    if (int rc = dev_alloc(c, &c->W, (size_t)B * c->nd * 2 * c->nd)) return rc;
    if (int rc = dev_alloc(c, &c->invT, (size_t)B * c->nd * c->nd)) return rc;
    {   // the level-0 smoother passes with the trailing stage need more than the default 64 KB of dynamic LDS
        const int lds = 18 * s0_row_bytes(8) + 9 * (S0_W / 2 + 2) * 8;
        HIPCHK(hipFuncSetAttribute((const void*)k_sweep0m<2, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_sweep0m<2, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_sweep0m<1, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIPCHK(hipFuncSetAttribute((const void*)k_sweep0m<1, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    c->nblk = (int)std::min<size_t>(256, std::max<size_t>(1, (len0 + 4 * RBLK - 1) / (4 * RBLK)));
    {
        int nblk_apply = ((l0.nj + AP_OUT - 1) / AP_OUT) * ((l0.ni + 31) / 32 + 1);   // smallest band height: 32 rows
        nblk_apply = std::max(nblk_apply, ((l0.nj + 99) / 100) * ((l0.ni + 1 + 31) / 32 + 1));   // k_sweep0m's trailing stage (strips >= 108 columns)
        if (int rc = dev_alloc(c, &c->partials, (size_t)B * 3 * std::max(c->nblk, nblk_apply))) return rc;
    }
    if (int rc = dev_alloc(c, &c->sc, (size_t)B)) return rc;
    if (int rc = dev_alloc(c, &c->active, (size_t)B)) return rc;
    if (int rc = dev_alloc(c, &c->func3, (size_t)B * 3)) return rc;
    HIPCHK(hipHostMalloc((void**)&c->h_active, B * sizeof(int)));
    HIPCHK(hipHostMalloc((void**)&c->h_sc, B * sizeof(PairScalars)));
    HIPCHK(hipHostMalloc((void**)&c->h_func3, B * 3 * sizeof(double)));
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreate(&c->ev_batch[i]));
    return 0;
}

int vof_create(vof_ctx** out, int device_id, int n_i, int n_j, int max_pairs_in_flight, void* stream) {
    if (!out) { g_create_error = "out is NULL"; return -1; }
    *out = nullptr;
    vof_ctx* c = new (std::nothrow) vof_ctx();
    if (!c) { g_create_error = "out of host memory"; return -1; }
    int rc = create_impl(c, device_id, n_i, n_j, max_pairs_in_flight, stream);
    if (rc) {
        g_create_error = c->err;
        vof_destroy(c);
        return rc;
    }
    *out = c;
    return 0;
}

// Two-phase solve of a stack with warm starts (the reference warm-starts pair k from pair k-1, OF.py:803-806, which
// serialises the pairs; here every stride-th pair is solved first from the constant initial fields, then all the others
// start from the solution of their nearest solved neighbour).  Pairs are addressed through the PairParam table
// (frame / output slot), so both phases are ordinary batches.
static int solve_stack_two_phase(vof_ctx* c, const double* movie, int P, double* v_x, double* v_y, double* remodelling,
                                 double* speed, vof_pair_stats* stats, int stride) {
    const vof_params prm = c->prm;
    const size_t len = 3 * c->L[0].npts;
    const int B = c->B;
    std::vector<int> first, rest;
    for (int k = 0; k < P; ++k) (k % stride == 0 ? first : rest).push_back(k);
    const int n1 = (int)first.size();
    if (!c->pp_buf) { if (int rc = dev_alloc(c, &c->pp_buf, (size_t)B)) return rc; }
    if (!c->warm_src) { if (int rc = dev_alloc(c, &c->warm_src, (size_t)B)) return rc; }
    if (c->warm_cap < (size_t)n1 * len) {
        if (int rc = dev_free(c, c->warm_x)) return rc;   // (round 2 kept every outgrown buffer until vof_destroy)
        c->warm_x = nullptr; c->warm_cap = 0;
        if (int rc = dev_alloc(c, &c->warm_x, (size_t)n1 * len)) return rc;
        c->warm_cap = (size_t)n1 * len;
    }
    std::vector<PairParam> hp((size_t)B);
    std::vector<int> hsrc((size_t)B);
    std::vector<vof_pair_stats> st((size_t)B);
    std::vector<char> usable((size_t)n1, 0);   // phase-1 solutions that may seed a neighbour: converged and finite
    auto run = [&](const std::vector<int>& list, bool phase2) -> int {
        for (size_t o = 0; o < list.size(); o += (size_t)B) {
            const int np = (int)std::min<size_t>((size_t)B, list.size() - o);
            for (int i = 0; i < np; ++i) {
                const int k = list[o + i];
                hp[i] = PairParam{prm.speed_alpha, prm.remodelling_alpha, k, k};
                const int src = std::min((k + stride / 2) / stride, n1 - 1);
                hsrc[i] = usable[src] ? src : -1;   // -1: constant initial fields (a failed pair must not poison its neighbours)
            }
            HIPCHK(hipStreamSynchronize(c->stream));   // the host tables are re-used
            HIPCHK(hipMemcpyAsync(c->pp_buf, hp.data(), (size_t)np * sizeof(PairParam), hipMemcpyHostToDevice, c->stream));
            if (phase2) HIPCHK(hipMemcpyAsync(c->warm_src, hsrc.data(), (size_t)np * sizeof(int), hipMemcpyHostToDevice, c->stream));
            c->pp = c->pp_buf;
            c->guess_src = phase2 ? c->warm_src : nullptr;
            int rc = solve_batch(c, movie, np, v_x, v_y, remodelling, speed, st.data());
            c->pp = nullptr;
            c->guess_src = nullptr;
            if (rc) return rc;
            if (!phase2) {
                HIPCHK(hipMemcpyAsync(c->warm_x + o * len, c->kx, (size_t)np * len * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                for (int i = 0; i < np; ++i) usable[o + i] = st[i].converged && std::isfinite(st[i].relative_residual);
            }
            if (stats)
                for (int i = 0; i < np; ++i) stats[list[o + i]] = st[i];
        }
        return 0;
    };
    if (int rc = run(first, false)) return rc;
    if (int rc = run(rest, true)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Solve the listed pairs (frame index / output slot relative to `frames`, v_x ...) with the direct preconditioner, in batches
// of what its buffers hold.  stats: per list entry.
static int direct_solve_list(vof_ctx* c, const double* frames, const std::vector<PairParam>& items, double* v_x, double* v_y,
                             double* remodelling, double* speed, vof_pair_stats* stats) {
    if (items.empty()) return 0;
    const int cap = direct_capacity(c, std::min<int>(c->B, (int)items.size()));
    if (cap < 1) { c->err = "the direct preconditioner does not fit into the free device memory (3 n_j x 3 n_j doubles per image row and pair)"; return -3; }
    if (int rc = direct_alloc(c, cap)) return rc;
    if (!c->pp_buf) { if (int rc = dev_alloc(c, &c->pp_buf, (size_t)c->B)) return rc; }
    std::vector<vof_pair_stats> st((size_t)c->dir_cap);
    for (size_t o = 0; o < items.size(); o += (size_t)c->dir_cap) {
        const int np = (int)std::min<size_t>((size_t)c->dir_cap, items.size() - o);
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpyAsync(c->pp_buf, items.data() + o, (size_t)np * sizeof(PairParam), hipMemcpyHostToDevice, c->stream));
        c->pp = c->pp_buf;
        c->direct_on = true;
        int rc = solve_batch(c, frames, np, v_x, v_y, remodelling, speed, st.data());
        c->direct_on = false;
        c->pp = nullptr;
        if (rc) return rc;
        if (stats)
            for (int i = 0; i < np; ++i) stats[o + i] = st[i];
    }
    return 0;
}

// Solve pairs 0 .. P-1 of a device-resident range of frames (P <= any size): two-phase warm start when it pays, plain
// batches otherwise.  Outputs are indexed by the pair's position in the range.
// preconditioner 1: every pair with the direct preconditioner; 2 (default): the multigrid cycle, and the pairs it leaves
// unconverged (the grad-div dominated regimes) once more with the direct preconditioner if that fits.
static int solve_range_dev(vof_ctx* c, const double* frames, int P, double* v_x, double* v_y, double* remodelling,
                           double* speed, vof_pair_stats* stats) {
    const vof_params prm = c->prm;
    const int stride = prm.warm_start_stride;
    const size_t fs = frame_stride(c);
    if (prm.preconditioner == 1 && c->L.size() > 1) {   // (a one-level grid is solved by its dense inverse anyway)
        std::vector<PairParam> items((size_t)P);
        for (int k = 0; k < P; ++k) items[k] = PairParam{prm.speed_alpha, prm.remodelling_alpha, k, k};
        return direct_solve_list(c, frames, items, v_x, v_y, remodelling, speed, stats);
    }
    std::vector<vof_pair_stats> local;
    if (!stats && prm.preconditioner == 2) { local.resize((size_t)P); stats = local.data(); }   // the fallback needs the flags
    // two phases double the latency-bound part of a solve (set-up, small coarse levels): worth it once the first phase
    // alone keeps the chip busy (>= 16 Mpixel of frame pairs; measured: 128^2 x 8 loses 45 %, 512^2 x 64 is neutral,
    // 1024^2 x 129 gains 22 %)
    if (stride > 1 && P >= 2 * stride && (double)(P / stride) * (double)c->Ni * (double)c->Nj >= 16e6) {
        if (int rc = solve_stack_two_phase(c, frames, P, v_x, v_y, remodelling, speed, stats, stride)) return rc;
    } else {
        for (int k0 = 0; k0 < P; k0 += c->B) {
            int np = std::min(c->B, P - k0);
            int rc = solve_batch(c, frames + (size_t)k0 * fs, np, v_x + (size_t)k0 * fs, v_y + (size_t)k0 * fs,
                                 remodelling + (size_t)k0 * fs, speed ? speed + (size_t)k0 * fs : nullptr,
                                 stats ? stats + k0 : nullptr);
            if (rc) return rc;
        }
    }
    if (prm.preconditioner == 2 && stats) {
        std::vector<PairParam> items;
        std::vector<int> which;
        for (int k = 0; k < P; ++k)
            if (!stats[k].converged && std::isfinite(stats[k].relative_residual)) {   // a NaN frame stays a reported failure
                items.push_back(PairParam{prm.speed_alpha, prm.remodelling_alpha, k, k});
                which.push_back(k);
            }
        if (!items.empty() && direct_ok_for_fallback(c)) {
            std::vector<vof_pair_stats> st(items.size());
            if (int rc = direct_solve_list(c, frames, items, v_x, v_y, remodelling, speed, st.data())) {
                if (rc != -3) return rc;     // -3: no room / no rocSOLVER: keep the reported non-convergence
                c->err.clear();
            } else {
                for (size_t i = 0; i < which.size(); ++i) {
                    st[i].iterations += stats[which[i]].iterations;   // Krylov steps of both attempts
                    st[i].batch_ms += stats[which[i]].batch_ms / std::max(1, stats[which[i]].batch_pairs) * st[i].batch_pairs;   // and their time (per-pair share kept)
                    stats[which[i]] = st[i];
                }
            }
        }
    }
    return 0;
}

int vof_solve_stack_dev(vof_ctx* c, const double* movie, int n_frames, const vof_params* p, double* v_x, double* v_y,
                        double* remodelling, double* speed, vof_pair_stats* stats) {
    if (!c) return -1;
    if (!movie || !v_x || !v_y || !remodelling) { c->err = "NULL array pointer"; return -1; }
    if (n_frames < 2) { c->err = "need at least two frames"; return -1; }
    if (int rc = check_params(c, p)) return rc;
    HIPCHK(hipSetDevice(c->device));
    if (int rc = solve_range_dev(c, movie, n_frames - 1, v_x, v_y, remodelling, speed, stats)) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Device staging of the host API: frames of one batch (+1) and TWO sets of output buffers, so that the device-to-host
// copies of batch k (copy stream) overlap the solve of batch k+1 (main stream).
static int ensure_staging(vof_ctx* c, bool double_buffer) {
    const size_t fs = frame_stride(c);
    if (!c->st_movie) {
        if (int rc = dev_alloc(c, &c->st_movie, (size_t)(c->B + 1) * fs)) return rc;
        for (int i = 0; i < 4; ++i)
            if (int rc = dev_alloc(c, &c->st_out[i], (size_t)c->B * fs)) return rc;
    }
    if (double_buffer && !c->st_out2[0]) {
        for (int i = 0; i < 4; ++i)
            if (int rc = dev_alloc(c, &c->st_out2[i], (size_t)c->B * fs)) return rc;
        if (int rc = dev_alloc(c, &c->st_movie2, (size_t)(c->B + 1) * fs)) return rc;
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipEventCreateWithFlags(&c->ev_solved[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_uploaded[i], hipEventDisableTiming));
        }
    }
    return 0;
}

int vof_solve_stack_host(vof_ctx* c, const double* movie, int n_frames, const vof_params* p, double* v_x,
                         double* v_y, double* remodelling, double* speed, vof_pair_stats* stats) {
    if (!c) return -1;
    if (!movie || !v_x || !v_y || !remodelling) { c->err = "NULL array pointer"; return -1; }
    if (n_frames < 2) { c->err = "need at least two frames"; return -1; }
    if (int rc = check_params(c, p)) return rc;
    HIPCHK(hipSetDevice(c->device));
    const size_t fs = frame_stride(c);
    const int P = n_frames - 1;
    const bool multi = P > c->B;                       // more than one batch: overlap the copies with the solves
    if (int rc = ensure_staging(c, multi)) return rc;
    // Batch schedule: full batches, and the remainder split so that the LAST batch is small - its device-to-host copies
    // are the only ones that cannot hide under a solve.
    struct Batch { int k0, np; };
    std::vector<Batch> batches;
    for (int k0 = 0; k0 < P;) {
        int np = std::min(c->B, P - k0);
        const int rest = P - k0;
        if (multi && rest <= c->B && rest > 48) np = rest - std::max(16, rest / 4);   // e.g. 127 -> 96 + 31
        batches.push_back({k0, np});
        k0 += np;
    }
    const int nb = (int)batches.size();
    // Pageable host-to-device copies run at ~2 GB/s on this platform, pinned ones at ~55 GB/s: the caller's movie is
    // pinned in place for the duration of the call (0.04 s/GB) - the frames of the first batch right away, the rest by a
    // helper thread while the first batch is solved.  The split point is page aligned so that the two registrations do
    // not share a page.  Unpinned parts fall back to pageable copies.
    const char* mbase = (const char*)movie;
    const size_t movie_bytes = (size_t)n_frames * fs * sizeof(double);
    size_t split = movie_bytes;
    if (nb > 1) {
        size_t want = (size_t)(batches[0].np + 1) * fs * sizeof(double);
        size_t addr = ((size_t)(uintptr_t)mbase + want + 4095) & ~(size_t)4095;
        split = std::min(movie_bytes, addr - (size_t)(uintptr_t)mbase);
    }
    const bool htrace = getenv("VOF_TRACE_HOST") != nullptr;
    const auto ht0 = std::chrono::steady_clock::now();
    auto hmark = [&](const char* what) {
        if (htrace) fprintf(stderr, "[vof host] %8.1f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ht0).count(), what);
    };
    const bool pinned_a = hipHostRegister((void*)mbase, split, hipHostRegisterDefault) == hipSuccess;
    if (!pinned_a) (void)hipGetLastError();
    hmark("first movie part pinned");
    bool pinned_b = false;
    // Freshly allocated output arrays (np.empty) are not resident yet: first-touch page faults would serialise with
    // the device-to-host copies (0.6 s for 8 GB).  Helper threads prepare them while the GPU solves the first batch
    // (the arrays are outputs: every byte is overwritten below): they are pinned in place (which faults them in), so
    // that the copies run at the pinned rate and asynchronously; arrays below 1 MB are only touched.
    double* outs[4] = {v_x, v_y, remodelling, speed};
    const size_t out_bytes = (size_t)P * fs * sizeof(double);
    std::vector<std::thread> helpers;
    std::thread movie_helper;
    if (split < movie_bytes)
        movie_helper = std::thread([&pinned_b, mbase, split, movie_bytes, dev = c->device]() {
            if (hipSetDevice(dev) == hipSuccess &&
                hipHostRegister((void*)(mbase + split), movie_bytes - split, hipHostRegisterDefault) == hipSuccess) pinned_b = true;
        });
    // Output regions: array i is cut at the batch boundaries (moved up to the next page so that no two registrations share
    // a page); region (bi, i) is what batch bi's copy of array i writes, apart from the < 4 KB before its first page, which
    // belongs to region bi - 1.  A pool of helper threads prepares the regions IN BATCH ORDER while the GPU solves: a region
    // is pinned in place (which faults its pages in; 8.6 GB of fresh np.empty pages cost ~0.4 s of first-touch faults -
    // round 2 pinned each array whole and the first batch's copies waited for all of it), arrays below 1 MB are only
    // touched.  The copies of batch bi wait for batch bi's regions alone.
    const bool pin_outputs = out_bytes >= ((size_t)1 << 20);   // pageable device-to-host copies can drop to ~2 GB/s
    std::vector<size_t> bound((size_t)nb + 1);                 // byte offsets of the region boundaries inside an output array
    struct Region { char* ptr; size_t bytes; bool pinned; };
    std::vector<Region> regions((size_t)nb * 4, Region{nullptr, 0, false});
    std::vector<std::atomic<int>> batch_ready((size_t)nb);
    for (auto& a : batch_ready) a.store(0);
    int n_outs = 0;
    for (int i = 0; i < 4; ++i) if (outs[i]) ++n_outs;
    for (int i = 0; i < 4; ++i) {
        if (!outs[i]) continue;
        const uintptr_t base = (uintptr_t)outs[i];
        for (int bi = 0; bi <= nb; ++bi) {
            size_t o = bi == nb ? out_bytes : (size_t)batches[bi].k0 * fs * sizeof(double);
            if (bi > 0 && bi < nb) o = std::min(out_bytes, (size_t)(((base + o + 4095) & ~(uintptr_t)4095) - base));
            bound[(size_t)bi] = o;   // (the same for every array only if their bases share the page offset: kept per array below)
            if (bi > 0) regions[(size_t)(bi - 1) * 4 + i].bytes = o;   // provisional: end offset
        }
        size_t prev = 0;
        for (int bi = 0; bi < nb; ++bi) {
            Region& r = regions[(size_t)bi * 4 + i];
            const size_t end = r.bytes;
            r.ptr = (char*)outs[i] + prev;
            r.bytes = end - prev;
            prev = end;
        }
    }
    std::atomic<int> next_task{0};
    const int n_tasks = nb * 4;
    // The helpers only TOUCH the pages (plain stores: no runtime call, no lock shared with the launching thread - eight
    // threads inside hipHostRegister slowed the solver's kernel launches threefold); registering resident pages afterwards
    // takes ~2 ms per GB and is done by the calling thread just before the batch's copies.
    auto helper_body = [&regions, &batch_ready, &next_task, n_tasks]() {
        for (;;) {
            const int t = next_task.fetch_add(1);
            if (t >= n_tasks) return;
            Region& r = regions[(size_t)t];
            if (r.ptr && r.bytes) {
                volatile char* q = (volatile char*)r.ptr;
                for (size_t o = 0; o < r.bytes; o += 4096) q[o] = 0;
                q[r.bytes - 1] = 0;
            }
            batch_ready[(size_t)(t / 4)].fetch_add(1);
        }
    };
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int n_helpers = (int)std::min<unsigned>(std::min<unsigned>(8u, hw), (unsigned)n_tasks);
        for (int t = 0; t < n_helpers; ++t) helpers.emplace_back(helper_body);
    }
    int regions_pinned_upto = 0;
    auto wait_batch_regions = [&](int bi) {   // every region of batch bi (and of the batches before it) is resident and pinned
        for (int b = 0; b <= bi; ++b)
            while (batch_ready[(size_t)b].load() < 4) std::this_thread::yield();
        for (; regions_pinned_upto <= bi; ++regions_pinned_upto)
            for (int i = 0; i < 4 && pin_outputs; ++i) {
                Region& r = regions[(size_t)regions_pinned_upto * 4 + i];
                if (!r.ptr || !r.bytes) continue;
                if (hipHostRegister((void*)r.ptr, r.bytes, hipHostRegisterDefault) == hipSuccess) r.pinned = true;
                else (void)hipGetLastError();
            }
    };
    auto join_helpers = [&]() { for (auto& t : helpers) if (t.joinable()) t.join(); };
    auto fail_msg = [&](const char* what, hipError_t e) { c->err = std::string(what) + ": " + hipGetErrorString(e); return -2; };
    int rc_all = 0;
    double* frames_buf[2] = {c->st_movie, multi ? c->st_movie2 : c->st_movie};
    auto upload = [&](int bi, hipStream_t st) {   // frames k0 .. k0 + np of batch bi; a copy never straddles the two pinned regions
        const Batch& bt = batches[bi];
        const size_t o0 = (size_t)bt.k0 * fs * sizeof(double), o1 = o0 + (size_t)(bt.np + 1) * fs * sizeof(double);
        char* dst = (char*)frames_buf[bi & 1];
        hipError_t e = hipSuccess;
        if (o0 < split) e = hipMemcpyAsync(dst, mbase + o0, std::min(o1, split) - o0, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && o1 > split) {
            const size_t a = std::max(o0, split);
            e = hipMemcpyAsync(dst + (a - o0), mbase + a, o1 - a, hipMemcpyHostToDevice, st);
        }
        return e;
    };
    {
        hipError_t e = upload(0, c->stream);
        if (e != hipSuccess) rc_all = fail_msg("H2D copy failed", e);
    }
    for (int bi = 0; bi < nb && !rc_all; ++bi) {
        const Batch& bt = batches[bi];
        const int set = bi & 1;
        double** so = (multi && set) ? c->st_out2 : c->st_out;
        hipError_t e;
        if (bi + 1 < nb) {   // frames of the next batch: uploaded on the copy stream while this batch is solved
            if (movie_helper.joinable()) movie_helper.join();
            // its buffer was last read by batch bi - 1
            if (bi >= 1 && (e = hipStreamWaitEvent(c->copy_stream, c->ev_solved[(bi + 1) & 1], 0)) != hipSuccess) { rc_all = fail_msg("stream wait failed", e); break; }
            if ((e = upload(bi + 1, c->copy_stream)) != hipSuccess ||
                (e = hipEventRecord(c->ev_uploaded[(bi + 1) & 1], c->copy_stream)) != hipSuccess) { rc_all = fail_msg("H2D copy failed", e); break; }
        }
        if (bi > 0 && (e = hipStreamWaitEvent(c->stream, c->ev_uploaded[set], 0)) != hipSuccess) { rc_all = fail_msg("stream wait failed", e); break; }
        if (multi && bi >= 2 && (e = hipStreamWaitEvent(c->stream, c->ev_copied[set], 0)) != hipSuccess) {   // output set free again
            rc_all = fail_msg("stream wait failed", e);
            break;
        }
        int rc = solve_range_dev(c, frames_buf[set], bt.np, so[0], so[1], so[2], so[3], stats ? stats + bt.k0 : nullptr);
        if (rc) { rc_all = rc; break; }
        hmark("batch solved");
        wait_batch_regions(bi);
        hmark("output regions of the batch ready");
        hipStream_t cs = multi ? c->copy_stream : c->stream;
        if (multi) {
            if ((e = hipEventRecord(c->ev_solved[set], c->stream)) != hipSuccess ||
                (e = hipStreamWaitEvent(cs, c->ev_solved[set], 0)) != hipSuccess) { rc_all = fail_msg("event record failed", e); break; }
        }
        for (int i = 0; i < 4 && !rc_all; ++i)
            if (outs[i]) {
                // the batch's bytes [o0, o1) of array i: the part before region bi's first page lies in region bi - 1
                const size_t o0 = (size_t)bt.k0 * fs * sizeof(double), o1 = o0 + (size_t)bt.np * fs * sizeof(double);
                const size_t r0 = (size_t)(regions[(size_t)bi * 4 + i].ptr - (char*)outs[i]);
                const size_t cut = std::min(std::max(r0, o0), o1);
                if (cut > o0) {
                    e = hipMemcpyAsync((char*)outs[i] + o0, (const char*)so[i], cut - o0, hipMemcpyDeviceToHost, cs);
                    if (e != hipSuccess) { rc_all = fail_msg("D2H copy failed", e); break; }
                }
                if (o1 > cut) {
                    e = hipMemcpyAsync((char*)outs[i] + cut, (const char*)so[i] + (cut - o0), o1 - cut, hipMemcpyDeviceToHost, cs);
                    if (e != hipSuccess) rc_all = fail_msg("D2H copy failed", e);
                }
            }
        if (multi) {
            if (!rc_all && (e = hipEventRecord(c->ev_copied[set], cs)) != hipSuccess) rc_all = fail_msg("event record failed", e);
        } else if ((e = hipStreamSynchronize(c->stream)) != hipSuccess && !rc_all) rc_all = fail_msg("stream synchronize failed", e);
    }
    join_helpers();
    if (movie_helper.joinable()) movie_helper.join();
    hmark("all batches enqueued");
    if (multi && hipStreamSynchronize(c->copy_stream) != hipSuccess && !rc_all) { c->err = "copy stream synchronize failed"; rc_all = -2; }
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc_all) { c->err = "stream synchronize failed"; rc_all = -2; }
    hmark("copies done");
    for (auto& r : regions)
        if (r.pinned) (void)hipHostUnregister((void*)r.ptr);
    if (pinned_a) (void)hipHostUnregister((void*)mbase);
    if (pinned_b) (void)hipHostUnregister((void*)(mbase + split));
    hmark("unpinned");
    return rc_all;
}

int vof_blur_stack_dev(vof_ctx* c, const double* in, double* out, int n_frames, const double* weights, int radius) {
    if (!c) return -1;
    if (!in || !out || !weights) { c->err = "NULL pointer"; return -1; }
    if (n_frames < 1 || radius < 0 || radius > 4096) { c->err = "bad n_frames / radius"; return -1; }
    HIPCHK(hipSetDevice(c->device));
    size_t fs = frame_stride(c);
    const int chunk = std::min(n_frames, 16);
    if (!c->blur_tmp) {
        if (int rc = dev_alloc(c, &c->blur_tmp, (size_t)16 * fs)) return rc;
        if (int rc = dev_alloc(c, &c->blur_w, (size_t)2 * 4096 + 1)) return rc;
    }
    HIPCHK(hipMemcpyAsync(c->blur_w, weights, (size_t)(2 * radius + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        int nf = std::min(chunk, n_frames - f0);
        dim3 g = grid2d(c->Ni, c->Nj, nf);
        Prof p(c, VOF_K_RHS, 0);
        k_blur1d<0><<<g, blk2d, 0, c->stream>>>(in + (size_t)f0 * fs, c->blur_tmp, c->Ni, c->Nj, c->blur_w, radius);
        k_blur1d<1><<<g, blk2d, 0, c->stream>>>(c->blur_tmp, out + (size_t)f0 * fs, c->Ni, c->Nj, c->blur_w, radius);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int vof_blur_stack_host(vof_ctx* c, const double* in, double* out, int n_frames, const double* weights, int radius) {
    if (!c) return -1;
    if (!in || !out || !weights) { c->err = "NULL pointer"; return -1; }
    HIPCHK(hipSetDevice(c->device));
    size_t fs = frame_stride(c);
    if (!c->blur_io) { if (int rc = dev_alloc(c, &c->blur_io, (size_t)2 * 16 * fs)) return rc; }
    for (int f0 = 0; f0 < n_frames; f0 += 16) {
        int nf = std::min(16, n_frames - f0);
        HIPCHK(hipMemcpyAsync(c->blur_io, in + (size_t)f0 * fs, (size_t)nf * fs * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (int rc = vof_blur_stack_dev(c, c->blur_io, c->blur_io + (size_t)16 * fs, nf, weights, radius)) return rc;
        HIPCHK(hipMemcpyAsync(out + (size_t)f0 * fs, c->blur_io + (size_t)16 * fs, (size_t)nf * fs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// sum (x - shift), sum (x - shift)^2 over n device doubles -> out2 (host); uses ctx->partials / func3 as scratch
static int moments_pass(vof_ctx* c, const double* x, size_t n, double shift, double* out2) {
    const int nb = c->nblk;
    Prof p(c, VOF_K_REDUCE, 0, 8.0 * n);
    k_moments<<<nb, RBLK, 0, c->stream>>>(x, n, shift, c->partials);
    k_sum3<<<1, 64, 0, c->stream>>>(c->partials, nb, c->func3);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_func3, c->func3, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    out2[0] = c->h_func3[0];
    out2[1] = c->h_func3[1];
    return 0;
}

// (count, mean, M2 = sum (x - mean)^2) of one chunk, exact two-pass; chunks are merged with Chan's formula
struct Moments {
    double n = 0, mean = 0, m2 = 0;
    void merge(double nb, double meanb, double m2b) {
        if (nb == 0) return;
        double nt = n + nb, d = meanb - mean;
        m2 += m2b + d * d * n * nb / nt;
        mean += d * nb / nt;
        n = nt;
    }
};

static int chunk_moments(vof_ctx* c, const double* x, size_t n, Moments* acc) {
    double s[2];
    if (int rc = moments_pass(c, x, n, 0.0, s)) return rc;
    double mean = s[0] / (double)n;
    if (int rc = moments_pass(c, x, n, mean, s)) return rc;
    mean += s[0] / (double)n;                                      // first-order correction of the rounded mean
    acc->merge((double)n, mean, s[1] - s[0] * s[0] / (double)n);
    return 0;
}

int vof_field_moments_dev(vof_ctx* c, const double* field, size_t n, double* mean, double* variance) {
    if (!c) return -1;
    if (!field || n == 0) { c->err = "empty field"; return -1; }
    HIPCHK(hipSetDevice(c->device));
    Moments m;
    if (int rc = chunk_moments(c, field, n, &m)) return rc;
    if (mean) *mean = m.mean;
    if (variance) *variance = m.m2 / m.n;
    return 0;
}

int vof_subsample_dev(vof_ctx* c, const double* field, int n_fields, int box, int offset, double* out) {
    if (!c) return -1;
    if (!field || !out) { c->err = "NULL pointer"; return -1; }
    if (n_fields < 1 || box < 1 || offset < 0 || offset >= box) { c->err = "bad n_fields / box / offset"; return -1; }
    HIPCHK(hipSetDevice(c->device));
    int nbx = c->Ni / box, nby = c->Nj / box;
    if (nbx < 1 || nby < 1) return 0;
    k_subsample<<<grid2d(nbx, nby, n_fields), blk2d, 0, c->stream>>>(field, c->Ni, c->Nj, box, offset, nbx, nby, out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// One batch of "virtual pairs" (own alpha / beta / frame / output slot each) of a device-resident movie, with the
// preconditioner policy of solve_range_dev: direct only, or multigrid with the direct re-solve of what it leaves unconverged.
static int solve_virtual_pairs(vof_ctx* c, const double* dmovie, const std::vector<PairParam>& hp, int np, double* const* outs,
                               vof_pair_stats* st) {
    if (c->prm.preconditioner == 1 && c->L.size() > 1) {
        std::vector<PairParam> items(hp.begin(), hp.begin() + np);
        return direct_solve_list(c, dmovie, items, outs[0], outs[1], outs[2], outs[3], st);
    }
    if (!c->pp_buf) { if (int rc = dev_alloc(c, &c->pp_buf, (size_t)c->B)) return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));   // hp is re-used by the caller: the previous upload must have completed
    HIPCHK(hipMemcpyAsync(c->pp_buf, hp.data(), (size_t)np * sizeof(PairParam), hipMemcpyHostToDevice, c->stream));
    c->pp = c->pp_buf;
    int rc = solve_batch(c, dmovie, np, outs[0], outs[1], outs[2], outs[3], st);
    c->pp = nullptr;
    if (rc) return rc;
    if (c->prm.preconditioner == 2) {
        std::vector<PairParam> items;
        std::vector<int> which;
        for (int i = 0; i < np; ++i)
            if (!st[i].converged && std::isfinite(st[i].relative_residual)) { items.push_back(hp[i]); which.push_back(i); }
        if (!items.empty() && direct_ok_for_fallback(c)) {
            std::vector<vof_pair_stats> s2(items.size());
            int rc2 = direct_solve_list(c, dmovie, items, outs[0], outs[1], outs[2], outs[3], s2.data());
            if (rc2 == 0) {
                for (size_t i = 0; i < which.size(); ++i) { s2[i].iterations += st[which[i]].iterations; st[which[i]] = s2[i]; }
            } else if (rc2 != -3) {
                return rc2;
            } else {
                c->err.clear();
            }
        }
    }
    return 0;
}

int vof_vary_regularisation_host(vof_ctx* c, const double* movie, int n_frames, const vof_params* base,
                                 const double* speed_alphas, int n_sa, const double* remodelling_alphas, int n_ra,
                                 const double* blur_weights, int blur_radius, vof_variation_stats* out) {
    if (!c) return -1;
    if (!movie || !base || !out || !speed_alphas || !remodelling_alphas) { c->err = "NULL pointer"; return -1; }
    if (n_frames < 2) { c->err = "need at least two frames"; return -1; }
    if (n_sa < 0 || n_ra < 0) { c->err = "negative grid size"; return -1; }
    if (int rc = check_params(c, base)) return rc;
    HIPCHK(hipSetDevice(c->device));
    const size_t fs = frame_stride(c);
    const int P = n_frames - 1;
    if (int rc = ensure_staging(c, false)) return rc;   // per-batch output staging shared with the host API
    // the whole movie stays resident for the sweep (freed on return)
    double* dmovie = nullptr;
    HIPCHK(hipMalloc((void**)&dmovie, (size_t)n_frames * fs * sizeof(double)));
    auto fail = [&](int rc) { (void)hipFree(dmovie); return rc; };
    {   // pinned in place for the upload (pageable copies run at ~2 GB/s here, see vof_solve_stack_host)
        const size_t movie_bytes = (size_t)n_frames * fs * sizeof(double);
        const bool pinned = hipHostRegister((void*)movie, movie_bytes, hipHostRegisterDefault) == hipSuccess;
        if (!pinned) (void)hipGetLastError();
        hipError_t e = hipMemcpyAsync(dmovie, movie, movie_bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (pinned) (void)hipHostUnregister((void*)movie);
        if (e != hipSuccess) { c->err = std::string("H2D copy failed: ") + hipGetErrorString(e); return fail(-2); }
    }
    if (blur_weights)
        if (int rc = vof_blur_stack_dev(c, dmovie, dmovie, n_frames, blur_weights, blur_radius)) return fail(rc);
    auto summarise = [&](vof_variation_stats& o, const vof_pair_stats* st, const Moments& ms, const Moments& mr) {
        memset(&o, 0, sizeof o);
        o.speed_mean = ms.mean; o.speed_variance = ms.m2 / ms.n;
        o.remodelling_mean = mr.mean; o.remodelling_variance = mr.m2 / mr.n;
        o.converged_all = 1;
        for (int k = 0; k < P; ++k) {
            o.L1_functional += st[k].L1_functional;
            o.speed_functional += st[k].speed_functional;
            o.remodelling_functional += st[k].remodelling_functional;
            o.max_relative_residual = std::max(o.max_relative_residual, st[k].relative_residual);
            o.max_iterations_used = std::max(o.max_iterations_used, st[k].iterations);
            o.converged_all &= st[k].converged;
        }
        o.converged_last = st[P - 1].converged;
    };
    const int n_comb = n_sa * n_ra;
    for (int t = 0; t < n_comb; ++t) {   // validate every combination before any work
        vof_params q = *base;
        q.speed_alpha = speed_alphas[t / n_ra];
        q.remodelling_alpha = remodelling_alphas[t % n_ra];
        if (int rc = check_params(c, &q)) return fail(rc);
    }
    if (int rc = check_params(c, base)) return fail(rc);
    if (P <= c->B) {
        // Short movies leave the chip idle (16 pairs of 512^2 fill a third of it): solve G combinations at once as
        // G * P "virtual pairs" of one batch - pair v = (combination v / P, frame pair v % P) reads its own
        // (alpha, beta) and frame index from the PairParam table.
        const int G = std::max(1, c->B / P);
        if (!c->pp_buf) { if (int rc = dev_alloc(c, &c->pp_buf, (size_t)c->B)) return fail(rc); }
        std::vector<PairParam> hp((size_t)G * P);
        std::vector<vof_pair_stats> st((size_t)G * P);
        for (int t0 = 0; t0 < n_comb; t0 += G) {
            const int g = std::min(G, n_comb - t0), np = g * P;
            for (int u = 0; u < g; ++u)
                for (int k = 0; k < P; ++k)
                    hp[(size_t)u * P + k] = PairParam{speed_alphas[(t0 + u) / n_ra], remodelling_alphas[(t0 + u) % n_ra], k, u * P + k};
            if (int rc = solve_virtual_pairs(c, dmovie, hp, np, c->st_out, st.data())) return fail(rc);
            for (int u = 0; u < g; ++u) {
                Moments ms, mr;
                if (int rc2 = chunk_moments(c, c->st_out[3] + (size_t)u * P * fs, (size_t)P * fs, &ms)) return fail(rc2);
                if (int rc2 = chunk_moments(c, c->st_out[2] + (size_t)u * P * fs, (size_t)P * fs, &mr)) return fail(rc2);
                summarise(out[t0 + u], st.data() + (size_t)u * P, ms, mr);
            }
        }
    } else {
        std::vector<vof_pair_stats> st((size_t)P);
        for (int t = 0; t < n_comb; ++t) {
            vof_params q = *base;
            q.speed_alpha = speed_alphas[t / n_ra];
            q.remodelling_alpha = remodelling_alphas[t % n_ra];
            if (int rc = check_params(c, &q)) return fail(rc);
            Moments ms, mr;
            std::vector<PairParam> hp2((size_t)c->B);
            for (int k0 = 0; k0 < P; k0 += c->B) {
                int np = std::min(c->B, P - k0);
                for (int i = 0; i < np; ++i) hp2[i] = PairParam{q.speed_alpha, q.remodelling_alpha, k0 + i, i};
                if (int rc = solve_virtual_pairs(c, dmovie, hp2, np, c->st_out, st.data() + k0)) return fail(rc);
                if (int rc = chunk_moments(c, c->st_out[3], (size_t)np * fs, &ms)) return fail(rc);
                if (int rc = chunk_moments(c, c->st_out[2], (size_t)np * fs, &mr)) return fail(rc);
            }
            summarise(out[t], st.data(), ms, mr);
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "stream synchronize failed"; return fail(-2); }
    return fail(0);
}

int vof_texture_stack_dev(vof_ctx* c, double* out, int n_frames, const double* mode_params, int n_modes,
                          const double* frame_offsets, double period, double scale) {
    if (!c) return -1;
    if (!out || !mode_params || !frame_offsets) { c->err = "NULL pointer"; return -1; }
    if (n_frames < 1 || n_modes < 1 || n_modes > 4096 || !(period > 0.0)) { c->err = "bad n_frames / n_modes / period"; return -1; }
    HIPCHK(hipSetDevice(c->device));
    const int width = std::max(c->Ni, c->Nj);
    const int chunk = std::max(1, std::min(n_frames, (int)(((size_t)64 << 20) / ((size_t)4 * n_modes * width * sizeof(double)))));
    const size_t need = (size_t)chunk * 4 * n_modes * width + (size_t)4 * n_modes + (size_t)2 * chunk;
    if (c->tex_cap < need) {
        if (int rc = dev_alloc(c, &c->tex_tab, need)) return rc;
        c->tex_cap = need;
    }
    double* prm = c->tex_tab + (size_t)chunk * 4 * n_modes * width;
    double* offs = prm + (size_t)4 * n_modes;
    HIPCHK(hipMemcpyAsync(prm, mode_params, (size_t)4 * n_modes * sizeof(double), hipMemcpyHostToDevice, c->stream));
    for (int t0 = 0; t0 < n_frames; t0 += chunk) {
        const int nf = std::min(chunk, n_frames - t0);
        HIPCHK(hipMemcpyAsync(offs, frame_offsets + (size_t)2 * t0, (size_t)2 * nf * sizeof(double), hipMemcpyHostToDevice, c->stream));
        k_texture_tables<<<dim3((width + 255) / 256, n_modes, nf), 256, 0, c->stream>>>(c->tex_tab, width, c->Ni, c->Nj, n_modes, prm,
                                                                                       offs, period);
        k_texture_sum<<<dim3((c->Nj + BX - 1) / BX, (c->Ni + BY * TEX_ROWS - 1) / (BY * TEX_ROWS), nf), blk2d, 0, c->stream>>>(
            c->tex_tab, width, c->Ni, c->Nj, n_modes, scale, out + (size_t)t0 * frame_stride(c));
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));   // offs / the tables are re-used by the next chunk; host offsets may be freed
    }
    return 0;
}

int vof_bench_sweeps_dev(vof_ctx* c, const double* movie, int n_pairs, const vof_params* p, int n_sweeps) {
    if (!c) return -1;
    if (!movie) { c->err = "NULL movie"; return -1; }
    if (n_pairs < 1 || n_pairs > c->B) { c->err = "n_pairs must be in [1, max_pairs_in_flight]"; return -1; }
    if (c->L.size() < 2) { c->err = "grid too small for the sweep benchmark"; return -1; }
    if (int rc = check_params(c, p)) return rc;
    HIPCHK(hipSetDevice(c->device));
    c->frames = movie;
    c->npairs = n_pairs;
    c->cur_units = n_pairs;
    Level& f = c->L[0];
    {
        Prof pr(c, VOF_K_RHS, 0);
        k_rhs<<<grid2d(f.ni, f.nj, n_pairs), blk2d, 0, c->stream>>>(movie, frame_stride(c), c->Nj, f.ni, f.nj, c->kb, nullptr);
    }
    HIPCHK(hipMemsetAsync(c->kx, 0, (size_t)n_pairs * 3 * f.npts * sizeof(double), c->stream));
    if (c->vfloat) {
        size_t n = (size_t)n_pairs * 3 * f.npts;
        k_convert<double, float><<<1024, 256, 0, c->stream>>>(c->kb, (float*)c->b32, n);
        smooth_level_t<float>(c, 0, (float*)c->ky, (float*)f.x2, (const float*)c->b32, n_sweeps, true, false, n_pairs, nullptr);
    } else {
        smooth_level_t<double>(c, 0, c->kx, (double*)f.x2, c->kb, n_sweeps, true, false, n_pairs, nullptr);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int vof_profile_enable(vof_ctx* c, int on) {
    if (!c) return -1;
    if (!on) prof_collect(c);
    c->prof = on != 0;
    return 0;
}

int vof_profile_filter(vof_ctx* c, int kid, int level) {
    if (!c) return -1;
    if (kid >= VOF_K_COUNT || level > 15) { c->err = "bad kernel id / level"; return -1; }
    c->prof_kid = kid;
    c->prof_level = level;
    return 0;
}

int vof_profile_get_units(vof_ctx* c, int kid, int level, int64_t* pair_launches) {
    if (!c) return -1;
    if (kid < 0 || kid >= VOF_K_COUNT || level > 15) { c->err = "bad kernel id / level"; return -1; }
    prof_collect(c);
    long long u = 0;
    for (int l = 0; l < 16; ++l)
        if (level < 0 || l == level) u += c->prof_units[kid][l];
    if (pair_launches) *pair_launches = u;
    return 0;
}

int vof_profile_get_bytes(vof_ctx* c, int kid, int level, double* algorithmic_bytes) {
    if (!c) return -1;
    if (kid < 0 || kid >= VOF_K_COUNT || level > 15) { c->err = "bad kernel id / level"; return -1; }
    prof_collect(c);
    double u = 0;
    for (int l = 0; l < 16; ++l)
        if (level < 0 || l == level) u += c->prof_bytes[kid][l];
    if (algorithmic_bytes) *algorithmic_bytes = u;
    return 0;
}

int vof_profile_get_moved(vof_ctx* c, int kid, int level, double* moved_bytes) {
    if (!c) return -1;
    if (kid < 0 || kid >= VOF_K_COUNT || level > 15) { c->err = "bad kernel id / level"; return -1; }
    prof_collect(c);
    double u = 0;
    for (int l = 0; l < 16; ++l)
        if (level < 0 || l == level) u += c->prof_moved[kid][l];
    if (moved_bytes) *moved_bytes = u;
    return 0;
}

int vof_profile_reset(vof_ctx* c) {
    if (!c) return -1;
    prof_collect(c);
    memset(c->prof_ms, 0, sizeof c->prof_ms);
    memset(c->prof_n, 0, sizeof c->prof_n);
    memset(c->prof_units, 0, sizeof c->prof_units);
    memset(c->prof_bytes, 0, sizeof c->prof_bytes);
    memset(c->prof_moved, 0, sizeof c->prof_moved);
    c->prof_dropped = 0;
    return 0;
}

int vof_profile_get(vof_ctx* c, int kid, int level, int64_t* launches, double* total_ms) {
    if (!c) return -1;
    if (kid < 0 || kid >= VOF_K_COUNT || level > 15) { c->err = "bad kernel id / level"; return -1; }
    prof_collect(c);
    long long n = 0; double ms = 0;
    for (int l = 0; l < 16; ++l)
        if (level < 0 || l == level) { n += c->prof_n[kid][l]; ms += c->prof_ms[kid][l]; }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return 0;
}

// ---------------------------------------------------------------------------------- debug entry points
static int dbg_ready(vof_ctx* c) {
    if (!c) return -1;
    if (!c->frames || c->npairs < 1) { c->err = "call vof_debug_setup first"; return -1; }
    return 0;
}

int vof_debug_setup(vof_ctx* c, const double* movie_host, int n_pairs, const vof_params* p) {
    if (!c) return -1;
    if (!movie_host) { c->err = "NULL movie"; return -1; }
    if (n_pairs < 1 || n_pairs > c->B) { c->err = "n_pairs must be in [1, max_pairs_in_flight]"; return -1; }
    if (int rc = check_params(c, p)) return rc;
    HIPCHK(hipSetDevice(c->device));
    size_t fs = frame_stride(c);
    if (int rc = ensure_staging(c, false)) return rc;
    HIPCHK(hipMemcpyAsync(c->st_movie, movie_host, (size_t)(n_pairs + 1) * fs * sizeof(double), hipMemcpyHostToDevice,
                          c->stream));
    if (int rc = setup_batch(c, c->st_movie, n_pairs)) return rc;
    c->vcoarse32 = p->vcycle_precision == 3;   // vof_debug_vcycle* run the cycle as a solve would (the per-level entry points: float64)
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int vof_debug_check_canaries(vof_ctx* c) {
    if (!c) return -1;
    if (!c->dbg_canary) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    std::string rep;
    const int bad = dbg_check_canaries(c, &rep);
    if (bad != 0) { c->err = "VOF_DEBUG_CANARY: " + rep; return -5; }
    return 0;
}

int vof_debug_level_shape(vof_ctx* c, int level, int* n_i, int* n_j) {
    if (!c) return -1;
    if (level < 0 || level >= (int)c->L.size()) { c->err = "bad level"; return -1; }
    if (n_i) *n_i = c->L[level].ni;
    if (n_j) *n_j = c->L[level].nj;
    return 0;
}

#define DBG_LEVEL(level)                                                              \
    if (int rc_ = dbg_ready(c)) return rc_;                                           \
    if (level < 0 || level >= (int)c->L.size()) { c->err = "bad level"; return -1; } \
    Level& lv = c->L[level];                                                          \
    size_t nbytes = (size_t)c->npairs * 3 * lv.npts * sizeof(double);                 \
    (void)nbytes;

// Device -> pageable host memory through a pinned bounce buffer on the context's stream (the debug entry points used the
// runtime's blocking hipMemcpy on the null stream, which pins the destination on the fly for copies above 1 MiB).
constexpr size_t BOUNCE_BYTES = (size_t)8 << 20;
static int d2h_bounced(vof_ctx* c, void* host, const void* dev, size_t bytes) {
    if (!c->h_bounce) HIPCHK(hipHostMalloc((void**)&c->h_bounce, BOUNCE_BYTES));
    for (size_t off = 0; off < bytes; off += BOUNCE_BYTES) {
        const size_t n = std::min(BOUNCE_BYTES, bytes - off);
        HIPCHK(hipMemcpyAsync(c->h_bounce, (const char*)dev + off, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy((char*)host + off, c->h_bounce, n);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// The debug API moves host float64 arrays in and out of V-typed device buffers (kp, kv, kt; staging: krh).
static int dbg_up(vof_ctx* c, void* dst_v, const double* host, size_t n) {
    if (c->vfloat) {
        HIPCHK(hipMemcpyAsync(c->krh, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        k_convert<double, float><<<256, 256, 0, c->stream>>>(c->krh, (float*)dst_v, n);
    } else {
        HIPCHK(hipMemcpyAsync(dst_v, host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    return 0;
}
static int dbg_down(vof_ctx* c, double* host, const void* src_v, size_t n) {
    if (c->vfloat) {
        k_convert<float, double><<<256, 256, 0, c->stream>>>((const float*)src_v, c->krh, n);
        HIPCHK(hipMemcpyAsync(host, c->krh, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(host, src_v, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int vof_debug_rhs(vof_ctx* c, double* b_host) {
    DBG_LEVEL(0)
    k_rhs<<<grid2d(lv.ni, lv.nj, c->npairs), blk2d, 0, c->stream>>>(c->frames, frame_stride(c), c->Nj, lv.ni, lv.nj, c->kb, nullptr);
    HIPCHK(hipMemcpyAsync(b_host, c->kb, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int vof_debug_apply(vof_ctx* c, int level, const double* x_host, double* y_host) {
    DBG_LEVEL(level)
    size_t n = nbytes / sizeof(double);
    if (int rc = dbg_up(c, c->kp, x_host, n)) return rc;
    if (level == 0) {   // the Krylov product: V-typed x, FP64 result
        krylov_apply(c, c->kp, c->kv, c->npairs, nullptr);
        HIPCHK(hipMemcpyAsync(y_host, c->kv, nbytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipGetLastError());
        return 0;
    }
    VDISPATCH(c, apply_level_t<VT>(c, level, (const VT*)c->kp, (const VT*)nullptr, (VT*)c->kv, 0, c->npairs, nullptr));
    return dbg_down(c, y_host, c->kv, n);
}

int vof_debug_gs(vof_ctx* c, int level, double* x_host, const double* b_host, int colour) {
    DBG_LEVEL(level)
    if (colour < 0 || colour > 3) { c->err = "bad colour"; return -1; }
    if (c->vfloat) { c->err = "the per-colour reference smoother works on float64 vectors only"; return -1; }
    HIPCHK(hipMemcpyAsync(c->kp, x_host, nbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->kv, b_host, nbytes, hipMemcpyHostToDevice, c->stream));
    gs_colour(c, level, c->kp, c->kv, colour, c->npairs, nullptr);
    HIPCHK(hipMemcpyAsync(x_host, c->kp, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int vof_debug_sweep(vof_ctx* c, int level, double* x_host, const double* b_host, int reverse, int from_zero) {
    DBG_LEVEL(level)
    if (level + 1 >= (int)c->L.size()) { c->err = "coarsest level has no smoother"; return -1; }
    size_t n = nbytes / sizeof(double);
    if (int rc = dbg_up(c, c->kp, x_host, n)) return rc;
    if (int rc = dbg_up(c, c->kv, b_host, n)) return rc;
    VDISPATCH(c, sweep_level_t<VT>(c, level, from_zero ? (const VT*)nullptr : (const VT*)c->kp, (VT*)c->kt,
                                   (const VT*)c->kv, reverse != 0, c->npairs, nullptr));
    return dbg_down(c, x_host, c->kt, n);
}

int vof_debug_smooth(vof_ctx* c, int level, double* x_host, const double* b_host, int nu, int reverse, int from_zero) {
    DBG_LEVEL(level)
    if (level + 1 >= (int)c->L.size()) { c->err = "coarsest level has no smoother"; return -1; }
    if (nu < 1) { c->err = "nu must be >= 1"; return -1; }
    size_t n = nbytes / sizeof(double);
    if (int rc = dbg_up(c, c->kp, x_host, n)) return rc;
    if (int rc = dbg_up(c, c->kv, b_host, n)) return rc;
    VDISPATCH(c, smooth_level_t<VT>(c, level, (VT*)c->kp, (VT*)c->kt, (const VT*)c->kv, nu, from_zero != 0, reverse != 0, c->npairs, nullptr));
    return dbg_down(c, x_host, c->kp, n);
}

int vof_set_fused_sweeps(vof_ctx* c, int on) {
    if (!c) return -1;
    c->fused = on != 0;
    if (!c->fused) c->vfloat = false;
    return 0;
}

int vof_debug_restrict(vof_ctx* c, int level, const double* fine_host, double* coarse_host) {
    DBG_LEVEL(level)
    if (level + 1 >= (int)c->L.size()) { c->err = "no coarser level"; return -1; }
    Level& k = c->L[level + 1];
    if (int rc = dbg_up(c, c->kp, fine_host, nbytes / sizeof(double))) return rc;
    VDISPATCH(c, restrict_level_t<VT>(c, level, (const VT*)c->kp, (VT*)c->kv, c->npairs, nullptr));
    return dbg_down(c, coarse_host, c->kv, (size_t)c->npairs * 3 * k.npts);
}

int vof_debug_resrestrict_u(vof_ctx* c, int level, const double* x_new_host, const double* x_old_host, double* coarse_host) {
    DBG_LEVEL(level)
    if (level < 1 || level + 1 >= (int)c->L.size() || !lv.C) { c->err = "a stored level with a coarser level is needed"; return -1; }
    Level& k = c->L[level + 1];
    if (int rc = dbg_up(c, c->kp, x_new_host, nbytes / sizeof(double))) return rc;
    if (x_old_host)
        if (int rc = dbg_up(c, c->kt, x_old_host, nbytes / sizeof(double))) return rc;
    VDISPATCH(c, resrestrict_u_t<VT>(c, level, (const VT*)c->kp, x_old_host ? (const VT*)c->kt : (const VT*)nullptr, (VT*)c->kv, c->npairs, nullptr));
    return dbg_down(c, coarse_host, c->kv, (size_t)c->npairs * 3 * k.npts);
}

int vof_debug_prolong_add(vof_ctx* c, int level, double* fine_host, const double* coarse_host) {
    DBG_LEVEL(level)
    if (level + 1 >= (int)c->L.size()) { c->err = "no coarser level"; return -1; }
    Level& k = c->L[level + 1];
    if (int rc = dbg_up(c, c->kp, fine_host, nbytes / sizeof(double))) return rc;
    if (int rc = dbg_up(c, c->kv, coarse_host, (size_t)c->npairs * 3 * k.npts)) return rc;
    VDISPATCH(c, prolong_add_level_t<VT>(c, level, (VT*)c->kp, (const VT*)c->kv, c->npairs, nullptr));
    return dbg_down(c, fine_host, c->kp, nbytes / sizeof(double));
}

int vof_debug_stencil(vof_ctx* c, int level, double* c_host) {
    DBG_LEVEL(level)
    if (!lv.C) { c->err = "level has no stored stencil"; return -1; }
    const CLay L(lv.ni, lv.nj);
    const int fmt = level > 0 ? c->cfmt : 0;
    const int planes = fmt == 3 ? 30 : (fmt == 2 ? 45 : 81);
    const size_t n = (size_t)c->npairs * planes * L.plane;
    // colour-split device layout -> row-major [pair][81][n_i][n_j]
    auto unpack = [&](auto get) {
        for (int k = 0; k < c->npairs; ++k)
            for (int pl = 0; pl < 81; ++pl)
                for (int p = 0; p < lv.ni; ++p)
                    for (int q = 0; q < lv.nj; ++q)
                        c_host[((size_t)k * 81 + pl) * lv.npts + (size_t)p * lv.nj + q] = get((size_t)k * planes * L.plane, pl, L.idx(p, q));
    };
    if (fmt == 2) {   // 36 planes of packed bfloat16 pairs (off-diagonal blocks) + 9 float32 planes (diagonal block)
        std::vector<uint32_t> tw(n);
        if (int rc = d2h_bounced(c, tw.data(), lv.C, n * sizeof(uint32_t))) return rc;
        unpack([&](size_t base, int pl, size_t idx) -> double {
            const int d = pl / 9, e = pl % 9;
            uint32_t bits;
            if (d == 4) bits = tw[base + (size_t)(36 + e) * L.plane + idx];
            else {
                const int j = (d < 4 ? d : d - 1) * 9 + e;
                const uint32_t v = tw[base + (size_t)(j >> 1) * L.plane + idx];
                bits = (j & 1) ? (v & 0xFFFF0000u) : (v << 16);
            }
            float f;
            memcpy(&f, &bits, 4);
            return f;
        });
    } else if (fmt == 3) {   // 18 planes of four 8-bit floats (1-4-3, bias 7, no infinities) + 9 float32 planes + 3 planes of units
        std::vector<uint32_t> tw(n);
        if (int rc = d2h_bounced(c, tw.data(), lv.C, n * sizeof(uint32_t))) return rc;
        unpack([&](size_t base, int pl, size_t idx) -> double {
            const int d = pl / 9, e = pl % 9;
            if (d == 4) {
                float f;
                memcpy(&f, &tw[base + (size_t)(18 + e) * L.plane + idx], 4);
                return f;
            }
            const int j = (d < 4 ? d : d - 1) * 9 + e;
            const uint32_t v = (tw[base + (size_t)(j >> 2) * L.plane + idx] >> (8 * (j & 3))) & 0xFFu;
            const int ex = (int)((v >> 3) & 0xFu), m = (int)(v & 7u);
            const double mag = ex ? std::ldexp(1.0 + m / 8.0, ex - 7) : std::ldexp(m / 8.0, -6);
            const int eb = (int)((tw[base + (size_t)(27 + e / 3) * L.plane + idx] >> (8 * (e % 3))) & 0xFFu);
            return ((v & 0x80u) ? -mag : mag) * std::ldexp(1.0, eb - 127);
        });
    } else if (fmt == 1) {
        std::vector<float> tf(n);
        if (int rc = d2h_bounced(c, tf.data(), lv.C, n * sizeof(float))) return rc;
        unpack([&](size_t base, int pl, size_t idx) -> double { return tf[base + (size_t)pl * L.plane + idx]; });
    } else {
        std::vector<double> tmp(n);
        if (int rc = d2h_bounced(c, tmp.data(), lv.C, n * sizeof(double))) return rc;
        unpack([&](size_t base, int pl, size_t idx) -> double { return tmp[base + (size_t)pl * L.plane + idx]; });
    }
    return 0;
}

int vof_debug_vcycle(vof_ctx* c, const double* r_host, double* e_host) {
    DBG_LEVEL(0)
    size_t n = nbytes / sizeof(double);
    if (int rc = dbg_up(c, c->kp, r_host, n)) return rc;
    vcycle(c, &c->ky, c->kp, c->npairs, nullptr);
    return dbg_down(c, e_host, c->ky, n);
}

// One multigrid cycle y = M r followed by the Krylov product v = A y with the dot products (v, r) and (v, v) per pair, as the
// BiCGStab loop runs them (fused into the cycle's last smoothing pass when that path applies; `fused` reports it).
int vof_debug_vcycle_apply(vof_ctx* c, const double* r_host, double* y_host, double* v_host, double* dots_host, int* fused) {
    DBG_LEVEL(0)
    size_t n = nbytes / sizeof(double);
    if (int rc = dbg_up(c, c->kp, r_host, n)) return rc;
    const int np = c->npairs;
    c->cur_units = np;
    HIPCHK(hipMemcpyAsync(c->krh, r_host, nbytes, hipMemcpyHostToDevice, c->stream));   // dot partner (float64)
    c->trail_req = S0Trail{c->kv, c->krh, 1, c->partials};
    c->trail_set = true; c->trail_done = false;
    vcycle(c, &c->ky, c->kp, np, nullptr);
    c->trail_set = false;
    int nb = c->trail_done ? c->trail_nblk : krylov_apply(c, c->ky, c->kv, np, nullptr, c->krh, 1);
    if (fused) *fused = c->trail_done ? 1 : 0;
    if (!nb) { c->err = "the operator kernel did not fuse the dot products"; return -1; }
    std::vector<double> part((size_t)np * 3 * nb);
    HIPCHK(hipMemcpyAsync(part.data(), c->partials, part.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (int rc = dbg_down(c, y_host, c->ky, n)) return rc;
    if (int rc = d2h_bounced(c, v_host, c->kv, nbytes)) return rc;
    for (int k = 0; k < np; ++k)
        for (int sl = 0; sl < 2; ++sl) {
            double a = 0;
            for (int i = 0; i < nb; ++i) a += part[((size_t)k * 3 + sl) * nb + i];
            dots_host[2 * k + sl] = a;
        }
    return 0;
}

int vof_debug_coarse_solve(vof_ctx* c, const double* r_host, double* e_host) {
    int last = c ? (int)c->L.size() - 1 : 0;
    DBG_LEVEL(last)
    size_t n = nbytes / sizeof(double);
    if (int rc = dbg_up(c, c->kp, r_host, n)) return rc;
    VDISPATCH(c, coarse_solve_t<VT>(c, (const VT*)c->kp, (VT*)c->kv, c->npairs, nullptr));
    return dbg_down(c, e_host, c->kv, n);
}

}  // extern "C"
