// Internal declarations shared by the libdfhip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <functional>
#include "../../include/dfhip.h"

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

void dfh_set_error(const char* fmt, ...);

#define DFH_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) {                                                            \
      dfh_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return DFH_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define DFH_TRY(call)                 \
  do {                                \
    int rc__ = (call);                \
    if (rc__ != DFH_OK) return rc__;  \
  } while (0)

#define DFH_ARG(cond)                                                         \
  do {                                                                        \
    if (!(cond)) {                                                            \
      dfh_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond);    \
      return DFH_ERR_BAD_ARG;                                                 \
    }                                                                         \
  } while (0)

#define DFH_LAUNCH_CHECK() DFH_HIP(hipGetLastError())

// A device allocation that is kept for the life of its owner (ctx scratch or gp state).
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// matrices factored in lock-step by one batched cholesky_device call (one pivot flag each)
constexpr int CHOL_MAX_BATCH = 64;
// function attributes (dynamic LDS limits) are per device: the "already set" flags are indexed by it
constexpr int DFH_MAX_DEVICES = 64;

struct dfh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // every kernel is launched on this stream ...
  hipStream_t main_stream = nullptr; // (== stream except inside StreamSwap scopes)
  hipStream_t side = nullptr;        // high-priority panel stream of the look-ahead Cholesky
  hipStream_t bulk = nullptr;        // low-priority stream: next chunk's cross kernel + TRSM (TS)
  hipStream_t aux = nullptr;         // off-critical-path work of the factorisation (block inverses)
  hipStream_t bulk_normal = nullptr; // normal-priority twin of `side` (experiments: DFH_CHOL_LR_NORMAL_PRIO)
  std::vector<hipEvent_t> evpool;    // untimed events for cross-stream ordering
  // factorisations that were repeated on the schedule without inter-workgroup hand-offs (a bounded wait
  // expired, or a block inverse was too poor for the resident panels) -- dfh_ctx_counters; after two in a row
  // the next calls go straight to that schedule (chol_cooldown of them), so that a crowded device does not pay
  // the time-out of ~1 s on every fit
  int64_t chol_fallbacks = 0;
  int chol_fallback_streak = 0, chol_cooldown = 0;
  // labels of the last tuning call, resident (lml_batch_wg): host copy to compare with, device copy, sum y and sum y^2
  std::vector<double> ycache_host;
  const double* ycache_dev = nullptr;
  double ycache_sum = 0.0, ycache_sum2 = 0.0;
  int lml_team_cooldown = 0;       // tuning batches that take one workgroup per candidate after a team's hand-off timed out
  hipEvent_t ev0 = nullptr, ev1 = nullptr;         // dfh_timer_begin / end
  // scratch pool: grow-only named slots reused across calls (no hipMalloc in hot loops)
  std::vector<DevBuf> scratch;
  int64_t* d_info = nullptr;                         // device int64[CHOL_MAX_BATCH + 16]: pivot flag per batch matrix, 8 debug words, hand-off status
  int64_t* h_info = nullptr;                         // pinned mirror
  void* h_stage = nullptr; size_t h_stage_bytes = 0; // pinned, grow-only: small per-call blobs and results
  // section timing
  bool timing = false;
  double t_ms[DFH_T_COUNT] = {0};
  hipEvent_t tev0[DFH_T_COUNT] = {nullptr}, tev1[DFH_T_COUNT] = {nullptr};   // one pair per section (nestable)
  char name[256] = {0};
  int n_cu = 256;
  double chunk_cap_gib = 0.0;        // posterior chunk cap of this context (api.hip: pick_chunk), set on first use
  // per-launch HIP-event profile of the GEMM kernel (bench.py roofline numbers)
  // When set, 128x128 GEMM launches request > 80 KB of LDS so that only ONE workgroup fits per
  // CU: the other half of every CU stays available to latency-critical kernels of another stream.
  bool gemm_half_occupancy = false;
  // The next symmetric single-part Gram matrix is wanted as its lower triangle only (tiles on and below the diagonal;
  // the rest of the buffer is left as it is): the fit path, whose factorisation reads nothing above the diagonal.
  bool km_lower_only = false;
  // Extended GEMM launches of the factorisation (gemm_f64.hip, chol.hip), consumed by the next
  // gemm_f64 call(s) while set: a device-side skip condition (the launch exits unless *gemm_cond >
  // gemm_cond_thr) and the look-ahead tile order with its completion counters.
  const double* gemm_cond = nullptr; double gemm_cond_thr = 0.0;
  int* gemm_la_cnt = nullptr;
  bool gemm_prof = false;
  struct GemmRec { hipEvent_t e0, e1; double flops, bytes; int variant; };
  std::vector<GemmRec> gemm_recs;
  size_t gemm_used = 0;
  hipEvent_t gemm_base = nullptr;   // time origin of the per-launch intervals
  // Block cache behind dev_alloc / dev_release: hipMalloc + hipFree of anything beyond the
  // runtime's small-block arena costs ~0.3 ms (2 MiB), which is most of a GP fit at n < 1000 --
  // the hyper-parameter searches of the reference fit and drop thousands of those in a row.
  struct PoolBlock { void* p; size_t cap; };
  std::vector<PoolBlock> pool_idle;                 // released blocks, ready for reuse
  std::unordered_map<void*, size_t> pool_caps;      // capacity of every block handed out or idle
  size_t pool_idle_bytes = 0;
  size_t pool_idle_limit = size_t(2) << 30;         // DFH_POOL_MAX_MIB (0 disables the cache)
};

enum ScratchSlot {
  SCR_STAGE_A = 0,  // host->device staging of inputs
  SCR_STAGE_B,
  SCR_STAGE_C,
  SCR_STAGE_D,
  SCR_XS,           // scaled / gathered copies
  SCR_KCT,          // candidate-by-train cross kernel / V^T chunk
  SCR_XS2,          // second parity of SCR_XS / SCR_KCT / SCR_VEC (pipelined Thompson sampling)
  SCR_KCT2,
  SCR_VECB,
  SCR_STAGE_A2,
  SCR_TMP,          // TRSM block temp
  SCR_TMP2,
  SCR_VEC,          // small vectors
  SCR_VEC2,
  SCR_VEC3,
  SCR_RED,          // reduction partials
  SCR_CHOLW,        // cholesky panel workspace
  SCR_CHOLINV,      // cholesky diag-block inverses (transient use)
  SCR_CHOLT,        // inverse-assembly temporaries
  SCR_TSK,          // TS block covariance
  SCR_TSL,          // TS block factor
  SCR_AUG,          // hallucination: augmented rows
  SCR_AUG2,
  SCR_OUT,          // device result staging for host outputs
  SCR_OUT2,
  SCR_DELTA,        // quality of the kept block inverses (one double per diagonal block)
  SCR_CHOLSYNC,     // resident look-ahead Cholesky: per-panel counters
  SCR_CHOLX,        // ... its inverse-based panel solve: solution
  SCR_CHOLR,        // ... and residual
  SCR_REFINE,       // residual of the refined row solves (trsm_rows*)
  SCR_CHOLKEEP,     // ... block inverses when the caller keeps none
  SCR_MUPART,       // cross matrix with the posterior mean fused in: per-block partial sums
  SCR_RED2,         // two-stage log-determinant: per-workgroup partial sums
  SCR_YCACHE,       // labels of the last tuning call (kept across calls: a fitter asks thousands of times with the same y)
  SCR_LMLCTL,       // tuning group: results | failed pivots | status | team flags, one block (one memset, one copy back)
  SCR_COUNT
};

// Device memory that outlives a call (GP state, caller-visible buffers): cached by size on release.
// The context's stream is idle whenever a block is released (callers synchronise first), so a
// reused block is never still being written by an earlier launch.
int dev_alloc(dfh_ctx* ctx, size_t bytes, void** out);
void dev_release(dfh_ctx* ctx, void* p);      // ctx may be null / already destroyed: plain hipFree

// pinned host staging memory of at least `bytes` (contents undefined; valid until the next call)
int pinned_get(dfh_ctx* ctx, size_t bytes, void** out);

// returns a device pointer with at least `bytes` capacity (contents undefined)
int scratch_get(dfh_ctx* ctx, int slot, size_t bytes, void** out);
bool ctx_is_live(const dfh_ctx* ctx);    // false once dfh_ctx_destroy ran (or for a foreign pointer)

// Resolve a user pointer: if it is a host pointer, copy `bytes` to scratch slot `slot` and
// return the device copy; if it is a device pointer return it unchanged.
int to_device(dfh_ctx* ctx, const void* p, size_t bytes, int slot, const double** out);
bool is_device_ptr(const void* p);
// MT19937 state n_words further down the stream, by jump-ahead on the host (mtjump.hip)
int mt19937_advance_host(uint32_t* key, int32_t* pos, int64_t n_words);
// Copy a device result to a user pointer that may be host or device.
int from_device(dfh_ctx* ctx, void* user_dst, const void* dev_src, size_t bytes);

// Temporarily route launches to another stream of the same context.
struct StreamSwap {
  dfh_ctx* c; hipStream_t old;
  StreamSwap(dfh_ctx* ctx, hipStream_t s) : c(ctx), old(ctx->stream) { c->stream = s; }
  ~StreamSwap() { c->stream = old; }
};
int ctx_event(dfh_ctx* ctx, size_t idx, hipEvent_t* out);   // lazily created, untimed
// index ranges of the context's event pool: the Thompson pipeline's five events come first, the
// factorisation's 2 + 5 per panel follow without an upper limit (the pool grows on demand), so the
// largest factorisation is set by memory, not by the pool
constexpr size_t EV_TS_BASE = 0;
constexpr size_t EV_CHOL_BASE = 8;

struct SectionTimer {
  dfh_ctx* ctx; int which; bool on;
  SectionTimer(dfh_ctx* c, int w);
  ~SectionTimer();
};

// ---------------------------------------------------------------------------------------
// device-level building blocks (all asynchronous on ctx->stream, device pointers only)
// ---------------------------------------------------------------------------------------
#define GEMM_LOWER   1   // compute only tiles intersecting the lower triangle (row >= col)
#define GEMM_KTRI_B  2   // B is [N x K] lower triangular (B[j][k]=0 for k>j): clip k-range
#define GEMM_TRANSB  4   // B is [K x N]

// Two-level batch: blockIdx.z = b2 * count + b1 ; operand offset = b1 * s? + b2 * s?2
struct GemmBatch {
  int count = 1; int64_t sA = 0, sB = 0, sCin = 0, sCout = 0;
  int count2 = 1; int64_t sA2 = 0, sB2 = 0, sCin2 = 0, sCout2 = 0;
};

// Cout = beta*Cin + alpha * A * op(B).  Cin may equal Cout.  Cin may be null iff beta == 0.
int gemm_f64(dfh_ctx* ctx, int flags, int64_t M, int64_t N, int64_t K, double alpha,
             const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
             const double* Cin, int64_t ldcin, double* Cout, int64_t ldc,
             const GemmBatch* batch = nullptr);

// Few-row variant (m <= 256, K a multiple of 64, NT only): streams B once, see gemm_f64.hip.
// Acopy (optional, 16-byte aligned rows, even ld): A is also copied there -- it must not overlap
// anything the call reads or writes.
bool gemm_skinny_applies(int64_t m, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                         int64_t ldb);
int gemm_skinny_nt(dfh_ctx* ctx, int64_t m, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double beta, const double* Cin, int64_t ldcin, double* Cout,
                   int64_t ldc, double* Acopy = nullptr, int64_t ldacopy = 0);

// Flattened kernel description.  A "part" is one SE / Matern kernel over a subset of the input
// columns: SE and Matern kernels have one part, an additive kernel one part per group
// (dragonfly/gp/kernel.py:484-494).  Inputs are pre-scaled once into a packed layout
//   Xp[n][P],  P = sum_parts pad4(|cols_part|),  Xp[i][poff+c] = X[i][cols[c]] / bw[c]
// (get_scaled_repr, kernel.py:179-181,255-257) with zero padding, plus the per-part squared
// row norms Np[n][n_parts] ((X**2).sum(axis=1), general_utils.py:66-67).
#define DFH_KERNEL_DIST 3   // internal: clip(dist_sq) itself (dfh_dist_squared)
struct PartDev {
  int kind;        // DFH_KERNEL_SE | DFH_KERNEL_MATERN | DFH_KERNEL_DIST | DFH_KERNEL_POLY | DFH_KERNEL_EXPDECAY
  int poff;        // first packed column
  int kc;          // packed (padded to 4) column count
  int p;           // Matern: int(nu) ; Poly: order ; ExpDecay: number of (real) columns
  double scale_c;  // SE / Poly / ExpDecay: scale ; Matern: scale * norm_constant (kernel.py:298)
  double s8, s2;   // Matern: sqrt(8 nu), sqrt(2 nu)
  double gfac;     // Matern: Gamma(p+1)/Gamma(2p+1) ; ExpDecay: offset
  double coeff[8]; // Matern: (p+i)!/(i!(p-i)!) ; ExpDecay: powers
  double k0;       // SE / Matern: k_part(x, x) (distance 0); unused for the non-stationary kinds
  // A part of an ADDITIVE FACTOR of a product kernel (an AdditiveKernel among the kernels of a
  // CoordinateProductKernel: the multi-fidelity GP with an additive domain model, gp/euclidean_gp.py:696-707):
  // the factor's parts are adjacent; they are summed (0 + k_1 + k_2 ..., kernel.py:490-493), the sum is
  // scaled (kernel.py:494) and only then multiplied into the product (kernel.py:588).
  int fmode;       // 0: a factor of its own ; FM_IN (| FM_BEGIN | FM_END): inside an additive factor
  int fpad;
  double fscale;   // FM_END: the additive factor's scale
};
constexpr int FM_IN = 4, FM_BEGIN = 1, FM_END = 2;
constexpr int EXPDECAY_MAX_DIM = 8;
struct KernDev {
  int kind = 0, dim = 0, n_parts = 0, P = 0;
  bool multi = false;          // additive: sum over parts then outer scale; product: scale * prod over parts
  bool product = false;        // (multi only) combine the parts by multiplication (kernel.py:584-588)
  bool nested = false;         // (product only) some factor is a sum of parts (PartDev::fmode)
  double outer_scale = 1.0;
  std::vector<PartDev> parts;
  std::vector<int> cols;       // [P] source column per packed column (-1 = padding)
  std::vector<int> lcols;      // [P] column index local to the part (for pre-gathered inputs)
  std::vector<double> bw;      // [P]
  PartDev* d_parts = nullptr;
  int* d_cols = nullptr;
  int* d_lcols = nullptr;
  double* d_bw = nullptr;
  void* d_blob = nullptr;      // the one device allocation behind the four pointers (owned if set)
  double kxx = 0.0;            // prior variance k(x,x) of a stationary kernel
  bool stationary = true;      // false with a Poly / ExpDecay part: k(x,x) depends on x (prior_diag)
};
int kerndev_build(dfh_ctx* ctx, const dfh_kernel_desc* k, KernDev* out);
// host-only part of kerndev_build, and the upload of several descriptors with one copy into a
// caller-provided device blob (kerndev_blob_bytes each, in order); such KernDevs own no memory
int kerndev_build_host(const dfh_kernel_desc* k, KernDev* out);
int kerndev_stage_many(KernDev* kds, int count, char* host, void* d_blob, size_t blob_bytes);
size_t kerndev_blob_bytes(const KernDev& kd);
int kerndev_upload_many(dfh_ctx* ctx, KernDev* kds, int count, void* d_blob, size_t blob_bytes);
// One-launch tuning objective for small problems (kernmat.hip: k_lml_tiny): applies when
// n <= TINY_MAX_N and every candidate's packed width / part count fits the LDS budget.
constexpr int64_t TINY_MAX_N = 128;
constexpr int64_t TINY64_MAX_N = 63;      // k_lml_tiny64: the system (n + 1 rows) is one 64 x 64 tile
constexpr int TINY_MAX_P = 64, TINY_MAX_PARTS = 16;
bool lml_tiny_applies(const KernDev* kds, int count, int64_t n);
// The blob of a small-problem tuning call in the context's pinned (mapped, coherent) buffer:
// TinyCand[count] | kernel images | y[n] | 10^-11 .. 10^4 | (64-byte aligned) results [count][4]
struct TinyBlob {
  char* host = nullptr; size_t bytes = 0, y_off = 0, pow_off = 0;
  double* res = nullptr;
  int Pmax = 1, parts_max = 1;
};
int tiny_blob_build(dfh_ctx* ctx, const KernDev* kds, int count, int64_t n, const double* y_host,
                    const double* noise_vars, const double* mean_consts, TinyBlob* tb);
int tiny_poll_results(dfh_ctx* ctx, volatile double* vres, int count, const char* what);
// TINY64_MAX_N < n <= LMLF_MAX_N, a handful of candidates: Gram matrix, factorisation and forward solve of each candidate in ONE
// launch by one workgroup (chol.hip: lml_wgf_kernel), descriptors and results through the pinned buffer.
// info[c]: 0 = logdet_dot[2c], [2c+1] are valid; otherwise the candidate is for the lock-step schedule (a failed pivot:
// the ladder; no noise: nothing bounds the augmented pivot).
constexpr int64_t LMLF_MAX_N = 128;     // (beyond, the team schedule -- one copy up, one memset, one copy back per group since round 6 -- is as fast: 101 us
                                         //  at n = 129 .. 191 against the one workgroup's 105 .. 143; the kernel itself is tested up to n = 255 through DFH_LML_FUSED_MAX_N)
bool lml_wg_fused_applies(const KernDev* kds, int count, int64_t n);
int lml_wg_fused_batch(dfh_ctx* ctx, const KernDev* kds, int count, const double* dX, int64_t n, int64_t ldx,
                       const double* y_host, const double* noise_vars, const double* mean_consts,
                       double* logdet_dot, long long* info);
int lml_tiny_batch(dfh_ctx* ctx, const KernDev* kds, int count, const double* dX, int64_t n, int64_t ldx,
                   const double* y_host, const double* noise_vars, const double* mean_consts,
                   bool allow_jitter, double* logdet_dot, int32_t* powers);
int kerndev_build_dist(dfh_ctx* ctx, int dim, KernDev* out);
int kerndev_clone(dfh_ctx* ctx, const KernDev& src, KernDev* out);   // deep copy with its own device image
void kerndev_free(KernDev* kd);
double kerndev_part_kxx(const KernDev& kd, int part);
// out[i] = k(x_i, x_i) from the packed inputs of m points (any kernel; needed when !kd.stationary)
// part_lo / part_hi (default: all parts): the prior variance of those groups of an additive kernel only
int prior_diag(dfh_ctx* ctx, const KernDev& kd, const double* Xp, const double* Np, int64_t m, double* out, int part_lo = 0,
               int part_hi = -1);

// Xp[n][P] / Np[n][n_parts] for parts [part_lo, part_hi) (other parts' columns untouched).
// pre_gathered: X holds only the columns of part_lo (ldx >= |cols|), as in add-UCB group
// candidates (gpb_acquisitions.py:164-166).
// count > 1: a lock-step batch -- kernel images sBlob bytes apart (kd is the first), outputs sXp /
// sNp doubles apart.
int pack_scaled(dfh_ctx* ctx, const KernDev& kd, int part_lo, int part_hi, bool pre_gathered,
                const double* X, int64_t n, int64_t ldx, double* Xp, double* Np, int count = 1,
                int64_t sBlob = 0, int64_t sXp = 0, int64_t sNp = 0);
int kernmat_sym_batch(dfh_ctx* ctx, const KernDev& kd, int count, int64_t sBlob, const double* Xp,
                      int64_t sXp, const double* Np, int64_t sNp, int64_t n, const double* d_diag_adds,
                      double* K, int64_t sK, int64_t ldk);

// K[n1 x n2] (ldk) = sum over parts [part_lo,part_hi) of k_part (times outer scale if multi).
// symmetric: Xp2/Np2 == Xp1/Np1 and diag_add is added on the diagonal.
int kernmat_packed(dfh_ctx* ctx, const KernDev& kd, int part_lo, int part_hi, bool apply_outer,
                   const double* Xp1, const double* Np1, int64_t n1, const double* Xp2,
                   const double* Np2, int64_t n2, bool symmetric, double diag_add, double* K,
                   int64_t ldk, const double* mu_alpha = nullptr, double* mu_out = nullptr, bool* mu_done = nullptr);
// (mu_alpha, mu_out, mu_done: ask for mu_out[n1] = K mu_alpha from the same pass; *mu_done tells whether
//  the kernel that ran could do it -- the caller multiplies itself otherwise)

// Blocked Cholesky, in place on the lower triangle of the row-major matrix A (upper part of the
// off-diagonal blocks is left untouched; the upper part of the 64x64 diagonal blocks is zeroed).
// keep_inv (optional): receives the inverses of the CHOL_NB x CHOL_NB diagonal blocks of L,
// block b at keep_inv + b*CHOL_NB*CHOL_NB, row-major with ld = CHOL_NB, zero upper part.
// *info_pivot = 0 on success, else the 1-based index of the first non-positive pivot.
constexpr int64_t CHOL_NB = 512;
// nbatch > 1: nbatch independent matrices of the same size at A + b*strideA are factored in lock
// step (one launch sequence, every kernel batched): the latency-bound pivot chain is paid once.
// info_pivot has nbatch entries.
// keep_inv buffers hold inv_buffer_doubles(n) doubles per matrix: the block inverses, then clean
// copies of the diagonal blocks themselves (lower triangle, zero elsewhere, row stride CHOL_NB).
// refine_out (host, [nbatch][nblk], optional): iterative-refinement steps a solve should take with
// each block, from the measured quality max|I - inv L_bb| of its inverse (chol.hip: refine_steps).
inline int64_t inv_buffer_doubles(int64_t n) { return 2 * ((n + CHOL_NB - 1) / CHOL_NB) * CHOL_NB * CHOL_NB; }
// inv64_only: keep_inv receives only the inverses of the 64 x 64 diagonal blocks (on the diagonal of
// each 512-block slot, zero elsewhere) -- no 512-block assembly, no quality measurement, refine_out
// all zero: for callers that substitute with 64-blocks themselves (the small tuning objective).
// rebuild (optional): restores the input in A; enables the schedules that may need a second attempt
// (chol.hip: resident look-ahead, hand-off timeouts) -- without it those failures are DFH_ERR_HIP.
int cholesky_device(dfh_ctx* ctx, double* A, int64_t n, int64_t lda, double* keep_inv,
                    int64_t* info_pivot, int nbatch = 1, int64_t strideA = 0, int64_t strideKeep = 0,
                    int* refine_out = nullptr, bool inv64_only = false,
                    const std::function<int()>* rebuild = nullptr);
constexpr int DFH_INTERNAL_RETRY = 1000;   // chol.hip internal: never crosses the C-ABI (a hand-off wait expired)
constexpr int DFH_INTERNAL_RETRY_COND = 1001;   // ... a block inverse too poor for the inverse-based panel solve (deterministic)

// The tuning objective of `count` candidates, one workgroup per candidate (chol.hip: lml_wg_kernel): Cholesky of the
// augmented matrix [[K, .], [(y - m)^T, c]] of each, sum(log L_ii) and |L^-1 (y - m)|^2 out.  K: matrices padded to
// order 64 * ceil((n + 1) / 64) (only the n x n part has to be filled), sK doubles apart, row stride ld.
constexpr int64_t LMLWG_MAX_N = 2047;
// team > 1: that many workgroups per candidate (for groups far smaller than the device); *d_status != 0 afterwards
// means a hand-off between them timed out and the launch's results are void (repeat with team = 1).
int lml_wg_batch(dfh_ctx* ctx, double* K, int64_t sK, int64_t ld, int64_t n, int count, const double* d_y,
                 const double* d_par, double* d_out2, long long* d_info, int team = 1,
                 unsigned long long* d_status = nullptr, int* d_sync_zeroed = nullptr);
constexpr int LMLT_SYNC_INTS_PER_CANDIDATE = 64;     // (= chol.hip's LMLT_SYNC_INTS: flags of a team, per candidate)

// alpha-solves with the factor and its diagonal-block inverses (in place on x[n]):
//   forward : x <- L^{-1} x          backward : x <- L^{-T} x
// refine (host, one entry per diagonal block, or null): refinement steps per block, see above
int trsv_both(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv, double* x, const int* refine);
int trsv_forward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                 double* x, const int* refine = nullptr);
int trsv_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                  double* x, const int* refine = nullptr);
// Rows-as-RHS solve used by the posterior:  Vt[m x n] <- Kct[m x n] * L^{-T}  (in place),
// i.e. each row v of Vt satisfies L v = k  (solve_lower_triangular(L, K_tetr.T), gp_core.py:180)
// diag_override: where the clean copies of L's diagonal blocks are, when not right behind its inverses
int trsm_rows(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
              double* Kct, int64_t m, int64_t ldk, const int* refine = nullptr, const double* diag_override = nullptr);
// Xt[m x n] <- Bt[m x n] * L^{-1} (in place): each row x satisfies L^T x = b
int trsm_rows_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                       double* Bt, int64_t m, int64_t ldb, const int* refine = nullptr);
// inverses of the CHOL_NB diagonal blocks of an existing lower factor L (layout as keep_inv)
// diag (optional): where the clean diagonal-block copies go (default: right behind the inverses)
int tri_block_inverses(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, double* inv,
                       int* refine_out = nullptr, double* diag = nullptr);

// projection of a symmetric matrix onto {eigenvalues >= eps} (psdproj.hip); out may alias M
int psd_project_device(dfh_ctx* ctx, const double* M, int64_t n, int64_t ldm, double eps, double* out, int64_t ldo);

// small helpers (elementwise / reductions)
int fill_f64(dfh_ctx* ctx, double* p, int64_t n, double v);
int zero_upper(dfh_ctx* ctx, double* A, int64_t n, int64_t lda);
int diag_max(dfh_ctx* ctx, const double* A, int64_t n, int64_t lda, double* host_out);
int add_diag(dfh_ctx* ctx, double* A, int64_t n, int64_t lda, double v);
int copy_matrix(dfh_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd,
                int64_t rows, int64_t cols);
int transpose_matrix(dfh_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd,
                     int64_t rows, int64_t cols);
// yout[m] = beta*yin + alpha * A[m x n] x[n]   (yin may equal yout; may be null iff beta == 0)
// tri_lower: row i only uses columns j <= i (lower-triangular A)
int gemv_rows(dfh_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, const double* x,
              double alpha, const double* yin, double beta, double* yout, bool tri_lower = false);
// yout[n] = beta*yin + alpha * A[m x n]^T x[m]
int gemv_cols(dfh_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, const double* x,
              double alpha, const double* yin, double beta, double* yout);
// out[m] = sum_j A[i][j]^2
int row_sumsq(dfh_ctx* ctx, const double* A, int64_t m, int64_t n, int64_t lda, double* out);
// host result: sum(log(diag(L))) and dot(a,b)
int logdet_and_dot_device(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* a,
                          const double* b, double* d_out2);
int logdet_and_dot(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* a,
                   const double* b, double* host_logdet, double* host_dot);
