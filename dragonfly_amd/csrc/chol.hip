// Blocked right-looking Cholesky + the triangular solves built on its diagonal-block inverses.
//
// Replaces np.linalg.cholesky (LAPACK dpotrf) at dragonfly/utils/general_utils.py:178,190 and
// scipy.linalg.solve_triangular (dtrtrs) at general_utils.py:213 as used by
// GP.build_posterior / GP.eval (dragonfly/gp/gp_core.py:159-163,180).
//
// Structure (row-major lower, n x n):
//   outer panels of CHOL_NB = 512 columns; inside a panel the 512 x 512 diagonal block is
//   factored with 64-wide steps: one launch (diag_step64) factors the 64 x 64 pivot block in
//   registers and solves the 64-wide column below it by substitution, then the rest of the
//   diagonal block is updated by an MFMA SYRK.  The eight 64-block inverses (trtri64, one
//   launch) are then merged into the inverse of the 512 block by three
//   levels of batched GEMMs ([[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]), so the panel
//   solve  L21 = A21 L11^-T  and the trailing update  A22 -= L21 L21^T  are two large MFMA_F64
//   GEMMs -- where n^3/3 of the flops are.  The 512-block inverses are kept: the posterior
//   solve (trsm_rows) and the alpha solves (trsv_*) reuse them, which turns every triangular
//   solve on the hot path into GEMM / GEMV work.
#include "common.h"
#include <cmath>
#include <functional>
#include <utility>
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int PBP = 65;      // LDS row stride

// Hand-off status word of a factorisation (d_info[CHOL_MAX_BATCH + 8], zeroed per call): set when a
// bounded wait expired.  The host then repeats the factorisation on the schedule without
// inter-workgroup hand-offs (cholesky_device), or reports DFH_ERR_HIP when the input is gone.
constexpr unsigned SYNC_ST_FUSED = 2;     // a strip of the one-launch panel waited too long for another strip
constexpr unsigned SYNC_ST_GATE = 4;      // a gate kernel / resident diagonal kernel waited too long for another launch
constexpr int SPIN_LIMIT_DEFAULT = 1 << 21;   // polls of ~0.5 us each: a second, a few hundred times the longest legitimate wait

#include "factor64.h"   // factor64_waves and its helpers (PB, SYNC_ST_RING, fast_rcp, perm16, quad_sum, ...)
#include "kerneval.h"   // kernel evaluation (the fused tuning objective builds its Gram matrix itself)

// One 64-wide step of the diagonal-block factorisation, one launch:
//   workgroup 0     : factors the nb x nb pivot block at D and writes the factor to Lout;
//   workgroup b >= 1: factors the same block redundantly (bit-identical, no inter-workgroup
//                     hand-off needed) and solves 64 rows of the column below it,
//                     P[r,:] <- P[r,:] L^-T, by forward substitution with four lanes per row.
// info[0] <- pivot_base + j + 1 for the first non-positive / NaN pivot.
constexpr int SPP = 66;      // row stride of the column-permuted factor image (16-byte aligned rows)
__global__ __launch_bounds__(256) void diag_step64_kernel(double* __restrict__ D, long lda, int nb,
                                                          int rows_below, long pivot_base,
                                                          long long* info, double* __restrict__ Lout,
                                                          long strideD, long strideL,
                                                          double* __restrict__ LinvOut, long strideI) {
  // batch element = blockIdx.y
  D += (long)blockIdx.y * strideD;
  Lout += (long)blockIdx.y * strideL;
  if (LinvOut) LinvOut += (long)blockIdx.y * strideI;
  long long* const info_dbg = info + CHOL_MAX_BATCH;   // debug words follow the pivot flags
  info += blockIdx.y;
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* Sp = dsm;                       // [64][SPP] factor, columns permuted by perm16
  double* R = dsm + PB * SPP;             // [64][65] panel rows
  double* colbuf = dsm + PB * SPP + PB * PBP;      // [64] reciprocal diagonal, then the column ring and its flags
  const int tid = threadIdx.x;
  const int k = tid & 63, w = tid >> 6;

  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // stage the pivot block (identity padding beyond nb, zero above the diagonal), coalesced; it
  // sits in the Sp region, which wave 0 overwrites with the factor image only after reading it
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = w + 4 * r;
    double v = (i == k) ? 1.0 : 0.0;
    if (i < nb && k < nb) v = (k <= i) ? D[i * lda + k] : 0.0;
    Sp[i * SPP + k] = v;
  }
  __shared__ int s_badv[4];
  __shared__ int s_ring_timeout;
  double* ring = colbuf + PB;                      // [64][64] published (unscaled) columns
  double* tbuf0 = ring + PB * PB;                  // 3 x [64][17] layout buffers (later 4 x [16][17] solve tiles)
  double* lbb = tbuf0 + 3 * PB * 17;               // 4 x [16][17] diagonal 16-blocks of the factor
  double* linv = lbb + 4 * 16 * 17;                // 4 x [16][17] their inverses
  double* rdiag = colbuf;                          // [64] 1 / L[c][c]
  if (tid < PB) ring[tid * PB] = 0.0;              // row-0 entries double as the "published" flags
  if (tid == 0) s_ring_timeout = 0;
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const int r0 = ((int)blockIdx.x - 1) * PB;
  double* Pn = D + (long)(nb + r0) * lda;
  {
    double a[16];
    long long stamp[4] = {0, 0, 0, 0};
    const bool dbg = info_dbg[7] != 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0;
    double* tbuf = tbuf0 + (w > 0 ? (w - 1) : 0) * PB * 17;
    const int bad = factor64_waves(a, k, w, Sp, tbuf, ring, lbb, linv, rdiag, &s_ring_timeout, dbg ? stamp : nullptr);
    if (k == 0) s_badv[w] = bad;
    if (dbg && k == 0) {
      // wave 3: start / end of its own 16 columns, end of scaling, end of the 16 x 16 inverse (debug hook only)
      if (w == 3) { info_dbg[0] = stamp[0] - (long long)t1; info_dbg[1] = stamp[1] - (long long)t1;
                    info_dbg[5] = stamp[2] - (long long)t1; info_dbg[6] = stamp[3] - (long long)t1; }
    }
    // factor image for the row solves: Sp[row][perm16(col)], and the reciprocal diagonal
#pragma unroll
    for (int j = 0; j < 16; ++j) Sp[k * SPP + perm16(16 * w + j)] = a[j];
    if (w == 0 && blockIdx.x > 0) {
      // wave 0 is done after the first 16 columns: it stages this workgroup's 64 panel rows
      // (coalesced along the row) while the other waves go on with the factorisation
      // (16 loads in flight at a time: one load per iteration would cost a full memory latency each)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double tmp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 16 * c + r;
          tmp[r] = (r0 + i < rows_below && k < nb) ? Pn[i * lda + k] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) R[(16 * c + r) * PBP + k] = tmp[r];
      }
    }
  }
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (info_dbg[7] != 0 && tid == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) { info_dbg[2] = (long long)(t1 - t0); info_dbg[3] = (long long)(t2 - t1); }
  const int s_bad = (s_badv[0] >= 0) ? s_badv[0] : (s_badv[1] >= 0) ? s_badv[1] : (s_badv[2] >= 0) ? s_badv[2] : s_badv[3];
  if (s_ring_timeout && tid == 0) atomicOr((unsigned long long*)(info_dbg + 8), (unsigned long long)SYNC_ST_RING);
  if (s_bad >= 0) {
    if (blockIdx.x == 0 && tid == 0 && info[0] == 0) info[0] = pivot_base + s_bad + 1;
    return;
  }
  if (blockIdx.x == 0) {
    // The factor goes to a scratch block (64 x 64, ld 64), NOT back into D: the other workgroups
    // of this launch re-read the unfactored pivot block from D and may start arbitrarily later.
    // trtri64_kernel moves it into place once the whole diagonal block is done.
    const int pk = perm16(k);
#pragma unroll
    for (int r = 0; r < 16; ++r) Lout[(w + 4 * r) * PB + k] = Sp[(w + 4 * r) * SPP + pk];
    // ... and the inverses of its four 16 x 16 diagonal blocks, for the strip kernel (same layout as in LDS)
    if (LinvOut)
      for (int i = tid; i < 4 * 16 * 17; i += 256) LinvOut[i] = linv[i];
    return;
  }
  {
    // Row solve X L^T = P on the matrix cores, 16-column blocks: for b = 0..3
    //     X_b = (P_b - sum_{b' < b} X_b' L[b][b']^T) Linv_bb^T
    // with the inverses of the four 16 x 16 diagonal blocks (computed by the waves that own them,
    // see factor64_waves).  Wave w takes rows 16w .. 16w+15 of the workgroup's 64: no dependence
    // between waves.  X_b goes back into R, from where the later blocks read it as an A operand;
    // the only layout change is accumulator -> A operand for the multiplication by Linv_bb^T, through
    // a 16 x 16 LDS tile private to the wave.  40 MFMAs per wave; the substitution this replaces was
    // a 64-step dependent chain (16k cycles).
    double* Tt = tbuf0 + w * (16 * 17);
    const int kq = k >> 4, l15 = k & 15;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      // two accumulators per product: a dependent MFMA waits ~64 cycles for its predecessor
      double4_t acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = R[(16 * w + kq + 4 * r) * PBP + 16 * b + l15];
#pragma unroll
      for (int bp = 0; bp < b; ++bp)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const double av = R[(16 * w + l15) * PBP + 16 * bp + 4 * st + kq];                 // X_b'[i][k]
          const double bv = -Sp[(16 * b + l15) * SPP + perm16(16 * bp + 4 * st + kq)];       // -L[16b+j][16b'+k]
          if (st & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) Tt[(kq + 4 * r) * 17 + l15] = acc[r] + acc2[r];
      COMPILER_BARRIER();                            // same wave: LDS executes its operations in order
      double4_t x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const double av = Tt[l15 * 17 + 4 * st + kq];                                        // T[i][k]
        const double bv = linv[b * (16 * 17) + l15 * 17 + 4 * st + kq];                      // Linv_bb[j][k]
        if (st & 1) x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x2, 0, 0, 0);
        else x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
      }
      COMPILER_BARRIER();
#pragma unroll
      for (int r = 0; r < 4; ++r) R[(16 * w + kq + 4 * r) * PBP + 16 * b + l15] = x[r] + x2[r];
      COMPILER_BARRIER();
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = w + 4 * r;
    if (r0 + i < rows_below && k < nb) Pn[i * lda + k] = R[i * PBP + k];
  }
  if (info_dbg[7] != 0 && tid == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) info_dbg[4] = (long long)(__builtin_amdgcn_s_memtime() - t2);
}
static_assert(SPP == SPP_STAGE, "staging stride");
constexpr int DIAG_STEP_SMEM = (PB * SPP + PB * PBP + PB + PB * PB + 3 * PB * 17 + 8 * 16 * 17) * 8;   // image, panel rows, rdiag, ring, layout buffers, 16-blocks + inverses

// ---------------------------------------------------------------------------------------------
// The whole tuning objective of one hyper-parameter candidate in ONE workgroup (round 5): Cholesky
// factor of the candidate's (n + 1) x (n + 1) AUGMENTED matrix
//     [ K + s2 I   . ]        L_aug = [ L    0 ]      z = L^-1 (y - m)
//     [ (y - m)^T  c ]                [ z^T  * ]
// so that the forward solve of GP.build_posterior (gp_core.py:161-162) is finished when the factor is
// -- the log marginal likelihood (gp_core.py:222-227) needs sum(log L_ii) and z.z only.  Replaces, for
// 128 < n <= 2047 and lock-step groups of candidates (GPFitter._tuning_objective, gp_core.py:551-574),
// the batched schedule of cholesky_device: at n = 1000 x 64 that was 44 launches, two one-launch panels
// whose 1024 workgroups queue for 256 CUs, and 1.1 ms of 512-block inverses the likelihood never uses
// (profiles/r05_lml_batch_before.txt).  Here a candidate never leaves its CU:
//   left-looking over 64-column blocks j:  T_ij = A_ij - sum_{k<j} L_ik L_jk^T for the tile rows i >= j,
//   two tile rows at a time (each wave owns 16 rows of every tile: its A operand goes straight from L2 / HBM
//   into MFMA fragments, the B operand -- block row j, shared by the four waves -- through a double-buffered
//   LDS image; lmlwg_gemm); then the diagonal tile through factor64_waves and the tiles below it through the
//   16-column MFMA substitution of diag_step64_kernel; every tile of A is read once, every tile of L written once.
// The matrix is stored padded to NP = 64 ceil((n + 1) / 64) rows and columns; rows beyond n are not read
// from memory but generated (row n: y - m and the diagonal entry c = 1 + |y - m|^2 / s2 > z.z, rows
// beyond: identity), so the Gram kernel only has to fill the n x n part.
// One workgroup per CU (the factor64_waves / substitution LDS images take 141 KB): the latency-bound
// diagonal steps are NOT hidden behind another candidate's products -- the price of never waiting for
// another workgroup.
struct LmlWgArgs {
  double* K; long sK; long ld;      // padded matrices, sK doubles apart, row stride ld
  int n, nbt;                       // observations; tile rows = ceil((n + 1) / 64)
  const double* y;                  // [n]
  const double* par;                // [count] augmented diagonal entry c, then [count] prior mean m
  int count;
  double* out2;                     // [count][2]: sum(log L_ii), z.z
  long long* info;                  // [count]: 1-based index of the first failing pivot (0: none)
  // lml_team_kernel only
  int T;                            // workgroups per candidate
  int* sync;                        // [count][LMLT_SYNC_INTS], zeroed per launch: diag[j], then brow[j] (see the kernel)
  double* linvbuf;                  // [count][nbt][LMLT_LINV]: inverses of the diagonal tiles' 16 x 16 blocks, handed on
  unsigned long long* status;       // hand-off status word (SYNC_ST_*)
  int spin_limit;
#ifdef DFH_DEBUG_HOOKS
  long long* stamps = nullptr;      // dfh_debug_lmlt_stamps: [workgroup][32 columns][16] s_memrealtime (100 MHz) at the LSTAMP points
#endif
};
constexpr int LMLT_SYNC_INTS = 64, LMLT_LINV = 4 * 16 * 17;
#ifdef DFH_DEBUG_HOOKS
#define LSTAMP(a, j, e) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[((long)blockIdx.x * 32 + (j)) * 16 + (e)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
long long* g_lmlt_stamps = nullptr;
#else
#define LSTAMP(a, j, e) do {} while (0)
#endif

// this wave's 16 x 64 slice of tile (i, j), as MFMA accumulators: acc[t][r] = element (64 i + 16 w + kq + 4 r, 64 j + 16 t + l15)
__device__ __forceinline__ void lmlwg_load_tile(const LmlWgArgs& a, const double* __restrict__ Km, double mean, double cdiag,
                                                int i, int j, int w, int kq, int l15, double4_t (&acc)[4]) {
  const int n = a.n;
  const long ld = a.ld;
  if (64 * (i + 1) <= n) {                             // (uniform) every row of the tile is a row of K
    const double* p = Km + (long)(64 * i + 16 * w + kq) * ld + 64 * j + l15;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = p[(long)(4 * r) * ld + 16 * t];
  } else {
    // (unconditional loads from clamped addresses, then selects: a load under a condition is waited for on the
    //  spot, and sixteen memory latencies in a row per tile were a tenth of the kernel's time)
    double kv[4][4], yv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int gj = min(64 * j + 16 * t + l15, n - 1);
      yv[t] = a.y[gj];
#pragma unroll
      for (int r = 0; r < 4; ++r) kv[t][r] = Km[(long)min(64 * i + 16 * w + kq + 4 * r, n - 1) * ld + gj];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = 64 * i + 16 * w + kq + 4 * r, gj = 64 * j + 16 * t + l15;
        const double on_row_n = (gj < n) ? yv[t] - mean : (gj == n ? cdiag : 0.0);
        const double in_k = (gj < n) ? kv[t][r] : 0.0;
        acc[t][r] = (gi < n) ? in_k : (gi == n ? on_row_n : (gi == gj ? 1.0 : 0.0));
      }
  }
}

// acc[q] += L[tile row i0 + q * istep][0 : 64 j] L[tile row brow][0 : 64 j]^T for q < RG, this wave's 16 rows of each
// tile (brow: j, or j + 1 for the look-ahead product of the next diagonal tile).  Sixteen columns per step.
//   A operand (this wave's own rows): straight from L2 / HBM into MFMA fragments -- lane (kq, l15) holds columns
//     2 kq, 2 kq + 1 and 8 + 2 kq, 9 + 2 kq of its row (two 16-byte loads, 64 contiguous bytes per row and
//     instruction) and the four MFMAs of a step contract over the columns {c, 2 + c, 4 + c, 6 + c} + {0, 8}: any
//     assignment of columns to k-slots is a valid product as long as both operands use the same one.
//   B operand (the 64 rows of tile row brow, the same for all four waves): through a double-buffered LDS image
//     Bs[2][64][LG_BKP], 32 bytes per thread and step, one barrier per step.  (Round 5's first version had every
//     wave load all of B itself: ten loads per sixteen MFMAs at one tile row per wave, and the products ran at a
//     third of the matrix pipe's rate -- tools/dbg_lmlt.py.)
// Loads are issued NS steps ahead and unconditionally (a load under a condition makes the compiler drain the whole
// queue -- s_waitcnt vmcnt(0) -- before every step; the last steps therefore re-load the final step's operands).
// Called by all four waves together (barriers inside); ends behind a barrier: Bs is free again.
constexpr int LG_BKP = 18;                             // row stride of the B image (doubles): 16-byte aligned rows
template <int RG>
__device__ __forceinline__ void lmlwg_gemm(const double* __restrict__ Km, long ld, int j, int i0, int w, int kq, int l15,
                                           double4_t (&acc)[2][4], double* Bs, int istep = 1, int brow = -1) {
  const int nch = 4 * j;                               // (a multiple of NS)
  if (nch <= 0) return;
  if (brow < 0) brow = j;
  const int tid = threadIdx.x;
  const double* pa[RG];
#pragma unroll
  for (int q = 0; q < RG; ++q) pa[q] = Km + (long)(64 * (i0 + q * istep) + 16 * w + l15) * ld + 2 * kq;
  const double* pbg = Km + (long)(64 * brow + (tid >> 2)) * ld + 4 * (tid & 3);   // staging: row tid / 4, four columns
  double* bst = Bs + (tid >> 2) * LG_BKP + 4 * (tid & 3);
  const double* bfr = Bs + l15 * LG_BKP + 2 * kq;      // fragments: row 16 t + l15, columns 2 kq (+ 8)
  constexpr int NS = 4;
  double2_t fa[NS][RG][2], gb[NS][2];
  auto load_a = [&](int c, double2_t (&xa)[RG][2]) {
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      xa[q][0] = *reinterpret_cast<const double2_t*>(pa[q] + 16 * c);
      xa[q][1] = *reinterpret_cast<const double2_t*>(pa[q] + 16 * c + 8);
    }
  };
  auto load_b = [&](int c, double2_t (&xb)[2]) {
    xb[0] = *reinterpret_cast<const double2_t*>(pbg + 16 * c);
    xb[1] = *reinterpret_cast<const double2_t*>(pbg + 16 * c + 2);
  };
  auto stage_b = [&](const double2_t (&xb)[2], int buf) {
    *reinterpret_cast<double2_t*>(bst + buf * (64 * LG_BKP)) = xb[0];
    *reinterpret_cast<double2_t*>(bst + buf * (64 * LG_BKP) + 2) = xb[1];
  };
  auto mma = [&](const double2_t (&xa)[RG][2], int buf) {
    double2_t xb[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xb[t][0] = *reinterpret_cast<const double2_t*>(bfr + buf * (64 * LG_BKP) + 16 * t * LG_BKP);
      xb[t][1] = *reinterpret_cast<const double2_t*>(bfr + buf * (64 * LG_BKP) + 16 * t * LG_BKP + 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int q = 0; q < RG; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double av = (s & 1) ? xa[q][s >> 1].y : xa[q][s >> 1].x;
          const double bv = (s & 1) ? xb[t][s >> 1].y : xb[t][s >> 1].x;
          acc[q][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[q][t], 0, 0, 0);
        }
  };
#pragma unroll
  for (int u = 0; u < NS; ++u) { load_a(min(u, nch - 1), fa[u]); load_b(min(u, nch - 1), gb[u]); }
  stage_b(gb[0], 0);
  load_b(min(NS, nch - 1), gb[0]);
  __syncthreads();
  for (int c = 0; c < nch; c += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      // step c + u: its B image is in buffer u & 1 (NS is even), its A fragments in fa[u]
      stage_b(gb[(u + 1) % NS], (u + 1) & 1);          // the next step's image (read last in the step before this one)
      load_b(min(c + u + 1 + NS, nch - 1), gb[(u + 1) % NS]);
      mma(fa[u], u & 1);
      load_a(min(c + u + NS, nch - 1), fa[u]);
      __syncthreads();
    }
  }
}

// X = T L_jj^-T for this wave's 16 x 64 slice T (in acc), by the 16-column substitution of diag_step64_kernel
// (factor image Sp with perm16 columns, the inverses linv of its 16 x 16 diagonal blocks); X goes to the wave's
// rows Rw of the LDS row buffer and from there to G (row stride ld), a 512-byte row segment per store.
// SC1: the rows go out with write-through stores (another workgroup reads them: lml_team_kernel).
template <bool SC1 = false>
__device__ __forceinline__ void lmlwg_solve_store(const double4_t (&acc)[4], const double* Sp, const double* linv,
                                                  double* Rw, double* Tt, double* __restrict__ G, long ld, int lane) {
  const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    double4_t a1 = acc[b], a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int bp = 0; bp < b; ++bp)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const double av = Rw[l15 * PBP + 16 * bp + 4 * st + kq];                               // X_b'[i][k]
        const double bv = -Sp[(16 * b + l15) * SPP_STAGE + perm16(16 * bp + 4 * st + kq)];     // -L[16b+j][16b'+k]
        if (st & 1) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a2, 0, 0, 0);
        else a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a1, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) Tt[(kq + 4 * r) * 17 + l15] = a1[r] + a2[r];
    COMPILER_BARRIER();                              // same wave: LDS executes its operations in order
    double4_t x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const double av = Tt[l15 * 17 + 4 * st + kq];                                            // T[i][k]
      const double bv = linv[b * (16 * 17) + l15 * 17 + 4 * st + kq];                          // Linv_bb[j][k]
      if (st & 1) x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x2, 0, 0, 0);
      else x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
    }
    COMPILER_BARRIER();
#pragma unroll
    for (int r = 0; r < 4; ++r) Rw[(kq + 4 * r) * PBP + 16 * b + l15] = x[r] + x2[r];
    COMPILER_BARRIER();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (SC1) __hip_atomic_store(G + (long)i * ld + lane, Rw[i * PBP + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else G[(long)i * ld + lane] = Rw[i * PBP + lane];
  }
  COMPILER_BARRIER();                                // (the next tile's substitution overwrites Rw)
}

// The last diagonal tile of lml_wg_body, staged in Sp by the caller (ring flags zeroed, barrier passed): columns 0 .. klast
// only, no inverses; leaves the factor image in Sp (perm16 columns).  Returns this wave's first bad column or -1.
// NOT inlined, for lmlt_factor_tile's reason: next to factor64_waves in one body the register allocator put sixteen
// registers of the pivot chain into scratch.  LDS pointers formed here, from the dynamic LDS base (lml_wg_body's layout).
__device__ __attribute__((noinline)) int lmlwg_factor_last(int klast, int* ring_timeout) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* Sp = dsm;
  double* ring = dsm + PB * SPP_STAGE + PB * PBP + PB;
  double* tbuf0 = ring + PB * PB;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double av[16];
  int bad = tiny64_factor(av, lane, w, Sp, tbuf0 + (w > 0 ? (w - 1) : 0) * PB * 17, ring, klast, ring_timeout);
  bad = (bad > klast) ? -1 : bad;
  const bool active = 16 * w <= klast;
#pragma unroll
  for (int q = 0; q < 16; ++q) Sp[lane * SPP_STAGE + perm16(16 * w + q)] = active ? av[q] : 0.0;   // (columns nobody reads: defined values all the same)
  return bad;
}

// FUSED (round 6): the workgroup first builds its candidate's Gram matrix itself -- descriptor, inputs and labels as
// k_lml_tiny takes them (kernmat.hip; LmlFuse), scaled inputs in the LDS the factorisation uses later -- writes the n x n
// lower triangle to Km, and publishes {sum log L_ii, z.z, failed pivot or 0, done} per candidate the way tiny_publish does:
// a small group of mid-sized candidates (a slice sampler's call at 64 <= n <= 128) is then ONE launch with no copy at all.
struct LmlFuse {
  ExpConsts ec;
  const double* X; long ldx;       // [n x d] raw inputs (device)
  const char* blob;                // TinyCand[count] | kernel images | y[n]   (pinned host memory when direct)
  long y_off;
  double* ybuf;                    // [n] device copy of y (every workgroup writes the same values)
  double* out4;                    // [count][4]
  int direct;
};
constexpr int LMLF_LDS_DOUBLES = PB * PB + 3 * PB * 17 + 8 * 16 * 17;     // ring .. linv: free until the first factorisation

template <bool FUSED>
__device__ __forceinline__ void lml_wg_body(const LmlWgArgs& a, const LmlFuse& f) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* Sp = dsm;                                  // [64][SPP] staged diagonal tile, then the factor image (perm16 columns)
  double* R = dsm + PB * SPP_STAGE;                  // [64][65] solved rows, 16 per wave
  double* colbuf = R + PB * PBP;                     // [64] reciprocal diagonal
  double* ring = colbuf + PB;                        // [64][64] published columns of factor64_waves
  double* tbuf0 = ring + PB * PB;                    // 3 x [64][17] layout buffers, then 4 x [16][17] substitution tiles
  double* lbb = tbuf0 + 3 * PB * 17;                 // 4 x [16][17]
  double* linv = lbb + 4 * 16 * 17;                  // 4 x [16][17] inverses of the factor's 16 x 16 diagonal blocks
  double* rdiag = colbuf;
  __shared__ int s_badv[4];
  __shared__ int s_ring_timeout;
  __shared__ double s_red[8];
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
  double* __restrict__ Km = a.K + (long)c * a.sK;
  const long ld = a.ld;
  const int n = a.n, nbt = a.nbt;
  double cdiag, mean;
  if constexpr (FUSED) {
    __shared__ PartDev parts[TINY_MAX_PARTS];
    __shared__ double s_fz[4];
    const TinyCand cand = reinterpret_cast<const TinyCand*>(f.blob)[c];
    const char* image = f.blob + cand.image;
    const int P = cand.P, n_parts = cand.n_parts;
    const size_t off_bw = (sizeof(PartDev) * n_parts + 15) & ~size_t(15);
    const size_t off_cols = off_bw + ((sizeof(double) * (P ? P : 1) + 15) & ~size_t(15));
    for (int q = tid; q < n_parts * (int)(sizeof(PartDev) / sizeof(int)); q += 256)
      reinterpret_cast<int*>(parts)[q] = reinterpret_cast<const int*>(image)[q];
    const double* bw = reinterpret_cast<const double*>(image + off_bw);
    const int* cols = reinterpret_cast<const int*>(image + off_cols);
    const double* yb = reinterpret_cast<const double*>(f.blob + f.y_off);
    double* Xp = ring;                               // [n][P], then Np [n][n_parts]
    double* Np = Xp + n * P;
    double r2 = 0.0;
    for (int j = tid; j < n; j += 256) { const double v = yb[j]; f.ybuf[j] = v; r2 = fma(v - cand.mean, v - cand.mean, r2); }
    for (int idx = tid; idx < n * P; idx += 256) {
      const int row = idx / P, pc = idx - row * P;
      const int col = cols[pc];
      Xp[idx] = col >= 0 ? f.X[(long)row * f.ldx + col] / bw[pc] : 0.0;       // kernel.py:179-181
    }
    for (int off = 32; off > 0; off >>= 1) r2 += __shfl_down(r2, off, 64);
    if (lane == 0) s_fz[w] = r2;
    __syncthreads();
    for (int idx = tid; idx < n * n_parts; idx += 256) {
      const int row = idx / n_parts, part = idx - row * n_parts;
      const PartDev& pd = parts[part];
      int nreal = 0;
      for (int q = 0; q < pd.kc; ++q) nreal += cols[pd.poff + q] >= 0;
      Np[idx] = np_sumsq(Xp + row * P + pd.poff, nreal);                     // general_utils.py:66-67
    }
    __syncthreads();
    // the augmented row's diagonal entry c = 1 + |y - m|^2 / s2 > z.z (a hair of slack for the sum's rounding)
    mean = cand.mean;
    const double rr = ((s_fz[0] + s_fz[1]) + (s_fz[2] + s_fz[3])) * (1.0 + 1e-6);
    cdiag = 1.0 + rr / cand.noise;
    if (!(cand.noise > 0.0) || !(cdiag < INFINITY)) {          // (uniform) nothing bounds z.z: the host takes this candidate elsewhere
      if (tid == 0) tiny_publish(f.out4 + 4 * (long)c, f.direct != 0, NAN, NAN, -1.0, 1.0);
      return;
    }
    // K + noise I, lower triangle (gp_core.py:843)
    tiny_gram_lower(cand, parts, n_parts, Xp, P, Np, n, f.ec,
                    [&](int i, int j, double v) { Km[(long)i * ld + j] = (i == j) ? v + cand.noise : v; });
    // the matrix and the labels are out: every wave drains its stores, then all of them may read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    cdiag = a.par[c];
    mean = a.par[a.count + c];
  }
  double* Rw = R + 16 * w * PBP;
  double* Tt = tbuf0 + w * (16 * 17);
  for (int j = 0; j < nbt; ++j) {
    double4_t acc[2][4];
    // (round 6) the LAST diagonal tile is read only as far as the last observation's column: sum log L_ii runs over the
    // rows of K, and row n of the factor -- z -- is final in column k as soon as column k is.  Its factorisation stops
    // there, and when the tile holds the augmented row alone (n a multiple of 64) the whole block column is not needed.
    const int klast = (j == nbt - 1) ? n - 1 - 64 * j : 63;
    if (klast < 0) break;
    // ---- tile rows j (the diagonal tile) and j + 1 ----
    const bool two = j + 1 < nbt;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[1][t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    lmlwg_load_tile(a, Km, mean, cdiag, j, j, w, kq, l15, acc[0]);
    if (two) lmlwg_load_tile(a, Km, mean, cdiag, j + 1, j, w, kq, l15, acc[1]);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[q][t] = -acc[q][t];       // the products ADD: -T = -A + sum L L^T
    if (two) lmlwg_gemm<2>(Km, ld, j, j, w, kq, l15, acc, ring);
    else lmlwg_gemm<1>(Km, ld, j, j, w, kq, l15, acc, ring);
    // the diagonal tile -> staged block (lower triangle, zero above)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + kq + 4 * r, col = 16 * t + l15;
        Sp[row * SPP_STAGE + col] = (col <= row) ? -acc[0][t][r] : 0.0;
      }
    if (tid < PB) ring[tid * PB] = 0.0;                // row-0 entries double as the "published" flags
    if (tid == 0) s_ring_timeout = 0;
    __syncthreads();
    {
      double av[16];
      double* tbuf = tbuf0 + (w > 0 ? (w - 1) : 0) * PB * 17;
      if (klast < 63) {
        const int bad = lmlwg_factor_last(klast, &s_ring_timeout);
        if (lane == 0) s_badv[w] = bad;
      } else {
        const int bad = factor64_waves<false>(av, lane, w, Sp, tbuf, ring, lbb, linv, rdiag, &s_ring_timeout);
        if (lane == 0) s_badv[w] = bad;
#pragma unroll
        for (int q = 0; q < 16; ++q) Sp[lane * SPP_STAGE + perm16(16 * w + q)] = av[q];
      }
    }
    __syncthreads();
    const int s_bad = (s_badv[0] >= 0) ? s_badv[0] : (s_badv[1] >= 0) ? s_badv[1] : (s_badv[2] >= 0) ? s_badv[2] : s_badv[3];
    if (s_bad >= 0 || s_ring_timeout) {                // (uniform) not positive definite as it stands: the host takes the ladder
      if (tid == 0) {
        if constexpr (FUSED) tiny_publish(f.out4 + 4 * (long)c, f.direct != 0, NAN, NAN, (double)(64ll * j + (s_bad >= 0 ? s_bad : 0) + 1), 1.0);
        else a.info[c] = 64ll * j + (s_bad >= 0 ? s_bad : 0) + 1;
      }
      return;
    }
    {
      // L_jj to its place (the likelihood reads its diagonal; later block columns never read a diagonal tile)
      const int pk = perm16(lane);
      double* Ljj = Km + (long)(64 * j) * ld + 64 * j;
#pragma unroll
      for (int r = 0; r < 16; ++r) Ljj[(long)(w + 4 * r) * ld + lane] = Sp[(w + 4 * r) * SPP_STAGE + pk];
    }
    if (two) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[1][t] = -acc[1][t];
      lmlwg_solve_store(acc[1], Sp, linv, Rw, Tt, Km + (long)(64 * (j + 1) + 16 * w) * ld + 64 * j, ld, lane);
    }
    // ---- the tile rows below, two at a time ----
    for (int i0 = j + 2; i0 < nbt; i0 += 2) {
      const bool two2 = i0 + 1 < nbt;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[1][t] = (double4_t){0.0, 0.0, 0.0, 0.0};
      lmlwg_load_tile(a, Km, mean, cdiag, i0, j, w, kq, l15, acc[0]);
      if (two2) lmlwg_load_tile(a, Km, mean, cdiag, i0 + 1, j, w, kq, l15, acc[1]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = -acc[q][t];
      if (two2) lmlwg_gemm<2>(Km, ld, j, i0, w, kq, l15, acc, ring);
      else lmlwg_gemm<1>(Km, ld, j, i0, w, kq, l15, acc, ring);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q][t] = -acc[q][t];
      lmlwg_solve_store(acc[0], Sp, linv, Rw, Tt, Km + (long)(64 * i0 + 16 * w) * ld + 64 * j, ld, lane);
      if (two2) lmlwg_solve_store(acc[1], Sp, linv, Rw, Tt, Km + (long)(64 * (i0 + 1) + 16 * w) * ld + 64 * j, ld, lane);
    }
    // block column j is out: every wave drains its stores, then all of them may read it as an operand
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // sum(log L_ii) over the rows of K, z.z over row n (fixed order: deterministic)
  double ldv = 0.0, dt = 0.0;
  for (int i = tid; i < n; i += 256) {
    ldv += log(Km[(long)i * ld + i]);
    const double z = Km[(long)n * ld + i];
    dt = fma(z, z, dt);
  }
  for (int off = 32; off > 0; off >>= 1) { ldv += __shfl_down(ldv, off, 64); dt += __shfl_down(dt, off, 64); }
  if (lane == 0) { s_red[w] = ldv; s_red[4 + w] = dt; }
  __syncthreads();
  if (tid == 0) {
    if constexpr (FUSED) {
      tiny_publish(f.out4 + 4 * (long)c, f.direct != 0, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]),
                   (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]), 0.0, 1.0);
    } else {
      a.out2[2 * c] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
      a.out2[2 * c + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }
  }
}

__global__ __launch_bounds__(256, 1) void lml_wg_kernel(LmlWgArgs a) { lml_wg_body<false>(a, LmlFuse()); }
__global__ __launch_bounds__(256, 1) void lml_wgf_kernel(LmlWgArgs a, LmlFuse f) { lml_wg_body<true>(a, f); }

// The same objective with a TEAM of T workgroups per candidate (few candidates: one workgroup each would leave
// most of the device idle and take 3 ms at n = 1000).  Tile row i belongs to member i mod T; in block column j
//   every member with rows >= j:  waits for brow[j] (tile row j -- the B operand -- is complete up to column
//                                 j - 1), accumulates its tiles of the column two at a time;
//   the owner of row j:           takes the diagonal tile first, factors it, hands L_jj (in its place) and the
//                                 inverses of its 16 x 16 blocks on under diag[j];
//   the others:                   wait for diag[j] after their first products, fetch that image, substitute;
//   the owner of row j + 1:       announces brow[j + 1] as soon as its tile (j + 1, j) is out.
// Hand-offs as in the one-launch panel: write-through (sc1) stores of whatever another member reads, every
// wave drains its stores, barrier, relaxed flag; the reader polls, takes ONE agent-scope acquire (its CU's L1)
// and reads with plain loads -- no line is ever read by a member before its final contents are written, so no
// stale copy can sit in another XCD's L2.  Every wait is bounded (status word -> the host repeats the group with
// one workgroup per candidate).  A failed pivot is handed on as diag[j] = 2: every member that still has rows
// waits for exactly that flag and leaves.  Nothing here assumes where a workgroup runs; co-residency of the
// T * count <= CUs workgroups is what makes it fast, the bounded waits are what makes it safe.
__device__ __forceinline__ int lmlt_wait(const int* p, const LmlWgArgs& a, int* s_val) {
  if (threadIdx.x == 0) {
    int spins = 0, v;
    while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
      if (++spins > a.spin_limit) { atomicOr(a.status, (unsigned long long)SYNC_ST_FUSED); v = -1; break; }
      if ((spins & 63) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { v = -1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    *s_val = v;
  }
  __syncthreads();
  const int v = *s_val;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // this CU's L1 holds nothing older than the flag
  __syncthreads();                                     // (s_val may be rewritten by the next wait)
  return v;
}
__device__ __forceinline__ void lmlt_publish(int* p, int v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every wave: its write-through stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS images of the one-workgroup objective (dynamic LDS, the layout of diag_step64_kernel, two words behind it)
struct LmltLds { double *Sp, *R, *ring, *tbuf0, *lbb, *linv, *rdiag; int *badv, *ring_timeout; };
constexpr int LMLT_SMEM = DIAG_STEP_SMEM + 64;
__device__ __forceinline__ LmltLds lmlt_lds() {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  LmltLds L;
  L.Sp = dsm;                                        // [64][SPP] staged diagonal tile, then the factor image (perm16 columns)
  L.R = dsm + PB * SPP_STAGE;                        // [64][65] solved rows, 16 per wave
  double* colbuf = L.R + PB * PBP;                   // [64] reciprocal diagonal
  L.ring = colbuf + PB;                              // [64][64] published columns of factor64_waves
  L.tbuf0 = L.ring + PB * PB;                        // 3 x [64][17] layout buffers, then 4 x [16][17] substitution tiles
  L.lbb = L.tbuf0 + 3 * PB * 17;                     // 4 x [16][17]
  L.linv = L.lbb + 4 * 16 * 17;                      // 4 x [16][17] inverses of the factor's 16 x 16 diagonal blocks
  L.rdiag = colbuf;
  L.badv = reinterpret_cast<int*>(dsm + DIAG_STEP_SMEM / 8);
  L.ring_timeout = L.badv + 4;
  return L;
}

// Stage a diagonal tile (this wave's 16 x 64 slice in d0 .. d3), factor it (factor64_waves) and leave the factor image
// (perm16 columns) in Sp and the inverses of its 16 x 16 blocks in linv.  Returns -1, or the first bad column (64: an
// LDS ring flag never came up).
// NOT inlined: inside the team kernel's loop nest the register allocator spilled two dwords of every pivot step of
// factor64_waves to scratch -- a memory round trip per column on the one chain that is pure latency: 53 us per tile
// instead of ~10 (tools/dbg_lmlt.py).  As a function of its own the step has the register file to itself; its LDS
// pointers are formed here, from the dynamic LDS base, so that they stay LDS pointers (passed as arguments they
// would be generic ones: flat_load instead of ds_read).
__device__ __attribute__((noinline)) int lmlt_factor_tile(double4_t d0, double4_t d1, double4_t d2, double4_t d3) {
  const LmltLds L = lmlt_lds();
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * w + kq + 4 * r;
    L.Sp[row * SPP_STAGE + l15] = (l15 <= row) ? d0[r] : 0.0;
    L.Sp[row * SPP_STAGE + 16 + l15] = (16 + l15 <= row) ? d1[r] : 0.0;
    L.Sp[row * SPP_STAGE + 32 + l15] = (32 + l15 <= row) ? d2[r] : 0.0;
    L.Sp[row * SPP_STAGE + 48 + l15] = (48 + l15 <= row) ? d3[r] : 0.0;
  }
  if (tid < PB) L.ring[tid * PB] = 0.0;                // row-0 entries double as the "published" flags
  if (tid == 0) *L.ring_timeout = 0;
  __syncthreads();
  {
    double av[16];
    double* tbuf = L.tbuf0 + (w > 0 ? (w - 1) : 0) * PB * 17;
    const int bad = factor64_waves<false>(av, lane, w, L.Sp, tbuf, L.ring, L.lbb, L.linv, L.rdiag, L.ring_timeout);
    if (lane == 0) L.badv[w] = bad;
#pragma unroll
    for (int q = 0; q < 16; ++q) L.Sp[lane * SPP_STAGE + perm16(16 * w + q)] = av[q];
  }
  __syncthreads();
  const int s_bad = (L.badv[0] >= 0) ? L.badv[0] : (L.badv[1] >= 0) ? L.badv[1] : (L.badv[2] >= 0) ? L.badv[2] : L.badv[3];
  return *L.ring_timeout ? 64 : s_bad;
}

// ... and hand it on: factor image and inverses to global memory with write-through stores, *flag = 1 -- or 2 after a
// failed pivot (then info is set and the caller leaves).  Returns false on failure.
__device__ __forceinline__ bool lmlt_factor_publish(const LmlWgArgs& a, const LmltLds& L, const double4_t (&dacc)[4],
                                                    double* __restrict__ Km, long ld, int jj, int c, double* linvg,
                                                    int* flag) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  LSTAMP(a, jj, 10);
  const int s_bad = lmlt_factor_tile(dacc[0], dacc[1], dacc[2], dacc[3]);
  LSTAMP(a, jj, 13);
  if (s_bad >= 0) {
    if (tid == 0) a.info[c] = 64ll * jj + (s_bad & 63) + 1;
    lmlt_publish(flag, 2);                             // nobody may hang: the waiters leave on 2
    return false;
  }
  const int pk = perm16(lane);
  double* Ljj = Km + (long)(64 * jj) * ld + 64 * jj;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __hip_atomic_store(Ljj + (long)(w + 4 * r) * ld + lane, L.Sp[(w + 4 * r) * SPP_STAGE + pk], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid; i < LMLT_LINV; i += 256)
    __hip_atomic_store(linvg + (long)jj * LMLT_LINV + i, L.linv[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  LSTAMP(a, jj, 14);
  lmlt_publish(flag, 1);
  LSTAMP(a, jj, 15);
  return true;
}

// factor image of block column jj (announced and waited for before) from global memory into Sp / linv
__device__ __forceinline__ void lmlt_fetch_image(const LmltLds& L, const double* __restrict__ Km, long ld, int jj,
                                                 const double* linvg) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const double* Ljj = Km + (long)(64 * jj) * ld + 64 * jj;
  const int pk = perm16(lane);
  double pre[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) pre[r] = Ljj[(long)(w + 4 * r) * ld + lane];
#pragma unroll
  for (int r = 0; r < 16; ++r) L.Sp[(w + 4 * r) * SPP_STAGE + pk] = pre[r];
  for (int i = tid; i < LMLT_LINV; i += 256) L.linv[i] = linvg[(long)jj * LMLT_LINV + i];
  __syncthreads();
}

// Look-ahead (the member that owns tile row j + 1, while block column j is being finished): the NEXT diagonal tile
// is accumulated over the columns before j while the member waits for L_jj anyway, takes the product with the
// just-solved tile (j + 1, j) straight from the LDS row buffer, and is factored and announced before the member
// turns to the rest of its rows of column j -- so that a block column's critical path is
//     L_jj announced -> fetch -> substitution of ONE tile -> K = 64 product -> 64 x 64 factorisation -> announce
// whatever j (the left-looking products over 64 j columns had been on it: 1.7 j us per column).
__global__ __launch_bounds__(256, 1) void lml_team_kernel(LmlWgArgs a) {
  const LmltLds L = lmlt_lds();
  __shared__ int s_wait;
  __shared__ double s_red[8];
  const int T = a.T;
  const int c = blockIdx.x / T, t = blockIdx.x - c * T;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
  double* __restrict__ Km = a.K + (long)c * a.sK;
  const long ld = a.ld;
  const int n = a.n, nbt = a.nbt;
  const double cdiag = a.par[c], mean = a.par[a.count + c];
  int* dflag = a.sync + (long)c * LMLT_SYNC_INTS;
  int* bflag = dflag + 32;
  double* linvg = a.linvbuf + (long)c * nbt * LMLT_LINV;
  double* Rw = L.R + 16 * w * PBP;
  double* Tt = L.tbuf0 + w * (16 * 17);
  int lds_image = -1;                                  // block column whose factor image Sp / linv hold
  int factored = -1;                                   // last diagonal tile this member has factored and announced
  double4_t dacc[2][4];                                // [0]: the look-ahead diagonal tile ([1] unused: lmlwg_gemm's signature)
  for (int j = 0; j < nbt; ++j) {
    const int own_j = j % T;
    const bool owner = own_j == t;
    const bool next_owner = (j + 1 < nbt) && ((j + 1) % T == t);
    const int i_first = j + ((t - own_j + T) % T);     // this member's first tile row >= j
    if (i_first >= nbt) break;                         // no rows left in this or any later column
    LSTAMP(a, j, 0);
    if (next_owner) {
      // look-ahead, part 1: the next diagonal tile over the columns before j (tile row j + 1 on both sides: this member's
      // own tiles) -- BEFORE the wait for tile row j, whose owner is busy with this column's diagonal tile right now
      lmlwg_load_tile(a, Km, mean, cdiag, j + 1, j + 1, w, kq, l15, dacc[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) dacc[0][q] = -dacc[0][q];
      lmlwg_gemm<1>(Km, ld, j, j + 1, w, kq, l15, dacc, L.ring, T, j + 1);
      LSTAMP(a, j, 3);
    }
    if (j > 0 && !owner) {
      // tile row j (the B operand of this column) was completed by its owner in column j - 1
      if (lmlt_wait(bflag + j, a, &s_wait) != 1) return;
    }
    LSTAMP(a, j, 1);
    if (owner && factored < j) {
      // (j = 0, or a team of one: no look-ahead has prepared this tile)
      lmlwg_load_tile(a, Km, mean, cdiag, j, j, w, kq, l15, dacc[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) dacc[0][q] = -dacc[0][q];
      lmlwg_gemm<1>(Km, ld, j, j, w, kq, l15, dacc, L.ring, T);
#pragma unroll
      for (int q = 0; q < 4; ++q) dacc[0][q] = -dacc[0][q];
      __syncthreads();
      if (!lmlt_factor_publish(a, L, dacc[0], Km, ld, j, c, linvg, dflag + j)) return;
      factored = j; lds_image = j;
    }
    bool waited = owner;                               // diag[j] seen (the owner wrote it)
    for (int i0 = owner ? j + T : i_first, inext; i0 < nbt; i0 = inext) {
      const int i1 = i0 + T;
      const bool two = i1 < nbt;
      inext = i1 + T;
      const bool la = next_owner && i0 == j + 1;       // this group starts with tile (j + 1, j)
      double4_t acc[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[1][q] = (double4_t){0.0, 0.0, 0.0, 0.0};
      lmlwg_load_tile(a, Km, mean, cdiag, i0, j, w, kq, l15, acc[0]);
      if (two) lmlwg_load_tile(a, Km, mean, cdiag, i1, j, w, kq, l15, acc[1]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) acc[q][q2] = -acc[q][q2];
      if (two) lmlwg_gemm<2>(Km, ld, j, i0, w, kq, l15, acc, L.ring, T);
      else lmlwg_gemm<1>(Km, ld, j, i0, w, kq, l15, acc, L.ring, T);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) acc[q][q2] = -acc[q][q2];
      if (i0 == (owner ? j + T : i_first)) LSTAMP(a, j, 2);
      if (!waited) {
        const int v = lmlt_wait(dflag + j, a, &s_wait);
        if (v != 1) return;                            // 2: not positive definite (reported by the owner); -1: gave up
        waited = true;
        LSTAMP(a, j, 4);
      }
      if (lds_image != j) { lmlt_fetch_image(L, Km, ld, j, linvg); lds_image = j; }
      if (i0 == (owner ? j + T : i_first)) LSTAMP(a, j, 5);
      lmlwg_solve_store<true>(acc[0], L.Sp, L.linv, Rw, Tt, Km + (long)(64 * i0 + 16 * w) * ld + 64 * j, ld, lane);
      if (i0 == (owner ? j + T : i_first)) LSTAMP(a, j, 6);
      if (la) {
        // -dacc += X X^T, X = tile (j + 1, j): this wave's rows and all 64 rows from the LDS row buffer
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const double av = Rw[l15 * PBP + 4 * ks + kq];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            dacc[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, L.R[(16 * q + l15) * PBP + 4 * ks + kq], dacc[0][q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dacc[0][q] = -dacc[0][q];
        // tile row j + 1 is complete up to column j: the next column's B operand (the barrier inside also frees the row buffer)
        lmlt_publish(bflag + j + 1, 1);
        LSTAMP(a, j, 7);
      }
      if (two) lmlwg_solve_store<true>(acc[1], L.Sp, L.linv, Rw, Tt, Km + (long)(64 * i1 + 16 * w) * ld + 64 * j, ld, lane);
      if (la) {
        // the next diagonal tile now, ahead of this member's other rows of column j
        __syncthreads();                               // every wave is done with the image of column j
        if (!lmlt_factor_publish(a, L, dacc[0], Km, ld, j + 1, c, linvg, dflag + j + 1)) return;
        factored = j + 1; lds_image = j + 1;
        LSTAMP(a, j, 8);
      }
    }
    // this member's tiles of column j are out: its own waves may read them as operands
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    LSTAMP(a, j, 9);
  }
  // the owner of the last tile row holds row n (z) and has waited for every diagonal tile
  if (t != (nbt - 1) % T) return;
  double ldv = 0.0, dt = 0.0;
  for (int i = tid; i < n; i += 256) {
    ldv += log(Km[(long)i * ld + i]);
    const double z = Km[(long)n * ld + i];
    dt = fma(z, z, dt);
  }
  for (int off = 32; off > 0; off >>= 1) { ldv += __shfl_down(ldv, off, 64); dt += __shfl_down(dt, off, 64); }
  if (lane == 0) { s_red[w] = ldv; s_red[4 + w] = dt; }
  __syncthreads();
  if (tid == 0) {
    a.out2[2 * c] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    a.out2[2 * c + 1] = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
  }
}

// ---------------------------------------------------------------------------------------------
// Panel strips.  Once the 512 x 512 diagonal block of a panel is factored, the rows below it are
//     L21 = A21 L11^-T ,
// and a strip of 64 of those rows needs nothing from any other strip: workgroup s keeps its
// 64 x 512 strip in the MFMA accumulators (16 rows x 512 columns per wave, 128 doubles per lane)
// and runs the eight 64-column steps locally --
//     X_j = (A_j - sum_{i<j} X_i L_ji^T) L_jj^-T      (the sum is already in the accumulators)
//     A_c -= X_j L_cj^T  for the later blocks c > j   (right-looking inside the strip)
// -- with the SAME row solve as diag_step64_kernel (16-column sub-blocks, the inverses of the
// 16 x 16 diagonal blocks exported by the workgroup that factored L_jj).  One launch replaces, for
// the rows below the diagonal block, the eight pivot steps (each workgroup re-factoring the pivot
// block) and the eight K = 64 updates that stream the rows x 448 panel through HBM.  Used for
// lock-step batches, where those two are what the pivot steps cost (cholesky_device); for a single
// matrix the chain is bound by launch latency and the strips do not pay.
// One wave per SIMD (the strip fills the register file): the 28 updates run at the fp64 matrix
// pipe's rate (64 cycles per MFMA), the row solves and the tile hand-over are latency -- 134 us per
// strip against 61 us of MFMA time.  Fully unrolled over (j, c): the accumulator index has to be
// static.
// LDS strides: an MFMA operand read takes element (row l15, column 4 st + kq) of a tile; with a
// row stride = 4 mod 32 doubles the 64 lanes cover all banks exactly twice (the minimum for 8 B).
constexpr int SK_LD = 68;      // 64-column tiles
constexpr int SK_TD = 20;      // 16-column tiles
constexpr int STRIP_SMEM = (2 * PB * SK_LD + 4 * 16 * SK_LD + 4 * 16 * SK_TD + 4 * 16 * SK_TD) * 8;

// The 36 tiles of L a strip consumes, in order: for j = 0..7 the diagonal factor block L_jj (with the
// inverses of its 16 x 16 diagonal blocks), then the sub-diagonal blocks L_cj, c = j+1..7.  They are
// fetched two tiles ahead into registers and handed over through two LDS buffers, one barrier per
// tile: the blocks were written by other XCDs a moment ago, every fetch is a full fabric round trip.
constexpr int SK_TILES = 36;
constexpr int sk_tile_j(int q) { int j = 0; while (q >= 8 - j) { q -= 8 - j; ++j; } return j; }
constexpr int sk_tile_c(int q) { int j = 0; while (q >= 8 - j) { q -= 8 - j; ++j; } return j + q; }

struct StripCtx {
  const double* Dblk; long lda; const double* Lfac; const double* Linv16;
  double* Pw; long rows_left;
  double* buf;        // 2 x [64][SK_LD]
  double* Xw; double* Tt; double* li;
};

template <int Q>
__device__ __forceinline__ void strip_fetch(const StripCtx& s, double (&pre)[16], double (&prei)[4]) {
  constexpr int J = sk_tile_j(Q), C = sk_tile_c(Q);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if constexpr (C == J) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = s.Lfac[J * PB * PB + (w + 4 * r) * PB + lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = tid + 256 * r;                   // element of the [4][16][16] inverses
      prei[r] = s.Linv16[J * (4 * 16 * 17) + (i >> 8) * (16 * 17) + ((i >> 4) & 15) * 17 + (i & 15)];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = s.Dblk[(long)(C * PB + w + 4 * r) * s.lda + J * PB + lane];
  }
}

template <int Q>
__device__ __forceinline__ void strip_tile(const StripCtx& s, double4_t (&acc)[32], double (&xa)[16],
                                           double (&pre)[2][16], double (&prei)[2][4]) {
  constexpr int J = sk_tile_j(Q), C = sk_tile_c(Q);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
  double* L = s.buf + (Q & 1) * (PB * SK_LD);
  // hand the prefetched tile over (the buffer was last read two tiles ago: one barrier in between)
#pragma unroll
  for (int r = 0; r < 16; ++r) L[(w + 4 * r) * SK_LD + lane] = pre[Q & 1][r];
  if constexpr (C == J) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = tid + 256 * r;
      s.li[(i >> 8) * (16 * SK_TD) + ((i >> 4) & 15) * SK_TD + (i & 15)] = prei[Q & 1][r];
    }
  }
  if constexpr (Q + 2 < SK_TILES) strip_fetch<Q + 2>(s, pre[Q & 1], prei[Q & 1]);
  __syncthreads();
  if constexpr (C == J) {
    // ---- row solve of block J: X_b = (T_b - sum_{b'<b} X_b' L[b][b']^T) Linv_bb^T, b = 0..3 ----
    double* Xw = s.Xw; double* Tt = s.Tt;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double4_t a1 = acc[4 * J + b], a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int bp = 0; bp < b; ++bp)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const double av = Xw[l15 * SK_LD + 16 * bp + 4 * st + kq];
          const double bv = -L[(16 * b + l15) * SK_LD + 16 * bp + 4 * st + kq];
          if (st & 1) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a2, 0, 0, 0);
          else a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a1, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) Tt[(kq + 4 * r) * SK_TD + l15] = a1[r] + a2[r];
      COMPILER_BARRIER();                            // same wave: LDS executes its operations in order
      double4_t x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const double av = Tt[l15 * SK_TD + 4 * st + kq];
        const double bv = s.li[b * (16 * SK_TD) + l15 * SK_TD + 4 * st + kq];
        if (st & 1) x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x2, 0, 0, 0);
        else x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
      }
      COMPILER_BARRIER();
#pragma unroll
      for (int r = 0; r < 4; ++r) Xw[(kq + 4 * r) * SK_LD + 16 * b + l15] = x[r] + x2[r];
      COMPILER_BARRIER();
    }
    // ---- the solved block is final: out to HBM, a full 512-byte row segment per store ----
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < s.rows_left) s.Pw[(long)i * s.lda + J * PB + lane] = Xw[i * SK_LD + lane];
    if constexpr (J < 7) {
#pragma unroll
      for (int st = 0; st < 16; ++st) xa[st] = -Xw[l15 * SK_LD + 4 * st + kq];   // -X_j: A operand of the later blocks
    }
  } else {
    // one wave per SIMD: nothing else hides the LDS latency, so the B operands are read a few MFMAs
    // ahead of their use (software pipeline pinned with scheduling groups; reading further ahead
    // only adds spills: 152 spilled VGPRs at sixteen ahead, 61 at four, same time)
    constexpr int AHEAD = 4;
    double bq[64];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) bq[i] = L[(16 * (i & 3) + l15) * SK_LD + 4 * (i >> 2) + kq];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (i + AHEAD < 64) bq[i + AHEAD] = L[(16 * ((i + AHEAD) & 3) + l15) * SK_LD + 4 * ((i + AHEAD) >> 2) + kq];
      acc[4 * C + (i & 3)] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i >> 2], bq[i], acc[4 * C + (i & 3)], 0, 0, 0);
      if (i + AHEAD < 64) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one DS read ...
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // ... one MFMA
    }
  }
  if constexpr (Q + 1 < SK_TILES) strip_tile<Q + 1>(s, acc, xa, pre, prei);
}

__global__ __launch_bounds__(256, 1) void panel_strip_kernel(const double* __restrict__ Dblk, long lda,
                                                             const double* __restrict__ Lfac,
                                                             const double* __restrict__ Linv16,
                                                             double* __restrict__ P, int rows, long strideD,
                                                             long strideL, long strideI) {
  extern __shared__ __attribute__((aligned(16))) double ssm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
  StripCtx s;
  s.Dblk = Dblk + (long)blockIdx.y * strideD; s.lda = lda;
  s.Lfac = Lfac + (long)blockIdx.y * strideL;
  s.Linv16 = Linv16 + (long)blockIdx.y * strideI;
  const long r0 = (long)blockIdx.x * PB + 16 * w;    // this wave's first row
  s.Pw = P + (long)blockIdx.y * strideD + r0 * lda;
  s.rows_left = (long)rows - r0;
  s.buf = ssm;                                                       // 2 x [64][SK_LD] tiles of L
  s.Xw = ssm + 2 * PB * SK_LD + w * (16 * SK_LD);                    // per wave: [16][SK_LD] the solved block X_j
  s.Tt = ssm + 2 * PB * SK_LD + 4 * 16 * SK_LD + w * (16 * SK_TD);   // per wave: [16][SK_TD]
  s.li = ssm + 2 * PB * SK_LD + 4 * 16 * SK_LD + 4 * 16 * SK_TD;     // [4][16][SK_TD] inverses
  double pre[2][16], prei[2][4], xa[16];
  strip_fetch<0>(s, pre[0], prei[0]);
  strip_fetch<1>(s, pre[1], prei[1]);
  double4_t acc[32];
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[t][r] = (kq + 4 * r < s.rows_left) ? s.Pw[(long)(kq + 4 * r) * lda + 16 * t + l15] : 0.0;
  strip_tile<0>(s, acc, xa, pre, prei);
}

// ---------------------------------------------------------------------------------------------
// One launch per panel (single matrices).  Workgroup g < 8 owns the 64-row strip g of the 512 x 512
// diagonal block, workgroup g >= 8 a strip of the rows below it; every strip lives in the MFMA
// accumulators as in panel_strip_kernel.  The eight diagonal strips form the dependent chain,
//   strip s:  for j < s: wait for L_jj, X_sj = (A_sj - ...) L_jj^-T, update the blocks (j, s];
//             then factor its own 64 x 64 diagonal block (factor64_waves) and publish L_ss,
// handing data on through HBM with two sets of per-matrix counters: flag[s] = epoch once L_ss and
// the inverses of its 16 x 16 blocks are out, prog[s] = j + 1 once X_sj is.  Readers spin on the
// counter (relaxed agent-scope loads, then one acquire fence), writers publish with a barrier and a
// release store.  Workgroups only ever wait for workgroups with a smaller index of the same launch,
// which the dispatcher starts first.  This replaces the sixteen dependent launches of a panel
// (eight pivot steps, eight K = 64 updates) whose gaps and HBM round trips were the factorisation's
// chain.
constexpr int FUSED_SYNC_INTS = 24;
struct FusedArgs {
  double* D; long lda;
  double* Lfac; double* Linv16;
  int* sync;                 // per matrix: flag[8], prog[8], flag2[8] (the inverse of L_ss's last 16 x 16 block, published late)
  int epoch;
  int nbk;                   // order of the diagonal block (512, or less for the last panel: identity padding)
  int rows_below;            // rows under the diagonal block (only below a full one)
  long long* info; long pivot_base;
  long strideD, strideL, strideI;
  unsigned long long* status;   // hand-off status word (SYNC_ST_*)
  int spin_limit;               // polls before a wait gives up (and reports)
  // resident look-ahead schedule (cholesky_device): the launch announces every workgroup that has
  // started in *resident, and -- wait_ptr != null -- only then waits for *wait_ptr >= wait_target
  // (the trailing update of another stream has written the diagonal block) before touching the matrix
  int* resident; const int* wait_ptr; int wait_target;
  // hand-off protocol inside the launch.  0: plain stores + release fence (buffer_wbl2: the XCD's whole L2 is
  // written back, dirty C tiles of a trailing update running beside the panel included) / acquire fence
  // (buffer_inv) + plain loads.  1: what is handed over goes out with write-through stores and comes in
  // with agent-coherent loads (sc1), the flags are relaxed: no cache-wide maintenance on the chain.
  int sc1;
  int prog_sleep;               // s_sleep argument of the polls that are not on the chain (waits for another strip's X_cJ)
  int g0;                       // strip index of the launch's first workgroup (0; or 8: a launch of the rows-below strips alone)
#ifdef DFH_DEBUG_HOOKS
  long long* stamps = nullptr;  // dfh_debug_panel_stamps: [strip][64] s_memrealtime (100 MHz) at the points marked FSTAMP
#endif
};
#ifdef DFH_DEBUG_HOOKS
#define FSTAMP(a, g, i) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[(g) * 64 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FSTAMP(a, g, i) do {} while (0)
#endif

// Bounded: a workgroup only ever waits for workgroups of the same launch with a smaller index (started
// before it by the dispatcher as observed, not by contract) or for another launch that needs none of
// its resources; should a wait expire all the same, the status word says so, the strip goes on with
// whatever it finds (nobody hangs) and the host repeats the factorisation without hand-offs.
__device__ __forceinline__ void fused_wait(const int* p, int target, const FusedArgs& a, unsigned what = SYNC_ST_FUSED) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > a.spin_limit) { atomicOr(a.status, (unsigned long long)what); break; }
      // somebody else has already given up: this factorisation is going to be repeated anyway
      if ((spins & 63) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // later loads see what the writer released
}

__device__ __forceinline__ void fused_publish(int* p, int value) {
  __syncthreads();                                    // every wave's stores are issued and acknowledged
  if (threadIdx.x == 0) __hip_atomic_store(p, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// the sc1 protocol (FusedArgs::sc1): data with write-through stores / agent-coherent loads ...
__device__ __forceinline__ void st_out(double* p, double v, bool sc1) {
  if (sc1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
__device__ __forceinline__ double ld_in(const double* p, bool sc1) {
  return sc1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// ... the flag goes up once every wave's write-through stores have been acknowledged (s_barrier alone does not
// wait for outstanding stores on gfx950: each wave drains its own first), and is read without a fence
__device__ __forceinline__ void fused_publish_sc1(int* p, int value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (relaxed: true for the waits off the chain -- every poll is a round trip to the memory side of the fabric, and
//  sixty strips polling one line at full rate slow the loads and stores of the strip that IS on the chain)
__device__ __forceinline__ void fused_wait_sc1(const int* p, int target, const FusedArgs& a, bool relaxed = false,
                                               unsigned what = SYNC_ST_FUSED) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > a.spin_limit) { atomicOr(a.status, (unsigned long long)what); break; }
      if ((spins & 63) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      if (relaxed) {
        for (int z = 0; z < a.prog_sleep; ++z) __builtin_amdgcn_s_sleep(1);
      } else {
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

template <int J>
__device__ __forceinline__ void fused_step(const FusedArgs& a, double4_t (&acc)[32], int s, bool diag, double* Rw,
                                           long rows_left, double* ssm) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
  double* Lj = ssm;                                                // [64][SK_LD]
  double* Lc = ssm + PB * SK_LD;                                   // [64][SK_LD]
  double* Xall = ssm + 2 * PB * SK_LD;                             // [64][SK_LD]: the four waves' solved rows
  double* Xw = Xall + w * (16 * SK_LD);
  double* Tt = ssm + 2 * PB * SK_LD + 4 * 16 * SK_LD + w * (16 * SK_TD);
  double* li = ssm + 2 * PB * SK_LD + 4 * 16 * SK_LD + 4 * 16 * SK_TD;
  int* flag = a.sync; int* prog = a.sync + 8;
  if (J < s) {
    fused_wait(flag + J, a.epoch, a);
    FSTAMP(a, blockIdx.x + a.g0, 1 + 4 * J);
#pragma unroll
    for (int r = 0; r < 16; ++r) Lj[(w + 4 * r) * SK_LD + lane] = a.Lfac[J * PB * PB + (w + 4 * r) * PB + lane];
    for (int i = tid; i < 4 * 16 * 16; i += 256)
      li[(i >> 8) * (16 * SK_TD) + ((i >> 4) & 15) * SK_TD + (i & 15)] =
          a.Linv16[J * (4 * 16 * 17) + (i >> 8) * (16 * 17) + ((i >> 4) & 15) * 17 + (i & 15)];
    __syncthreads();
    FSTAMP(a, blockIdx.x + a.g0, 2 + 4 * J);
    // ---- row solve of block J (as in panel_strip_kernel) ----
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double4_t a1 = acc[4 * J + b], a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int bp = 0; bp < b; ++bp)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const double av = Xw[l15 * SK_LD + 16 * bp + 4 * st + kq];
          const double bv = -Lj[(16 * b + l15) * SK_LD + 16 * bp + 4 * st + kq];
          if (st & 1) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a2, 0, 0, 0);
          else a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a1, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) Tt[(kq + 4 * r) * SK_TD + l15] = a1[r] + a2[r];
      COMPILER_BARRIER();
      double4_t x = {0.0, 0.0, 0.0, 0.0}, x2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const double av = Tt[l15 * SK_TD + 4 * st + kq];
        const double bv = li[b * (16 * SK_TD) + l15 * SK_TD + 4 * st + kq];
        if (st & 1) x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x2, 0, 0, 0);
        else x = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, x, 0, 0, 0);
      }
      COMPILER_BARRIER();
#pragma unroll
      for (int r = 0; r < 4; ++r) Xw[(kq + 4 * r) * SK_LD + 16 * b + l15] = x[r] + x2[r];
      COMPILER_BARRIER();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < rows_left) Rw[(long)i * a.lda + J * PB + lane] = Xw[i * SK_LD + lane];
    if (diag) fused_publish(prog + s, a.epoch * 16 + J + 1);   // X_sJ is out (later diagonal strips and the rows below read it)
    FSTAMP(a, blockIdx.x + a.g0, 3 + 4 * J);
    double xa[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) xa[st] = -Xw[l15 * SK_LD + 4 * st + kq];
    const int c_hi = diag ? s : 7;                     // last block this strip still needs
#pragma unroll
    for (int c = J + 1; c < 8; ++c) {
      if (c > c_hi) break;
      const double* L;
      if (diag && c == s) {
        __syncthreads();                               // all four waves' rows of X_sJ are in Xall
        L = Xall;                                      // own diagonal block: A_ss -= X_sJ X_sJ^T
      } else {
        fused_wait(prog + c, a.epoch * 16 + J + 1, a); // strip c has published X_cJ (its barrier also frees Lc)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Lc[(w + 4 * r) * SK_LD + lane] = a.D[(long)(c * PB + w + 4 * r) * a.lda + J * PB + lane];
        __syncthreads();
        L = Lc;
      }
#pragma unroll
      for (int i = 0; i < 64; ++i)
        acc[4 * c + (i & 3)] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[i >> 2], L[(16 * (i & 3) + l15) * SK_LD + 4 * (i >> 2) + kq],
                                                                    acc[4 * c + (i & 3)], 0, 0, 0);
    }
    __syncthreads();                                   // Xall / Lj / li are free for the next step
    FSTAMP(a, blockIdx.x + a.g0, 4 + 4 * J);
  }
}

// Round 4: the same step with the strip held TRANSPOSED in the accumulators,
//     accT[t][r] of lane (kq, l15) = strip element (row l15 of the wave's 16, column 16 t + kq + 4 r),
// i.e. the D layout of v_mfma_f64_16x16x4 for the transposed tile: D[kq + 4 r][l15] = T[l15][kq + 4 r].  In
// that layout an accumulator tile IS the B operand of a product that contracts over the strip's columns
// (B[k][n], k = kq + 4 r for the r-th of four MFMAs, n = l15), so
//     X_b^T = Linv_bb T_b^T,     T_b'^T -= L_b'b X_b^T  (b' > b, right-looking),     A_c^T -= L_cJ X^T
// take their A operands (rows of L / Linv, one ds_read_b64 per lane) from LDS and their B operands straight
// from registers: the row solve no longer changes layout through LDS between its four 16-column stages.
// Measured on the row-per-accumulator form (tools/dbg_panel.py, 100 MHz stamps): 4.25 us of a 22.5 us hop
// were this solve -- 40 MFMAs at ~200 cycles each, eight LDS write -> read round trips.
// acc^T[4 C + i] -= L_i X^T over the k-steps [KS0, KS1) of a 64-column block (k-step ks: columns 4 ks + kq of
// X = -xn), i = 0 .. 3: the order of panel_strip_kernel's update (k-step outer, the four tiles inner, ONE
// accumulator chain per tile), with the A operands read AHEAD MFMAs early: one wave per SIMD, nothing else
// hides the LDS latency.
// (C: the block; a compile-time constant at every call site after unrolling -- the accumulator index must be static)
template <int KS0, int KS1>
__device__ __forceinline__ void strip_update_t(double4_t (&acc)[32], const int C, const double* L, const double4_t (&xn)[4], int l15, int kq) {
  constexpr int N = (KS1 - KS0) * 4, AHEAD = 6;
  double aq[N];
#pragma unroll
  for (int q = 0; q < AHEAD && q < N; ++q) aq[q] = L[(16 * (q & 3) + l15) * SK_LD + 4 * (KS0 + (q >> 2)) + kq];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (q + AHEAD < N) aq[q + AHEAD] = L[(16 * ((q + AHEAD) & 3) + l15) * SK_LD + 4 * (KS0 + ((q + AHEAD) >> 2)) + kq];
    const int ks = KS0 + (q >> 2);
    acc[4 * C + (q & 3)] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[q], xn[ks >> 2][ks & 3], acc[4 * C + (q & 3)], 0, 0, 0);
#ifndef DFH_NO_SGB
    if (q + AHEAD < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one DS read ...
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // ... one MFMA
#endif
  }
}

template <int J>
__device__ __forceinline__ void fused_step_t(const FusedArgs& a, double4_t (&acc)[32], int s, bool diag, double* Rw,
                                             long rows_left, double* ssm) {
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform, and known to be: the row-store addresses built on it stay scalar.  (With a vector w the strip's
  // row pointer was spilled and every one of the sixteen X row stores waited -- vmcnt(0) -- for its reload behind
  // the previous write-through store: 5 us per step.)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, l15 = lane & 15;
  double* Lj = ssm;                                                // [64][SK_LD]
  double* Lc = ssm + PB * SK_LD;                                   // [64][SK_LD]
  double* Xall = ssm + 2 * PB * SK_LD;                             // [64][SK_LD]: the four waves' solved rows
  double* Xw = Xall + w * (16 * SK_LD);
  double* li = ssm + 2 * PB * SK_LD + 4 * 16 * SK_LD + 4 * 16 * SK_TD;
  int* flag = a.sync; int* prog = a.sync + 8;
  constexpr bool sc1 = true;        // the transposed panel always hands over with write-through stores / coherent loads
  if (J < s) {
    fused_wait_sc1(flag + J, a.epoch, a, !(diag && J == s - 1));
    FSTAMP(a, blockIdx.x + a.g0, 1 + 4 * J);
    {
      double pre[16], prei[4];                         // all loads in flight before the first LDS write
#pragma unroll
      for (int r = 0; r < 16; ++r) pre[r] = ld_in(a.Lfac + J * PB * PB + (w + 4 * r) * PB + lane, sc1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tid + 256 * r;
        prei[r] = ld_in(a.Linv16 + J * (4 * 16 * 17) + (i >> 8) * (16 * 17) + ((i >> 4) & 15) * 17 + (i & 15), sc1);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Lj[(w + 4 * r) * SK_LD + lane] = pre[r];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tid + 256 * r;
        li[(i >> 8) * (16 * SK_TD) + ((i >> 4) & 15) * SK_TD + (i & 15)] = prei[r];
      }
    }
    __syncthreads();
    FSTAMP(a, blockIdx.x + a.g0, 2 + 4 * J);
    // ---- row solve of block J, 16-column stages, right-looking; xn[b] = -X_b^T ----
    // (the accumulator tiles of block J are only ever read from here on: the solved block lives in xn and in LDS, so
    //  that no accumulator tile is written by the vector unit -- they stay in the AGPR half of the register file)
    double4_t xn[4], odd[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) odd[b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // the strip's LAST step (the one on the chain): X_s,s-1 is announced late, and three quarters of the
    // own-block product A_ss -= X X^T run BEFORE the fourth solve stage, while the inverse that stage needs is
    // still on its way (second flag of strip J)
    const bool last_step = diag && J == s - 1;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (b == 3 && last_step) {
        FSTAMP(a, blockIdx.x + a.g0, 48);
        __syncthreads();                               // columns 0 .. 47 of all four waves' X_sJ are in Xall
        FSTAMP(a, blockIdx.x + a.g0, 49);
        if constexpr (J + 1 < 8) strip_update_t<0, 12>(acc, J + 1, Xall, xn, l15, kq);
      }
      if (b == 3 && last_step) FSTAMP(a, blockIdx.x + a.g0, 50);
      // T_b^T = the tile (which carries the even-numbered k-steps of the earlier stages' updates) + the odd chain:
      // the association of panel_strip_kernel / diag_step64_kernel / fused_step (two accumulators per product,
      // added at the end), so that every schedule of the factorisation rounds alike -- results do not depend on
      // how a candidate set is cut into shards, chunks and lock-step groups (tests/test_gpu_mgpu.py).
      // It leaves the accumulator file for the vector registers here: the tile is dead from now on, so the
      // product below does not need a 33rd tile of accumulators (which the compiler found by spilling one)
      const double4_t tsum = (b == 0) ? acc[4 * J + b] : acc[4 * J + b] + odd[b];
      double tb0 = tsum[0], tb1 = tsum[1], tb2 = tsum[2], tb3 = tsum[3];
      asm volatile("" : "+v"(tb0), "+v"(tb1), "+v"(tb2), "+v"(tb3));
      double i0, i1, i2, i3;                                       // Linv_bb[l15][k], k = kq + 4 r
      if (b < 3) {
        const double* lib = li + b * (16 * SK_TD) + l15 * SK_TD + kq;
        i0 = lib[0]; i1 = lib[4]; i2 = lib[8]; i3 = lib[12];
      } else {
        // the last block's inverse arrives under its own flag (see the end of panel_fused_kernel), straight
        // into registers: every wave polls for itself, no workgroup barrier on the way
        if (lane == 0) {
          int spins = 0;
          while (__hip_atomic_load(a.sync + 16 + J, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.epoch) {
            if (++spins > a.spin_limit) { atomicOr(a.status, (unsigned long long)SYNC_ST_FUSED); break; }
            if ((spins & 63) == 0 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        // (the inverse's loads below are coherent loads issued after the poll in program order; the compiler must
        //  not move them above it either)
        asm volatile("" ::: "memory");
        if (last_step) FSTAMP(a, blockIdx.x + a.g0, 51);
        const double* gi = a.Linv16 + J * (4 * 16 * 17) + 3 * (16 * 17) + l15 * 17 + kq;
        i0 = ld_in(gi, sc1); i1 = ld_in(gi + 4, sc1); i2 = ld_in(gi + 8, sc1); i3 = ld_in(gi + 12, sc1);
      }
      double4_t p = {0.0, 0.0, 0.0, 0.0}, p2 = {0.0, 0.0, 0.0, 0.0};
      p = __builtin_amdgcn_mfma_f64_16x16x4f64(i0, tb0, p, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, tb1, p2, 0, 0, 0);
      p = __builtin_amdgcn_mfma_f64_16x16x4f64(i2, tb2, p, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(i3, tb3, p2, 0, 0, 0);
      p = p + p2;                                                  // X_b^T
      xn[b] = -p;
      // the solved block's row-major image in LDS (coalesced store below; A operand of the own-block product)
#pragma unroll
      for (int r = 0; r < 4; ++r) Xw[l15 * SK_LD + 16 * b + kq + 4 * r] = p[r];
#pragma unroll
      for (int b2 = b + 1; b2 < 4; ++b2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double lv = Lj[(16 * b2 + l15) * SK_LD + 16 * b + kq + 4 * r];                 // L[16 b2 + l15][16 b + k]
          if (r & 1) odd[b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, xn[b][r], odd[b2], 0, 0, 0);
          else acc[4 * J + b2] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, xn[b][r], acc[4 * J + b2], 0, 0, 0);
        }
    }
    COMPILER_BARRIER();                              // same wave: LDS executes its operations in order
    if (rows_left >= 16) {                             // (wave-uniform)
#pragma unroll
      for (int i = 0; i < 16; ++i) st_out(Rw + (long)i * a.lda + J * PB + lane, Xw[i * SK_LD + lane], sc1);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < rows_left) st_out(Rw + (long)i * a.lda + J * PB + lane, Xw[i * SK_LD + lane], sc1);
    }
    // X_sJ is out (later diagonal strips and the rows below read it).  In the strip's LAST step -- the one on the
    // chain -- the announcement waits until the own-block product below has been issued: nobody needs X_s,s-1
    // before L_ss exists, and the wait for the stores' acknowledgement (1.5 us) leaves the chain.
    if (diag && !last_step) fused_publish_sc1(prog + s, a.epoch * 16 + J + 1);
    FSTAMP(a, blockIdx.x + a.g0, 3 + 4 * J);
    const int c_hi = diag ? s : 7;                     // last block this strip still needs
#pragma unroll
    for (int c = J + 1; c < 8; ++c) {
      if (c > c_hi) break;
      const double* L;
      const bool own = diag && c == s;
      if (own) {
        if (last_step) FSTAMP(a, blockIdx.x + a.g0, 52);
        __syncthreads();                               // all four waves' rows of X_sJ are in Xall
        if (last_step) FSTAMP(a, blockIdx.x + a.g0, 53);
        L = Xall;                                      // own diagonal block: A_ss -= X_sJ X_sJ^T (lower tiles only)
      } else {
        // strip c has published X_cJ (its barrier also frees Lc)
        fused_wait_sc1(prog + c, a.epoch * 16 + J + 1, a, true);
        double pre[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pre[r] = ld_in(a.D + (long)(c * PB + w + 4 * r) * a.lda + J * PB + lane, sc1);
#pragma unroll
        for (int r = 0; r < 16; ++r) Lc[(w + 4 * r) * SK_LD + lane] = pre[r];
        __syncthreads();
        L = Lc;
      }
      // (the own block's tiles right of the diagonal are computed too -- garbage nobody reads, as in the other
      //  schedules; the waves that would skip them wait for wave 3 at the next barrier anyway)
      if (own && last_step) strip_update_t<12, 16>(acc, c, L, xn, l15, kq);      // k-steps 0 .. 11: before the fourth solve stage (above)
      else strip_update_t<0, 16>(acc, c, L, xn, l15, kq);
    }
    // (the last step's X_s,s-1 is announced from the kernel's tail, behind the staging barrier: by then the
    //  stores' acknowledgement, 1.5 - 2 us for write-through, has arrived without anybody waiting for it)
    if (last_step) FSTAMP(a, blockIdx.x + a.g0, 54);
    __syncthreads();                                   // Xall / Lj / li are free for the next step
    FSTAMP(a, blockIdx.x + a.g0, 4 + 4 * J);
  }
}

template <bool TR>
__global__ __launch_bounds__(256, 1) void panel_fused_kernel(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double ssm[];
  a.D += (long)blockIdx.y * a.strideD;
  a.Lfac += (long)blockIdx.y * a.strideL;
  a.Linv16 += (long)blockIdx.y * a.strideI;
  a.sync += (long)blockIdx.y * FUSED_SYNC_INTS;
  a.info += blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  // (NOT through readfirstlane here, unlike in fused_step_t.  With a scalar w the diagnostics build of this kernel
  //  came out with columns 0..3 of strip 0's block right and NaN from column 4 on.  Root cause, read off the ISA
  //  (docs/NOTES_r05.md section 2): a register-allocation fault of the compiler under this kernel's pressure (512 of
  //  512 registers, 288 spilled dwords).  The spill of acc[0] was split into scratch_store_dwordx3 (dwords 0-2), an
  //  AGPR copy a191 of dword 3 and four AGPR copies of dwords 4-7; the reload in the strip-0 leaf of the block select
  //  below restores seven of the eight -- a191 is written once and read nowhere -- so the high half of acc[0][1],
  //  i.e. columns 4..7 of the staged block, is whatever a195 last held.  Nothing in the source or the hardware; it
  //  comes and goes with the allocation, so tools/isa_audit.py looks for its signature (a register written and never
  //  read) in every function of the built library and tests/test_isa_audit.py runs that on each build.)
  const int w = tid >> 6;
  const int kq = lane >> 4, l15 = lane & 15;
  const int g = blockIdx.x + a.g0;   // (g0: the launch may hold only the rows-below strips, see cholesky_device_impl)
  const int nd = (a.nbk + PB - 1) / PB;              // diagonal strips (8 for a full panel)
  if (TR) a.sc1 = 1;                                 // the transposed panel's hand-offs are write-through / coherent loads
  if (a.resident) {
    if (tid == 0) __hip_atomic_fetch_add(a.resident, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.wait_ptr) fused_wait(a.wait_ptr, a.wait_target, a, SYNC_ST_GATE);
  }
  const bool diag = g < nd;
  const int s = diag ? g : nd;
  double* Rw = a.D + ((long)g * PB + 16 * w) * a.lda;               // this wave's 16 rows of the panel
  const long rows_left = (long)(a.nbk + a.rows_below) - ((long)g * PB + 16 * w);
  double4_t acc[32];
  const bool full_tr = TR && a.nbk == 8 * PB && rows_left >= 16;      // (wave-uniform)
  if (full_tr) {
    // full panel, all sixteen rows present: one base address per lane and immediate offsets; the blocks a
    // diagonal strip never touches are not loaded, its own block is cut to the lower triangle
    const double* rp = Rw + (long)l15 * a.lda + kq;
    const int row = 16 * w + l15;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const bool need = !diag || c <= s;
      const bool own = diag && c == s;
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        double4_t v = {0.0, 0.0, 0.0, 0.0};
        if (need) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rp[16 * (4 * c + ti) + 4 * r];
          if (own) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (16 * ti + kq + 4 * r <= row) ? v[r] : 0.0;
          }
        }
        acc[4 * c + ti] = v;
      }
    }
  } else
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // TR: lane (kq, l15) holds row l15, columns 16 t + kq + 4 r; otherwise rows kq + 4 r, column 16 t + l15
      const int rl = TR ? l15 : kq + 4 * r;              // row within the wave's 16
      const int cl = TR ? kq + 4 * r : l15;              // column within the 16-column tile
      const int row = 16 * w + rl, c = t >> 2, col = 16 * (t & 3) + cl;
      // a diagonal strip needs its blocks up to its own, of that one only the lower triangle; rows and
      // columns beyond the block's order (last panel) are identity padding
      const bool want = (rl < rows_left) && (16 * t + cl < a.nbk) && (!diag || c < s || (c == s && col <= row));
      const bool pad_one = diag && c == s && col == row && (rl >= rows_left);
      acc[t][r] = want ? Rw[(long)rl * a.lda + 16 * t + cl] : (pad_one ? 1.0 : 0.0);
    }
  FSTAMP(a, g, 0);
  if constexpr (TR) {
    fused_step_t<0>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<1>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<2>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<3>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<4>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<5>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<6>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step_t<7>(a, acc, s, diag, Rw, rows_left, ssm);
  } else {
    fused_step<0>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<1>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<2>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<3>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<4>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<5>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<6>(a, acc, s, diag, Rw, rows_left, ssm);
    fused_step<7>(a, acc, s, diag, Rw, rows_left, ssm);
  }
  if (!diag) return;
  // ---- factor the strip's own diagonal block and publish it (LDS of the strip machinery is free) ----
  double* Sp = ssm;                                  // [64][SPP] staged block, then the factor image
  double* colbuf = ssm + PB * SPP;                   // [64]
  double* ring = colbuf + PB;                        // [64][64]
  double* tbuf0 = ring + PB * PB;                    // 3 x [64][17]
  double* lbb = tbuf0 + 3 * PB * 17;                 // 4 x [16][17]
  double* linv = lbb + 4 * 16 * 17;                  // 4 x [16][17]
  double* rdiag = colbuf;
  __shared__ int s_badv[4];
  __shared__ int s_ring_timeout;
  if (tid == 0) s_ring_timeout = 0;
  // the 64 x 64 block s of the accumulators -> staged block (lower triangle, zero above)
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    double4_t v;
    // accumulator index has to be static: select the block by a chain of uniform branches
    switch (s) {
      case 0: v = acc[0 + t4]; break; case 1: v = acc[4 + t4]; break; case 2: v = acc[8 + t4]; break;
      case 3: v = acc[12 + t4]; break; case 4: v = acc[16 + t4]; break; case 5: v = acc[20 + t4]; break;
      case 6: v = acc[24 + t4]; break; default: v = acc[28 + t4]; break;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * w + (TR ? l15 : kq + 4 * r), col = 16 * t4 + (TR ? kq + 4 * r : l15);
      Sp[row * SPP + col] = (col <= row) ? v[r] : 0.0;
    }
  }
  if (tid < PB) ring[tid * PB] = 0.0;
  if (TR && s > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of X_s,s-1 are out
  __syncthreads();
  if (TR && s > 0 && tid == 0)                         // ... all four waves': announce them (see fused_step_t)
    __hip_atomic_store(a.sync + 8 + s, a.epoch * 16 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  FSTAMP(a, g, 40);
  double av[16];
  double* tbuf = tbuf0 + (w > 0 ? (w - 1) : 0) * PB * 17;
  // TR: the inverse of the last 16 x 16 block -- the only one of the four that is not hidden behind later columns,
  // ~4.3k cycles at the end of the chain -- is computed AFTER the factor has been handed on, and published under a
  // second flag: the next strip needs it for the last of its four solve stages only, ~3 us after it saw the first
  double my_r3 = 1.0;
  const int bad = factor64_waves<TR>(av, lane, w, Sp, tbuf, ring, lbb, linv, rdiag, &s_ring_timeout, nullptr, &my_r3);
  if (lane == 0) s_badv[w] = bad;
#pragma unroll
  for (int j = 0; j < 16; ++j) Sp[lane * SPP + perm16(16 * w + j)] = av[j];
  FSTAMP(a, g, 41);        // (thread 0 = wave 0: done with its columns long before wave 3)
  __syncthreads();
  FSTAMP(a, g, 42);
  const int s_bad = (s_badv[0] >= 0) ? s_badv[0] : (s_badv[1] >= 0) ? s_badv[1] : (s_badv[2] >= 0) ? s_badv[2] : s_badv[3];
  if (s_ring_timeout && tid == 0) atomicOr(a.status, (unsigned long long)SYNC_ST_RING);
  if (s_bad >= 0 && tid == 0) {
    // first failing pivot of the matrix: strips run in order, so the first writer wins
    unsigned long long expect = 0ull;
    __hip_atomic_compare_exchange_strong((unsigned long long*)a.info, &expect,
                                         (unsigned long long)(a.pivot_base + (long)s * PB + s_bad + 1), __ATOMIC_RELAXED,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {
    double* Lout = a.Lfac + (long)s * PB * PB;
    const int pk = perm16(lane);
    const bool sc1 = TR || a.sc1 != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) st_out(Lout + (w + 4 * r) * PB + lane, Sp[(w + 4 * r) * SPP + pk], sc1);
    double* Iout = a.Linv16 + (long)s * (4 * 16 * 17);
    for (int i = tid; i < (TR ? 3 : 4) * 16 * 17; i += 256) st_out(Iout + i, linv[i], sc1);
  }
  // published even after a failed pivot: nobody may hang
  if (TR || a.sc1) fused_publish_sc1(a.sync + s, a.epoch); else fused_publish(a.sync + s, a.epoch);
  FSTAMP(a, g, 43);
  if constexpr (TR) {
    if (w == 3) {
      factor64_inverse16(av, lane, 3, lbb, linv, rdiag, my_r3);
      COMPILER_BARRIER();                              // same wave: LDS executes its operations in order
      double* Iout3 = a.Linv16 + (long)s * (4 * 16 * 17) + 3 * (16 * 17);
      for (int i = lane; i < 16 * 17; i += 64) st_out(Iout3 + i, linv[3 * (16 * 17) + i], true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the whole wave's stores are acknowledged
      if (lane == 0) __hip_atomic_store(a.sync + 16 + s, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

constexpr int FUSED_SMEM_STRIP = (2 * PB * SK_LD + 4 * 16 * SK_LD + 4 * 16 * SK_TD + 4 * 16 * SK_TD) * 8;
constexpr int FUSED_SMEM_FACTOR = (PB * SPP + PB + PB * PB + 3 * PB * 17 + 8 * 16 * 17) * 8;
constexpr int FUSED_SMEM = FUSED_SMEM_STRIP > FUSED_SMEM_FACTOR ? FUSED_SMEM_STRIP : FUSED_SMEM_FACTOR;

// Stream gate: the launches behind it in its stream start only once *p >= target -- another stream's
// kernel has reached the point that raises the word.  One wave, polling politely; bounded like every
// wait of the factorisation (status word, then the host's fallback).
__global__ __launch_bounds__(64) void k_gate(const int* __restrict__ p, int target, unsigned long long* status, int spin_limit) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > spin_limit) { atomicOr(status, (unsigned long long)SYNC_ST_GATE); break; }
      if ((spins & 63) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

// p[b * stride + i] = 0 for i < count, b = blockIdx.y: one launch instead of one memset per batch matrix
__global__ void k_zero_strided(double* __restrict__ p, long count, long stride) {
  double* q = p + (long)blockIdx.y * stride;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) q[i] = 0.0;
}

// Inverses of the 64 x 64 lower-triangular diagonal blocks of an nbk x nbk factor block (one
// workgroup per block).  Padded rows/cols are identity so the full 64-block is safe.
// Lscr != null: block b's factor is read from Lscr + b*64*64 (ld 64) and also copied into its
// place on the diagonal of D (upper part zero); Lscr == null: the factor is read from D.
// (two waves per SIMD in the bounds: at most 256 registers, so that a workgroup fits beside ONE resident
//  GEMM workgroup -- with 340 registers it had to wait for a CU to drain completely: 1 - 1.8 ms behind a
//  running trailing update, against 36 us alone)
__global__ __launch_bounds__(256, 2) void trtri64_kernel(double* __restrict__ D, long lda, int nbk,
                                                      double* __restrict__ inv, long ldinv,
                                                      const double* __restrict__ Lscr, long strideD,
                                                      long strideInv, long strideL, int zero_rest) {
  __shared__ double S[PB * PBP];
  __shared__ double Tt[32 * 33];
  __builtin_amdgcn_s_setprio(3);      // beside a GEMM wave on the same SIMD the chain's kernel goes first
  D += (long)blockIdx.y * strideD;
  if (inv) inv += (long)blockIdx.y * strideInv;
  if (Lscr) Lscr += (long)blockIdx.y * strideL;
  const int tid = threadIdx.x;
  const int k = tid & 63, w = tid >> 6;
  const int j0 = blockIdx.x * PB;
  const int nb = min(PB, nbk - j0);
  double* A = D + (long)j0 * lda + j0;
  if (zero_rest && inv) {
    // this block's 64 rows of the 512-wide inverse, except the diagonal block written below
    for (int idx = tid; idx < PB * (int)CHOL_NB; idx += 256) {
      const int i = idx / (int)CHOL_NB, c = idx - i * (int)CHOL_NB;
      if (c < j0 || c >= j0 + PB) inv[(long)(j0 + i) * ldinv + c] = 0.0;
    }
  }
  double a[16];
  if (Lscr) {
    const double* src = Lscr + (long)blockIdx.x * PB * PB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = w + 4 * r;
      a[r] = src[i * PB + k];
      if (i < nb && k < nb) A[i * lda + k] = a[r];
    }
    if (!inv) return;                 // commit of the factor only (no inverses wanted)
  } else {
    load_block64(A, lda, nb, w, k, a);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) S[(w + 4 * r) * PBP + k] = a[r];
  __syncthreads();
  // [[A, 0], [B, C]]^-1 = [[A^-1, 0], [-C^-1 B A^-1, C^-1]] with 32 x 32 blocks: the two triangular
  // inverses by back substitution on rows (x_r L = e_r) in the two halves of wave 0 -- 32 values per
  // lane instead of 64: the kernel stays far below 256 registers --, the off-diagonal block by two
  // 32^3 products of the whole workgroup; everything in place in S.
  if (tid < 64) {
    const int r = tid & 31, o = 32 * (tid >> 5);
    double x[32];
#pragma unroll
    for (int c = 31; c >= 0; --c) {
      double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int kk = c + 1; kk < 32; ++kk) s = fma(-x[kk], S[(o + kk) * PBP + o + c], s);
      x[c] = s / S[(o + c) * PBP + o + c];
    }
    // (one wave, identical trip counts: every lane has read the factor entries before any is overwritten)
#pragma unroll
    for (int c = 0; c < 32; ++c) S[(o + r) * PBP + o + c] = (c <= r) ? x[c] : 0.0;
  }
  __syncthreads();
  {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // T = B A^-1  (A^-1 lower: k >= j)
      const int idx = tid + 256 * q, i = idx >> 5, j = idx & 31;
      double acc = 0.0;
      for (int kk = j; kk < 32; ++kk) acc = fma(S[(32 + i) * PBP + kk], S[kk * PBP + j], acc);
      t[q] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, i = idx >> 5, j = idx & 31;
      Tt[i * 33 + j] = t[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // X = -C^-1 T  (C^-1 lower: k <= i)
      const int idx = tid + 256 * q, i = idx >> 5, j = idx & 31;
      double acc = 0.0;
      for (int kk = 0; kk <= i; ++kk) acc = fma(-S[(32 + i) * PBP + 32 + kk], Tt[kk * 33 + j], acc);
      S[(32 + i) * PBP + j] = acc;
    }
  }
  __syncthreads();
  double* out = inv + (long)j0 * ldinv + j0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = w + 4 * r;
    if (i < nb && k < nb) out[(long)i * ldinv + k] = (k <= i) ? S[i * PBP + k] : 0.0;
  }
}

// Merge the 64-block inverses on the diagonal of Linv (ld = CHOL_NB) into the inverse of the
// nbk x nbk lower-triangular block D (ld = lda) by recursive doubling:
//   [[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]
int assemble_block_inverse(dfh_ctx* ctx, const double* D, int64_t lda, int64_t nbk, double* Linv,
                           double* T, int nbatch = 1, int64_t strideD = 0, int64_t strideInv = 0,
                           int64_t strideT = 0) {
  const int64_t NB = CHOL_NB;
  for (int64_t s = PB; s < nbk; s *= 2) {
    const int64_t full_pairs = nbk / (2 * s);
    if (full_pairs > 0) {
      GemmBatch b1, b2;
      b1.count = b2.count = (int)full_pairs;
      b1.count2 = b2.count2 = nbatch;
      b1.sA2 = strideD; b1.sB2 = strideInv; b1.sCout2 = strideT;
      b2.sA2 = strideInv; b2.sB2 = strideT; b2.sCout2 = strideInv;
      // T_q = B_q * A_q^-1 : B_q = D[hi rows, lo cols], A_q^-1 = Linv[lo, lo]
      b1.sA = 2 * s * (lda + 1); b1.sB = 2 * s * (NB + 1); b1.sCout = s * s;
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, s, s, s, 1.0, D + s * lda, lda, Linv, NB, 0.0, nullptr, 0,
                       T, s, &b1));
      // X_q = -C_q^-1 * T_q -> Linv[hi rows, lo cols]
      b2.sA = 2 * s * (NB + 1); b2.sB = s * s; b2.sCout = 2 * s * (NB + 1);
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, s, s, s, -1.0, Linv + s * (NB + 1), NB, T, s, 0.0, nullptr,
                       0, Linv + s * NB, NB, &b2));
    }
    const int64_t lo = full_pairs * 2 * s, hi = lo + s;
    if (hi < nbk) {                                   // trailing partial pair
      const int64_t hs = nbk - hi;
      GemmBatch c1, c2;
      c1.count = c2.count = nbatch;
      c1.sA = strideD; c1.sB = strideInv; c1.sCout = strideT;
      c2.sA = strideInv; c2.sB = strideT; c2.sCout = strideInv;
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, hs, s, s, 1.0, D + hi * lda + lo, lda, Linv + lo * (NB + 1),
                       NB, 0.0, nullptr, 0, T, s, &c1));
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, hs, s, hs, -1.0, Linv + hi * (NB + 1), NB, T, s, 0.0,
                       nullptr, 0, Linv + hi * NB + lo, NB, &c2));
    }
  }

  return DFH_OK;
}


// ---------------------------------------------------------------------------------------------
// Quality of an explicit block inverse, and what the solves do about it.
// Multiplying by the explicit inverse M of a 512 x 512 diagonal block L_bb leaves a residual of
// order eps * cond(L_bb) instead of the order eps of a substitution.  For the matrices a GP with
// noise >= Var(Y)/20 produces that is invisible; the reference's tuners also pick noise variances
// of 1e-8 Var(Y), where cond(L_bb) reaches 1e5 and more (tests/test_gpu_conditioning.py).  Per
// block the factorisation therefore keeps, next to M, a clean copy of L_bb (lower triangle, zero
// elsewhere, row stride NB) and measures delta = max |I - M L_bb|.  A solve with block b then runs
// refine_steps(delta) steps of iterative refinement in working precision, x <- x + M (r - L_bb x):
// each step multiplies the residual by delta, and it bottoms out at the residual of a backward
// stable solve (the rounding of r - L_bb x itself) -- the same GEMM / GEMV kernels, no
// substitution chain.  Well-conditioned blocks (delta <= tol) take no step and cost nothing.
// ---------------------------------------------------------------------------------------------
__global__ void k_copy_lower_block(const double* __restrict__ D, long lda, int nbk, double* __restrict__ out,
                                   long strideD, long strideOut) {
  __builtin_amdgcn_s_setprio(3);
  D += (long)blockIdx.z * strideD;
  out += (long)blockIdx.z * strideOut;
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= (int)CHOL_NB) return;
  out[(long)i * CHOL_NB + j] = (i < nbk && j <= i) ? D[(long)i * lda + j] : 0.0;
}

// delta[blockIdx.x * strideDelta] = max_ij |E_ij - [i == j]| over the nbk x nbk block E (ld NB); +inf if any
// NaN.  blockIdx.y = slice of the elements; the slices combine through an atomic max on the bit
// pattern (non-negative doubles order like their bit patterns; the slot is zeroed beforehand).
constexpr int DELTA_SLICES = 32;
__global__ __launch_bounds__(256) void k_inv_delta(const double* __restrict__ E, int nbk, long strideE,
                                                    double* __restrict__ delta, long strideDelta) {
  __shared__ double sm[256];
  __builtin_amdgcn_s_setprio(3);
  E += (long)blockIdx.x * strideE;
  double m = 0.0;
  bool bad = false;
  const long total = (long)nbk * nbk;
  for (long idx = (long)blockIdx.y * 256 + threadIdx.x; idx < total; idx += (long)DELTA_SLICES * 256) {
    const int i = (int)(idx / nbk), j = (int)(idx % nbk);
    const double v = fabs(E[(long)i * CHOL_NB + j] - (i == j ? 1.0 : 0.0));
    if (v != v) bad = true;
    m = v > m ? v : m;
  }
  sm[threadIdx.x] = bad ? INFINITY : m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(delta + (long)blockIdx.x * strideDelta),
              (unsigned long long)__double_as_longlong(sm[0]));
}

// clean copy of the diagonal block + delta, asynchronous on ctx->stream (T: NB*NB doubles per matrix)
int block_inverse_quality(dfh_ctx* ctx, const double* D, int64_t lda, int64_t nbk, const double* Linv, double* Ldiag,
                          double* T, double* d_delta, int nbatch = 1, int64_t strideD = 0, int64_t strideInv = 0,
                          int64_t strideT = 0, int64_t strideDelta = 0) {
  const int64_t NB = CHOL_NB;
  hipLaunchKernelGGL(k_copy_lower_block, dim3((unsigned)(NB / 256), (unsigned)NB, (unsigned)nbatch), dim3(256), 0,
                     ctx->stream, D, (long)lda, (int)nbk, Ldiag, (long)strideD, (long)strideInv);
  DFH_LAUNCH_CHECK();
  GemmBatch b;
  b.count = nbatch; b.sA = strideInv; b.sB = strideInv; b.sCout = strideT;
  DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, nbk, nbk, nbk, 1.0, Linv, NB, Ldiag, NB, 0.0, nullptr, 0, T, NB, &b));
  hipLaunchKernelGGL(k_zero_strided, dim3(1, (unsigned)nbatch), dim3(64), 0, ctx->stream, d_delta, 1L, (long)strideDelta);
  DFH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_inv_delta, dim3((unsigned)nbatch, DELTA_SLICES), dim3(256), 0, ctx->stream, T, (int)nbk,
                     (long)strideT, d_delta, (long)strideDelta);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

int refine_steps(double delta) {
  static const double tol = []() { const char* e = getenv("DFH_REFINE_TOL"); double v = e ? atof(e) : 1e-13; return v; }();
  // test hook (tests/test_gpu_refine_steps.py): every block takes this many steps whatever its inverse is like
  static const int forced = []() { const char* e = getenv("DFH_REFINE_FORCE_STEPS"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced > 8 ? 8 : forced;
  if (!(delta > tol)) return 0;
  if (!(delta < 0.25)) return 8;                    // the inverse is barely an inverse: as many steps as we allow
  const int k = (int)ceil(log(1e-15) / log(delta)) - 1;
  return k < 1 ? 1 : (k > 8 ? 8 : k);
}

}  // namespace

// Look-ahead schedule.  Two streams: P (high priority, "panel") and M (the context's main
// stream, "trailing").  For panel k
//   P: factor the 512 diagonal block, invert it, solve the panel L21(k), then update ONLY block
//      column k+1 with it (after M's trailing update k-1, which also wrote that column);
//   M: once L21(k) is ready, update the rest of the trailing matrix (columns >= k+2, lower).
// P therefore factors panel k+1 while M is still busy with the big SYRK of panel k: the
// latency-bound diagonal work leaves the critical path while the trailing update is long enough
// to cover it.
//
// Resident look-ahead (round 3; single matrices while at least DFH_CHOL_LR_MIN_REM rows are left).
// The schedule above does not overlap in practice: a panel kernel needs whole CUs (122 KB of LDS, all
// 512 registers of its lanes) and the trailing update refills every slot the moment it frees, so
// the panel of k+1 used to start when the update of k had drained.  Now
//   * the diagonal block of panel k+1 is factored by a launch of its own (panel_fused_kernel with no
//     rows below: eight workgroups) that is enqueued BEFORE the trailing update of panel k may start:
//     a one-wave gate kernel holds stream M until all eight have announced themselves (resident
//     counter).  They then wait -- bounded -- for the update's first sixteen tiles, the next
//     diagonal block, which the update computes first (look-ahead tile order, gemm_f64.hip) and
//     announces through a counter; 3 % of the CUs idle for ~0.1 ms per panel;
//   * the rows below the diagonal block are solved by GEMM with the explicit 512-block inverse the
//     posterior keeps anyway, L21 = A21 M^T, refined in working precision when the inverse's measured
//     quality asks for it (device-side condition: the refinement launches exit at once for a
//     well-conditioned block) -- GEMM workgroups share the CUs with the update's, which the panel
//     strips (one per CU) could not;
//   * the update of panel k is ONE launch over the whole trailing matrix (the look-ahead block
//     column is its first tiles instead of a launch of its own behind a stream barrier).
// The chain diag(k+1) -> inverse -> solve then runs beside update(k), and the updates follow each
// other on M as long as one lasts longer than the chain (~0.6 ms: n - k0 > ~8000).  Below that the
// schedule above takes over (it is the faster one when the chain is all there is).
namespace {

// refinement step s (1-based) of a solve with a block inverse of quality delta is due iff
// refine_steps(delta) >= s  <=>  delta > LR_REFINE_THR[s - 1]   (a NaN delta runs every step)
constexpr int LR_REFINE_MAX = 1;   // (each step is two launches on the chain, usually no-ops of ~45 us beside the update)
const double LR_REFINE_THR[3] = {0.0 /* = the tolerance, filled in at run time */, 3.1622776601683794e-8, 1e-5};

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

}  // namespace

static int cholesky_device_impl(dfh_ctx* ctx, double* A, int64_t n, int64_t lda, double* keep_inv,
                                int64_t* info_pivot, int nbatch, int64_t strideA, int64_t strideKeep, int* refine_out,
                                bool inv64_only, bool allow_lr, bool safe, int64_t clean_blocks = 0) {
  DFH_ARG(nbatch >= 1 && nbatch <= CHOL_MAX_BATCH);
  if (info_pivot) for (int b = 0; b < nbatch; ++b) info_pivot[b] = 0;
  if (n <= 0) return DFH_OK;
  const int64_t NB = CHOL_NB;
  long long* d_info = reinterpret_cast<long long*>(ctx->d_info);
  unsigned long long* d_status = reinterpret_cast<unsigned long long*>(d_info + CHOL_MAX_BATCH + 8);
  // Three streams.  P (high priority): the dependent chain -- 64-wide pivot steps over the panel,
  // each solving ALL rows below it by substitution, then the update of the next block column.
  // M (the caller's stream): the big trailing updates.  X (aux): everything the chain does not
  // need -- moving the pivot-block factors into place and the explicit 512-block inverses kept
  // for the triangular solves of the posterior.
  hipStream_t M = ctx->stream, P = ctx->side, X = ctx->aux;
  DFH_HIP(hipMemsetAsync(d_info, 0, 8 * (size_t)nbatch, M));
  DFH_HIP(hipMemsetAsync(d_status, 0, 8, M));
  // test hook: DFH_TEST_SPIN_LIMIT=0 makes every inter-workgroup wait expire at once (the fallback's test)
  static const int spin_limit = env_int("DFH_TEST_SPIN_LIMIT", SPIN_LIMIT_DEFAULT);
  static const int lr_on = env_int("DFH_CHOL_LR", 1);
  static const long lr_min_rem = env_int("DFH_CHOL_LR_MIN_REM", 7680);
  if (!keep_inv && !inv64_only && lr_on && allow_lr && !safe && nbatch == 1 && n - NB >= (lr_min_rem > 640 ? lr_min_rem : 640)) {
    // the resident schedule solves the panels with the 512-block inverses: a caller that keeps none gets them from scratch
    DFH_TRY(scratch_get(ctx, SCR_CHOLKEEP, (size_t)inv_buffer_doubles(n) * 8, (void**)&keep_inv));
    refine_out = nullptr;
  }

  const int64_t strideInv = keep_inv ? strideKeep : 0;          // between the batch matrices' inverse blocks
  const int64_t strideL = 2 * (NB / PB) * PB * PB;               // factor scratch: [parity][8][64][64] per matrix
  const int64_t strideT = NB * NB;
  const int64_t strideI = 2 * (NB / PB) * (4 * 16 * 17);         // 16 x 16 inverses of those blocks, same parity scheme
  double* Lscr_all = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_CHOLINV, ((size_t)nbatch * (strideL + strideI) + (size_t)nbatch * FUSED_SYNC_INTS / 2 + 2) * 8, (void**)&Lscr_all));
  double* Iscr_all = Lscr_all + (int64_t)nbatch * strideL;
  int* fsync_all = reinterpret_cast<int*>(Iscr_all + (int64_t)nbatch * strideI);     // [nbatch][FUSED_SYNC_INTS] counters of the fused panels
  // One launch per panel (panel_fused_kernel) for single matrices / small batches: n = 4096 2.92 -> 2.50 ms,
  // 8192 8.05 -> 7.07, 16384 34.8 -> 33.4.  Large lock-step batches keep the pivot steps + strips: their
  // workgroups would spend the diagonal chain's 190 us spinning.
  static const int fused_on = []() { const char* e = getenv("DFH_CHOL_FUSED"); return e ? atoi(e) : 1; }();
  // (round 3: lock-step batches of up to 16 gained 5-14 % from it as well, 32 and more lost.  Round 4, with the
  //  transposed panel -- tools/time_lml_batch.py, tools/r4_run18.sh: 32 matrices gain 4 % (n = 3000) to 20 % (n = 600),
  //  64 matrices 7-21 % up to n = 1500 and nothing at n = 3000: up to 32 matrices always, up to 64 while n <= 2048)
  static const int fused_max_batch_env = []() { const char* e = getenv("DFH_CHOL_FUSED_MAX_BATCH"); return e ? atoi(e) : -1; }();
  const int fused_max_batch = fused_max_batch_env >= 0 ? fused_max_batch_env : (n <= 2048 ? 64 : 32);
  const bool fused_mode = fused_on && nbatch <= fused_max_batch && !safe;   // safe: no inter-workgroup hand-offs
  // DFH_CHOL_FUSED_TR=0: the round-3 form of the one-launch panel (strip rows in the accumulators' rows)
  static const bool fused_tr = env_int("DFH_CHOL_FUSED_TR", 1) != 0;
  void (*const fused_kernel)(FusedArgs) = fused_tr ? panel_fused_kernel<true> : panel_fused_kernel<false>;
  // DFH_CHOL_FUSED_SC1=0: hand-offs inside the launch with release / acquire fences (FusedArgs::sc1)
  static const int fused_sc1 = env_int("DFH_CHOL_FUSED_SC1", 1) != 0 ? 1 : 0;
  static const int fused_prog_sleep = env_int("DFH_CHOL_PROG_SLEEP", 8);
  if (fused_mode) DFH_HIP(hipMemsetAsync(fsync_all, 0, (size_t)nbatch * FUSED_SYNC_INTS * sizeof(int), ctx->stream));
  // Panel strips (panel_strip_kernel) for lock-step batches: there the pivot steps are throughput-bound
  // (64 matrices x 64 workgroups, each re-factoring the pivot block, one workgroup per CU) and the
  // K = 64 panel updates HBM-bound (1.85 GB per step).  A single matrix keeps the pivot-step / GEMM
  // pairs: its chain is bound by launch latency, which the strips do not shorten.
  static const int strips_on = []() { const char* e = getenv("DFH_CHOL_STRIPS"); return e ? atoi(e) : 1; }();
  static const int strips_max_wg = []() { const char* e = getenv("DFH_CHOL_STRIPS_MAX_WG"); return e ? atoi(e) : (1 << 30); }();
  // from how many row strips on: 129 in a lock-step batch (round 3, tools/prof_lml.py: 20 - 64 matrices of
  // n = 600 ... 2000 gain 5 - 20 % over the 513 of round 2; below ~100 strips the pivot steps win), 513 for a
  // single matrix (which takes the one-launch panel anyway unless that is switched off)
  static const int strips_min_env = []() { const char* e = getenv("DFH_CHOL_STRIPS_MIN_WG"); return e ? atoi(e) : -1; }();
  const int strips_min_wg = strips_min_env >= 0 ? strips_min_env : (nbatch > 1 ? 129 : 513);
  double* T = nullptr;
  if (keep_inv) DFH_TRY(scratch_get(ctx, SCR_CHOLT, (size_t)nbatch * strideT * 8, (void**)&T));
  const int64_t nblk_all = (n + NB - 1) / NB;
  // blocks between an inverse and the clean copy of its diagonal block: the block count of the matrix
  // keep_inv was laid out for (this one, unless the call factors a diagonal sub-block of a larger one)
  if (clean_blocks <= 0) clean_blocks = nblk_all;
  double* d_delta = nullptr;                  // [nbatch][nblk] quality of the kept block inverses
  if (keep_inv) DFH_TRY(scratch_get(ctx, SCR_DELTA, (size_t)nbatch * nblk_all * 8, (void**)&d_delta));
  GemmBatch bA;                               // every operand inside the batch matrices
  bA.count = nbatch; bA.sA = bA.sB = bA.sCin = bA.sCout = strideA;

  static bool attr_set_dev[DFH_MAX_DEVICES] = {false};
  bool& attr_set = attr_set_dev[ctx->device];
  if (!attr_set) {
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(diag_step64_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_STEP_SMEM));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_strip_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, STRIP_SMEM));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_SMEM));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fused_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_SMEM));
    attr_set = true;
  }
  const int64_t nblk = (n + NB - 1) / NB;
  static const bool pair_on = []() { const char* e = getenv("DFH_CHOL_PAIR"); return e ? atoi(e) != 0 : true; }();
  static const long pair_min_rem = []() { const char* e = getenv("DFH_CHOL_PAIR_MIN_REM"); return e ? atol(e) : 6144L; }();

  // ---- resident look-ahead: which panels, and what it needs (nothing may allocate inside the loop:
  //      hipMalloc can wait for the device, and a gate kernel may be waiting for a launch not yet enqueued) ----
  int64_t kb_lr = 0;                          // panels [0, kb_lr) take the resident schedule
  if (lr_on && allow_lr && !safe && nbatch == 1 && keep_inv && !inv64_only && fused_mode) {
    const int64_t floor_rem = lr_min_rem > 640 ? lr_min_rem : 640;     // >= 5 tile rows for the look-ahead order
    while (n - (kb_lr + 1) * NB >= floor_rem) ++kb_lr;
    kb_lr &= ~(int64_t)1;                     // the schedule below pairs panels from an even index on
  }
  int* lr_sync = nullptr;                     // per panel {diag workgroups resident, next diagonal block's tiles done, next block column's tiles done, -}
  double *lr_X = nullptr, *lr_R = nullptr;    // the solved rows (two buffers, alternating), their residual (rows below the first panel x 512 each)
  const int64_t lr_xstride = (n - NB) * NB;
  if (kb_lr > 0) {
    DFH_TRY(scratch_get(ctx, SCR_CHOLSYNC, (size_t)kb_lr * 4 * sizeof(int), (void**)&lr_sync));
    DFH_HIP(hipMemsetAsync(lr_sync, 0, (size_t)kb_lr * 4 * sizeof(int), M));
    DFH_TRY(scratch_get(ctx, SCR_CHOLX, (size_t)2 * lr_xstride * 8, (void**)&lr_X));
    DFH_TRY(scratch_get(ctx, SCR_CHOLR, (size_t)(n - NB) * NB * 8, (void**)&lr_R));
  }
  static const double refine_tol = []() { const char* e = getenv("DFH_REFINE_TOL"); return e ? atof(e) : 1e-13; }();
  hipEvent_t ev_start, ev_done;
  DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 0, &ev_start));
  DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 1, &ev_done));
  DFH_HIP(hipEventRecord(ev_start, M));
  DFH_HIP(hipStreamWaitEvent(P, ev_start, 0));
  DFH_HIP(hipStreamWaitEvent(X, ev_start, 0));
  if (ctx->bulk_normal) DFH_HIP(hipStreamWaitEvent(ctx->bulk_normal, ev_start, 0));

  for (int64_t kb = 0; kb < nblk; ++kb) {
    const int64_t k0 = kb * NB;
    const int64_t nbk = (n - k0 < NB) ? n - k0 : NB;
    double* Linv = keep_inv ? keep_inv + kb * NB * NB : nullptr;
    double* D = A + k0 * lda + k0;
    double* Lscr = Lscr_all + (kb & 1) * (NB / PB) * PB * PB;
    double* Iscr = Iscr_all + (kb & 1) * (NB / PB) * (4 * 16 * 17);
    const int64_t rem = n - k0 - nbk;
    // paired trailing updates while the trailing matrix is large (decided per pair, on the rows left
    // below its FIRST panel, so that both panels of a pair see the same answer)
    const bool paired = pair_on && ((kb & 1) ? rem + NB : rem) > pair_min_rem;
    const bool strips = strips_on && rem > 0 && nbk == NB && (int64_t)nbatch * ((rem + PB - 1) / PB) <= strips_max_wg &&
                        (int64_t)nbatch * ((rem + PB - 1) / PB) >= strips_min_wg;
    hipEvent_t e_panel, e_trail, e_aux, e_diag, e_trail_prev = nullptr, e_aux_prev2 = nullptr;
    hipEvent_t e_copy, e_copy_prev = nullptr, e_copy_prev2 = nullptr;
    DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 2 + 5 * kb, &e_panel));
    DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 3 + 5 * kb, &e_trail));
    DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 4 + 5 * kb, &e_aux));
    DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 5 + 5 * kb, &e_diag));
    DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 6 + 5 * kb, &e_copy));
    if (kb >= 1) DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 3 + 5 * (kb - 1), &e_trail_prev));
    if (kb >= 2) DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 4 + 5 * (kb - 2), &e_aux_prev2));
    if (kb >= 1) DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 6 + 5 * (kb - 1), &e_copy_prev));
    if (kb >= 2) DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 6 + 5 * (kb - 2), &e_copy_prev2));
    // ---- off the chain (stream X): factor blocks into place, 64-block inverses, 512-block inverse, its quality ----
    auto aux_block = [&](hipEvent_t after, hipStream_t X) -> int {     // (X: the stream it runs on)
      StreamSwap on_x(ctx, X);
      if (after) DFH_HIP(hipStreamWaitEvent(X, after, 0));
      const bool zero_in_trtri = X != ctx->aux && nbk == NB;     // resident panels: one launch less on the chain
      if (Linv && !zero_in_trtri) {
        if (nbatch == 1) {
          DFH_HIP(hipMemsetAsync(Linv, 0, (size_t)NB * NB * 8, X));
        } else {
          hipLaunchKernelGGL(k_zero_strided, dim3(64, (unsigned)nbatch), dim3(256), 0, X, Linv, (long)(NB * NB), (long)strideInv);
          DFH_LAUNCH_CHECK();
        }
      }
      hipLaunchKernelGGL(trtri64_kernel, dim3((unsigned)((nbk + PB - 1) / PB), (unsigned)nbatch), dim3(256), 0, X,
                         D, (long)lda, (int)nbk, Linv, (long)NB, Lscr, (long)strideA, (long)strideInv,
                         (long)strideL, zero_in_trtri ? 1 : 0);
      DFH_LAUNCH_CHECK();
      if (Linv && !inv64_only) {
        DFH_TRY(assemble_block_inverse(ctx, D, lda, nbk, Linv, T, nbatch, strideA, strideInv, strideT));
        // clean copy of the block behind the inverses (keep_inv + nblk*NB*NB + ...) and delta = max|I - M L_bb|
        DFH_TRY(block_inverse_quality(ctx, D, lda, nbk, Linv, Linv + clean_blocks * NB * NB, T, d_delta + kb, nbatch,
                                      strideA, strideInv, strideT, nblk_all));
      }
      DFH_HIP(hipEventRecord(e_aux, X));
      return DFH_OK;
    };
    if (kb < kb_lr) {
      // =============== resident look-ahead panel (see the comment above cholesky_device_impl) ===============
      int* sy = lr_sync + 4 * kb;
      static const bool lr_chain_normal_prio = env_int("DFH_CHOL_LR_NORMAL_PRIO", 0) != 0;
      hipStream_t Pc = lr_chain_normal_prio ? ctx->bulk_normal : P;
      double* A21 = A + (k0 + NB) * lda + k0;          // rem x 512: the rows below the diagonal block
      double* Xk = lr_X + (kb & 1) * lr_xstride;
      {
        StreamSwap on_p(ctx, Pc);
        if (e_aux_prev2) DFH_HIP(hipStreamWaitEvent(Pc, e_aux_prev2, 0));     // factor scratch of this parity is free again
        // Not before update(kb-2) has finished: the eight workgroups need whole CUs, and a high-priority
        // launch that is PENDING because it does not fit throttles the dispatch of the running update
        // (measured: update(0) 2.42 ms with this launch enqueued after it, 2.52 / 2.71 ms with it pending
        // for the last 0.45 / 1.2 ms).  update(kb-1) cannot start before update(kb-2) has ended anyway.
        if (kb >= 2) {
          hipEvent_t e_trail_prev2;
          DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 3 + 5 * (kb - 2), &e_trail_prev2));
          DFH_HIP(hipStreamWaitEvent(Pc, e_trail_prev2, 0));
        }
        FusedArgs fa;
        fa.D = D; fa.lda = lda; fa.Lfac = Lscr; fa.Linv16 = Iscr; fa.sync = fsync_all; fa.epoch = (int)kb + 1;
        fa.nbk = (int)NB; fa.rows_below = 0; fa.info = d_info; fa.pivot_base = (long)k0;
        fa.strideD = strideA; fa.strideL = strideL; fa.strideI = strideI;
        fa.status = d_status; fa.spin_limit = spin_limit;
        fa.sc1 = fused_sc1; fa.prog_sleep = fused_prog_sleep; fa.g0 = 0;
        fa.resident = sy;
        fa.wait_ptr = kb > 0 ? sy - 4 + 1 : nullptr;   // the sixteen... ten lower tiles of this diagonal block, out of update(kb-1)
        fa.wait_target = 10;
        hipLaunchKernelGGL(fused_kernel, dim3((unsigned)(NB / PB), 1), dim3(256), FUSED_SMEM, Pc, fa);
        DFH_LAUNCH_CHECK();
        DFH_HIP(hipEventRecord(e_diag, Pc));
      }
      // the inverse is ON the chain here (the panel solve multiplies by it): it runs on the priority
      // stream -- on the auxiliary stream its small kernels wait for a slot behind the trailing update's
      // pending workgroups (trtri64: 36 us alone, 1.3 - 1.8 ms beside the update)
      DFH_TRY(aux_block(nullptr, Pc));
      {
        StreamSwap on_p(ctx, Pc);
        if (kb > 0) {
          // the whole block column has to be out of update(kb-1): 4 T - 6 look-ahead tiles over T tile rows
          const int64_t Tprev = (rem + NB + 127) / 128;
          hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, Pc, (const int*)(sy - 4 + 2), (int)(4 * Tprev - 6), d_status, spin_limit);
          DFH_LAUNCH_CHECK();
        }
        const double* Mi = Linv;                               // inverse of the diagonal block (lower, ld NB)
        const double* Lbb = Linv + clean_blocks * NB * NB;      // its clean copy
        // the solved rows go to a panel buffer of their own (two, alternating): the trailing update reads
        // them from there (contiguous, ld 512) and the copy into the factor happens off the chain
        if (e_copy_prev2) DFH_HIP(hipStreamWaitEvent(Pc, e_copy_prev2, 0));      // the buffer's previous contents are in place
        DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, rem, NB, NB, 1.0, A21, lda, Mi, NB, 0.0, nullptr, 0, Xk, NB));
        for (int st = 0; st < LR_REFINE_MAX; ++st) {
          // X <- X + (A21 - X L_bb^T) M^T, the right-hand side untouched; skipped on the device unless due
          ctx->gemm_cond = d_delta + kb;
          ctx->gemm_cond_thr = st == 0 ? refine_tol : LR_REFINE_THR[st];
          int rc_r = gemm_f64(ctx, GEMM_KTRI_B, rem, NB, NB, -1.0, Xk, NB, Lbb, NB, 1.0, A21, lda, lr_R, NB);
          if (rc_r == DFH_OK) rc_r = gemm_f64(ctx, GEMM_KTRI_B, rem, NB, NB, 1.0, lr_R, NB, Mi, NB, 1.0, Xk, NB, Xk, NB);
          ctx->gemm_cond = nullptr;
          DFH_TRY(rc_r);
        }
        static const bool lr_inplace = env_int("DFH_CHOL_LR_INPLACE", 0) != 0;     // ablation: operands of the update in place
        if (lr_inplace) DFH_TRY(copy_matrix(ctx, Xk, NB, A21, lda, rem, NB));
        DFH_HIP(hipEventRecord(e_panel, Pc));
      }
      {
        StreamSwap on_x(ctx, X);
        DFH_HIP(hipStreamWaitEvent(X, e_panel, 0));
        DFH_TRY(copy_matrix(ctx, Xk, NB, A21, lda, rem, NB));
        DFH_HIP(hipEventRecord(e_copy, X));
      }
      // ---- M: the whole trailing update with this panel, next block column first ----
      DFH_HIP(hipStreamWaitEvent(M, e_panel, 0));
      const bool next_resident = kb + 1 < kb_lr;
      if (next_resident) {
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, M, (const int*)(sy + 4), (int)(NB / PB), d_status, spin_limit);
        DFH_LAUNCH_CHECK();
      }
      {
        double* C = A + (k0 + NB) * lda + (k0 + NB);
        ctx->gemm_la_cnt = next_resident ? sy + 1 : nullptr;
        static const bool lr_inplace_u = env_int("DFH_CHOL_LR_INPLACE", 0) != 0;
        const double* Uop = lr_inplace_u ? A21 : Xk;
        const int64_t ldu = lr_inplace_u ? lda : NB;
        const int rc_u = gemm_f64(ctx, GEMM_LOWER, rem, rem, NB, -1.0, Uop, ldu, Uop, ldu, 1.0, C, lda, C, lda);
        ctx->gemm_la_cnt = nullptr;
        DFH_TRY(rc_u);
      }
      DFH_HIP(hipEventRecord(e_trail, M));
      continue;
    }
    {
      StreamSwap on_p(ctx, P);
      if (e_aux_prev2) DFH_HIP(hipStreamWaitEvent(P, e_aux_prev2, 0));     // factor scratch of this parity is free again
      // the last resident panel's update covered this block column too: it has to be complete, and
      // the last two resident panels' solved rows have to be in place
      if (kb == kb_lr && kb > 0 && e_trail_prev) {
        DFH_HIP(hipStreamWaitEvent(P, e_trail_prev, 0));
        if (e_copy_prev) DFH_HIP(hipStreamWaitEvent(P, e_copy_prev, 0));
        if (e_copy_prev2) DFH_HIP(hipStreamWaitEvent(P, e_copy_prev2, 0));
      }
      const bool fused = fused_mode;                  // full panels, and the (last) partial one: identity padding
      // (experiment, default OFF: measured slower -- n = 4096 2.16 -> 2.81 ms, 8192 6.28 -> 8.97 -- because the
      //  rows-below launch on a normal-priority stream queues behind the trailing update's pending workgroups,
      //  and a second high-priority stream made every schedule slower, presumably by sharing hardware queues;
      //  tools/r4_run24.sh, docs/NOTES_r04.md)
      static const bool split_on = env_int("DFH_CHOL_FUSED_SPLIT", 0) != 0;
      const bool split = fused && split_on && nbatch == 1 && nbk == NB && rem > 0;
      hipStream_t Q = ctx->bulk_normal;
      // Experiment switch (off): no look-ahead at all above DFH_CHOL_SERIAL_MIN_REM rows -- the one-launch
      // panel cannot be placed while the update runs and, pending, slows it (K = 1024 updates at 52 TF/s
      // against 64 alone); starting it only when the update has finished nevertheless LOSES (n = 16384
      // 34.5 -> 35.1 ms, 8192 7.06 -> 7.9): the panel does overlap the update's last wave of tiles.
      static const long serial_min_rem = env_int("DFH_CHOL_SERIAL_MIN_REM", 1 << 30);
      if (fused && kb > 0 && e_trail_prev && rem + nbk > serial_min_rem) DFH_HIP(hipStreamWaitEvent(P, e_trail_prev, 0));
      if (fused) {
        // ---- the whole panel in one launch: diagonal block by eight flag-synchronised strips, rows below alongside ----
        FusedArgs fa;
        fa.D = D; fa.lda = lda; fa.Lfac = Lscr; fa.Linv16 = Iscr; fa.sync = fsync_all; fa.epoch = (int)kb + 1;
        fa.nbk = (int)nbk; fa.rows_below = (int)rem; fa.info = d_info; fa.pivot_base = (long)k0;
        fa.strideD = strideA; fa.strideL = strideL; fa.strideI = strideI;
        fa.status = d_status; fa.spin_limit = spin_limit;
        fa.resident = nullptr; fa.wait_ptr = nullptr; fa.wait_target = 0; fa.sc1 = fused_sc1; fa.prog_sleep = fused_prog_sleep;
        fa.g0 = 0;
        if (!split) {
          hipLaunchKernelGGL(fused_kernel, dim3((unsigned)((nbk + PB - 1) / PB + (rem + PB - 1) / PB), (unsigned)nbatch),
                             dim3(256), FUSED_SMEM, P, fa);
          DFH_LAUNCH_CHECK();
        } else {
          // Round 4: the eight diagonal strips and the strips of the rows below as TWO launches of the same kernel
          // (the second with g0 = 8, on stream Q behind the larger part of the previous panel's look-ahead product):
          // the diagonal chain of this panel then waits only for the 512 x 512 block it factors, not for the whole
          // block column (rem x 512 x 512 at 24 TF/s in 64 x 64 tiles: 55 - 80 us of every panel of an n = 4096
          // factorisation, tools/r4_run23.sh).  Both launches talk through the panel's flags as before; Q takes
          // over everything P was made to wait for through e_copy.
          DFH_HIP(hipEventRecord(e_copy, P));
          hipLaunchKernelGGL(fused_kernel, dim3((unsigned)(NB / PB), 1), dim3(256), FUSED_SMEM, P, fa);
          DFH_LAUNCH_CHECK();
          DFH_HIP(hipStreamWaitEvent(Q, e_copy, 0));
          fa.g0 = (int)(NB / PB);
          hipLaunchKernelGGL(fused_kernel, dim3((unsigned)((rem + PB - 1) / PB), 1), dim3(256), FUSED_SMEM, Q, fa);
          DFH_LAUNCH_CHECK();
          DFH_HIP(hipEventRecord(e_diag, Q));
          DFH_HIP(hipStreamWaitEvent(P, e_diag, 0));
        }
      }
      // ---- 64-wide pivot steps: factor, solve every row below, update the rest of the panel ----
      for (int64_t j0 = 0; j0 < (fused ? 0 : nbk); j0 += PB) {
        const int w = (int)((nbk - j0 < PB) ? nbk - j0 : PB);
        double* Djj = D + j0 * lda + j0;
        const int64_t cols_left = nbk - j0 - w;            // panel columns still to be factored
        // every row below the pivot block -- or, with strips, only those inside the diagonal block
        const int64_t rows = cols_left + (strips ? 0 : rem);
        const unsigned nwg = 1 + (unsigned)((rows + PB - 1) / PB);
        hipLaunchKernelGGL(diag_step64_kernel, dim3(nwg, (unsigned)nbatch), dim3(256), DIAG_STEP_SMEM, P, Djj,
                           (long)lda, w, (int)rows, (long)(k0 + j0), d_info, Lscr + (j0 / PB) * PB * PB,
                           (long)strideA, (long)strideL, strips ? Iscr + (j0 / PB) * (4 * 16 * 17) : (double*)nullptr,
                           (long)strideI);
        DFH_LAUNCH_CHECK();
        if (cols_left > 0) {
          // A[r, c] -= L[r, j] L[c, j]^T for the rows below and the panel columns to the right
          // (the part above the diagonal of the block is scratch: only the lower triangle is L)
          double* Pn = D + (j0 + w) * lda + j0;                      // rows x w, already solved
          double* D22 = D + (j0 + w) * lda + (j0 + w);
          DFH_TRY(gemm_f64(ctx, 0, rows, cols_left, w, -1.0, Pn, lda, Pn, lda, 1.0, D22, lda, D22, lda, &bA));
        }
      }
      if (strips && !fused) {
        // ---- the rows below the diagonal block: L21 = A21 L11^-T, 64 rows per workgroup, one launch ----
        hipLaunchKernelGGL(panel_strip_kernel, dim3((unsigned)((rem + PB - 1) / PB), (unsigned)nbatch), dim3(256),
                           STRIP_SMEM, P, D, (long)lda, Lscr, Iscr, A + (k0 + nbk) * lda + k0, (int)rem,
                           (long)strideA, (long)strideL, (long)strideI);
        DFH_LAUNCH_CHECK();
      }
      DFH_HIP(hipEventRecord(e_panel, P));
      // ---- the next block column, so that the next panel can start before the trailing update ----
      if (rem > 0) {
        // kw = width of the panels whose contribution is still missing to the right of this one:
        // this panel alone, or -- in paired mode, after the second panel of a pair -- both
        const int64_t kw = (paired && (kb & 1)) ? nbk + NB : nbk;
        const double* A21 = A + (k0 + nbk) * lda + (k0 + nbk - kw);    // rem x kw, final
        if (e_trail_prev) DFH_HIP(hipStreamWaitEvent(P, e_trail_prev, 0));
        const int64_t nb1 = rem < NB ? rem : NB;
        double* C1 = A + (k0 + nbk) * lda + (k0 + nbk);   // rows k+1.., block column k+1
        if (!split || rem <= nb1) {
          DFH_TRY(gemm_f64(ctx, 0, rem, nb1, kw, -1.0, A21, lda, A21, lda, 1.0, C1, lda, C1, lda, &bA));
        } else {
          // the next diagonal block on the chain's stream, the rows under it on Q (the next panel's rows-below
          // launch follows them there)
          DFH_TRY(gemm_f64(ctx, GEMM_LOWER, nb1, nb1, kw, -1.0, A21, lda, A21, lda, 1.0, C1, lda, C1, lda, &bA));
          StreamSwap on_q(ctx, Q);
          DFH_HIP(hipStreamWaitEvent(Q, e_panel, 0));
          if (e_trail_prev) DFH_HIP(hipStreamWaitEvent(Q, e_trail_prev, 0));
          DFH_TRY(gemm_f64(ctx, 0, rem - nb1, nb1, kw, -1.0, A21 + nb1 * lda, lda, A21, lda, 1.0, C1 + nb1 * lda, lda,
                           C1 + nb1 * lda, lda, &bA));
        }
      }
    }
    DFH_TRY(aux_block(e_panel, X));
    // Paired mode: the trailing update runs after every SECOND panel, 1024 wide (the K = 512 update
    // reads and writes the C tile once per 512 columns of operand: 54 TF/s at n = 15872 against 63 for
    // K = 1024, tools/syrk_k.py).  The first panel of a pair only updates the next block column (the
    // look-ahead product above).  Its pivot chain then has no trailing update to run beside, which
    // costs most of what the wider update wins: n = 16384 35.2 -> 34.6 ms, n = 4096 2.82 -> 2.90 ms
    // -- hence only while more than DFH_CHOL_PAIR_MIN_REM rows are left.
    const bool trail_now = !paired || (kb & 1) || rem <= NB;
    if (rem > NB && trail_now) {
      const int64_t rem2 = rem - NB;
      const int64_t kw = (paired && (kb & 1)) ? nbk + NB : nbk;
      const double* A31 = A + (k0 + nbk + NB) * lda + (k0 + nbk - kw);   // rows k+2.. of the panel(s)
      double* A33 = A + (k0 + nbk + NB) * lda + (k0 + nbk + NB);
      DFH_HIP(hipStreamWaitEvent(M, e_panel, 0));
      {
        // DFH_CHOL_HALF_OCC=1: trailing update at one workgroup per CU, leaving slots for the
        // latency-bound panel kernels of the look-ahead stream
        static const bool half = []() { const char* e = getenv("DFH_CHOL_HALF_OCC"); return e && atoi(e) != 0; }();
        const bool old = ctx->gemm_half_occupancy;
        if (half) ctx->gemm_half_occupancy = true;
        const int rc_t = gemm_f64(ctx, GEMM_LOWER, rem2, rem2, kw, -1.0, A31, lda, A31, lda, 1.0, A33, lda, A33, lda, &bA);
        ctx->gemm_half_occupancy = old;
        DFH_TRY(rc_t);
      }
      DFH_HIP(hipEventRecord(e_trail, M));
    } else {
      // nothing for M to do: keep the event chain well-formed for the next panel's wait
      DFH_HIP(hipEventRecord(e_trail, M));
    }
  }
  DFH_HIP(hipEventRecord(ev_done, P));
  DFH_HIP(hipStreamWaitEvent(M, ev_done, 0));
  {
    hipEvent_t e_aux_last;
    DFH_TRY(ctx_event(ctx, EV_CHOL_BASE + 4 + 5 * (nblk - 1), &e_aux_last));
    DFH_HIP(hipStreamWaitEvent(M, e_aux_last, 0));
  }

  DFH_HIP(hipMemcpyAsync(ctx->h_info, d_info, 8 * (size_t)nbatch, hipMemcpyDeviceToHost, M));
  DFH_HIP(hipMemcpyAsync(ctx->h_info + CHOL_MAX_BATCH + 8, d_status, 8, hipMemcpyDeviceToHost, M));
  std::vector<double> deltas;
  if (keep_inv && refine_out && inv64_only) {
    for (size_t i = 0; i < (size_t)nbatch * nblk_all; ++i) refine_out[i] = 0;
  } else if (keep_inv && (refine_out || kb_lr > 0)) {
    deltas.resize((size_t)nbatch * nblk_all);
    DFH_HIP(hipMemcpyAsync(deltas.data(), d_delta, deltas.size() * 8, hipMemcpyDeviceToHost, M));
  }
  DFH_HIP(hipStreamSynchronize(M));
  if (refine_out) for (size_t i = 0; i < deltas.size(); ++i) refine_out[i] = refine_steps(deltas[i]);
  const unsigned long long sync_status = (unsigned long long)ctx->h_info[CHOL_MAX_BATCH + 8];
  if (sync_status != 0) {
    // a bounded wait expired: whatever was computed after it is not to be trusted
    dfh_set_error("Cholesky: an inter-workgroup hand-off timed out (status %llx)", sync_status);
    return DFH_INTERNAL_RETRY;
  }
  int rc = DFH_OK;
  for (int b = 0; b < nbatch; ++b) {
    const int64_t piv = ctx->h_info[b];
    if (info_pivot) info_pivot[b] = piv;
    if (piv != 0 && rc == DFH_OK) {
      dfh_set_error("Matrix is not positive definite (pivot %lld)", (long long)piv);
      rc = DFH_ERR_NOT_PD;
    }
  }
  if (kb_lr > 0 && (rc == DFH_OK || rc == DFH_ERR_NOT_PD)) {
    // a resident panel solved its rows with the block inverse and at most LR_REFINE_MAX refinement steps
    // on the device; an inverse so poor that more are due sends the matrix through the substitution
    // schedule.  That also holds when a LATER pivot came out non-positive: the inaccurately solved
    // panel may be what broke it, and whether the matrix is positive definite (whether the caller's
    // stable_cholesky ladder adds jitter, general_utils.py:183-203) is for the substitution schedule
    // to decide.  The failed pivot's own block and those after it hold no meaningful delta.
    int64_t kb_hi = kb_lr;
    if (rc == DFH_ERR_NOT_PD) kb_hi = std::min<int64_t>(kb_lr, (ctx->h_info[0] - 1) / CHOL_NB);  // pivots are 1-based
    for (int64_t kb = 0; kb < kb_hi; ++kb)
      if (refine_steps(deltas[(size_t)kb]) > LR_REFINE_MAX) {
        dfh_set_error("Cholesky: diagonal block %lld too ill-conditioned for the inverse-based panel solve", (long long)kb);
        return DFH_INTERNAL_RETRY_COND;
      }
  }
  return rc;
}

// rebuild (optional): re-creates the input matrix in A (a failed or abandoned factorisation destroys
// it).  With it the call may use the schedules whose rare failure modes need a second attempt -- the
// resident look-ahead with its inverse-based panel solve; a hand-off timeout -- and repeats itself on
// the conservative schedule (no inter-workgroup waits, substitution only) when one occurs.  Without
// it such a failure is DFH_ERR_HIP.
int cholesky_device(dfh_ctx* ctx, double* A, int64_t n, int64_t lda, double* keep_inv,
                    int64_t* info_pivot, int nbatch, int64_t strideA, int64_t strideKeep, int* refine_out,
                    bool inv64_only, const std::function<int()>* rebuild) {
  static const bool force_safe = env_int("DFH_CHOL_SAFE", 0) != 0;
  // (Tried and dropped, round 3: a two-way recursion -- L11, the posterior's GEMM-based row solve for
  //  L21, ONE update of depth n/2, L22.  The deep update does run at 66 TF/s, but the row solve of only
  //  n/2 rows is a chain of 256-tile launches at half occupancy: n = 16384 34.6 ms against 31.8 with
  //  the resident look-ahead, n = 8192 8.7 against 7.0.)
  auto attempt = [&](bool allow_lr, bool safe) -> int {
    return cholesky_device_impl(ctx, A, n, lda, keep_inv, info_pivot, nbatch, strideA, strideKeep, refine_out,
                                inv64_only, allow_lr, safe);
  };
  const bool cooling = ctx->chol_cooldown > 0;
  if (cooling) --ctx->chol_cooldown;
  const bool safe_first = force_safe || cooling;
  int rc = attempt(rebuild != nullptr && !cooling, safe_first);
  static const bool verbose = env_int("DFH_CHOL_VERBOSE", 0) != 0;
  if (rc != DFH_INTERNAL_RETRY && rc != DFH_INTERNAL_RETRY_COND) { if (!cooling) ctx->chol_fallback_streak = 0; return rc; }
  ++ctx->chol_fallbacks;
  // only hand-off time-outs (a crowded device) feed the cool-down: an ill-conditioned block is a property of the
  // matrix, and two of those in a row -- common at the extremes of a hyper-parameter search -- must not push the next
  // 32 factorisations onto the slow schedule
  if (rc == DFH_INTERNAL_RETRY && ++ctx->chol_fallback_streak >= 2) { ctx->chol_cooldown = 32; ctx->chol_fallback_streak = 0; }
  if (verbose) fprintf(stderr, "dfhip: factorisation of n = %lld repeated on the safe schedule: %s\n", (long long)n, dfh_last_error());
  if (!rebuild || force_safe) return DFH_ERR_HIP;
  DFH_TRY((*rebuild)());
  rc = attempt(false, true);
  return (rc == DFH_INTERNAL_RETRY || rc == DFH_INTERNAL_RETRY_COND) ? DFH_ERR_HIP : rc;
}

// Right-looking block substitution: once x_i is final it is pushed into every remaining row
// (wide, short GEMVs -> thousands of independent rows per launch instead of one long dependent
// chain).  The pass is HBM-bound: the lower triangle of L is read once per solve.
// ---------------------------------------------------------------------------------------------
// The two substitutions of GP.build_posterior (gp_core.py:161-163) on the 512-block inverses, round 6.
// A step of either direction is a chain of two dependent launches -- the block's solve by its explicit inverse, then the
// block's contribution to everything it feeds -- and the old steps spent their time INSIDE their kernels (trace of round
// 5: one workgroup per row of 4 KB in the forward update, 15 us; a partial + reduce pair of 16 + 6 us per transposed
// product; a device-to-device copy per step because the block's solve ran in place).  Here: r is updated in place, the
// solution goes to a vector of its own (no copy), and each kernel is shaped for its operand:
//   k_trsv_blk_fwd   z_b = M_b r_b             one wave per row of the lower-triangular inverse, columns <= row only
//   k_trsv_upd_fwd   r_i -= L[i, b] z_b        eight rows per wave, z_b in registers, 32 KB of loads in flight per wave
//   k_trsv_blk_bwd   a_b = M_b^T r_b           64 columns per workgroup, rows dealt to sixteen waves, one LDS reduce
//   k_trsv_upd_bwd   r_j -= L[b, j]^T a_b      64 columns per workgroup, a_b in LDS, sixteen row loads in flight per wave
// Sums run in a fixed order (deterministic); a block that needs refinement steps takes the general route below.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsv_blk_fwd(const double* __restrict__ M, int w, const double* __restrict__ r,
                                                      double* __restrict__ z) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= w) return;
  const double2_t* a = reinterpret_cast<const double2_t*>(M + (long)row * CHOL_NB);
  const double2_t* x = reinterpret_cast<const double2_t*>(r);
  double s0 = 0.0, s1 = 0.0;
  for (int j = lane; 2 * j <= row; j += 64) {          // (the inverse is exactly zero above its diagonal)
    const double2_t av = a[j], xv = x[j];
    s0 = fma(av.x, xv.x, s0);
    s1 = fma(av.y, (2 * j + 1 < w) ? xv.y : 0.0, s1);
  }
  double sum = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) z[row] = sum;
}

template <int TRSV_RPW>                                // rows per wave of the forward update
__global__ __launch_bounds__(256) void k_trsv_upd_fwd(const double* __restrict__ Lp, long ldl, long rows,
                                                      const double* __restrict__ z, double* __restrict__ r) {
  // Lp: the panel below the block (rows x 512, stride ldl); z: the block's solution (512); r: the rows' residuals
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * 4 + wv) * TRSV_RPW;
  if (row0 >= rows) return;
  double2_t zv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) zv[k] = reinterpret_cast<const double2_t*>(z)[lane + 64 * k];
  double2_t av[TRSV_RPW][4];
#pragma unroll
  for (int q = 0; q < TRSV_RPW; ++q) {
    const long row = row0 + q < rows ? row0 + q : rows - 1;     // (clamped: unconditional loads, results of the extra rows dropped)
    const double2_t* a = reinterpret_cast<const double2_t*>(Lp + row * ldl);
#pragma unroll
    for (int k = 0; k < 4; ++k) av[q][k] = a[lane + 64 * k];
  }
  double sum[TRSV_RPW];
#pragma unroll
  for (int q = 0; q < TRSV_RPW; ++q) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s0 = fma(av[q][k].x, zv[k].x, s0); s1 = fma(av[q][k].y, zv[k].y, s1); }
    sum[q] = s0 + s1;
  }
#pragma unroll
  for (int q = 0; q < TRSV_RPW; ++q)
    for (int off = 32; off > 0; off >>= 1) sum[q] += __shfl_down(sum[q], off, 64);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < TRSV_RPW; ++q)
      if (row0 + q < rows) r[row0 + q] -= sum[q];
  }
}

// (the two transposed products walk DOWN 512 rows per column: with four waves a lane's chain of row loads is eight memory
//  latencies long -- 9.4 us even for the smallest update; sixteen waves of 32 rows each make it two)
constexpr int TRSV_BW = 16;                            // waves per workgroup of the backward kernels
// CW columns per workgroup (64: a lane per column; 16: four row phases inside the wave as well, for the block's own solve
// and the short updates -- eight workgroups of 256 KB each are bound by what ONE CU can pull from HBM, 8 us a launch)
template <int CW>
__device__ __forceinline__ void trsv_colsum(const double* __restrict__ p0, long ld, int row0, int w, const double* s_x,
                                            double (*s_p)[64], double& out, bool& writer) {
  // p0: column `cc` of the first row; rows row0 .. w - 1; lane (rp, c): row phase rp of 64 / CW, column c
  constexpr int RP = 64 / CW, PH = TRSV_BW * RP;       // row phases: per wave, per workgroup
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rp = lane / CW;
  double s0 = 0.0, s1 = 0.0;
  int i = row0 + wv * RP + rp;
  for (; i + 15 * PH < w; i += 16 * PH) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p0[(long)(i + PH * u) * ld];
#pragma unroll
    for (int u = 0; u < 16; u += 2) { s0 = fma(v[u], s_x[i + PH * u], s0); s1 = fma(v[u + 1], s_x[i + PH * (u + 1)], s1); }
  }
  for (; i < w; i += PH) s0 = fma(p0[(long)i * ld], s_x[i], s0);
  s_p[wv][lane] = s0 + s1;
  __syncthreads();
  writer = threadIdx.x < CW;
  double t = 0.0;
  if (writer) {
#pragma unroll
    for (int q = 0; q < TRSV_BW; ++q)
#pragma unroll
      for (int h = 0; h < RP; ++h) t += s_p[q][h * CW + lane];
  }
  out = t;
}

template <int CW>
__global__ __launch_bounds__(1024) void k_trsv_blk_bwd(const double* __restrict__ M, int w, const double* __restrict__ r,
                                                       double* __restrict__ out) {
  __shared__ double s_r[CHOL_NB];
  __shared__ double s_p[TRSV_BW][64];
  for (int i = threadIdx.x; i < CHOL_NB; i += 64 * TRSV_BW) s_r[i] = i < w ? r[i] : 0.0;
  __syncthreads();
  // out[col] = sum_{i >= col} M[i][col] r[i]: the inverse is exactly zero above its diagonal, so the rows start at
  // the workgroup's first column
  const int col = blockIdx.x * CW + (threadIdx.x & 63) % CW;
  const int cc = col < w ? col : w - 1;
  double t; bool writer;
  trsv_colsum<CW>(M + cc, CHOL_NB, blockIdx.x * CW, w, s_r, s_p, t, writer);
  if (writer && col < w) out[col] = t;
}

template <int CW>
__global__ __launch_bounds__(1024) void k_trsv_upd_bwd(const double* __restrict__ Lr, long ldl, int w, long cols,
                                                       const double* __restrict__ a, double* __restrict__ r) {
  // Lr: the block row (w x cols, stride ldl); a: the block's solution (w); r: the residuals of the columns before it
  __shared__ double s_a[CHOL_NB];
  __shared__ double s_p[TRSV_BW][64];
  for (int i = threadIdx.x; i < CHOL_NB; i += 64 * TRSV_BW) s_a[i] = i < w ? a[i] : 0.0;
  __syncthreads();
  const long col = (long)blockIdx.x * CW + (threadIdx.x & 63) % CW;
  const long cc = col < cols ? col : cols - 1;
  double t; bool writer;
  trsv_colsum<CW>(Lr + cc, ldl, 0, w, s_a, s_p, t, writer);
  if (writer && col < cols) r[col] -= t;
}

static bool trsv_fast_applies(const double* L, int64_t n, int64_t ldl, const double* inv, const double* x, const int* refine) {
  static const int on = env_int("DFH_TRSV_FAST", 1);
  if (!on || (ldl & 1) || ((reinterpret_cast<uintptr_t>(L) | reinterpret_cast<uintptr_t>(inv) | reinterpret_cast<uintptr_t>(x)) & 15)) return false;
  for (int64_t b = 0; refine && b < (n + CHOL_NB - 1) / CHOL_NB; ++b)
    if (refine[b] > 0) return false;
  return true;
}

// z = L^-1 r (r is used up) -- the forward half
static int trsv_fast_forward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv, double* r, double* z) {
  const int64_t NB = CHOL_NB;
  for (int64_t c0 = 0; c0 < n; c0 += NB) {
    const int64_t w = std::min<int64_t>(NB, n - c0), below = n - c0 - w;
    hipLaunchKernelGGL(k_trsv_blk_fwd, dim3((unsigned)((w + 3) / 4)), dim3(256), 0, ctx->stream, inv + (c0 / NB) * NB * NB,
                       (int)w, r + c0, z + c0);
    DFH_LAUNCH_CHECK();
    if (below > 4 * NB) {
      hipLaunchKernelGGL(k_trsv_upd_fwd<8>, dim3((unsigned)((below + 31) / 32)), dim3(256), 0, ctx->stream,
                         L + (c0 + w) * ldl + c0, (long)ldl, (long)below, z + c0, r + c0 + w);
      DFH_LAUNCH_CHECK();
    } else if (below > 0) {                            // a short panel: two rows per wave, four times the workgroups
      hipLaunchKernelGGL(k_trsv_upd_fwd<2>, dim3((unsigned)((below + 7) / 8)), dim3(256), 0, ctx->stream,
                         L + (c0 + w) * ldl + c0, (long)ldl, (long)below, z + c0, r + c0 + w);
      DFH_LAUNCH_CHECK();
    }
  }
  return DFH_OK;
}

// a = L^-T r (r is used up) -- the backward half
static int trsv_fast_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv, double* r, double* a) {
  const int64_t NB = CHOL_NB, nblk = (n + NB - 1) / NB;
  for (int64_t b = nblk - 1; b >= 0; --b) {
    const int64_t c0 = b * NB, w = std::min<int64_t>(NB, n - c0);
    hipLaunchKernelGGL(k_trsv_blk_bwd<16>, dim3((unsigned)((w + 15) / 16)), dim3(64 * TRSV_BW), 0, ctx->stream, inv + b * NB * NB,
                       (int)w, r + c0, a + c0);
    DFH_LAUNCH_CHECK();
    if (c0 > 0 && c0 < 8 * NB) {                       // fewer than 64 workgroups of 64 columns: 16 columns each
      hipLaunchKernelGGL(k_trsv_upd_bwd<16>, dim3((unsigned)((c0 + 15) / 16)), dim3(64 * TRSV_BW), 0, ctx->stream, L + c0 * ldl,
                         (long)ldl, (int)w, (long)c0, a + c0, r);
      DFH_LAUNCH_CHECK();
    } else if (c0 > 0) {
      hipLaunchKernelGGL(k_trsv_upd_bwd<64>, dim3((unsigned)((c0 + 63) / 64)), dim3(64 * TRSV_BW), 0, ctx->stream, L + c0 * ldl,
                         (long)ldl, (int)w, (long)c0, a + c0, r);
      DFH_LAUNCH_CHECK();
    }
  }
  return DFH_OK;
}

// x <- L^-T L^-1 x  (gp_core.py:161-163): the forward half leaves z in a scratch vector, the backward half reads it
// there and writes alpha to x -- no copy in between
int trsv_both(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv, double* x, const int* refine) {
  if (!trsv_fast_applies(L, n, ldl, inv, x, refine)) {
    DFH_TRY(trsv_forward(ctx, L, n, ldl, inv, x, refine));
    return trsv_backward(ctx, L, n, ldl, inv, x, refine);
  }
  double* z = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)std::max<int64_t>(2 * CHOL_NB, n) * 8, (void**)&z));
  DFH_TRY(trsv_fast_forward(ctx, L, n, ldl, inv, x, z));
  return trsv_fast_backward(ctx, L, n, ldl, inv, z, x);
}

int trsv_forward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                 double* x, const int* refine) {
  const int64_t NB = CHOL_NB;
  if (trsv_fast_applies(L, n, ldl, inv, x, refine)) {
    double* z = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)std::max<int64_t>(2 * NB, n) * 8, (void**)&z));
    DFH_TRY(trsv_fast_forward(ctx, L, n, ldl, inv, x, z));
    DFH_HIP(hipMemcpyAsync(x, z, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return DFH_OK;
  }
  const int64_t nblk = (n + NB - 1) / NB;
  const double* diag = inv + nblk * NB * NB;
  double* tmp = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)NB * 8 * 2, (void**)&tmp));
  double* res = tmp + NB;
  for (int64_t c0 = 0; c0 < n; c0 += NB) {
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    const double* Mi = inv + (c0 / NB) * NB * NB;
    const double* Lbb = diag + (c0 / NB) * NB * NB;
    // x_i <- Linv_ii x_i
    DFH_TRY(gemv_rows(ctx, Mi, w, w, NB, x + c0, 1.0, nullptr, 0.0, tmp));
    for (int s = 0; s < (refine ? refine[c0 / NB] : 0); ++s) {
      // res = b_i - L_ii x ; x += Linv_ii res
      DFH_TRY(gemv_rows(ctx, Lbb, w, w, NB, tmp, -1.0, x + c0, 1.0, res, true));
      DFH_TRY(gemv_rows(ctx, Mi, w, w, NB, res, 1.0, tmp, 1.0, tmp));
    }
    DFH_HIP(hipMemcpyAsync(x + c0, tmp, (size_t)w * 8, hipMemcpyDeviceToDevice, ctx->stream));
    // x[i+1:] <- x[i+1:] - L[i+1:, i] x_i
    const int64_t below = n - c0 - w;
    if (below > 0)
      DFH_TRY(gemv_rows(ctx, L + (c0 + w) * ldl + c0, below, w, ldl, x + c0, -1.0, x + c0 + w, 1.0, x + c0 + w));
  }
  return DFH_OK;
}

int trsv_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                  double* x, const int* refine) {
  const int64_t NB = CHOL_NB;
  if (trsv_fast_applies(L, n, ldl, inv, x, refine)) {
    double* a = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)std::max<int64_t>(2 * NB, n) * 8, (void**)&a));
    DFH_TRY(trsv_fast_backward(ctx, L, n, ldl, inv, x, a));
    DFH_HIP(hipMemcpyAsync(x, a, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return DFH_OK;
  }
  const int64_t nblk = (n + NB - 1) / NB;
  const double* diag = inv + nblk * NB * NB;
  double* tmp = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC3, (size_t)NB * 8 * 2, (void**)&tmp));
  double* res = tmp + NB;
  for (int64_t b = nblk - 1; b >= 0; --b) {
    const int64_t c0 = b * NB;
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    const double* Mi = inv + b * NB * NB;
    const double* Lbb = diag + b * NB * NB;
    // x_i <- Linv_ii^T x_i
    DFH_TRY(gemv_cols(ctx, Mi, w, w, NB, x + c0, 1.0, nullptr, 0.0, tmp));
    for (int s = 0; s < (refine ? refine[b] : 0); ++s) {
      // res = b_i - L_ii^T x ; x += Linv_ii^T res
      DFH_TRY(gemv_cols(ctx, Lbb, w, w, NB, tmp, -1.0, x + c0, 1.0, res));
      DFH_TRY(gemv_cols(ctx, Mi, w, w, NB, res, 1.0, tmp, 1.0, tmp));
    }
    DFH_HIP(hipMemcpyAsync(x + c0, tmp, (size_t)w * 8, hipMemcpyDeviceToDevice, ctx->stream));
    // x[:i] <- x[:i] - L[i, :i]^T x_i
    if (c0 > 0) DFH_TRY(gemv_cols(ctx, L + c0 * ldl, w, c0, ldl, x + c0, -1.0, x, 1.0, x));
  }
  return DFH_OK;
}

// residual buffer of the refined row solves (m x NB), only when some block takes a step
static int refine_scratch(dfh_ctx* ctx, const int* refine, int64_t nblk, int64_t m, double** out) {
  *out = nullptr;
  bool any = false;
  for (int64_t b = 0; refine && b < nblk; ++b) any = any || refine[b] > 0;
  if (any) DFH_TRY(scratch_get(ctx, SCR_REFINE, (size_t)m * CHOL_NB * 8, (void**)out));
  return DFH_OK;
}

// at most this many right-hand rows take the right-looking (wide, shallow) form of trsm_rows
constexpr int64_t TRSM_FEW_ROWS = 256;

int trsm_rows(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
              double* Kct, int64_t m, int64_t ldk, const int* refine, const double* diag_override) {
  if (m <= 0 || n <= 0) return DFH_OK;
  const int64_t NB = CHOL_NB;
  const double* diag = diag_override ? diag_override : inv + ((n + NB - 1) / NB) * NB * NB;      // clean copies of the diagonal blocks
  double *T = nullptr, *R2 = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_TMP, (size_t)m * NB * 8, (void**)&T));
  DFH_TRY(refine_scratch(ctx, refine, (n + NB - 1) / NB, m, &R2));
  if (m <= TRSM_FEW_ROWS) {
    // A handful of rows (single-point GP.eval calls, tree-search frontiers, hallucinated batches):
    // the left-looking form below would run each block as ONE tile row with a K loop over every
    // earlier column -- a few workgroups walking all of L serially.  Right-looking instead: solve
    // the block, then subtract its contribution from ALL later columns at once, a GEMM that is
    // (n - c0) / 128 tiles wide with K = 512, so L streams from HBM across the whole chip.
    for (int64_t c0 = 0; c0 < n; c0 += NB) {
      const int64_t w = (n - c0 < NB) ? n - c0 : NB;
      const int64_t rest = n - c0 - w;
      const double* Linv = inv + (c0 / NB) * NB * NB;
      const double* Lpanel = L + (c0 + w) * ldl + c0;
      // (the inverse block has an exactly zero upper part, so the full K range gives the same sum)
      // the solved block goes to T (a GEMM may not overwrite what other workgroups still read) and
      // has to end up in place as well: the few-row update below stages T anyway and writes the
      // copy on the side; only the last block, which has nothing to update, needs a copy launch
      const bool skinny_update = rest > 0 && gemm_skinny_applies(m, rest, w, T, NB, Lpanel, ldl) && (ldk % 2) == 0;
      if (gemm_skinny_applies(m, w, w, Kct + c0, ldk, Linv, NB))
        DFH_TRY(gemm_skinny_nt(ctx, m, w, w, 1.0, Kct + c0, ldk, Linv, NB, 0.0, nullptr, 0, T, NB));
      else
        DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, 1.0, Kct + c0, ldk, Linv, NB, 0.0, nullptr, 0, T, NB));
      for (int s = 0; s < (refine ? refine[c0 / NB] : 0); ++s) {
        // R <- B - X L_bb^T from the untouched right-hand side (it is only overwritten by the solution
        // below) ; X <- X + R Linv^T.  (Round 2 kept the residual IN PLACE of the right-hand side, which
        // is right for one step only: the second would subtract X0 L^T twice.)
        const double* Lbb = diag + (c0 / NB) * NB * NB;
        DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, -1.0, T, NB, Lbb, NB, 1.0, Kct + c0, ldk, R2, NB));
        DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, 1.0, R2, NB, Linv, NB, 1.0, T, NB, T, NB));
      }
      if (skinny_update) {
        DFH_TRY(gemm_skinny_nt(ctx, m, rest, w, -1.0, T, NB, Lpanel, ldl, 1.0, Kct + c0 + w, ldk,
                               Kct + c0 + w, ldk, Kct + c0, ldk));
      } else {
        DFH_TRY(copy_matrix(ctx, T, NB, Kct + c0, ldk, m, w));
        if (rest > 0)
          DFH_TRY(gemm_f64(ctx, 0, m, rest, w, -1.0, T, NB, Lpanel, ldl, 1.0, Kct + c0 + w, ldk,
                           Kct + c0 + w, ldk));
      }
    }
    return DFH_OK;
  }
  for (int64_t c0 = 0; c0 < n; c0 += NB) {
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    // T = Kct[:, c0:c0+w] - Vt[:, 0:c0] * L[c0:c0+w, 0:c0]^T      (K = 0 degenerates to a copy)
    DFH_TRY(gemm_f64(ctx, 0, m, w, c0, -1.0, Kct, ldk, L + c0 * ldl, ldl, 1.0, Kct + c0, ldk, T, NB));
    // Vt[:, c0:c0+w] = T * Linv_ii^T
    const double* Linv = inv + (c0 / NB) * NB * NB;
    DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, 1.0, T, NB, Linv, NB, 0.0, nullptr, 0, Kct + c0, ldk));
    for (int s = 0; s < (refine ? refine[c0 / NB] : 0); ++s) {
      // R <- T - X L_bb^T (the residual of the right-hand side T, which stays) ; X <- X + R Linv^T
      const double* Lbb = diag + (c0 / NB) * NB * NB;
      DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, -1.0, Kct + c0, ldk, Lbb, NB, 1.0, T, NB, R2, NB));
      DFH_TRY(gemm_f64(ctx, GEMM_KTRI_B, m, w, w, 1.0, R2, NB, Linv, NB, 1.0, Kct + c0, ldk, Kct + c0, ldk));
    }
  }
  return DFH_OK;
}

int tri_block_inverses(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, double* inv, int* refine_out,
                       double* diag) {
  const int64_t NB = CHOL_NB;
  const int64_t nblk = (n + NB - 1) / NB;
  double* T = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_CHOLT, (size_t)NB * NB * 8, (void**)&T));
  double* d_delta = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_DELTA, (size_t)nblk * 8, (void**)&d_delta));
  if (!diag) diag = inv + nblk * NB * NB;
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int64_t nbk = (n - k0 < NB) ? n - k0 : NB;
    double* Linv = inv + (k0 / NB) * NB * NB;
    DFH_HIP(hipMemsetAsync(Linv, 0, (size_t)NB * NB * 8, ctx->stream));
    const double* D = L + k0 * ldl + k0;
    hipLaunchKernelGGL(trtri64_kernel, dim3((unsigned)((nbk + PB - 1) / PB)), dim3(256), 0, ctx->stream,
                       const_cast<double*>(D), (long)ldl, (int)nbk, Linv, (long)NB, (const double*)nullptr, 0L, 0L, 0L, 0);
    DFH_LAUNCH_CHECK();
    DFH_TRY(assemble_block_inverse(ctx, D, ldl, nbk, Linv, T));
    DFH_TRY(block_inverse_quality(ctx, D, ldl, nbk, Linv, diag + (k0 / NB) * NB * NB, T, d_delta + k0 / NB));
  }
  if (refine_out) {
    std::vector<double> deltas((size_t)nblk);
    DFH_HIP(hipMemcpyAsync(deltas.data(), d_delta, (size_t)nblk * 8, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < nblk; ++b) refine_out[b] = refine_steps(deltas[b]);
  }
  return DFH_OK;
}

int trsm_rows_backward(dfh_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* inv,
                       double* Bt, int64_t m, int64_t ldb, const int* refine) {
  if (m <= 0 || n <= 0) return DFH_OK;
  const int64_t NB = CHOL_NB;
  const double* diag = inv + ((n + NB - 1) / NB) * NB * NB;
  double *T = nullptr, *R2 = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_TMP, (size_t)m * NB * 8, (void**)&T));
  const int64_t nblk = (n + NB - 1) / NB;
  DFH_TRY(refine_scratch(ctx, refine, nblk, m, &R2));
  for (int64_t b = nblk - 1; b >= 0; --b) {
    const int64_t c0 = b * NB;
    const int64_t w = (n - c0 < NB) ? n - c0 : NB;
    const int64_t below = n - c0 - w;
    // T = Bt[:, c0:c0+w] - Xt[:, c0+w:] * L[c0+w:, c0:c0+w]
    DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, m, w, below, -1.0, Bt + c0 + w, ldb, L + (c0 + w) * ldl + c0, ldl,
                     1.0, Bt + c0, ldb, T, NB));
    // Xt[:, c0:c0+w] = T * Linv_ii
    DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, m, w, w, 1.0, T, NB, inv + b * NB * NB, NB, 0.0, nullptr, 0,
                     Bt + c0, ldb));
    for (int s = 0; s < (refine ? refine[b] : 0); ++s) {
      // R <- T - X L_bb (the residual of the right-hand side T, which stays) ; X <- X + R Linv
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, m, w, w, -1.0, Bt + c0, ldb, diag + b * NB * NB, NB, 1.0, T, NB, R2, NB));
      DFH_TRY(gemm_f64(ctx, GEMM_TRANSB, m, w, w, 1.0, R2, NB, inv + b * NB * NB, NB, 1.0, Bt + c0, ldb, Bt + c0, ldb));
    }
  }
  return DFH_OK;
}

// Launch of lml_wg_kernel: `count` candidates, one workgroup each (K: padded matrices of order 64 * ceil((n + 1) / 64),
// see the kernel).  d_par: [count] augmented diagonal entries, then [count] prior means; d_out2: [count][2];
// d_info: [count] failing pivots (zeroed here).  Asynchronous on ctx->stream.
// team > 1: `team` workgroups per candidate (lml_team_kernel; team * count should not exceed the CUs).  d_status
// (device, 8 bytes, or null when team == 1) is zeroed here and non-zero afterwards iff a hand-off wait expired:
// the results of the launch are then void.
int lml_wg_batch(dfh_ctx* ctx, double* K, int64_t sK, int64_t ld, int64_t n, int count, const double* d_y,
                 const double* d_par, double* d_out2, long long* d_info, int team, unsigned long long* d_status,
                 int* d_sync_zeroed) {
  // d_sync_zeroed: the team's flags ([count][LMLT_SYNC_INTS]) in a block the caller has ALREADY zeroed together with
  // d_info and d_status (one memset per group instead of three); null: allocated and zeroed here.
  static_assert(LMLT_SYNC_INTS == LMLT_SYNC_INTS_PER_CANDIDATE, "common.h and chol.hip disagree on the flags per candidate");
  DFH_ARG(ctx && K && d_y && d_par && d_out2 && d_info && n >= 1 && n <= LMLWG_MAX_N && count >= 1 && team >= 1 &&
          team <= 32 && (team == 1 || d_status));
  const int64_t nbt = (n + 1 + PB - 1) / PB;
  DFH_ARG(ld >= nbt * PB && (ld & 1) == 0 && sK >= nbt * PB * ld && (reinterpret_cast<uintptr_t>(K) & 15) == 0);
  static bool attr_set_dev[DFH_MAX_DEVICES] = {false};
  bool& attr_set = attr_set_dev[ctx->device];
  if (!attr_set) {
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lml_wg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                DIAG_STEP_SMEM));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lml_team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                LMLT_SMEM));
    attr_set = true;
  }
  if (!d_sync_zeroed) DFH_HIP(hipMemsetAsync(d_info, 0, (size_t)count * 8, ctx->stream));
  LmlWgArgs a;
  a.K = K; a.sK = (long)sK; a.ld = (long)ld; a.n = (int)n; a.nbt = (int)nbt;
  a.y = d_y; a.par = d_par; a.count = count; a.out2 = d_out2; a.info = d_info;
  a.T = team; a.sync = nullptr; a.linvbuf = nullptr; a.status = d_status; a.spin_limit = 0;
  if (team == 1) {
    hipLaunchKernelGGL(lml_wg_kernel, dim3((unsigned)count), dim3(256), DIAG_STEP_SMEM, ctx->stream, a);
    DFH_LAUNCH_CHECK();
    return DFH_OK;
  }
  DFH_TRY(scratch_get(ctx, SCR_CHOLINV, (size_t)count * nbt * LMLT_LINV * 8, (void**)&a.linvbuf));
  if (d_sync_zeroed) {
    a.sync = d_sync_zeroed;
  } else {
    DFH_TRY(scratch_get(ctx, SCR_CHOLSYNC, (size_t)count * LMLT_SYNC_INTS * sizeof(int), (void**)&a.sync));
    DFH_HIP(hipMemsetAsync(a.sync, 0, (size_t)count * LMLT_SYNC_INTS * sizeof(int), ctx->stream));
    DFH_HIP(hipMemsetAsync(d_status, 0, 8, ctx->stream));
  }
  // a legitimate wait lasts well under a millisecond; a poll is ~1 us: give up after ~0.1 s (DFH_TEST_SPIN_LIMIT=0: at once, the fallback's test)
  static const int spin_limit = env_int("DFH_TEST_SPIN_LIMIT", 1 << 17);
  a.spin_limit = spin_limit;
#ifdef DFH_DEBUG_HOOKS
  a.stamps = g_lmlt_stamps;
#endif
  hipLaunchKernelGGL(lml_team_kernel, dim3((unsigned)(count * team)), dim3(256), LMLT_SMEM, ctx->stream, a);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

bool lml_wg_fused_applies(const KernDev* kds, int count, int64_t n) {
  static const int fused_max = env_int("DFH_LML_FUSED", 16);       // candidates per call; 0: off
  static const int max_n = std::min(255, env_int("DFH_LML_FUSED_MAX_N", (int)LMLF_MAX_N));
  if (count < 1 || count > fused_max || n < 1 || n > max_n) return false;
  for (int c = 0; c < count; ++c)
    if (kds[c].P > TINY_MAX_P || kds[c].n_parts > TINY_MAX_PARTS || kds[c].P < 1 || !kds[c].stationary ||
        n * (int64_t)(kds[c].P + kds[c].n_parts) > LMLF_LDS_DOUBLES) return false;
  return true;
}

int lml_wg_fused_batch(dfh_ctx* ctx, const KernDev* kds, int count, const double* dX, int64_t n, int64_t ldx,
                       const double* y_host, const double* noise_vars, const double* mean_consts,
                       double* logdet_dot, long long* info) {
  DFH_ARG(ctx && kds && dX && y_host && noise_vars && logdet_dot && info && lml_wg_fused_applies(kds, count, n));
  const int64_t nbt = (n + 1 + PB - 1) / PB, NP = PB * nbt, sK = NP * NP;
  static bool attr_set_dev[DFH_MAX_DEVICES] = {false};
  if (!attr_set_dev[ctx->device]) {
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lml_wgf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                DIAG_STEP_SMEM));
    attr_set_dev[ctx->device] = true;
  }
  TinyBlob tb;
  DFH_TRY(tiny_blob_build(ctx, kds, count, n, y_host, noise_vars, mean_consts, &tb));
  double *Kb = nullptr, *ybuf = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_KCT, (size_t)count * sK * 8, (void**)&Kb));
  DFH_TRY(scratch_get(ctx, SCR_VEC, (size_t)std::max<int64_t>(256, n * 8), (void**)&ybuf));
  LmlWgArgs a;
  a.K = Kb; a.sK = (long)sK; a.ld = (long)NP; a.n = (int)n; a.nbt = (int)nbt;
  a.y = ybuf; a.par = nullptr; a.count = count; a.out2 = nullptr; a.info = nullptr;
  a.T = 1; a.sync = nullptr; a.linvbuf = nullptr; a.status = nullptr; a.spin_limit = 0;
  LmlFuse f;
  f.ec = kExpConsts;
  f.X = dX; f.ldx = (long)ldx;
  f.blob = tb.host; f.y_off = (long)tb.y_off;
  f.ybuf = ybuf; f.out4 = tb.res; f.direct = 1;
  volatile double* vres = tb.res;
  for (int c = 0; c < count; ++c) vres[4 * c + 3] = -1.0;          // "not there yet"
  hipLaunchKernelGGL(lml_wgf_kernel, dim3((unsigned)count), dim3(256), DIAG_STEP_SMEM, ctx->stream, a, f);
  DFH_LAUNCH_CHECK();
  DFH_TRY(tiny_poll_results(ctx, vres, count, "lml_wgf_kernel"));
  for (int c = 0; c < count; ++c) {
    logdet_dot[2 * c] = vres[4 * c];
    logdet_dot[2 * c + 1] = vres[4 * c + 1];
    info[c] = (long long)vres[4 * c + 2];
    if (info[c] == 0 && (!std::isfinite(logdet_dot[2 * c]) || !std::isfinite(logdet_dot[2 * c + 1]))) info[c] = -2;
  }
  return DFH_OK;
}

#ifdef DFH_DEBUG_HOOKS
// Diagnostics: the team kernel's next launches stamp their progress into `dev_buf` (device, [workgroups][32][16] int64,
// zeroed by the caller); null switches it off.  tools/dbg_lmlt.py decodes the stamps.
extern "C" int dfh_debug_lmlt_stamps(void* dev_buf) { g_lmlt_stamps = reinterpret_cast<long long*>(dev_buf); return DFH_OK; }
#endif
#ifdef DFH_DEBUG_HOOKS      // diagnostics: built only with `python -m dragonfly_amd.build --debug-hooks` (include/dfhip_debug.h)
// Diagnostics hook (not part of the product path): times `reps` back-to-back launches of the
// 64-wide diagonal step on a synthetic SPD block and returns in-kernel cycle stamps.
extern "C" int dfh_debug_diag_step(dfh_ctx* ctx, int reps, int rows_below, double* ms_per_launch,
                                   long long* cycles_out /* [7]: load, factor, trsm, waves 0-3 */) {
  DFH_ARG(ctx && reps > 0 && rows_below >= 0 && rows_below <= 448);
  DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(diag_step64_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_STEP_SMEM));
  const int64_t nn = 512;
  double* A = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_TSK, (size_t)nn * nn * 8, (void**)&A));
  std::vector<double> h((size_t)nn * nn, 0.01);
  for (int64_t i = 0; i < nn; ++i) h[i * nn + i] = 10.0 + (double)(i % 7);
  DFH_HIP(hipMemcpyAsync(A, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  long long* d_info = reinterpret_cast<long long*>(ctx->d_info);
  long long init[8] = {0, 0, 0, 0, 0, 0, 0, 1};
  DFH_HIP(hipMemcpyAsync(d_info + CHOL_MAX_BATCH, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  const unsigned nwg = 1 + (unsigned)((rows_below + PB - 1) / PB);
  hipEvent_t e0, e1;
  DFH_HIP(hipEventCreate(&e0));
  DFH_HIP(hipEventCreate(&e1));
  DFH_HIP(hipEventRecord(e0, ctx->stream));
  for (int r = 0; r < reps; ++r) {
    // the block is re-factored from its own output (still SPD: L has a dominant diagonal)
    hipLaunchKernelGGL(diag_step64_kernel, dim3(nwg), dim3(256), DIAG_STEP_SMEM, ctx->stream, A, (long)nn, 64,
                       rows_below, 0L, d_info, A + 256 * nn, 0L, 0L, (double*)nullptr, 0L);
  }
  DFH_HIP(hipEventRecord(e1, ctx->stream));
  DFH_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  DFH_HIP(hipEventElapsedTime(&ms, e0, e1));
  long long out[8];
  DFH_HIP(hipMemcpy(out, d_info + CHOL_MAX_BATCH, sizeof(out), hipMemcpyDeviceToHost));
  if (ms_per_launch) *ms_per_launch = ms / reps;
  if (cycles_out) { cycles_out[0] = out[2]; cycles_out[1] = out[3]; cycles_out[2] = out[4];
                    cycles_out[3] = out[0]; cycles_out[4] = out[1]; cycles_out[5] = out[5]; cycles_out[6] = out[6]; }
  long long zero[8] = {0};
  DFH_HIP(hipMemcpy(d_info + CHOL_MAX_BATCH, zero, sizeof(zero), hipMemcpyHostToDevice));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return DFH_OK;
}
// Diagnostics hook: `reps` launches of the one-launch panel on a synthetic SPD 512 x 512 block with `rows_below`
// rows under it; ms_out[reps] per launch (HIP events), stamps_out [(8 + strips below)][64] from the last launch.
// A_out (optional, [(512 + rows_below) x 512]): the input block; Lfac_out (optional, [8][64][64]): the factored
// 64 x 64 diagonal blocks as the strips published them -- for a host-side comparison with LAPACK.
extern "C" int dfh_debug_panel_data(double* A_out, double* Lfac_out);
static std::vector<double> g_dbg_A, g_dbg_L;
extern "C" int dfh_debug_panel_data(double* A_out, double* Lfac_out) {
  if (A_out) for (size_t i = 0; i < g_dbg_A.size(); ++i) A_out[i] = g_dbg_A[i];
  if (Lfac_out) for (size_t i = 0; i < g_dbg_L.size(); ++i) Lfac_out[i] = g_dbg_L[i];
  return (int)g_dbg_A.size();
}
extern "C" int dfh_debug_panel_stamps(dfh_ctx* ctx, int reps, int rows_below, double* ms_out, long long* stamps_out) {
  DFH_ARG(ctx && reps > 0 && rows_below >= 0 && ms_out && stamps_out);
  DFH_HIP(hipSetDevice(ctx->device));
  static const bool fused_tr = env_int("DFH_CHOL_FUSED_TR", 1) != 0;
  void (*const fused_kernel)(FusedArgs) = fused_tr ? panel_fused_kernel<true> : panel_fused_kernel<false>;
  DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_SMEM));
  const int64_t nn = 512, rows = nn + rows_below;
  const int nwg = (int)(nn / PB + (rows_below + PB - 1) / PB);
  std::vector<double> h((size_t)rows * nn);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  for (int64_t i = 0; i < rows; ++i)
    for (int64_t j = 0; j < nn; ++j) h[i * nn + j] = (i < nn && j > i) ? 0.0 : 0.02 * (rnd() - 0.5);
  for (int64_t i = 0; i < nn; ++i) { for (int64_t j = 0; j < i; ++j) h[j * nn + i] = h[i * nn + j]; h[i * nn + i] = 3.0 + rnd(); }
  double *master = nullptr, *D = nullptr, *scr = nullptr;
  long long* d_st = nullptr;
  DFH_HIP(hipMalloc(&master, h.size() * 8));
  DFH_HIP(hipMalloc(&D, h.size() * 8));
  const size_t scr_doubles = 8 * PB * PB + 8 * (4 * 16 * 17) + 64;
  DFH_HIP(hipMalloc(&scr, scr_doubles * 8));
  DFH_HIP(hipMalloc(&d_st, (size_t)nwg * 64 * 8));
  DFH_HIP(hipMemcpy(master, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  DFH_HIP(hipMemset(scr, 0, scr_doubles * 8));
  long long* d_info = reinterpret_cast<long long*>(ctx->d_info);
  hipEvent_t e0, e1;
  DFH_HIP(hipEventCreate(&e0)); DFH_HIP(hipEventCreate(&e1));
  hipStream_t S = ctx->side;
  for (int r = 0; r < reps; ++r) {
    DFH_HIP(hipMemcpyAsync(D, master, h.size() * 8, hipMemcpyDeviceToDevice, S));
    DFH_HIP(hipMemsetAsync(d_info, 0, 8 * (CHOL_MAX_BATCH + 16), S));
    DFH_HIP(hipMemsetAsync(d_st, 0, (size_t)nwg * 64 * 8, S));
    FusedArgs fa;
    fa.D = D; fa.lda = nn; fa.Lfac = scr; fa.Linv16 = scr + 8 * PB * PB;
    fa.sync = reinterpret_cast<int*>(scr + 8 * PB * PB + 8 * (4 * 16 * 17)); fa.epoch = r + 1;
    fa.nbk = (int)nn; fa.rows_below = rows_below; fa.info = d_info; fa.pivot_base = 0;
    fa.strideD = 0; fa.strideL = 0; fa.strideI = 0;
    fa.status = reinterpret_cast<unsigned long long*>(d_info + CHOL_MAX_BATCH + 8); fa.spin_limit = SPIN_LIMIT_DEFAULT;
    fa.resident = nullptr; fa.wait_ptr = nullptr; fa.wait_target = 0; fa.sc1 = env_int("DFH_CHOL_FUSED_SC1", 1) != 0 ? 1 : 0;
    fa.prog_sleep = env_int("DFH_CHOL_PROG_SLEEP", 8); fa.g0 = 0;
    fa.stamps = d_st;
    DFH_HIP(hipEventRecord(e0, S));
    hipLaunchKernelGGL(fused_kernel, dim3((unsigned)nwg, 1), dim3(256), FUSED_SMEM, S, fa);
    DFH_HIP(hipEventRecord(e1, S));
    DFH_HIP(hipStreamSynchronize(S));
    float ms = 0.f;
    DFH_HIP(hipEventElapsedTime(&ms, e0, e1));
    ms_out[r] = ms;
  }
  DFH_HIP(hipMemcpy(stamps_out, d_st, (size_t)nwg * 64 * 8, hipMemcpyDeviceToHost));
  g_dbg_A = h;
  g_dbg_L.resize((size_t)8 * PB * PB);
  DFH_HIP(hipMemcpy(g_dbg_L.data(), scr, g_dbg_L.size() * 8, hipMemcpyDeviceToHost));
  long long bad = 0;
  DFH_HIP(hipMemcpy(&bad, d_info, 8, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(master); (void)hipFree(D); (void)hipFree(scr); (void)hipFree(d_st);
  if (bad != 0) { dfh_set_error("debug panel: pivot %lld failed", bad); return DFH_ERR_NOT_PD; }
  return DFH_OK;
}
#endif  // DFH_DEBUG_HOOKS
