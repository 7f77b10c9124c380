// The 64 x 64 pivot-block factorisation by the four waves of a workgroup (factor64_waves) and its helpers.
// Shared by chol.hip (diagonal steps, one-launch panels, the one-workgroup tuning objective) and kernmat.hip (the
// whole tuning objective of a small problem in one launch).  Include inside the translation unit's anonymous namespace.
#pragma once

constexpr int PB = 64;       // pivot block
constexpr unsigned SYNC_ST_RING = 1;      // an LDS column ring flag never came up (factor64_waves)

// ---------------------------------------------------------------------------------------------
// 64 x 64 pivot-block kernels.  The block is factored by the four waves of a workgroup without
// barriers (factor64_waves below); history of the alternatives measured on gfx950: a 256-thread
// column-per-thread version with one barrier per column (100 us), its rank-4 blocked form with 16
// barriers (58k cycles), a single-wave all-v_readlane form (~29 cycles per readlane pair + FMA:
// slower), a single wave with LDS-broadcast columns and deferred updates (36k cycles, issue-bound).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_block64(const double* __restrict__ A, long lda, int nb, int w,
                                              int k, double (&a)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = w + 4 * r;
    double v = (i == k) ? 1.0 : 0.0;                  // identity padding beyond nb
    if (i < nb && k < nb) v = (k <= i) ? A[i * lda + k] : 0.0;
    a[r] = v;
  }
}

// 1/d for a positive normal d: hardware reciprocal + two Newton steps (what the compiler's own
// division sequence does before its final correction).
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  return r;
}
// q ~= n/d with one residual correction (<= 1 ulp from the correctly rounded quotient)
__device__ __forceinline__ double fast_div(double n, double d, double r) {
  const double q = n * r;
  return fma(fma(-d, q, n), r, q);
}

// LDS image of a factor row: column j sits at perm16(j) so that the 16 columns of a residue class
// j = q (mod 4) a lane of the row substitution reads are 16 consecutive doubles (ds_read_b128).
__device__ __forceinline__ int perm16(int i) { return (i & 3) * 16 + (i >> 2); }

#define LDS_FENCE() asm volatile("" ::: "memory")   // keeps every LDS load above it, every use below

// 1/sqrt(d) and sqrt(d) for a positive normal d: v_rsq_f64 seed + two Newton steps + one
// residual correction each (<= ~1 ulp).
__device__ __forceinline__ void fast_rsqrt_sqrt(double d, double* rs, double* sq) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  double s = d * y;
  s = fma(fma(-s, s, d), 0.5 * y, s);        // sqrt(d)
  const double r = fast_div(1.0, s, y);      // 1/sqrt(d) consistent with s
  *rs = r;
  *sq = s;
}

// Factorisation of the 64 x 64 pivot block by the four waves of the workgroup.  Lane i of every
// wave holds row i; wave w owns the 16 columns 16w .. 16w+15 (16 doubles per lane).
//
// The elimination runs on UNSCALED columns (LDL^T style): with u[:,k] the column as it stands when
// it becomes the pivot column and d_k = u[k][k],
//     a[i][j] -= (u[i][k] / d_k) * u[j][k]          for j > k,
// and only at the very end L[:,k] = u[:,k] / sqrt(d_k).  The dependent chain from one pivot to the
// next is then  v_readlane d_k -> v_rcp_f64 + two Newton steps -> multiply -> one FMA  (7 dependent
// VALU operations; a dependent fp64 operation costs ~25 cycles here) -- the reciprocal square root
// with its ~20 dependent operations is off the chain: sixteen independent ones per wave at the end.
//   * The OWNER of the current 16 columns runs that chain inside the wave (the element u[k+1][k]
//     comes from lane k+1 by v_readlane), updates its next column eagerly and its other columns
//     one step late, and publishes every finished column u[:,k] and 1/d_k to an LDS ring, then
//     raises the column's flag.  A wave's LDS operations execute in order, so data -> flag needs
//     only a compiler barrier, no s_waitcnt.
//   * The waves owning LATER columns consume published columns as their flags come up: one
//     per-lane read (their row's element) + 8 broadcast ds_read_b128 + 16 FMAs per column.
//   * Waves owning earlier columns are finished (wave 0 then stages the panel rows).
// No workgroup barrier inside the 64 steps.  History of this kernel on gfx950: 256 threads with a
// barrier per column (100 us), rank-4 blocked with 16 barriers (58k cycles), one wave with
// LDS-broadcast columns (36k cycles, issue-bound: 7300 instructions x 4 cycles), four waves with
// the rsqrt on the chain (38k: latency-bound), this one.
// Branch-free: a non-positive / NaN pivot only raises a flag (columns and flags are still
// published, so nobody waits forever).
#define COMPILER_BARRIER() asm volatile("" ::: "memory")
constexpr int SPP_STAGE = 66;   // row stride of the staged pivot block (= SPP of the factor image, below)

// Owner step for column k = 16 w + KL.  The published ring slot of column k holds u[i][k] for the
// rows i >= 1 and, in row 0's place, 1/d_k (row 0 of a column k >= 1 lies above the diagonal and
// is never read as data; for k = 0 only row 0's own -- unused -- updates see it).  A non-zero
// row-0 entry doubles as the "published" flag: the ring's row-0 entries are zeroed beforehand, and
// a wave's ds_write_b64 lands as one LDS operation.
template <int KL>
__device__ __forceinline__ void f64_owner_step(double (&a)[16], int lane, int w, double* ring, int& bad,
                                               double& mprev) {
#define SB() __builtin_amdgcn_sched_barrier(0)
  const int k = 16 * w + KL;
  // Column k updates the next TWO own columns eagerly, through v_readlane (no LDS on the way to the
  // next pivot); the previous own column's update of columns KL+2 .. 15 comes one step late through
  // the LDS ring: those reads are issued first and are consumed inside the chain's stalls.
  double c[16];
  if (KL >= 1) {
    const double* cbp = ring + (k - 1) * PB + 16 * w;
#pragma unroll
    for (int j = KL + 2; j < 16; ++j) c[j] = cbp[j];
  }
  const int lo = __builtin_amdgcn_readlane(__double2loint(a[KL]), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(a[KL]), k);
  const double d = __hiloint2double(hi, lo);
  bad = (bad < 0 && !(d > 0.0)) ? k : bad;                  // uniform: d is the same in every lane
  // dependent chain readlane -> rcp -> 4 FMA -> mul -> FMA, with the deferred FMAs pinned into its
  // stalls (in-order issue: left to itself the compiler puts them behind the chain, or -- worse --
  // sinks them to where each column is needed)
  double rc = __builtin_amdgcn_rcp(d);
  double t1 = 0.0, t2 = 0.0;
  if (KL + 1 < 16) {
    t1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[KL]), k + 1),
                          __builtin_amdgcn_readlane(__double2loint(a[KL]), k + 1));      // u[k+1][k]
  }
  if (KL + 2 < 16) {
    t2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[KL]), k + 2),
                          __builtin_amdgcn_readlane(__double2loint(a[KL]), k + 2));      // u[k+2][k]
  }
  SB();
  if (KL >= 1) {
#pragma unroll
    for (int j = KL + 2; j < 16 && j < KL + 6; ++j) a[j] = fma(-mprev, c[j], a[j]);
  }
  double er = fma(-d, rc, 1.0);
  SB();
  if (KL >= 1) {
#pragma unroll
    for (int j = KL + 6; j < 16 && j < KL + 10; ++j) a[j] = fma(-mprev, c[j], a[j]);
  }
  rc = fma(rc, er, rc);
  SB();
  if (KL >= 1) {
#pragma unroll
    for (int j = KL + 10; j < 16; ++j) a[j] = fma(-mprev, c[j], a[j]);
  }
  er = fma(-d, rc, 1.0);
  SB();
  rc = fma(rc, er, rc);
  SB();
  const double m = a[KL] * rc;                                // u[i][k] / d_k
  if (KL + 1 < 16) a[KL + 1] = fma(-m, t1, a[KL + 1]);        // eager: the next pivot column
  SB();
  if (KL + 2 < 16) a[KL + 2] = fma(-m, t2, a[KL + 2]);        // and the one after it
  // publish: the column, with 1/d_k (never exactly zero here: it is the flag) in row 0's place
  const double rcpub = (rc == 0.0) ? 1.0 : rc;
  ring[k * PB + lane] = (lane == 0) ? rcpub : a[KL];
  mprev = m;
  SB();
#undef SB
}

template <int... KLs>
__device__ __forceinline__ void f64_owner_block(double (&a)[16], int lane, int w, double* ring, int& bad,
                                                std::integer_sequence<int, KLs...>) {
  double mprev = 0.0;
  (f64_owner_step<KLs>(a, lane, w, ring, bad, mprev), ...);
}

// Consumption of published columns by a wave that owns later columns: a rank-NV update of the
// wave's 64 x 16 block on the matrix cores,
//     acc[t] -= U[rows of tile t][k0 .. k0+3] * diag(1/d) * U[this wave's 16 rows][k0 .. k0+3]^T,
// one v_mfma_f64_16x16x4 per 16-row tile.  Every lane fetches ONE element per operand from the
// ring (per-lane addresses, 4 LDS clocks per read): 6 reads per four columns.  The earlier
// FMA formulation needed the sixteen u[j][k] in every lane -- 8 broadcast ds_read_b128 of 8 LDS
// clocks each per column; with up to three waves consuming every column that saturated the LDS
// (and slowed the owner's chain with it).  While it is a consumer, the wave keeps its block in the
// MFMA accumulator layout: acc[t][r] = element (row 16t + (lane>>4) + 4r, column 16w + (lane&15)).
// NV < 4: only the first NV of the four columns are published yet; the others are masked to zero.
// Flag and data come in ONE LDS round trip (the flag first: LDS serves a wave's reads in order, so
// if the flag was up the data behind it is valid; otherwise everything is re-read -- cheap now).
// The poll is bounded (about 0.2 s) so that a logic error could never hang the GPU: the owner of
// an earlier block never waits on anything, in practice a flag is up within a few hundred cycles.
template <int NV>
__device__ __forceinline__ void f64_consume_mfma(double4_t (&acc)[4], int lane, int w, int k0, int kvalid0,
                                                 const double* ring, int* ring_timeout) {
  // columns k0 + kvalid0 .. k0 + kvalid0 + NV - 1 are applied (the ones before were applied earlier)
  const int kq = lane >> 4, l15 = lane & 15;
  const double* col = ring + (k0 + kq) * PB;
  double rcv = 0.0, bu = 0.0, au[4] = {0.0, 0.0, 0.0, 0.0};
  bool up = false;
  for (int spins = 0; spins < (1 << 22); ++spins) {
    COMPILER_BARRIER();                                // LDS is re-read in every iteration
    const double flag = ring[(k0 + kvalid0 + NV - 1) * PB];
    COMPILER_BARRIER();                                // the flag read is issued before the data reads
    rcv = col[0];
    bu = col[16 * w + l15];
#pragma unroll
    for (int t = 0; t < 4; ++t) au[t] = col[16 * t + l15];
    if (flag != 0.0) { up = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  // (never seen: the owner of an earlier block waits on nothing) -- reported, not silently computed with
  if (!up && lane == 0) *ring_timeout = 1;
  const bool valid = (kq >= kvalid0) && (kq < kvalid0 + NV);
  const double bneg = valid ? -(bu * rcv) : 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(valid ? au[t] : 0.0, bneg, acc[t], 0, 0, 0);
}

__device__ __forceinline__ void f64_consume_block(double4_t (&acc)[4], int lane, int w, int kb, const double* ring,
                                                  int* ring_timeout) {
  const int kb0 = 16 * kb;
  f64_consume_mfma<4>(acc, lane, w, kb0, 0, ring, ring_timeout);
  f64_consume_mfma<4>(acc, lane, w, kb0 + 4, 0, ring, ring_timeout);
  f64_consume_mfma<4>(acc, lane, w, kb0 + 8, 0, ring, ring_timeout);
  // (taking the last four columns one at a time for the wave that owns the next block does not
  //  pay: every batch costs a full LDS round trip plus the MFMA latency, ~450 cycles)
  f64_consume_mfma<4>(acc, lane, w, kb0 + 12, 0, ring, ring_timeout);
}

// On return a[] holds this wave's 16 columns of L in the row-per-lane layout (zero above the
// diagonal); returns the first bad column of the wave's own block or -1.  stage: the pivot block as staged in LDS (row stride
// SPP); tbuf: 64 x 17 doubles of LDS private to this wave (layout change consumer -> owner).
// ring_timeout: a word of LDS, zeroed by the caller before its barrier, set if a column never came up.
// Inverse of wave w's 16 x 16 diagonal block of the factor (a[]: the wave's scaled columns, my_r = 1 / L[c][c] in
// lane c): see the comment inside factor64_waves.
__device__ __forceinline__ void factor64_inverse16(const double (&a)[16], int lane, int w, double* lbb, double* linv,
                                                   double* rdiag, double my_r) {
  if ((lane >> 4) == w) {
    const int il = lane & 15;
    double* lb = lbb + w * (16 * 17);
    double* li = linv + w * (16 * 17);
    rdiag[lane] = my_r;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) lb[il * 17 + kk] = a[kk];
    COMPILER_BARRIER();                              // same wave: LDS executes its operations in order
    double sv[16], rd[16], yv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = (i == il) ? 1.0 : 0.0; rd[i] = rdiag[16 * w + i]; }
    // (results are stored after the loop: a store inside it may alias the block's loads for all the
    //  compiler knows, which puts an LDS round trip into every step)
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const double y = sv[kk] * rd[kk];
      yv[kk] = y;
#pragma unroll
      for (int i = kk + 1; i < 16; ++i) sv[i] = fma(-lb[i * 17 + kk], y, sv[i]);
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) li[kk * 17 + il] = yv[kk];
  }
}

// DEFER3: wave 3 returns before the inverse of its 16 x 16 block (the last thing on the chain, ~4.3k cycles) with
// *my_r_out set; the caller runs factor64_inverse16 for it after the factor itself has been handed on.
template <bool DEFER3 = false>
__device__ __forceinline__ int factor64_waves(double (&a)[16], int lane, int w, const double* stage, double* tbuf,
                                              double* ring, double* lbb, double* linv, double* rdiag,
                                              int* ring_timeout, long long* dbg_stamp = nullptr, double* my_r_out = nullptr) {
  int bad = -1;
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = stage[lane * SPP_STAGE + j];       // lane <- row, columns 0..15
    __syncthreads();                                   // the staged block is read: its LDS may be reused
  } else {
    double4_t acc[4];
    const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = stage[(16 * t + kq + 4 * r) * SPP_STAGE + 16 * w + l15];
    __syncthreads();
    for (int kb = 0; kb < w; ++kb) f64_consume_block(acc, lane, w, kb, ring, ring_timeout);
    // accumulator layout -> row per lane, through this wave's private LDS block
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tbuf[(16 * t + kq + 4 * r) * 17 + l15] = acc[t][r];
    COMPILER_BARRIER();                                // same wave: LDS executes its operations in order
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = tbuf[lane * 17 + j];
  }
  if (dbg_stamp) dbg_stamp[0] = (long long)__builtin_amdgcn_s_memtime();
  f64_owner_block(a, lane, w, ring, bad, std::make_integer_sequence<int, 16>{});
  if (dbg_stamp) dbg_stamp[1] = (long long)__builtin_amdgcn_s_memtime();
  // scale the columns, L[:,k] = u[:,k] * sqrt(1/d_k): sixteen independent square-root chains,
  // written stage by stage across the sixteen so that no operation waits for its predecessor
  // (one chain after the other costs 16 x 15 dependent operations of ~25 cycles)
  double my_r = 1.0;
  {
    double x[16], y[16], h[16], e[16], sq[16];
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) x[kl] = ring[(16 * w + kl) * PB];      // 1/d_k as published
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) { y[kl] = __builtin_amdgcn_rsq(x[kl]); h[kl] = 0.5 * x[kl]; }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int kl = 0; kl < 16; ++kl) e[kl] = fma(-(h[kl] * y[kl]), y[kl], 0.5);
#pragma unroll
      for (int kl = 0; kl < 16; ++kl) y[kl] = fma(y[kl], e[kl], y[kl]);     // y = 1/sqrt(x) = sqrt(d_k) = L[k][k]
    }
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) sq[kl] = x[kl] * y[kl];
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) sq[kl] = fma(fma(-sq[kl], sq[kl], x[kl]), 0.5 * y[kl], sq[kl]);   // sqrt(x) = 1/L[k][k]
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) e[kl] = fma(-sq[kl], y[kl], 1.0);
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) y[kl] = fma(e[kl], y[kl], y[kl]);       // L[k][k], consistent with sq
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) {
      const int k = 16 * w + kl;
      a[kl] = (lane == k) ? y[kl] : ((lane > k) ? a[kl] * sq[kl] : 0.0);
      my_r = (lane == k) ? sq[kl] : my_r;              // 1 / L[k][k]
    }
  }
  if (dbg_stamp) dbg_stamp[2] = (long long)__builtin_amdgcn_s_memtime();
  // Inverse of the wave's 16 x 16 diagonal block (for the MFMA row solve): its sixteen rows sit in
  // lanes 16w .. 16w+15; lane 16w + j computes column j of the inverse by right-looking forward
  // substitution (two dependent operations per step), the block's entries coming as LDS broadcasts
  // among those sixteen lanes.  Waves 0-2 do this while later waves still factor; only wave 3's is
  // exposed.  (A helper wave following the owner's published, unscaled columns -- the inverse needs
  // only u and 1/d -- was tried to hide that one too; the follower ran at ~440 cycles per column
  // against the owner's 225 and ended later.)
  if (DEFER3 && w == 3) {
    *my_r_out = my_r;
    return bad;
  }
  factor64_inverse16(a, lane, w, lbb, linv, rdiag, my_r);
  if (dbg_stamp) dbg_stamp[3] = (long long)__builtin_amdgcn_s_memtime();
  return bad;
}

// ---- the factorisation of the first klast + 1 columns only, without the 16 x 16 inverses (round 6) ----
// (the tuning objective's last diagonal tile and the one-tile system of k_lml_tiny64: nothing reads the columns beyond the
//  last observation's, and no tile below needs the inverses)
template <int... KLs>
__device__ __forceinline__ void f64_owner_block_upto(double (&a)[16], int lane, int w, double* ring, int& bad, int klast,
                                                     std::integer_sequence<int, KLs...>) {
  double mprev = 0.0;
  ((16 * w + KLs <= klast ? f64_owner_step<KLs>(a, lane, w, ring, bad, mprev) : (void)0), ...);
}

// factor64_waves without the 16 x 16 inverses, columns 0 .. klast only.  a[]: this wave's sixteen columns of L, row per
// lane (garbage in columns beyond klast).  Returns the first non-positive pivot column of the wave's block or -1.
__device__ __forceinline__ int tiny64_factor(double (&a)[16], int lane, int w, const double* stage, double* tbuf,
                                             double* ring, int klast, int* ring_timeout) {
  int bad = -1;
  const bool active = 16 * w <= klast;                 // wave-uniform
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = stage[lane * SPP_STAGE + j];
    __syncthreads();
  } else {
    double4_t acc[4];
    const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = stage[(16 * t + kq + 4 * r) * SPP_STAGE + 16 * w + l15];
    __syncthreads();
    if (!active) return -1;
    for (int kb = 0; kb < w; ++kb) f64_consume_block(acc, lane, w, kb, ring, ring_timeout);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tbuf[(16 * t + kq + 4 * r) * 17 + l15] = acc[t][r];
    COMPILER_BARRIER();                                // same wave: LDS executes its operations in order
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = tbuf[lane * 17 + j];
  }
  f64_owner_block_upto(a, lane, w, ring, bad, klast, std::make_integer_sequence<int, 16>{});
  // L[:,k] = u[:,k] * sqrt(1/d_k), sixteen independent chains stage by stage (as factor64_waves)
  double x[16], y[16], h[16], e[16], sq[16];
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) { x[kl] = ring[(16 * w + kl) * PB]; x[kl] = (16 * w + kl <= klast) ? x[kl] : 1.0; }
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) { y[kl] = __builtin_amdgcn_rsq(x[kl]); h[kl] = 0.5 * x[kl]; }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) e[kl] = fma(-(h[kl] * y[kl]), y[kl], 0.5);
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) y[kl] = fma(y[kl], e[kl], y[kl]);
  }
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) sq[kl] = x[kl] * y[kl];
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) sq[kl] = fma(fma(-sq[kl], sq[kl], x[kl]), 0.5 * y[kl], sq[kl]);
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) e[kl] = fma(-sq[kl], y[kl], 1.0);
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) y[kl] = fma(e[kl], y[kl], y[kl]);
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) {
    const int k = 16 * w + kl;
    a[kl] = (lane == k) ? y[kl] : ((lane > k) ? a[kl] * sq[kl] : 0.0);
  }
  return bad;
}

// sum over the four lanes of a quad, in every lane (DPP quad_perm, no LDS crossbar)
__device__ __forceinline__ double quad_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int lo1 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  int hi1 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  v += __hiloint2double(hi1, lo1);
  lo = __double2loint(v); hi = __double2hiint(v);
  lo1 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
  hi1 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
  return v + __hiloint2double(hi1, lo1);
}

