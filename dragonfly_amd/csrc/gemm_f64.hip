// fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64).
//
//   Cout[M x N] = beta * Cin + alpha * A[M x K] * op(B)
//   op(B) = B^T with B[N x K]  ("NT", both operands K-contiguous: SYRK / TRSM-by-inverse)
//         = B   with B[K x N]  ("NN")
//
// This is the one dense contraction on the GP hot path: it stands in for the dgemm /
// dpotrf-trailing-update / dtrsm work NumPy+SciPy hand to OpenBLAS/LAPACK in the reference
// (dragonfly/utils/general_utils.py:68,178,213; dragonfly/gp/gp_core.py:174,180,181).
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as 2x2, each wave
// owns 64x64 = 4x4 MFMA tiles (16 accumulators x 4 f64 = 128 VGPRs).  K is consumed in
// chunks of 16 through a double-buffered, padded LDS image (one barrier per chunk); global
// loads of chunk c+1 are issued before the MFMAs of chunk c, and the MFMA fragments are read from
// LDS one k-pair ahead into a second register set, so every HBM/L2 and LDS latency hides under
// the matrix pipe.  2 workgroups per CU (73.7 KB LDS, 250 VGPRs): the second wave per SIMD fills
// the matrix pipe while the first sits at the barrier (measured: 1 workgroup per CU is ~10 % slower).
//
// MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3):
//   A operand : lane l holds A[i = l&15][k = l>>4]
//   B operand : lane l holds B[k = l>>4][j = l&15]
//   C/D       : reg r of lane l holds D[i = (l>>4) + 4r][j = l&15]
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <utility>

namespace {

constexpr int BK = 16;
constexpr int BKP = 18;    // row stride (doubles) of the K-contiguous LDS tiles: 2*(r*18+k) mod 64
                           // is distinct over r<16,k<2 -> ds_read_b64 conflict free
// Tile geometry, WT = MFMA tiles per wave per dimension: WT=4 -> 128x128 block tile (the
// throughput configuration), WT=2 -> 64x64 (small / latency-bound problems: 4x more workgroups).
template <int WT> struct Geo {
  static constexpr int BM = 32 * WT, BN = 32 * WT;
  static constexpr int BNP = BN + 16;                  // NN B tile [BK][BN] row stride: 2*BNP mod 64 = 32
  static constexpr int TILE_A = BM * BKP;              // doubles per A stage
  static constexpr int TILE_B = (BN * BKP > BK * BNP) ? BN * BKP : BK * BNP;
  static constexpr int SMEM_BYTES = 2 * (TILE_A + TILE_B) * 8;   // 73728 (WT=4), 36864 (WT=2)
  static constexpr int PA = BM / 32;                   // 32-row load passes per operand tile
  static constexpr int NN_LANES = BN / 2;              // lanes per k-row of the NN B tile
  static constexpr int NN_ROWS = 256 / NN_LANES;       // k-rows per pass
  static constexpr int NN_PASSES = BK / NN_ROWS;
};

constexpr unsigned COND_GRID = 64;     // workgroups of a conditional (usually no-op) launch

#ifndef DFH_GEMM_GROUP_M
#define DFH_GEMM_GROUP_M 8      // row tiles per group of the plain tile order (map_tile)
#endif

struct GemmArgs {
  int M, N, K;
  const double* A; long lda; long sA; long sA2;
  const double* B; long ldb; long sB; long sB2;
  const double* Cin; long ldcin; long sCin; long sCin2;
  double* Cout; long ldc; long sCout; long sCout2;
  int cnt1;
  double alpha, beta;
  int flags;
  int tiles_m, tiles_n;
  int bm, bn;
};
// extended launches only (gemm_f64_ext_kernel): a second kernel argument, so that the plain kernels'
// argument block -- and with it their code -- is what it has always been
struct GemmExt {
  const double* cond; double cond_thr;   // skip the whole launch unless *cond > cond_thr (NaN runs)
  int* la_cnt;                           // look-ahead order (LOWER): {diagonal-block tiles done, block-column tiles done}
  int la_pad;                            // workgroups reserved for the look-ahead tiles (multiple of 8)
  int vgrid;                             // tiles (incl. look-ahead padding) the launch walks
};

__device__ __forceinline__ void map_tile(const GemmArgs& p, unsigned b, unsigned nb, int& tm, int& tn) {
  // XCD-aware remap: hardware places block b on XCD b%8; give each XCD a contiguous span of
  // the linear tile order so neighbouring tiles (sharing A/B panels) share one L2.
  const unsigned q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
  unsigned lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  if (p.flags & GEMM_LOWER) {
    // row-by-row enumeration of the lower-triangular tile set.  (Round 2 tried 8 x 8 super-blocks
    // instead: rocprofv3 FETCH_SIZE of the K = 512 SYRK fell from 5.0 to 2.7 GB per launch, its time
    // did not move (54 TF/s either way) -- and the extra index arithmetic in this prologue shifted
    // the start times of the workgroups of the PLAIN tile order enough to nearly triple the L2
    // misses of the posterior TRSM products (3.0 -> 8.6 GB per 32768 x 512 x 8192 launch, +3 % time:
    // tiles that share an A panel miss together instead of one after the other).  Reverted; see
    // DESIGN.md section 7.)
    unsigned i = (unsigned)((sqrt(8.0 * (double)lin + 1.0) - 1.0) * 0.5);
    while ((unsigned long long)i * (i + 1) / 2 > lin) --i;
    while ((unsigned long long)(i + 1) * (i + 2) / 2 <= lin) ++i;
    tm = (int)i;
    tn = (int)(lin - (unsigned)((unsigned long long)i * (i + 1) / 2));
  } else {
    constexpr unsigned GROUP_M = DFH_GEMM_GROUP_M;
    const unsigned width = GROUP_M * (unsigned)p.tiles_n;
    const unsigned group = lin / width;
    const unsigned first_m = group * GROUP_M;
    const unsigned gsz = min((unsigned)p.tiles_m - first_m, GROUP_M);
    const unsigned in = lin % width;
    tm = (int)(first_m + in % gsz);
    tn = (int)(in / gsz);
  }
}

// Look-ahead tile order of the factorisation's trailing update (extended launches, LOWER): the first
// block column (four tile columns = the next panel) comes first -- workgroup b < la_pad takes tile
// row 8 (b >> 5) + (b & 7), tile column (b >> 3) & 3, so the four tiles of a row share an XCD (and its
// L2 copy of the row's A panel) and the sixteen tiles of the next diagonal block are the very first
// -- and raises the counters in la_cnt when it is done; the remaining tiles are the lower triangle
// of the (tiles_m - 4)-tile matrix behind it, in the plain row-by-row order with the same
// XCD-contiguous spans.  Returns false for a padding workgroup.
__device__ __forceinline__ bool map_tile_lookahead(const GemmArgs& p, int la_pad, unsigned b, unsigned vgrid, int& tm,
                                                   int& tn, bool& is_la) {
  if (b < (unsigned)la_pad) {
    const unsigned xcd = b & 7, idx = b >> 3;
    const unsigned i = (idx >> 2) * 8 + xcd, j = idx & 3;
    is_la = true;
    tm = (int)i; tn = (int)j;
    return i < (unsigned)p.tiles_m && j <= i;
  }
  is_la = false;
  const unsigned nb = vgrid - (unsigned)la_pad, bb = b - (unsigned)la_pad;
  const unsigned q = nb >> 3, r = nb & 7, xcd = bb & 7, idx = bb >> 3;
  const unsigned lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  unsigned i = (unsigned)((sqrt(8.0 * (double)lin + 1.0) - 1.0) * 0.5);
  while ((unsigned long long)i * (i + 1) / 2 > lin) --i;
  while ((unsigned long long)(i + 1) * (i + 2) / 2 <= lin) ++i;
  tm = (int)i + 4;
  tn = (int)(lin - (unsigned)((unsigned long long)i * (i + 1) / 2)) + 4;
  return true;
}

template <bool TRANSB, bool EDGE, int WT, bool EXT>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, int tm, int tn, bool is_la, int* la_cnt, double* smem) {
  using G = Geo<WT>;
  constexpr int BM = G::BM, BN = G::BN, BNP = G::BNP, TILE_A = G::TILE_A, TILE_B = G::TILE_B;
  constexpr int WS = WT * 16;                          // wave tile edge
  double* As = smem;                  // [2][TILE_A]
  double* Bs = smem + 2 * TILE_A;     // [2][TILE_B]
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  const long bz1 = blockIdx.z % p.cnt1, bz2 = blockIdx.z / p.cnt1;
  const double* __restrict__ A = p.A + bz1 * p.sA + bz2 * p.sA2;
  const double* __restrict__ B = p.B + bz1 * p.sB + bz2 * p.sB2;

  int kend = p.K;
  if (p.flags & GEMM_KTRI_B) kend = min(p.K, n0 + BN);   // B[j][k] = 0 for k > j
  const int nchunks = (kend + BK - 1) / BK;

  // C enters through the accumulators when alpha = +-1 (the SYRK / TRSM-update case): the tile of
  // beta*C is loaded at kernel entry, in flight together with the first operand chunk, instead of
  // being read -- one exposed HBM round trip per tile -- in the epilogue.  out = alpha * acc.
  const double* __restrict__ Cin0 = p.Cin ? p.Cin + bz1 * p.sCin + bz2 * p.sCin2 : nullptr;
  const bool c_in_acc = (Cin0 != nullptr) && (p.alpha == 1.0 || p.alpha == -1.0);
  double4_t acc[WT][WT];
  if (c_in_acc) {
    const double cscale = p.beta * p.alpha;
#pragma unroll
    for (int i = 0; i < WT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WS + i * 16 + l4 + 4 * r;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          const int col = n0 + wn * WS + j * 16 + l15;
          double v = 0.0;
          if (!EDGE || (row < p.M && col < p.N)) v = cscale * Cin0[(long)row * p.ldcin + col];
          acc[i][j][r] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  }

  // staging registers: 4 x 16B for A and 4 x 16B for B per thread per chunk
  double2_t ra[G::PA], rb[(G::PA > G::NN_PASSES) ? G::PA : G::NN_PASSES];

  const int a_row = tid >> 3, a_col = (tid & 7) * 2;      // K-contiguous tiles: 8 lanes per row
  const int b_krow = tid / G::NN_LANES, b_ncol = (tid % G::NN_LANES) * 2;   // NN B tile

  auto load_chunk = [&](int kc) {
    const int k0 = kc * BK;
#pragma unroll
    for (int q = 0; q < G::PA; ++q) {
      const int row = a_row + 32 * q;
      if (!EDGE) {
        ra[q] = *reinterpret_cast<const double2_t*>(A + (long)(m0 + row) * p.lda + k0 + a_col);
      } else {
        const int gr = m0 + row, gk = k0 + a_col;
        const double* src = A + (long)gr * p.lda + gk;
        ra[q].x = (gr < p.M && gk < kend) ? src[0] : 0.0;
        ra[q].y = (gr < p.M && gk + 1 < kend) ? src[1] : 0.0;
      }
    }
    if (!TRANSB) {
#pragma unroll
      for (int q = 0; q < G::PA; ++q) {
        const int row = a_row + 32 * q;
        if (!EDGE) {
          rb[q] = *reinterpret_cast<const double2_t*>(B + (long)(n0 + row) * p.ldb + k0 + a_col);
        } else {
          const int gr = n0 + row, gk = k0 + a_col;
          const double* src = B + (long)gr * p.ldb + gk;
          rb[q].x = (gr < p.N && gk < kend) ? src[0] : 0.0;
          rb[q].y = (gr < p.N && gk + 1 < kend) ? src[1] : 0.0;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < G::NN_PASSES; ++q) {
        const int krow = b_krow + G::NN_ROWS * q;
        if (!EDGE) {
          rb[q] = *reinterpret_cast<const double2_t*>(B + (long)(k0 + krow) * p.ldb + n0 + b_ncol);
        } else {
          const int gk = k0 + krow, gn = n0 + b_ncol;
          const double* src = B + (long)gk * p.ldb + gn;
          rb[q].x = (gk < kend && gn < p.N) ? src[0] : 0.0;
          rb[q].y = (gk < kend && gn + 1 < p.N) ? src[1] : 0.0;
        }
      }
    }
  };

  auto store_chunk = [&](int buf) {
    double* as = As + buf * TILE_A;
    double* bs = Bs + buf * TILE_B;
#pragma unroll
    for (int q = 0; q < G::PA; ++q)
      *reinterpret_cast<double2_t*>(as + (a_row + 32 * q) * BKP + a_col) = ra[q];
    if (!TRANSB) {
#pragma unroll
      for (int q = 0; q < G::PA; ++q)
        *reinterpret_cast<double2_t*>(bs + (a_row + 32 * q) * BKP + a_col) = rb[q];
    } else {
#pragma unroll
      for (int q = 0; q < G::NN_PASSES; ++q)
        *reinterpret_cast<double2_t*>(bs + (b_krow + G::NN_ROWS * q) * BNP + b_ncol) = rb[q];
    }
  };

  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();

  // Fragment registers for the two k-pairs of a chunk (a pair = two MFMA k-steps = 8 columns).
  // Schedule per chunk c (one wave per SIMD, so nothing else hides a stall):
  //   global loads of chunk c+1 | LDS reads of pair 1 (chunk c) | 32 MFMAs on pair 0 |
  //   LDS writes of chunk c+1, barrier | LDS reads of pair 0 (chunk c+1) | 32 MFMAs on pair 1
  // so every LDS read has a 32-MFMA (~2000 cycle) shadow and the only exposed work between the
  // two MFMA groups is the store + barrier.
  double fa[2][WT][2], fb[2][WT][2];
  auto read_frags = [&](int buf, int pair, int slot) {
    const double* as = As + buf * TILE_A + (wm * WS + l15) * BKP + l4 + pair * 8;
    const double* bs = TRANSB ? (Bs + buf * TILE_B + (l4 + pair * 8) * BNP + wn * WS + l15)
                              : (Bs + buf * TILE_B + (wn * WS + l15) * BKP + l4 + pair * 8);
#pragma unroll
    for (int t = 0; t < WT; ++t) {
      fa[slot][t][0] = as[t * 16 * BKP];
      fa[slot][t][1] = as[t * 16 * BKP + 4];
    }
#pragma unroll
    for (int t = 0; t < WT; ++t) {
      fb[slot][t][0] = TRANSB ? bs[t * 16] : bs[t * 16 * BKP];
      fb[slot][t][1] = TRANSB ? bs[4 * BNP + t * 16] : bs[t * 16 * BKP + 4];
    }
  };
  auto mfma_pair = [&](int slot) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i][ks], fb[slot][j][ks], acc[i][j], 0, 0, 0);
  };

  if (WT == 4) {
    if (nchunks > 0) read_frags(0, 0, 0);
    for (int kc = 0; kc < nchunks; ++kc) {
      const int buf = kc & 1;
      const bool more = kc + 1 < nchunks;
      if (more) load_chunk(kc + 1);
      read_frags(buf, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair(0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) store_chunk(buf ^ 1);
      __syncthreads();
      if (more) read_frags(buf ^ 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // small / latency-bound problems (64 x 64 tiles, few chunks): plain double-buffered loop
    for (int kc = 0; kc < nchunks; ++kc) {
      const int buf = kc & 1;
      if (kc + 1 < nchunks) load_chunk(kc + 1);
      read_frags(buf, 0, 0);
      read_frags(buf, 1, 1);
      mfma_pair(0);
      mfma_pair(1);
      if (kc + 1 < nchunks) store_chunk(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue
  const double alpha = p.alpha, beta = p.beta;
  const double* __restrict__ Cin = Cin0;
  double* __restrict__ Cout = p.Cout + bz1 * p.sCout + bz2 * p.sCout2;
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * WS + i * 16 + l4 + 4 * r;
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const int col = n0 + wn * WS + j * 16 + l15;
        if (EDGE && (row >= p.M || col >= p.N)) continue;
        double v = alpha * acc[i][j][r];
        if (!c_in_acc && beta != 0.0) v += beta * Cin[(long)row * p.ldcin + col];
        // a look-ahead tile is handed to the panel kernel of another CU (possibly another XCD): it goes
        // out with write-through (sc1) stores -- a release fence per tile instead writes back the XCD's
        // whole L2, full of the other workgroups' freshly written C tiles (496 tiles per launch: the
        // look-ahead order cost 5 % of the update that way)
        if (EXT && is_la) __hip_atomic_store(&Cout[(long)row * p.ldc + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else Cout[(long)row * p.ldc + col] = v;
      }
    }
  }
  if (EXT && is_la) {
    // announce the tile once it is out: s_barrier does NOT wait for outstanding global stores on
    // gfx950 (the compiler emits no vmcnt wait ahead of it), so every wave drains its own stores
    // first; only then may the counter go up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (tm < 4) __hip_atomic_fetch_add(la_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(la_cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Plain launches: one tile per workgroup.  Extended launches (EXT, the factorisation's): a device-side
// condition decides whether the launch does anything -- such a launch is a small persistent grid
// walking the tiles (vgrid of them), because even a workgroup that exits at once needs a free 74 KB LDS
// slot to START: a full-size grid of no-ops would queue behind the trailing update it runs beside;
// look-ahead launches have one workgroup per tile as usual.
template <bool TRANSB, bool EDGE, int WT>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // the 64 x 64-tile instantiation serves small, latency-bound products -- the factorisation's chain
  // among them, whose waves share SIMDs with the trailing update's: they go first
  if (WT == 2) __builtin_amdgcn_s_setprio(3);
  int tm, tn;
  map_tile(p, blockIdx.x, gridDim.x, tm, tn);
  gemm_tile<TRANSB, EDGE, WT, false>(p, tm, tn, false, nullptr, smem);
}

// look-ahead order: one tile per workgroup, exactly like the plain kernel but for the tile map and the
// completion counters (a tile-walking loop around the tile body cost 8 % on its own: 22.2 instead of
// 20.6 ms over the sixteen largest trailing updates of n = 16384)
template <bool EDGE>
__global__ __launch_bounds__(256, 2) void gemm_f64_la_kernel(GemmArgs p, GemmExt x) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  int tm, tn;
  bool is_la = false;
  if (!map_tile_lookahead(p, x.la_pad, blockIdx.x, gridDim.x, tm, tn, is_la)) return;
  // two copies of the tile body: the (few) look-ahead tiles take the one with the hand-off epilogue,
  // all others exactly the plain kernel's (with the hand-off merely compiled in, every tile was 4 % slower)
  if (is_la) gemm_tile<false, EDGE, 4, true>(p, tm, tn, true, x.la_cnt, smem);
  else gemm_tile<false, EDGE, 4, false>(p, tm, tn, false, nullptr, smem);
}

// conditional launch: a small grid walking the tiles, or nothing at all
template <bool EDGE>
__global__ __launch_bounds__(256, 2) void gemm_f64_cond_kernel(GemmArgs p, GemmExt x) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const double dv = *x.cond;
  if (dv <= x.cond_thr) return;                        // (a NaN runs)
  for (unsigned vb = blockIdx.x; vb < (unsigned)x.vgrid; vb += gridDim.x) {
    int tm, tn;
    map_tile(p, vb, (unsigned)x.vgrid, tm, tn);
    gemm_tile<false, EDGE, 4, false>(p, tm, tn, false, nullptr, smem);
    __syncthreads();                                   // the LDS images are free for the next tile
  }
}

template <bool TRANSB, bool EDGE, int WT, bool EXT = false>
int launch(dfh_ctx* ctx, const GemmArgs& p, dim3 grid, const GemmExt* x = nullptr) {
  static bool attr_set_dev[2][DFH_MAX_DEVICES] = {{false}, {false}};
  bool& attr_set = attr_set_dev[(EXT && x->la_cnt != nullptr) ? 1 : 0][ctx->device];
  const bool la = EXT && x->la_cnt != nullptr;
  const void* kern = !EXT ? reinterpret_cast<const void*>(gemm_f64_kernel<TRANSB, EDGE, WT>)
                          : (la ? reinterpret_cast<const void*>(gemm_f64_la_kernel<EDGE>)
                                : reinterpret_cast<const void*>(gemm_f64_cond_kernel<EDGE>));
  constexpr int HALF_OCC_SMEM = 84 * 1024;            // two of these do not fit in 160 KB
  constexpr int MAX_SMEM = (WT == 4) ? HALF_OCC_SMEM : Geo<WT>::SMEM_BYTES;
  if (!attr_set) {
    DFH_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_SMEM));
    attr_set = true;
  }
  // Optional half occupancy (> 80 KB of LDS keeps a second workgroup off the CU): used for bulk
  // GEMMs that run beside latency-critical kernels of another stream, which then find free slots.
  const int smem = (WT == 4 && ctx->gemm_half_occupancy) ? HALF_OCC_SMEM : Geo<WT>::SMEM_BYTES;
  dfh_ctx::GemmRec* rec = nullptr;
  if (ctx->gemm_prof) {
    if (ctx->gemm_used == ctx->gemm_recs.size()) {
      dfh_ctx::GemmRec r;
      DFH_HIP(hipEventCreate(&r.e0));
      DFH_HIP(hipEventCreate(&r.e1));
      r.flops = 0.0; r.bytes = 0.0; r.variant = 0;
      ctx->gemm_recs.push_back(r);
    }
    rec = &ctx->gemm_recs[ctx->gemm_used++];
    // algorithmic flops: only the triangle LOWER asks for, only the k-range a triangular B has
    double f = 2.0 * (double)p.M * (double)p.N * (double)p.K;
    if (p.flags & GEMM_LOWER) f = (double)p.M * ((double)p.M + 1.0) * (double)p.K;
    if (p.flags & GEMM_KTRI_B) f = (double)p.M * (double)p.N * ((double)p.N + 1.0);
    rec->flops = f * (double)grid.z;
    // algorithmic bytes: each operand once, the output tile written (and read when it is updated)
    const double c_elems = (p.flags & GEMM_LOWER) ? 0.5 * (double)p.M * ((double)p.M + 1.0) : (double)p.M * (double)p.N;
    const double b_elems = (p.flags & GEMM_KTRI_B) ? 0.5 * (double)p.N * ((double)p.N + 1.0) : (double)p.N * (double)p.K;
    rec->bytes = 8.0 * (double)grid.z * ((double)p.M * (double)p.K + (p.B == p.A ? 0.0 : b_elems) +
                                         c_elems * ((p.Cin != nullptr && p.beta != 0.0) ? 2.0 : 1.0));
    rec->variant = (TRANSB ? 4 : 0) | (EDGE ? 2 : 0) | (WT == 2 ? 1 : 0);
    // the factorisation's extended launches are kernels of their own (gemm_f64_la_kernel / _cond_kernel in a
    // rocprof trace) and are booked apart: slots 7 / 6, which no NN edge launch of the library's paths uses
    if (EXT) rec->variant = la ? 7 : 6;
    DFH_HIP(hipEventRecord(rec->e0, ctx->stream));
  }
  if constexpr (EXT) {
    if (la) hipLaunchKernelGGL(gemm_f64_la_kernel<EDGE>, grid, dim3(256), smem, ctx->stream, p, *x);
    else hipLaunchKernelGGL(gemm_f64_cond_kernel<EDGE>, grid, dim3(256), smem, ctx->stream, p, *x);
  } else hipLaunchKernelGGL((gemm_f64_kernel<TRANSB, EDGE, WT>), grid, dim3(256), smem, ctx->stream, p);
  DFH_LAUNCH_CHECK();
  if (rec) DFH_HIP(hipEventRecord(rec->e1, ctx->stream));
  return DFH_OK;
}

template <int WT>
int dispatch(dfh_ctx* ctx, GemmArgs& p, int count, bool edge, const GemmExt* ext = nullptr) {
  constexpr int BM = Geo<WT>::BM, BN = Geo<WT>::BN;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const long ntiles = (p.flags & GEMM_LOWER) ? (long)p.tiles_m * (p.tiles_m + 1) / 2 : (long)p.tiles_m * p.tiles_n;
  dim3 grid((unsigned)ntiles, 1, (unsigned)count);
  edge = edge || (p.M % BM) || (p.N % BN);
  if constexpr (WT == 4) {
    if (ext) {
      // extended launch (factorisation only): device-side skip condition and / or look-ahead order
      GemmExt x = *ext;
      if ((p.flags & GEMM_TRANSB) || count != 1) { dfh_set_error("extended GEMM launch: NT, unbatched only"); return DFH_ERR_BAD_ARG; }
      if (x.la_cnt) {
        if (!(p.flags & GEMM_LOWER) || p.tiles_m < 5) { dfh_set_error("look-ahead GEMM order needs a LOWER product of >= 5 tile rows"); return DFH_ERR_BAD_ARG; }
        x.la_pad = 32 * ((p.tiles_m + 7) / 8);
        const long rest = (long)(p.tiles_m - 4) * (p.tiles_m - 3) / 2;
        grid.x = (unsigned)(x.la_pad + rest);
      }
      x.vgrid = (int)grid.x;
      if (x.cond && x.la_cnt) { dfh_set_error("extended GEMM launch: condition and look-ahead order exclude each other"); return DFH_ERR_BAD_ARG; }
      if (x.cond && grid.x > COND_GRID) grid.x = COND_GRID;
      return edge ? launch<false, true, 4, true>(ctx, p, grid, &x) : launch<false, false, 4, true>(ctx, p, grid, &x);
    }
  }
  if (p.flags & GEMM_TRANSB)
    return edge ? launch<true, true, WT>(ctx, p, grid) : launch<true, false, WT>(ctx, p, grid);
  return edge ? launch<false, true, WT>(ctx, p, grid) : launch<false, false, WT>(ctx, p, grid);
}

// ---------------------------------------------------------------------------------------
// Few-row NT GEMM:  C[m x N] = beta * Cin + alpha * A[m x K] * B[N x K]^T  for m <= 256 rows.
//
// The posterior solve of a handful of points (single GP.eval calls, tree-search frontiers,
// block-row appends) is a chain of such products in which B is a 512-column panel of the factor L
// with up to n rows: the work is reading B once.  The square-tile kernel above is the wrong shape
// for it -- a 64x64 tile per workgroup whose K loop waits for HBM every 16 columns, with 63 of 64
// accumulator rows computing padding: 32 us per launch.  Here a wave owns 16 columns of C (16 rows
// of B) and ALL m rows: B streams from HBM through registers exactly once, 128 contiguous bytes per
// row and load, two K slabs in flight; A (tiny) is staged per 64-column K slab in LDS in the
// operand order of v_mfma_f64_16x16x4; the m/16 accumulator tiles stay in registers.
// K order inside a slab is permuted (lane group g takes columns 16t+4g..16t+4g+3) -- the same
// permutation for A and B, so the sum is over the same products -- to make both operand loads
// 32 contiguous bytes per lane.  A row's result depends on K and on nothing else: a point gets the
// same value alone or inside any batch.
// ---------------------------------------------------------------------------------------
constexpr int SK_SLAB = 64;                 // K columns per LDS slab
constexpr int SK_LDA = SK_SLAB + 4;         // slab row stride in LDS (doubles)
constexpr int SK_MAX_ROWS = 256;

struct SkinnyArgs {
  const double* A; long lda;
  const double* B; long ldb;
  const double* Cin; long ldcin;
  double* Cout; long ldc;
  double* Acopy; long ldacopy;       // optional: workgroup 0 also copies A there (A is staged anyway)
  int m, N, K;
  double alpha, beta;
};

// SPLIT: narrow C (the 512-column block solve: only eight 64-column workgroups' worth of work) --
// a workgroup then owns 16 columns and its four waves split the row tiles, four times as many
// workgroups each a quarter as long.
template <int MT, bool SPLIT>      // MT 16-row tiles of C per workgroup: m <= 16 * MT
__global__ __launch_bounds__(256) void gemm_skinny_kernel(SkinnyArgs p) {
  extern __shared__ __attribute__((aligned(32))) double slab[];     // [16 * MT][SK_LDA]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int m = p.m;
  constexpr int MTW = SPLIT ? MT / 4 : MT;                          // row tiles of this wave
  const int tile0 = SPLIT ? wave * MTW : 0;
  const long j = SPLIT ? (long)blockIdx.x * 16 + l15                 // this lane's row of B / column of C
                       : (long)blockIdx.x * 64 + wave * 16 + l15;
  const long jc = j < p.N ? j : p.N - 1;                            // clamp: loads stay in bounds
  const double* brow = p.B + jc * p.ldb + 4 * g;
  double4_t acc[MTW];
#pragma unroll
  for (int t = 0; t < MTW; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

  double4_t bnext[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) bnext[t] = *reinterpret_cast<const double4_t*>(brow + 16 * t);
  // A slab: 16 * MT rows x 64 columns = 2 * MT double2 per thread, prefetched like B so that no
  // iteration waits for a global load it has just issued
  // (the 256-row variant has no registers left for it and stages A straight from global memory)
  constexpr bool PREFETCH_A = MT <= 8;
  constexpr int AREGS = 2 * MT;
  double2_t anext[PREFETCH_A ? AREGS : 1];
  auto load_a = [&](int k0) {
    if (!PREFETCH_A) return;
#pragma unroll
    for (int q = 0; q < AREGS; ++q) {
      const int idx = tid + 256 * q, r = idx / (SK_SLAB / 2), c2 = idx - r * (SK_SLAB / 2);
      anext[PREFETCH_A ? q : 0] = (double2_t){0.0, 0.0};
      if (r < m) anext[PREFETCH_A ? q : 0] = *reinterpret_cast<const double2_t*>(p.A + (long)r * p.lda + k0 + 2 * c2);
    }
  };
  load_a(0);
  for (int k0 = 0; k0 < p.K; k0 += SK_SLAB) {
    double4_t bcur[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bcur[t] = bnext[t];
    __syncthreads();                        // the previous slab has been consumed
#pragma unroll
    for (int q = 0; q < AREGS; ++q) {
      const int idx = tid + 256 * q, r = idx / (SK_SLAB / 2), c2 = idx - r * (SK_SLAB / 2);
      double2_t v;
      if (PREFETCH_A) {
        v = anext[PREFETCH_A ? q : 0];
      } else {
        v = (double2_t){0.0, 0.0};
        if (r < m) v = *reinterpret_cast<const double2_t*>(p.A + (long)r * p.lda + k0 + 2 * c2);
      }
      *reinterpret_cast<double2_t*>(slab + r * SK_LDA + 2 * c2) = v;
      if (p.Acopy && blockIdx.x == 0 && r < m)
        *reinterpret_cast<double2_t*>(p.Acopy + (long)r * p.ldacopy + k0 + 2 * c2) = v;
    }
    if (k0 + SK_SLAB < p.K) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        bnext[t] = *reinterpret_cast<const double4_t*>(brow + k0 + SK_SLAB + 16 * t);
      load_a(k0 + SK_SLAB);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int it = 0; it < MTW; ++it) {
        const double4_t a = *reinterpret_cast<const double4_t*>(slab + ((tile0 + it) * 16 + l15) * SK_LDA + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], bcur[t][r], acc[it], 0, 0, 0);
      }
    }
  }
  if (j < p.N) {
#pragma unroll
    for (int it = 0; it < MTW; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = (tile0 + it) * 16 + g + 4 * r;
        if (i < m) {
          double v = p.alpha * acc[it][r];
          if (p.Cin) v += p.beta * p.Cin[(long)i * p.ldcin + j];
          p.Cout[(long)i * p.ldc + j] = v;
        }
      }
    }
  }
}

template <int MT, bool SPLIT>
int launch_skinny(dfh_ctx* ctx, const SkinnyArgs& p) {
  const int cols = SPLIT ? 16 : 64;
  hipLaunchKernelGGL((gemm_skinny_kernel<MT, SPLIT>), dim3((unsigned)((p.N + cols - 1) / cols)), dim3(256),
                     sizeof(double) * 16 * MT * SK_LDA, ctx->stream, p);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

}  // namespace

bool gemm_skinny_applies(int64_t m, int64_t N, int64_t K, const double* A, int64_t lda, const double* B,
                         int64_t ldb) {
  auto aligned32 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 31) == 0; };
  return m >= 1 && m <= SK_MAX_ROWS && N >= 1 && K >= SK_SLAB && K % SK_SLAB == 0 && (lda % 2) == 0 &&
         (ldb % 4) == 0 && aligned32(B) && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

int gemm_skinny_nt(dfh_ctx* ctx, int64_t m, int64_t N, int64_t K, double alpha, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double beta, const double* Cin, int64_t ldcin, double* Cout,
                   int64_t ldc, double* Acopy, int64_t ldacopy) {
  DFH_ARG(gemm_skinny_applies(m, N, K, A, lda, B, ldb));
  DFH_ARG(beta == 0.0 || Cin != nullptr);
  SkinnyArgs p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb;
  p.Cin = beta == 0.0 ? nullptr : Cin; p.ldcin = ldcin; p.Cout = Cout; p.ldc = ldc;
  p.Acopy = Acopy; p.ldacopy = ldacopy;
  p.m = (int)m; p.N = (int)N; p.K = (int)K; p.alpha = alpha; p.beta = beta;
  const int tiles = (int)((m + 15) / 16);
  static bool attr_set[DFH_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device]) {
    const int big = (int)(sizeof(double) * 256 * SK_LDA), mid = (int)(sizeof(double) * 128 * SK_LDA);
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<16, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<16, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<8, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, mid));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_kernel<8, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, mid));
    attr_set[ctx->device] = true;
  }
  // few columns and several row tiles: split the rows over the waves instead of the columns
  const bool split = tiles > 2 && (N + 63) / 64 <= ctx->n_cu / 8;
  if (tiles <= 1) return launch_skinny<1, false>(ctx, p);
  if (tiles <= 2) return launch_skinny<2, false>(ctx, p);
  if (tiles <= 4) return split ? launch_skinny<4, true>(ctx, p) : launch_skinny<4, false>(ctx, p);
  if (tiles <= 8) return split ? launch_skinny<8, true>(ctx, p) : launch_skinny<8, false>(ctx, p);
  return split ? launch_skinny<16, true>(ctx, p) : launch_skinny<16, false>(ctx, p);
}

int gemm_f64(dfh_ctx* ctx, int flags, int64_t M, int64_t N, int64_t K, double alpha,
             const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
             const double* Cin, int64_t ldcin, double* Cout, int64_t ldc,
             const GemmBatch* batch) {
  if (M <= 0 || N <= 0) return DFH_OK;
  DFH_ARG(M < (1LL << 30) && N < (1LL << 30) && K < (1LL << 30) && K >= 0);
  DFH_ARG(beta == 0.0 || Cin != nullptr);
  if (flags & GEMM_LOWER) DFH_ARG(M == N);
  GemmArgs p;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb;
  p.Cin = (beta == 0.0) ? nullptr : Cin; p.ldcin = ldcin; p.Cout = Cout; p.ldc = ldc;
  p.sA = p.sB = p.sCin = p.sCout = 0;
  p.sA2 = p.sB2 = p.sCin2 = p.sCout2 = 0;
  p.cnt1 = 1;
  int count = 1;
  if (batch) {
    p.cnt1 = batch->count; count = batch->count * batch->count2;
    p.sA = batch->sA; p.sB = batch->sB; p.sCin = batch->sCin; p.sCout = batch->sCout;
    p.sA2 = batch->sA2; p.sB2 = batch->sB2; p.sCin2 = batch->sCin2; p.sCout2 = batch->sCout2;
  }
  p.alpha = alpha; p.beta = beta; p.flags = flags;
  GemmExt ext;
  ext.cond = ctx->gemm_cond; ext.cond_thr = ctx->gemm_cond_thr; ext.la_cnt = ctx->gemm_la_cnt; ext.la_pad = 0; ext.vgrid = 0;
  // diagnostics (tools/chol_gemm_prof.py): DFH_GEMM_FORCE_LA=1 sends every eligible LOWER product through the
  // look-ahead tile order (counters in scratch) so that the extended kernel can be timed on its own
  static const bool force_la = getenv("DFH_GEMM_FORCE_LA") && atoi(getenv("DFH_GEMM_FORCE_LA")) != 0;
  if (force_la && !ext.la_cnt && (flags & GEMM_LOWER) && !batch && M >= 5 * 128) {
    int* dummy = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_RED, 64, (void**)&dummy));
    ext.la_cnt = dummy;
  }
  const bool use_ext = ext.cond != nullptr || ext.la_cnt != nullptr;
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool edge = (K % BK) || (lda & 1) || (ldb & 1) || !aligned16(A) || !aligned16(B) ||
                    ((p.sA | p.sB | p.sA2 | p.sB2) & 1);
  // Small problems are latency-bound on a single 128x128 tile per CU: use 64x64 tiles when the
  // 128-tiling would leave most of the 256 CUs idle.
  const long t128 = ((M + 127) / 128) * ((N + 127) / 128) * (long)count;
  if (use_ext) return dispatch<4>(ctx, p, count, edge, &ext);
  if (t128 < 192) return dispatch<2>(ctx, p, count, edge);
  // Tuning knob (tools/gemm_rows.py): cut a tall product into launches of this many rows.  Measured
  // on the posterior shape 262144 x 512 x 8192: L2 misses fall from 1.9x to 1.5x the operand bytes
  // (two-wave launches keep the tiles that share a panel in step), the time does not move (31.55 vs
  // 31.61 ms, 69.7 TF/s) -- the kernel is bound by the fp64 pipe, not by HBM.  Off by default.
  static const long split_rows = getenv("DFH_GEMM_SPLIT_ROWS") ? atol(getenv("DFH_GEMM_SPLIT_ROWS")) : 0;
  if (split_rows >= 128 && !batch && !(flags & GEMM_LOWER) && M > split_rows) {
    for (int64_t r0 = 0; r0 < M; r0 += split_rows) {
      const int64_t mr = (M - r0 < split_rows) ? M - r0 : split_rows;
      p.M = (int)mr; p.A = A + r0 * lda; p.Cin = Cin ? Cin + r0 * ldcin : nullptr; p.Cout = Cout + r0 * ldc;
      DFH_TRY(dispatch<4>(ctx, p, count, edge));
    }
    return DFH_OK;
  }
  if (!edge && !(flags & GEMM_LOWER) && (N % 128) == 0 && (M % 128) != 0 && M >= 1024) {
    // A ragged last row tile would send EVERY tile through the bounds-checked kernel (scalar,
    // predicated operand loads): the full row tiles take the fast kernel, the < 128 leftover rows a
    // launch of their own.  A row's sum runs over the same chunks in the same order either way.
    const int64_t M1 = (M / 128) * 128;
    DFH_TRY(gemm_f64(ctx, flags, M1, N, K, alpha, A, lda, B, ldb, beta, Cin, ldcin, Cout, ldc, batch));
    return gemm_f64(ctx, flags, M - M1, N, K, alpha, A + M1 * lda, lda, B, ldb, beta,
                    Cin ? Cin + M1 * ldcin : nullptr, ldcin, Cout + M1 * ldc, ldc, batch);
  }
  return dispatch<4>(ctx, p, count, edge);
}

// Enable / disable per-launch event timing of the GEMM kernel and fetch the totals.
// stats_out[8][5]: per kernel variant (bit2 = NN, bit1 = edge path, bit0 = 64x64 tiles; variant 0
// is the 128x128 NT throughput configuration; 7 / 6 = the factorisation's look-ahead-order /
// conditional launches, kernels of their own) {launches, sum of launch durations in ms,
// algorithmic flop, busy ms, algorithmic bytes}.  `busy` is the length of the union of the variant's launch
// intervals: launches on different streams overlap (look-ahead Cholesky, TS pipeline) and then
// share the CUs, so the plain sum counts that wall-clock more than once.
extern "C" int dfh_ctx_gemm_profile(dfh_ctx* ctx, int enable, double* stats_out) {
  DFH_ARG(ctx != nullptr);
  if (stats_out) {
    for (int i = 0; i < 40; ++i) stats_out[i] = 0.0;
    DFH_HIP(hipStreamSynchronize(ctx->main_stream));
    DFH_HIP(hipStreamSynchronize(ctx->side));
    DFH_HIP(hipStreamSynchronize(ctx->bulk));
    DFH_HIP(hipStreamSynchronize(ctx->aux));
    std::vector<std::pair<float, float>> iv[8];
    for (size_t i = 0; i < ctx->gemm_used; ++i) {
      float ms = 0.f, t0 = 0.f;
      const dfh_ctx::GemmRec& r = ctx->gemm_recs[i];
      DFH_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
      if (ctx->gemm_base) DFH_HIP(hipEventElapsedTime(&t0, ctx->gemm_base, r.e0));
      const int v = r.variant;
      stats_out[v * 5 + 0] += 1.0;
      stats_out[v * 5 + 1] += (double)ms;
      stats_out[v * 5 + 2] += r.flops;
      stats_out[v * 5 + 4] += r.bytes;
      iv[v].emplace_back(t0, t0 + ms);
    }
    for (int v = 0; v < 8; ++v) {
      std::sort(iv[v].begin(), iv[v].end());
      double busy = 0.0; float hi = -1e30f;
      for (const auto& x : iv[v]) {
        if (x.first >= hi) { busy += x.second - x.first; hi = x.second; }
        else if (x.second > hi) { busy += x.second - hi; hi = x.second; }
      }
      stats_out[v * 5 + 3] = busy;
    }
  }
  ctx->gemm_used = 0;
  ctx->gemm_prof = enable != 0;
  if (ctx->gemm_prof) {
    if (!ctx->gemm_base) DFH_HIP(hipEventCreate(&ctx->gemm_base));
    DFH_HIP(hipStreamSynchronize(ctx->main_stream));
    DFH_HIP(hipStreamSynchronize(ctx->side));
    DFH_HIP(hipStreamSynchronize(ctx->bulk));
    DFH_HIP(hipStreamSynchronize(ctx->aux));
    DFH_HIP(hipEventRecord(ctx->gemm_base, ctx->main_stream));
  }
  return DFH_OK;
}
