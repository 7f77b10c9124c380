// Kernel-matrix construction: SE / Matern / additive Gram and cross matrices in fp64.
//
// Replaces, fused into one pass over the output,
//   get_scaled_repr                 dragonfly/gp/kernel.py:179-181, 255-257   (pack_scaled)
//   dist_squared (+ clip at 0)      dragonfly/utils/general_utils.py:58-70
//   SEKernel._child_evaluate        dragonfly/gp/kernel.py:171-177
//   MaternKernel._child_evaluate    dragonfly/gp/kernel.py:259-270, 292-299
//   AdditiveKernel._child_evaluate  dragonfly/gp/kernel.py:484-494
//   K + noise_var*np.eye(n)         dragonfly/gp/gp_core.py:843               (diag_add)
//
// The reference materialises three n1 x n2 temporaries plus the dgemm output; here each
// 128 x 128 (128 x 64 for additive) output tile is produced in registers: the -2 X1 X2^T term
// runs on the fp64 matrix cores (same expansion as the reference, so the same rounding
// behaviour), the norms / clip / exp / Matern polynomial run on the VALU beside it, and the
// only HBM traffic is the coalesced write of K (the pass is HBM-write bound:
// 8*(n1*n2 + (n1+n2)*d) algorithmic bytes).
#include "common.h"
#include <atomic>
#include <chrono>
#include <cstring>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int KM_BM = 128;
constexpr int KM_KC = 32;          // packed columns per LDS chunk
constexpr int KM_KP = 34;          // LDS row stride (doubles); 34 = 2 mod 32 -> conflict-free b64 frag reads

#include "kerneval.h"   // ExpConsts, exp_fast, kern_eval, combine_nested, np_sumsq, TinyCand

struct KmArgs {
  ExpConsts ec;
  const double* Xp1; const double* Np1;
  const double* Xp2; const double* Np2;
  int n1, n2, P, n_parts_total;
  const PartDev* parts;
  int part_lo, part_hi;
  double outer;
  int apply_outer, symmetric;
  int product;                 // MULTI: parts are multiplied (CoordinateProductKernel) instead of summed
  int nt_stores;               // kernmat_sym_kernel: write the matrix with streaming stores
  int lower_only;              // kernmat_sym_kernel: tiles of the lower triangle only, no mirror images (the fit path: the factorisation reads nothing else)
  double diag_add;
  double* K; long ldk;
  // lock-step batch over blockIdx.z (symmetric single-part kernel only): element strides of the
  // packed inputs / output, byte stride between the device images of the kernels, one diagonal
  // term per batch element (NULL: diag_add)
  long sXp, sNp, sK, sBlob;
  const double* diag_adds;
  // strip kernel with the posterior mean fused in: mu_part[row][blk] = sum over the columns of block
  // blk (KM_MU_BLOCK columns) of K[row][col] * mu_alpha[col]
  const double* mu_alpha;
  double* mu_part;
  double* mu_out;
  int mu_nblk;
};
constexpr int KM_MU_BLOCK = 512;
// NS: the parts may be polynomial / exponential-decay kernels (an instance of its own: their pow()
// calls cost the stationary multi-part kernel its registers).  NESTED (with NS): a product kernel
// with additive factors.
template <int TJ, bool MULTI, bool NS = false, bool NESTED = false>
__global__ __launch_bounds__(256, 2) void kernmat_kernel(KmArgs p) {
  constexpr int BN = 2 * TJ * 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                           // [128][KM_KP]
  double* Bs = As + KM_BM * KM_KP;             // [BN][KM_KP]
  double* na = Bs + BN * KM_KP;                // [128]
  double* nb = na + KM_BM;                     // [BN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long m0 = (long)blockIdx.y * KM_BM, n0 = (long)blockIdx.x * BN;

  double4_t res[4][TJ];
  double4_t fsum[NESTED ? 4 : 1][NESTED ? TJ : 1];
  if (MULTI) {
    // additive: 0 + k_1 + k_2 ...; product: scale * k_1 * k_2 ... in the reference's order
    // (kernel.py:584-588: K = scale * ones; K *= kernel(...))
    const double r0 = p.product ? p.outer : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) res[i][j] = (double4_t){r0, r0, r0, r0};
  }
  if (NESTED) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) fsum[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  }

  for (int part = p.part_lo; part < p.part_hi; ++part) {
    const PartDev& pd = p.parts[part];          // stays in global memory: uniform scalar loads
    double4_t acc[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    for (int k0 = 0; k0 < pd.kc; k0 += KM_KC) {
      const int kc = min(KM_KC, pd.kc - k0);     // multiple of 4
      const int kh = kc >> 1;                    // double2 per row
      __syncthreads();                           // previous readers of As/Bs/na/nb are done
      for (int idx = tid; idx < KM_BM * kh; idx += 256) {
        const int r = idx / kh, c2 = (idx - r * kh) * 2;
        const long row = m0 + r;
        double2_t v = (double2_t){0.0, 0.0};
        if (row < p.n1) v = *reinterpret_cast<const double2_t*>(p.Xp1 + row * p.P + pd.poff + k0 + c2);
        *reinterpret_cast<double2_t*>(As + r * KM_KP + c2) = v;
      }
      for (int idx = tid; idx < BN * kh; idx += 256) {
        const int r = idx / kh, c2 = (idx - r * kh) * 2;
        const long row = n0 + r;
        double2_t v = (double2_t){0.0, 0.0};
        if (row < p.n2) v = *reinterpret_cast<const double2_t*>(p.Xp2 + row * p.P + pd.poff + k0 + c2);
        *reinterpret_cast<double2_t*>(Bs + r * KM_KP + c2) = v;
      }
      if (k0 == 0) {
        if (tid < KM_BM) {
          const long row = m0 + tid;
          na[tid] = row < p.n1 ? p.Np1[row * p.n_parts_total + part] : 0.0;
        } else if (tid - KM_BM < BN) {
          const long row = n0 + tid - KM_BM;
          nb[tid - KM_BM] = row < p.n2 ? p.Np2[row * p.n_parts_total + part] : 0.0;
        }
      }
      __syncthreads();
      const double* as = As + (wm * 64 + l15) * KM_KP + l4;
      const double* bs = Bs + (wn * TJ * 16 + l15) * KM_KP + l4;
      for (int kk = 0; kk < kc; kk += 4) {
        double a[4], b[TJ];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = as[t * 16 * KM_KP + kk];
#pragma unroll
        for (int t = 0; t < TJ; ++t) b[t] = bs[t * 16 * KM_KP + kk];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }

    // distances -> kernel values for this part
    const ExpConsts& ec = p.ec;                  // kernel arguments: scalar loads, SGPR-resident
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = wm * 64 + i * 16 + l4 + 4 * r;
        const double nai = na[lr];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int lc = wn * TJ * 16 + j * 16 + l15;
          double kv;
          if (NS && pd.kind == DFH_KERNEL_POLY) {
            kv = poly_eval(pd, acc[i][j][r]);
          } else if (NS && pd.kind == DFH_KERNEL_EXPDECAY) {
            // the part's columns (kc <= KM_KC: one chunk) are still in the operand tiles
            kv = expdecay_eval(pd, As + lr * KM_KP, Bs + lc * KM_KP);
          } else {
            double dsq = (nb[lc] + nai) - 2.0 * acc[i][j][r];     // general_utils.py:66-68
            dsq = dsq < 0.0 ? 0.0 : dsq;                           // np.clip(.,0,inf), NaN kept
            kv = kern_eval(pd, dsq, ec);
          }
          if (NESTED) {
            double rr = res[i][j][r], ff = fsum[i][j][r];
            combine_nested(pd, kv, rr, ff);
            res[i][j][r] = rr; fsum[i][j][r] = ff;
          } else if (MULTI) {
            res[i][j][r] = p.product ? res[i][j][r] * kv : res[i][j][r] + kv;   // kernel.py:493 / :588
          } else {
            acc[i][j][r] = kv;
          }
        }
      }
    }
    if (!MULTI) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) res[i][j] = acc[i][j];
    }
  }

  // store
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = m0 + wm * 64 + i * 16 + l4 + 4 * r;
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const long col = n0 + wn * TJ * 16 + j * 16 + l15;
        if (row < p.n1 && col < p.n2) {
          double v = res[i][j][r];
          if (MULTI && p.apply_outer && !p.product) v = p.outer * v;   // kernel.py:494
          if (p.symmetric && row == col) v += p.diag_add;         // gp_core.py:843
          p.K[row * p.ldk + col] = v;
        }
      }
    }
  }
}

// Symmetric Gram matrix K(X, X) + diag_add I, single-part kernels: only the tiles on and below the
// diagonal are computed; each off-diagonal tile is written twice, as itself and transposed into its
// mirror position.  Both images go through an LDS staging buffer so that every global store is a
// full 16-byte-per-lane row segment (the natural MFMA accumulator layout only offers 8-byte stores
// in 128-byte segments, and none at all for the transposed image).
// TS = tile edge: 64 (2x2 MFMA tiles per wave, ~35 KB LDS, 4 workgroups per CU -- the phases
// load / MFMA / exp / store of different workgroups overlap) or 128.
template <int TS, int KC, int SR, int OCC, bool SYM>
__global__ __launch_bounds__(256, OCC) void kernmat_sym_kernel(KmArgs p) {
  constexpr int WT = TS / 32;            // MFMA tiles per wave per dimension
  constexpr int WS = TS / 2;             // wave tile edge
  constexpr int SP = TS + 2;             // staging row stride (doubles): 16-byte aligned rows
  constexpr int NH = TS / SR;            // SR-row staging passes per image
  constexpr int KP = KC + 2;             // operand row stride: = 2 (mod 32) for KC = 32, 18 for KC = 16
  constexpr int OPER = 2 * TS * KP;      // doubles of the two operand tiles
  constexpr int STAGE = SR * SP;
  constexpr int BODY = (OPER > STAGE) ? OPER : STAGE;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                     // [TS][KP]
  double* Bs = As + TS * KP;             // [TS][KP]
  double* na = smem + BODY;              // [TS]
  double* nb = na + TS;                  // [TS]
  double* St = smem;                     // [SR][SP] staging, reuses the operand tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  unsigned ti, tj;
  if (SYM) {                              // lower-triangular tile enumeration
    const unsigned lin = blockIdx.x;
    ti = (unsigned)((sqrt(8.0 * (double)lin + 1.0) - 1.0) * 0.5);
    while ((unsigned long long)ti * (ti + 1) / 2 > lin) --ti;
    while ((unsigned long long)(ti + 1) * (ti + 2) / 2 <= lin) ++ti;
    tj = lin - (unsigned)((unsigned long long)ti * (ti + 1) / 2);
  } else {                                // cross matrix: plain 2-D grid
    ti = blockIdx.y; tj = blockIdx.x;
  }
  const long m0 = (long)ti * TS, n0 = (long)tj * TS;
  const long bz = SYM ? (long)blockIdx.z : 0;       // batch element (strides are 0 for a single matrix)
  const PartDev& pd = reinterpret_cast<const PartDev*>(reinterpret_cast<const char*>(p.parts) + bz * p.sBlob)[p.part_lo];
  const double* __restrict__ XpA = p.Xp1 + bz * p.sXp;
  const double* __restrict__ NpA = p.Np1 + bz * p.sNp;
  double* __restrict__ Kout = p.K + bz * p.sK;
  const double diag_add = p.diag_adds ? p.diag_adds[bz] : p.diag_add;
  const double* __restrict__ XpB = SYM ? XpA : p.Xp2;
  const double* __restrict__ NpB = SYM ? NpA : p.Np2;
  const long nB = SYM ? p.n1 : p.n2;

  double4_t acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < pd.kc; k0 += KC) {
    const int kc = min(KC, pd.kc - k0);
    const int kh = kc >> 1;
    __syncthreads();
    for (int idx = tid; idx < TS * kh; idx += 256) {
      const int r = idx / kh, c2 = (idx - r * kh) * 2;
      const long rowa = m0 + r, rowb = n0 + r;
      double2_t va = (double2_t){0.0, 0.0}, vb = (double2_t){0.0, 0.0};
      if (rowa < p.n1) va = *reinterpret_cast<const double2_t*>(XpA + rowa * p.P + pd.poff + k0 + c2);
      if (rowb < nB) vb = *reinterpret_cast<const double2_t*>(XpB + rowb * p.P + pd.poff + k0 + c2);
      *reinterpret_cast<double2_t*>(As + r * KP + c2) = va;
      *reinterpret_cast<double2_t*>(Bs + r * KP + c2) = vb;
    }
    if (k0 == 0) {
      if (tid < TS) {
        const long row = m0 + tid;
        na[tid] = row < p.n1 ? NpA[row * p.n_parts_total + p.part_lo] : 0.0;
      } else if (tid - TS < TS) {
        const long row = n0 + tid - TS;
        nb[tid - TS] = row < nB ? NpB[row * p.n_parts_total + p.part_lo] : 0.0;
      }
    }
    __syncthreads();
    const double* as = As + (wm * WS + l15) * KP + l4;
    const double* bs = Bs + (wn * WS + l15) * KP + l4;
    for (int kk = 0; kk < kc; kk += 4) {
      double a[WT], b[WT];
#pragma unroll
      for (int t = 0; t < WT; ++t) a[t] = as[t * 16 * KP + kk];
#pragma unroll
      for (int t = 0; t < WT; ++t) b[t] = bs[t * 16 * KP + kk];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // distances -> kernel values (in the accumulator registers).  SE: -dsq/2 is formed directly as
  // acc - (na/2 + nb/2): scaling by powers of two commutes with rounding, so this is bit-identical
  // to ((nb + na) - 2 acc) clipped at 0 and then halved and negated (general_utils.py:66-69,
  // kernel.py:176).  The diagonal term only exists in diagonal tiles.
  const bool se = (pd.kind == DFH_KERNEL_SE);
  const bool diag_tile = SYM && (ti == tj);
  const ExpConsts& ec = p.ec;                    // kernel arguments: scalar loads, SGPR-resident
#pragma unroll
  for (int i = 0; i < WT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = wm * WS + i * 16 + l4 + 4 * r;
      const double nai = na[lr];
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const int lc = wn * WS + j * 16 + l15;
        double kv;
        if (se) {
          double t = acc[i][j][r] - (0.5 * nb[lc] + 0.5 * nai);
          t = t > 0.0 ? 0.0 : t;
          kv = pd.scale_c * exp_fast_neg(t, ec);       // t <= 0: no exponent clamp needed (two VALU ops of ~26)
        } else {
          double dsq = (nb[lc] + nai) - 2.0 * acc[i][j][r];
          dsq = dsq < 0.0 ? 0.0 : dsq;
          kv = kern_eval(pd, dsq, ec);
        }
        if (diag_tile && lr == lc) kv += diag_add;
        acc[i][j][r] = kv;
      }
    }
  }

  // staged stores: passes [0, NH) = the tile itself, SR rows at a time; passes [NH, 2 NH) = the
  // mirror image (rows = original columns)
  const int npass = (!SYM || ti == tj || p.lower_only) ? NH : 2 * NH;
  for (int pass = 0; pass < npass; ++pass) {
    const bool mirror = pass >= NH;
    const int h = mirror ? pass - NH : pass;
    __syncthreads();                                   // staging buffer free (and operands dead)
    // image row of an accumulator element: direct -> wm*WS + i*16 + l4 + 4r ; mirror -> wn*WS + j*16 + l15
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          const int irow = mirror ? (wn * WS + j * 16 + l15) : (wm * WS + i * 16 + l4 + 4 * r);
          const int icol = mirror ? (wm * WS + i * 16 + l4 + 4 * r) : (wn * WS + j * 16 + l15);
          if (irow / SR == h) St[(irow - h * SR) * SP + icol] = acc[i][j][r];
        }
    __syncthreads();
    const long row_base = (mirror ? n0 : m0) + h * SR;
    const long col_base = mirror ? m0 : n0;
    constexpr int RP = TS / 2;                         // double2 per staged row
#pragma unroll
    for (int q = 0; q < (SR * RP) / 256; ++q) {
      const int idx = tid + 256 * q;
      const int r = idx / RP, c2 = (idx % RP) * 2;
      const long row = row_base + r, col = col_base + c2;
      const long nrow = mirror ? nB : p.n1, ncol = mirror ? p.n1 : nB;
      if (row < nrow && col + 1 < ncol) {
        // streaming (non-temporal) stores for the wide kernels: the matrix is written once and is far larger than
        // L2 + MALL; measured 16384^2: d = 32 SE 0.444 -> 0.428 ms, Matern 0.554 -> 0.53, but d = 6 Matern 0.402 ->
        // 0.418 (tools/r4_run15.sh) -- hence only from a packed width of 16 on (KmArgs::nt_stores)
        if (p.nt_stores)
          __builtin_nontemporal_store(*reinterpret_cast<const double2_t*>(St + r * SP + c2),
                                      reinterpret_cast<double2_t*>(Kout + row * p.ldk + col));
        else
          *reinterpret_cast<double2_t*>(Kout + row * p.ldk + col) =
              *reinterpret_cast<const double2_t*>(St + r * SP + c2);
      } else if (row < nrow && col < ncol) {
        Kout[row * p.ldk + col] = St[r * SP + c2];
        if (col + 1 < ncol) Kout[row * p.ldk + col + 1] = St[r * SP + c2 + 1];
      }
    }
  }
}

// Symmetric Gram matrix of a multi-part kernel with stationary parts (additive: scale * sum_g k_g,
// kernel.py:484-494; coordinate product of SE / Matern factors: kernel.py:578-589): the lower
// triangle of 64 x 64 tiles only, each tile stored twice through the LDS staging buffer as in
// kernmat_sym_kernel.  The parts' packed columns are adjacent, so one LDS fill takes as many whole
// parts as fit into KC columns (the groups of an additive model are a few columns wide: two barriers
// per KC columns instead of two per part), then each part runs its own MFMA dot product, its
// epilogue, and is combined into the running result in the reference's order.
// Half the tiles of the generic kernel, a quarter of its LDS, 20 KB per workgroup.
template <int KC, int SR, int OCC>
__global__ __launch_bounds__(256, OCC) void kernmat_symmulti_kernel(KmArgs p) {
  constexpr int TS = 64, WT = 2, WS = 32;
  constexpr int SP = TS + 2;
  constexpr int NH = TS / SR;
  constexpr int KP = KC + 2;
  constexpr int OPER = 2 * TS * KP;
  constexpr int STAGE = SR * SP;
  constexpr int BODY = (OPER > STAGE) ? OPER : STAGE;
  constexpr int MAXP = KC / 4;           // parts per fill (a part is at least 4 packed columns)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                     // [TS][KP]
  double* Bs = As + TS * KP;             // [TS][KP]
  double* na = smem + BODY;              // [MAXP][TS]
  double* nb = na + MAXP * TS;           // [MAXP][TS]
  double* St = smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const unsigned lin = blockIdx.x;
  unsigned ti = (unsigned)((sqrt(8.0 * (double)lin + 1.0) - 1.0) * 0.5);
  while ((unsigned long long)ti * (ti + 1) / 2 > lin) --ti;
  while ((unsigned long long)(ti + 1) * (ti + 2) / 2 <= lin) ++ti;
  const unsigned tj = lin - (unsigned)((unsigned long long)ti * (ti + 1) / 2);
  const long m0 = (long)ti * TS, n0 = (long)tj * TS;
  const ExpConsts& ec = p.ec;

  double4_t res[WT][WT];
  {
    const double r0 = p.product ? p.outer : 0.0;
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j) res[i][j] = (double4_t){r0, r0, r0, r0};
  }

  int part = p.part_lo;
  while (part < p.part_hi) {
    int pe = part, cols = 0;
    while (pe < p.part_hi && cols + p.parts[pe].kc <= KC) { cols += p.parts[pe].kc; ++pe; }
    const int c0 = p.parts[part].poff;
    const int ch = cols >> 1;
    __syncthreads();
    for (int idx = tid; idx < TS * ch; idx += 256) {
      const int r = idx / ch, c2 = (idx - r * ch) * 2;
      const long rowa = m0 + r, rowb = n0 + r;
      double2_t va = (double2_t){0.0, 0.0}, vb = (double2_t){0.0, 0.0};
      if (rowa < p.n1) va = *reinterpret_cast<const double2_t*>(p.Xp1 + rowa * p.P + c0 + c2);
      if (rowb < p.n1) vb = *reinterpret_cast<const double2_t*>(p.Xp1 + rowb * p.P + c0 + c2);
      *reinterpret_cast<double2_t*>(As + r * KP + c2) = va;
      *reinterpret_cast<double2_t*>(Bs + r * KP + c2) = vb;
    }
    for (int idx = tid; idx < (pe - part) * TS; idx += 256) {
      const int q = idx / TS, r = idx - q * TS;
      const long rowa = m0 + r, rowb = n0 + r;
      na[idx] = rowa < p.n1 ? p.Np1[rowa * p.n_parts_total + part + q] : 0.0;
      nb[idx] = rowb < p.n1 ? p.Np1[rowb * p.n_parts_total + part + q] : 0.0;
    }
    __syncthreads();
    for (int q = part; q < pe; ++q) {
      const PartDev& pd = p.parts[q];
      const int off = pd.poff - c0;
      double4_t acc[WT][WT];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
      const double* as = As + (wm * WS + l15) * KP + off + l4;
      const double* bs = Bs + (wn * WS + l15) * KP + off + l4;
      for (int kk = 0; kk < pd.kc; kk += 4) {
        double a[WT], b[WT];
#pragma unroll
        for (int t = 0; t < WT; ++t) a[t] = as[t * 16 * KP + kk];
#pragma unroll
        for (int t = 0; t < WT; ++t) b[t] = bs[t * 16 * KP + kk];
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      const bool se = (pd.kind == DFH_KERNEL_SE);
      const double* naq = na + (q - part) * TS;
      const double* nbq = nb + (q - part) * TS;
#pragma unroll
      for (int i = 0; i < WT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double nai = naq[wm * WS + i * 16 + l4 + 4 * r];
#pragma unroll
          for (int j = 0; j < WT; ++j) {
            const double nbj = nbq[wn * WS + j * 16 + l15];
            double kv;
            if (se) {                    // -dsq/2 = acc - (na/2 + nb/2), see kernmat_sym_kernel
              double t = acc[i][j][r] - (0.5 * nbj + 0.5 * nai);
              t = t > 0.0 ? 0.0 : t;
              kv = pd.scale_c * exp_fast(t, ec);
            } else {
              double dsq = (nbj + nai) - 2.0 * acc[i][j][r];
              dsq = dsq < 0.0 ? 0.0 : dsq;
              kv = kern_eval(pd, dsq, ec);
            }
            res[i][j][r] = p.product ? res[i][j][r] * kv : res[i][j][r] + kv;   // kernel.py:588 / :493
          }
        }
      }
    }
    part = pe;
  }

  const bool diag_tile = (ti == tj);
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        double v = res[i][j][r];
        if (p.apply_outer && !p.product) v = p.outer * v;                       // kernel.py:494
        if (diag_tile && (wm * WS + i * 16 + l4 + 4 * r) == (wn * WS + j * 16 + l15)) v += p.diag_add;
        res[i][j][r] = v;
      }

  const int npass = diag_tile ? NH : 2 * NH;
  for (int pass = 0; pass < npass; ++pass) {
    const bool mirror = pass >= NH;
    const int h = mirror ? pass - NH : pass;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          const int irow = mirror ? (wn * WS + j * 16 + l15) : (wm * WS + i * 16 + l4 + 4 * r);
          const int icol = mirror ? (wm * WS + i * 16 + l4 + 4 * r) : (wn * WS + j * 16 + l15);
          if (irow / SR == h) St[(irow - h * SR) * SP + icol] = res[i][j][r];
        }
    __syncthreads();
    const long row_base = (mirror ? n0 : m0) + h * SR;
    const long col_base = mirror ? m0 : n0;
    constexpr int RP = TS / 2;
#pragma unroll
    for (int q = 0; q < (SR * RP) / 256; ++q) {
      const int idx = tid + 256 * q;
      const int r = idx / RP, c2 = (idx % RP) * 2;
      const long row = row_base + r, col = col_base + c2;
      if (row < p.n1 && col + 1 < p.n1) {
        *reinterpret_cast<double2_t*>(p.K + row * p.ldk + col) = *reinterpret_cast<const double2_t*>(St + r * SP + c2);
      } else if (row < p.n1 && col < p.n1) {
        p.K[row * p.ldk + col] = St[r * SP + c2];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Cross matrix K(X1, X2), single-part SE / Matern kernels, packed width 8..32: "strip" kernel.
//
// What was measured on gfx950 (tools/km_bench.hip, 32768 x 16384, d = 32): the fp64 MFMA work of
// the distance expansion alone takes 0.56 ms, the fp64 VALU epilogue (clip, exp) alone 0.43 ms,
// both together 0.89 ms -- fp64 matrix and fp64 vector instructions share the SIMD's fp64 pipe on
// this part, they do not overlap -- and the 4.3 GB of output 0.72 ms.  The pass is bound by that
// pipe, so the kernel is organised to keep it fed: no LDS, no barriers, a wave keeps the operand
// fragments of its 32 rows in registers and walks along the columns in tiles of 64, loading the
// next tile's column fragments a whole tile ahead (register double buffer) while the current tile
// runs its 8 * C MFMAs and its epilogue; stores are fire-and-forget; two such waves per SIMD.
// Operand fragments come straight from L2: lane (l15, l4) of an MFMA holds, for row l15 of a
// 16-row tile, the packed columns [l4 * C, (l4 + 1) * C) -- which k of the dot product sits in
// which MFMA slot is free as long as both operands agree -- i.e. contiguous 16-byte loads.
// Same expansion as the reference ((|a|^2 + |b|^2) - 2 a.b, clipped at 0; general_utils.py:66-69),
// only the summation order inside a.b differs from the LDS kernel's.
// The 64 x 64-tile LDS kernel (kernmat_sym_kernel<..., false>) took 1.45 ms on this shape and
// 0.84 ms (Matern-2.5) for 65536 x 4096 at d = 6, this one 1.1-1.2 ms and 0.5 ms.
// ---------------------------------------------------------------------------------------------
// MU: the product of the strip with a vector (the posterior mean K(X*, X) alpha, gp_core.py:174) rides
// along: every lane accumulates K[row][col] * alpha[col] over the columns it owns, tile after tile;
// at the end of every block of KM_MU_BLOCK columns the 16 lanes that share a row add up (fixed
// butterfly) and the row's partial sum of that block is written out.  A second, tiny kernel adds
// the blocks in order.  Blocks are cut by column index alone and segments consist of whole blocks,
// so a row's mean does not depend on how many rows the call has or where they start (chunks,
// shards, Thompson blocks all give the same bits) -- and the 8 n m bytes of the cross matrix are
// not read again for it.
template <int KIND, int C, int MP, bool MU = false>
__global__ __launch_bounds__(256, 2) void kernmat_strip_kernel(KmArgs p, int tiles_per_seg) {
  // 32 rows x 64 (wide packed inputs: 32) columns per wave and tile, at least 2 waves per SIMD (measured
  // 1.40 -> 1.12 ms against 64 x 32 at one wave per SIMD: a lone wave has nothing to cover its own stalls)
  // (with the mean riding along, the 64-column tile of the narrow packings would spill registers)
  constexpr int WI = 2, WJ = ((C >= 6 || MU) ? 2 : 4);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const long m0 = ((long)blockIdx.y * 4 + wave) * (16 * WI);
  if (m0 >= p.n1) return;
  const long ntile = ((long)p.n2 + 16 * WJ - 1) / (16 * WJ);
  const long t0 = (long)blockIdx.x * tiles_per_seg;
  const long t1 = t0 + tiles_per_seg < ntile ? t0 + tiles_per_seg : ntile;
  if (t0 >= t1) return;
  const PartDev& pd = p.parts[p.part_lo];
  const ExpConsts& ec = p.ec;              // SE: scale_c already folded into the coefficients
  const int npt = p.n_parts_total, part = p.part_lo;
  const double* __restrict__ A = p.Xp1 + pd.poff + l4 * C;
  const double* __restrict__ B = p.Xp2 + pd.poff + l4 * C;
  double a[WI][C];
  double nah[WI][4];                       // SE: |a|^2 / 2 ; Matern: |a|^2
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    long row = m0 + i * 16 + l15;
    row = row < p.n1 ? row : p.n1 - 1;
    const double* src = A + row * p.P;
#pragma unroll
    for (int c = 0; c < C; c += 2) {
      const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
      a[i][c] = v.x; a[i][c + 1] = v.y;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      long rr = m0 + i * 16 + l4 + 4 * r;
      rr = rr < p.n1 ? rr : p.n1 - 1;
      const double v = p.Np1[rr * npt + part];
      nah[i][r] = KIND == DFH_KERNEL_SE ? 0.5 * v : v;
    }
  }
  // Matern constants (uniform)
  constexpr int mp = MP;                   // Matern: int(nu), compile time
  const double s8 = pd.s8, s2 = pd.s2, gsc = pd.scale_c * pd.gfac;
  const double c0 = pd.coeff[0], c1 = pd.coeff[1], c2 = pd.coeff[2], c3 = pd.coeff[3];
  double b[WJ][C], nbh[WJ];
  double alh[WJ], mu_acc[WI][4];
  constexpr int TPB = KM_MU_BLOCK / (16 * WJ);        // tiles per mean block
  if (MU) {
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) mu_acc[i][r] = 0.0;
  }
  auto load_b = [&](long t, double (&bb)[WJ][C], double (&nn)[WJ]) {
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      // MFMA tile j, lane column l15 <-> matrix column n0 + WJ l15 + j: a lane then owns WJ ADJACENT
      // columns of every row it holds and stores them with 16-byte instructions
      long col = t * (16 * WJ) + WJ * l15 + j;
      col = col < p.n2 ? col : p.n2 - 1;
      const double* src = B + col * p.P;
#pragma unroll
      for (int c = 0; c < C; c += 2) {
        const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
        bb[j][c] = v.x; bb[j][c + 1] = v.y;
      }
      const double v = p.Np2[col * npt + part];
      nn[j] = KIND == DFH_KERNEL_SE ? 0.5 * v : v;
    }
  };
  // per-lane element offset inside a 4-row group: the store address is a wave-uniform row-group base
  // plus this
  const unsigned voff = (unsigned)(l4 * p.ldk + WJ * l15);
  double* __restrict__ Kstrip = p.K + m0 * p.ldk;
  const bool rows_full = m0 + 16 * WI <= p.n1;
  load_b(t0, b, nbh);
  for (long t = t0; t < t1; ++t) {
    double bn[WJ][C], nbn[WJ];
    load_b(t + 1 < t1 ? t + 1 : t, bn, nbn);
    if (MU) {               // this tile's alpha (L2-resident): issued here, used after the MFMAs
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const long col = t * (16 * WJ) + WJ * l15 + j;
        alh[j] = col < p.n2 ? p.mu_alpha[col] : 0.0;
      }
    }
    double4_t acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
      for (int j = 0; j < WJ; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][c], b[j][c], acc[i][j], 0, 0, 0);
    const long n0 = t * (16 * WJ);
    double* __restrict__ Kt = Kstrip + n0;
    auto epilogue = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int i = 0; i < WI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double kvs[WJ];
#pragma unroll
          for (int j = 0; j < WJ; ++j) {
            double kv;
            if (KIND == DFH_KERNEL_SE) {
              // -dsq/2 directly: acc - (|b|^2/2 + |a|^2/2), clipped at 0 (scaling by 2 commutes with
              // rounding: the same number as ((nb + na) - 2 acc) clipped, halved and negated)
              double tt = acc[i][j][r] - (nbh[j] + nah[i][r]);
              tt = tt > 0.0 ? 0.0 : tt;
              kv = exp_fast_neg(tt, ec);                               // kernel.py:176, scale inside ec
            } else {
              double dsq = (nbh[j] + nah[i][r]) - 2.0 * acc[i][j][r];   // general_utils.py:66-68
              dsq = dsq < 0.0 ? 0.0 : dsq;
              const double dist = sqrt_fast(dsq);                       // kernel.py:296
              const double mult = s8 * dist;                            // kernel.py:265
              double u;                                                 // sum_i coeff_i mult^(p-i), kernel.py:266
              if (mp == 0) u = c0;
              else if (mp == 1) u = fma(c0, mult, c1);
              else if (mp == 2) u = fma(fma(c0, mult, c1), mult, c2);
              else u = fma(fma(fma(c0, mult, c1), mult, c2), mult, c3);
              kv = u * (gsc * exp_fast_neg(-s2 * dist, ec));             // kernel.py:268-269, 298
            }
            kvs[j] = kv;
            if (MU) mu_acc[i][r] = fma(kv, alh[j], mu_acc[i][r]);
          }
          double* __restrict__ rowp = Kt + (long)(i * 16 + 4 * r) * p.ldk;      // wave-uniform
          if (FULL) {
#pragma unroll
            for (int j = 0; j < WJ; j += 2)
              *reinterpret_cast<double2_t*>(rowp + voff + j) = (double2_t){kvs[j], kvs[j + 1]};
          } else {
            const long row = m0 + i * 16 + l4 + 4 * r, col = n0 + WJ * l15;
#pragma unroll
            for (int j = 0; j < WJ; ++j)
              if (row < p.n1 && col + j < p.n2) rowp[voff + j] = kvs[j];
          }
        }
      }
    };
    if (rows_full && n0 + 16 * WJ <= p.n2) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (MU && ((t + 1) % TPB == 0 || t + 1 == t1)) {          // a mean block is complete
      const long blk = t / TPB;
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = mu_acc[i][r];
          v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
          const long row = m0 + i * 16 + l4 + 4 * r;
          if (l15 == 0 && row < p.n1) p.mu_part[row * p.mu_nblk + blk] = v;
          mu_acc[i][r] = 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      nbh[j] = nbn[j];
#pragma unroll
      for (int c = 0; c < C; ++c) b[j][c] = bn[j][c];
    }
  }
}

// mu[row] = the mean blocks of the row added in order
__global__ void k_mu_finish(const double* __restrict__ part, long n, int nblk, double* __restrict__ mu) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += part[row * nblk + b];
  mu[row] = s;
}

template <int KIND, int MP>
int launch_strip(dfh_ctx* ctx, KmArgs a, int C) {
  // enough waves for 256 CUs x 4 SIMDs x 2: split the columns of a 64-row strip into segments
  const long strips = ((long)a.n1 + 31) / 32;
  const int tile_cols = (C >= 6 || a.mu_part) ? 32 : 64;
  const long ntile = ((long)a.n2 + tile_cols - 1) / tile_cols;
  static const long want_waves = []() { const char* e = getenv("DFH_KM_WAVES"); long v = e ? atol(e) : 8192; return v > 0 ? v : 8192; }();
  long segs = (want_waves + strips - 1) / strips;
  if (segs > ntile) segs = ntile;
  if (segs < 1) segs = 1;
  int tps = (int)((ntile + segs - 1) / segs);
  if (a.mu_part) {                                   // segments of whole mean blocks
    const int tpb = KM_MU_BLOCK / tile_cols;
    tps = (tps + tpb - 1) / tpb * tpb;
  }
  segs = (ntile + tps - 1) / tps;
  dim3 grid((unsigned)segs, (unsigned)((strips + 3) / 4));
  if (a.mu_part) {
    switch (C) {
      case 2: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 2, MP, true>), grid, dim3(256), 0, ctx->stream, a, tps); break;
      case 4: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 4, MP, true>), grid, dim3(256), 0, ctx->stream, a, tps); break;
      case 6: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 6, MP, true>), grid, dim3(256), 0, ctx->stream, a, tps); break;
      default: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 8, MP, true>), grid, dim3(256), 0, ctx->stream, a, tps); break;
    }
    DFH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_mu_finish, dim3((unsigned)((a.n1 + 255) / 256)), dim3(256), 0, ctx->stream, a.mu_part, (long)a.n1,
                       a.mu_nblk, a.mu_out);
    DFH_LAUNCH_CHECK();
    return DFH_OK;
  }
  switch (C) {
    case 2: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 2, MP>), grid, dim3(256), 0, ctx->stream, a, tps); break;
    case 4: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 4, MP>), grid, dim3(256), 0, ctx->stream, a, tps); break;
    case 6: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 6, MP>), grid, dim3(256), 0, ctx->stream, a, tps); break;
    default: hipLaunchKernelGGL((kernmat_strip_kernel<KIND, 8, MP>), grid, dim3(256), 0, ctx->stream, a, tps); break;
  }
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// ---- packing -----------------------------------------------------------------------------
__global__ void k_pack_cols(const double* __restrict__ X, long n, long ldx, int P, int c_lo, int c_hi,
                            const int* __restrict__ cols, const double* __restrict__ bw,
                            double* __restrict__ Xp, long sBlob, long sXp) {
  // batch element blockIdx.y: its kernel image sits sBlob bytes further, its output sXp doubles
  cols = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cols) + (long)blockIdx.y * sBlob);
  bw = reinterpret_cast<const double*>(reinterpret_cast<const char*>(bw) + (long)blockIdx.y * sBlob);
  Xp += (long)blockIdx.y * sXp;
  const int w = c_hi - c_lo;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = n * w;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; idx < total; idx += stride) {
    const long row = idx / w;
    const int pc = c_lo + (int)(idx - row * w);
    const int c = cols[pc];
    // kernel.py:181 (X / bandwidths); a negative entry is a polynomial kernel's scaling: X * s (kernel.py:383)
    const double b = bw[pc];
    Xp[row * P + pc] = c >= 0 ? (b < 0.0 ? X[row * ldx + c] * -b : X[row * ldx + c] / b) : 0.0;
  }
}


__global__ void k_pack_norms(const double* __restrict__ Xp, long n, int P, int n_parts_total,
                             const PartDev* __restrict__ parts, const int* __restrict__ cols,
                             int part_lo, int part_hi, double* __restrict__ Np, long sBlob, long sXp,
                             long sNp) {
  parts = reinterpret_cast<const PartDev*>(reinterpret_cast<const char*>(parts) + (long)blockIdx.y * sBlob);
  cols = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cols) + (long)blockIdx.y * sBlob);
  Xp += (long)blockIdx.y * sXp;
  Np += (long)blockIdx.y * sNp;
  const int np = part_hi - part_lo;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = n * np;
  if (idx >= total) return;
  const long row = idx / np;
  const int part = part_lo + (int)(idx - row * np);
  const PartDev pd = parts[part];
  int nreal = 0;
  for (int c = 0; c < pd.kc; ++c) nreal += cols[pd.poff + c] >= 0;   // padding is trailing
  Np[row * n_parts_total + part] = np_sumsq(Xp + row * P + pd.poff, nreal);
}

// Both passes in one launch (round 3: the Gram-matrix section of a fit is packing + norms + the Gram
// kernel, and at n = 16384 the two packing launches with the gaps around them were 3 - 4 % of it): a
// workgroup scales R rows, coalesced as k_pack_cols does, keeps the packed values in LDS and takes the
// norms from there -- the same operations in the same order, so Xp / Np are bit for bit what the two
// kernels above produce.
__global__ __launch_bounds__(256) void k_pack_fused(const double* __restrict__ X, long n, long ldx, int P, int c_lo, int c_hi,
                                                    const int* __restrict__ cols, const int* __restrict__ cols_all,
                                                    const double* __restrict__ bw, const PartDev* __restrict__ parts,
                                                    int part_lo, int part_hi, int n_parts_total, int R,
                                                    double* __restrict__ Xp, double* __restrict__ Np, long sBlob,
                                                    long sXp, long sNp) {
  extern __shared__ double pk[];               // [R][w]
  cols = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cols) + (long)blockIdx.y * sBlob);
  cols_all = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cols_all) + (long)blockIdx.y * sBlob);
  bw = reinterpret_cast<const double*>(reinterpret_cast<const char*>(bw) + (long)blockIdx.y * sBlob);
  parts = reinterpret_cast<const PartDev*>(reinterpret_cast<const char*>(parts) + (long)blockIdx.y * sBlob);
  Xp += (long)blockIdx.y * sXp;
  Np += (long)blockIdx.y * sNp;
  const int w = c_hi - c_lo;
  const long r0 = (long)blockIdx.x * R;
  const int rows = (int)((n - r0 < R) ? n - r0 : R);
  for (int idx = threadIdx.x; idx < rows * w; idx += blockDim.x) {
    const int lr = idx / w, pc = c_lo + (idx - lr * w);
    const long row = r0 + lr;
    const int c = cols[pc];
    const double b = bw[pc];
    const double v = c >= 0 ? (b < 0.0 ? X[row * ldx + c] * -b : X[row * ldx + c] / b) : 0.0;    // as k_pack_cols
    Xp[row * P + pc] = v;
    pk[idx] = v;
  }
  __syncthreads();
  const int np = part_hi - part_lo;
  for (int idx = threadIdx.x; idx < rows * np; idx += blockDim.x) {
    const int lr = idx / np, part = part_lo + (idx - lr * np);
    const PartDev pd = parts[part];
    int nreal = 0;
    for (int c = 0; c < pd.kc; ++c) nreal += cols_all[pd.poff + c] >= 0;   // padding is trailing
    Np[(r0 + lr) * n_parts_total + part] = np_sumsq(pk + lr * w + (pd.poff - c_lo), nreal);          // as k_pack_norms
  }
}

double factorial_d(int n) {
  double r = 1.0;
  for (int i = 2; i <= n; ++i) r *= (double)i;
  return r;
}

double part_value_at_zero(const PartDev& pd);

int fill_part(PartDev& pd, int kind, double scale, double nu) {
  pd.kind = kind;
  pd.p = 0; pd.s8 = pd.s2 = pd.gfac = 0.0; pd.k0 = 0.0;
  for (int i = 0; i < 8; ++i) pd.coeff[i] = 0.0;
  if (kind == DFH_KERNEL_SE || kind == DFH_KERNEL_DIST) {
    pd.scale_c = scale;
    pd.k0 = part_value_at_zero(pd);
    return DFH_OK;
  }
  if (kind == DFH_KERNEL_POLY) {               // nu carries the order
    if (!(nu >= 0.0 && nu <= 64.0 && nu == floor(nu))) {
      dfh_set_error("polynomial kernel: the order has to be an integer in [0, 64] (got %g)", nu);
      return DFH_ERR_BAD_ARG;
    }
    pd.p = (int)nu; pd.scale_c = scale;
    return DFH_OK;
  }
  if (kind == DFH_KERNEL_EXPDECAY) {           // nu carries the offset; powers are set by the caller
    pd.scale_c = scale; pd.gfac = nu;
    return DFH_OK;
  }
  // Matern: kernel.py:242-253, 259-270
  double frac = fmod(nu, 1.0);
  if (!(frac == 0.5) || nu < 0.5) {
    dfh_set_error("Matern kernel: nu has to be p + 0.5 where p is an integer (got %g)", nu);
    return DFH_ERR_BAD_ARG;
  }
  const int p = (int)nu;
  if (p > 7) {
    dfh_set_error("Matern kernel: nu = %g not supported (p <= 7)", nu);
    return DFH_ERR_BAD_ARG;
  }
  pd.p = p;
  for (int i = 0; i <= p; ++i)
    pd.coeff[i] = factorial_d(p + i) / (factorial_d(i) * factorial_d(p - i));
  pd.s8 = sqrt(8.0 * nu);
  pd.s2 = sqrt(2.0 * nu);
  pd.gfac = tgamma((double)p + 1.0) / tgamma(2.0 * p + 1.0);
  // norm_constant = 1 / _eval_kernel_values_unnormalised(0)   (kernel.py:253)
  double u0 = 0.0;
  const double mult0 = pd.s8 * 0.0;
  for (int i = 0; i <= p; ++i) u0 += pd.coeff[i] * pow(mult0, (double)(p - i));
  u0 *= (pd.gfac * exp(-pd.s2 * 0.0));
  const double norm_constant = 1.0 / u0;
  pd.scale_c = scale * norm_constant;
  pd.k0 = part_value_at_zero(pd);
  return DFH_OK;
}

double part_value_at_zero(const PartDev& pd) {
  // k_part(x, x): distance 0
  if (pd.kind == DFH_KERNEL_SE) return pd.scale_c * exp(-0.0 / 2);
  if (pd.kind == DFH_KERNEL_MATERN) {
    double u = 0.0;
    for (int i = 0; i <= pd.p; ++i) u += pd.coeff[i] * pow(0.0, (double)(pd.p - i));
    u *= (pd.gfac * exp(-pd.s2 * 0.0));
    return pd.scale_c * u;
  }
  return 0.0;
}

// device image of a KernDev: [parts | bw | cols | lcols], each section 16-byte aligned
static size_t pad16(size_t x) { return (x + 15) & ~(size_t)15; }
static void blob_layout(const KernDev& kd, size_t off[4], size_t* total) {
  const size_t P = kd.P ? kd.P : 1;
  off[0] = 0;
  off[1] = off[0] + pad16(sizeof(PartDev) * kd.parts.size());
  off[2] = off[1] + pad16(sizeof(double) * P);
  off[3] = off[2] + pad16(sizeof(int) * P);
  *total = off[3] + pad16(sizeof(int) * P);
}
static void blob_fill(const KernDev& kd, char* host) {
  size_t off[4], total;
  blob_layout(kd, off, &total);
  std::memcpy(host + off[0], kd.parts.data(), sizeof(PartDev) * kd.parts.size());
  std::memcpy(host + off[1], kd.bw.data(), sizeof(double) * kd.P);
  std::memcpy(host + off[2], kd.cols.data(), sizeof(int) * kd.P);
  std::memcpy(host + off[3], kd.lcols.data(), sizeof(int) * kd.P);
}
static void blob_point(KernDev* kd, char* dev) {
  size_t off[4], total;
  blob_layout(*kd, off, &total);
  kd->d_parts = reinterpret_cast<PartDev*>(dev + off[0]);
  kd->d_bw = reinterpret_cast<double*>(dev + off[1]);
  kd->d_cols = reinterpret_cast<int*>(dev + off[2]);
  kd->d_lcols = reinterpret_cast<int*>(dev + off[3]);
}

int upload(dfh_ctx* ctx, KernDev* kd) {
  size_t off[4], total;
  blob_layout(*kd, off, &total);
  std::vector<char> host(total, 0);
  blob_fill(*kd, host.data());
  DFH_HIP(hipMalloc(&kd->d_blob, total));
  blob_point(kd, static_cast<char*>(kd->d_blob));
  DFH_HIP(hipMemcpyAsync(kd->d_blob, host.data(), total, hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

void add_part_cols(KernDev* kd, PartDev& pd, const int* cols, const double* bw, int ncols) {
  pd.poff = kd->P;
  pd.kc = (ncols + 3) & ~3;
  for (int c = 0; c < pd.kc; ++c) {
    kd->cols.push_back(c < ncols ? cols[c] : -1);
    kd->lcols.push_back(c < ncols ? c : -1);
    kd->bw.push_back(c < ncols ? bw[c] : 1.0);
  }
  kd->P += pd.kc;
}

// ---------------------------------------------------------------------------------------
// The whole tuning objective of SMALL problems in one launch (n <= TINY_MAX_N): workgroup c packs
// the inputs for candidate c's kernel, builds K + noise I, factors it (stable_cholesky's jitter
// ladder included) and solves for the log marginal likelihood -- all in LDS, nothing but the two
// result numbers goes back to HBM.  Sequential hyper-parameter searches (the reference's slice
// sampler, its PDOO) ask for a handful of such values per call thousands of times; with one
// launch per stage a call costs ~0.3 ms of launches and synchronisations, far more than the
// arithmetic of a 50 x 50 Cholesky.
// ---------------------------------------------------------------------------------------
struct TinyArgs {
  ExpConsts ec;
  const double* X; long ldx;       // [n x d] raw inputs (device)
  const char* blob;                // TinyCand[count] | kernel images | y[n] | pow10[16]
  long y_off, pow_off;
  int n, count, allow_jitter;
  int direct;                      // blob and out are host memory mapped into the device (small groups: no copies)
#ifdef DFH_DEBUG_HOOKS
  long long* stamps;               // diagnostics (DFH_TINY_STAMPS=1): [count][16] s_memrealtime (100 MHz) of k_lml_tiny64's phases
#endif
  double* out;                     // [count][4] = {sum log L_ii, |L^-1 (y - m)|^2, jitter power or -100, status}
};

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower, j <= i

__global__ __launch_bounds__(256) void k_lml_tiny(TinyArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int c = blockIdx.x, tid = threadIdx.x, n = a.n;
  const TinyCand cand = reinterpret_cast<const TinyCand*>(a.blob)[c];
  const char* image = a.blob + cand.image;
  const int P = cand.P, n_parts = cand.n_parts;
  // kernel image sections (blob_layout): parts | bw | cols | lcols
  const size_t off_bw = (sizeof(PartDev) * n_parts + 15) & ~size_t(15);
  const size_t off_cols = off_bw + ((sizeof(double) * (P ? P : 1) + 15) & ~size_t(15));
  const PartDev* parts_g = reinterpret_cast<const PartDev*>(image);
  __shared__ PartDev parts[TINY_MAX_PARTS];          // the Gram loop reads them per entry: keep them off the global-load path
  for (int q = tid; q < n_parts * (int)(sizeof(PartDev) / sizeof(int)); q += 256)
    reinterpret_cast<int*>(parts)[q] = reinterpret_cast<const int*>(parts_g)[q];
  const double* bw = reinterpret_cast<const double*>(image + off_bw);
  const int* cols = reinterpret_cast<const int*>(image + off_cols);
  const double* y = reinterpret_cast<const double*>(a.blob + a.y_off);
  const double* pow10 = reinterpret_cast<const double*>(a.blob + a.pow_off);

  double* A = lds;                                   // packed lower triangle of the (n+1) x (n+1) system
  double* Xp = A + (n + 1) * (n + 2) / 2;            // [n][P]
  double* Np = Xp + n * P;                           // [n][n_parts]
  __shared__ int s_fail;
  __shared__ double s_pivot;

  // get_scaled_repr (kernel.py:179-181) and the squared row norms (general_utils.py:66-67)
  for (int idx = tid; idx < n * P; idx += 256) {
    const int row = idx / P, pc = idx - row * P;
    const int col = cols[pc];
    Xp[idx] = col >= 0 ? a.X[(long)row * a.ldx + col] / bw[pc] : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < n * n_parts; idx += 256) {
    const int row = idx / n_parts, part = idx - row * n_parts;
    const PartDev& pd = parts[part];
    int nreal = 0;
    for (int q = 0; q < pd.kc; ++q) nreal += cols[pd.poff + q] >= 0;
    Np[idx] = np_sumsq(Xp + row * P + pd.poff, nreal);
  }
  __syncthreads();

  const int tx = tid & 15, ty = tid >> 4;
  double max_diag = 0.0;                             // max(diag(K + noise I)), for the ladder
  int power = -100;                                  // -100: no jitter needed
  for (int attempt = 0; attempt < 17; ++attempt) {
    double jitter = 0.0;
    if (attempt > 0) {
      power = attempt - 12;                          // -11 ... 4 (general_utils.py:183-203)
      jitter = pow10[attempt - 1] * max_diag;
    }
    // K + noise I (+ jitter I), lower triangle; row n of the system is y - m
    for (int i = ty; i < n; i += 16) {
      for (int j = tx; j <= i; j += 16) {
        double res = cand.multi ? (cand.product ? cand.outer : 0.0) : 0.0;
        double fsum = 0.0;
        for (int part = 0; part < n_parts; ++part) {
          const PartDev& pd = parts[part];
          const double* xi = Xp + i * P + pd.poff;
          const double* xj = Xp + j * P + pd.poff;
          double dot = 0.0;
          for (int q = 0; q < pd.kc; ++q) dot = fma(xi[q], xj[q], dot);
          double dsq = (Np[j * n_parts + part] + Np[i * n_parts + part]) - 2.0 * dot;   // general_utils.py:66-68
          dsq = dsq < 0.0 ? 0.0 : dsq;
          const double kv = kern_eval(pd, dsq, a.ec);
          if (!cand.multi) res = kv;
          else if (!cand.product) res = res + kv;
          else combine_nested(pd, kv, res, fsum);          // (a plain factor: res * kv)
        }
        if (cand.multi && !cand.product) res = cand.outer * res;
        if (i == j) {
          res += cand.noise;                         // gp_core.py:843
          if (attempt > 0) res += jitter;            // M + diag_noise * np.eye(n)
        }
        A[tri(i, j)] = res;
      }
    }
    for (int j = tid; j < n; j += 256) A[tri(n, j)] = y[j] - cand.mean;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    if (attempt == 0) {                              // np.diag(M).max() of the un-jittered matrix
      double m = -INFINITY;
      bool any_nan = false;
      for (int i = 0; i < n; ++i) { const double v = A[tri(i, i)]; any_nan |= (v != v); m = v > m ? v : m; }
      max_diag = any_nan ? NAN : m;
    }
    // Left-looking Cholesky in panels of four columns; the extra row turns into z = L^-1 (y - m)
    // along the way.  Two threads per row.  For a panel starting at k0 the bulk of the work --
    // b[c] = sum_{j<k0} L[i][j] L[k0+c][j], c = 0..3 -- is one pass over the row: each L[i][j]
    // (a per-lane LDS load) feeds four FMAs, the four panel rows are broadcasts; nothing is stored
    // inside the pass.  The four columns are then finished one after the other from registers:
    // v = A[i][k] - b[c] - sum_{c'<c} L[i][k0+c'] L[k][k0+c'], pivot, scale -- two barriers each.
    const int half = tid & 1, slot = tid >> 1;
    bool failed = false;
    for (int k0 = 0; k0 < n && !failed; k0 += 4) {
      const int width = n - k0 < 4 ? n - k0 : 4;
      const int j0 = half ? (k0 + 1) / 2 : 0, j1 = half ? k0 : (k0 + 1) / 2;
      const double* prow[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) prow[c] = A + tri(k0 + (c < width ? c : 0), 0);
      // this thread's rows: i0 = k0 + slot and, only while more than 128 rows are left, i0 + 128
      double bulk[2][4], mine[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int i = k0 + slot + 128 * r;
#pragma unroll
        for (int c = 0; c < 4; ++c) { bulk[r][c] = 0.0; mine[r][c] = 0.0; }
        if (i > n) continue;
        const double* rowi = A + tri(i, 0);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        int j = j0;
        for (; j + 2 <= j1; j += 2) {
          const double x = rowi[j], y2 = rowi[j + 1];
          s0 = fma(x, prow[0][j], s0); s1 = fma(x, prow[1][j], s1);
          s2 = fma(x, prow[2][j], s2); s3 = fma(x, prow[3][j], s3);
          t0 = fma(y2, prow[0][j + 1], t0); t1 = fma(y2, prow[1][j + 1], t1);
          t2 = fma(y2, prow[2][j + 1], t2); t3 = fma(y2, prow[3][j + 1], t3);
        }
        if (j < j1) {
          const double x = rowi[j];
          s0 = fma(x, prow[0][j], s0); s1 = fma(x, prow[1][j], s1);
          s2 = fma(x, prow[2][j], s2); s3 = fma(x, prow[3][j], s3);
        }
        s0 += t0; s1 += t1; s2 += t2; s3 += t3;
        bulk[r][0] = s0 + __shfl_xor(s0, 1, 64); bulk[r][1] = s1 + __shfl_xor(s1, 1, 64);
        bulk[r][2] = s2 + __shfl_xor(s2, 1, 64); bulk[r][3] = s3 + __shfl_xor(s3, 1, 64);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c >= width || failed) continue;            // uniform
        const int k = k0 + c;
        const double* rowk = A + tri(k, 0);
        double v[2] = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int i = k0 + slot + 128 * r;
          if (i < k || i > n) continue;                // rows above this column's diagonal are done
          double acc = A[tri(i, k)] - bulk[r][c];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
            if (cc < c) acc -= mine[r][cc] * rowk[k0 + cc];
          v[r] = acc;
          if (i == k && half == 0) s_pivot = acc;
        }
        __syncthreads();                               // the pivot is published; row k has been read
        const double pivot = s_pivot;
        if (!(pivot > 0.0)) {                          // not positive definite (or NaN): uniform
          if (tid == 0) s_fail = 1;
          failed = true;
          continue;
        }
        const double lkk = sqrt(pivot);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int i = k0 + slot + 128 * r;
          if (i < k || i > n) continue;
          const double lik = (i == k) ? lkk : v[r] / lkk;
          mine[r][c] = lik;
          if (half == 0) A[tri(i, k)] = lik;
        }
        __syncthreads();
      }
    }
    __syncthreads();
    if (!s_fail) break;
    __syncthreads();
    if (!a.allow_jitter || attempt == 16) { power = attempt == 16 ? 99 : 98; break; }   // status below
  }

  double* out = a.out + 4 * (long)c;
  if (s_fail) {
    if (tid == 0) tiny_publish(out, a.direct != 0, NAN, NAN, (double)power, power == 98 ? 1.0 : 2.0);
    return;
  }
  double ld = 0.0, zz = 0.0;
  for (int i = tid; i < n; i += 256) {
    ld += log(A[tri(i, i)]);
    const double z = A[tri(n, i)];
    zz = fma(z, z, zz);
  }
  for (int o = 32; o > 0; o >>= 1) { ld += __shfl_down(ld, o, 64); zz += __shfl_down(zz, o, 64); }
  __shared__ double s_ld[4], s_zz[4];
  if ((tid & 63) == 0) { s_ld[tid >> 6] = ld; s_zz[tid >> 6] = zz; }
  __syncthreads();
  if (tid == 0)
    tiny_publish(out, a.direct != 0, (s_ld[0] + s_ld[1]) + (s_ld[2] + s_ld[3]), (s_zz[0] + s_zz[1]) + (s_zz[2] + s_zz[3]),
                 (double)power, 0.0);
}


#include "factor64.h"   // factor64's owner / consumer steps (shared with chol.hip)

// ---------------------------------------------------------------------------------------
// The same objective for n <= 63 with the factorisation on the 64 x 64 machinery of chol.hip (round 6).
// k_lml_tiny's column loop costs two workgroup barriers, an LDS round trip, a square root and a division per
// column, ~1100 cycles each: 39 us for n = 50 on an otherwise idle device (profiles/r06_small_calls_before.txt), most
// of a slice sampler's call.  Here the system [[K + s2 I, .], [(y - m)^T, 1]] is staged as ONE 64 x 64 tile (identity
// below row n) and factored by the four waves without barriers -- the owner chain of f64_owner_step is ~225 cycles
// per column -- and only as far as column n - 1: row n of the factor, z = L^-1 (y - m), is final in column k as soon as
// column k is, so the augmented row never has to be a pivot, and the waves whose sixteen columns lie beyond n - 1 sit
// the factorisation out.  Everything else -- packing, Gram entries, the jitter ladder, the results -- is k_lml_tiny's.
// ---------------------------------------------------------------------------------------
#ifdef DFH_DEBUG_HOOKS
#define TSTAMP(a, e) do { if ((a).stamps && threadIdx.x == 0) (a).stamps[(long)blockIdx.x * 16 + (e)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(a, e) do {} while (0)
#endif
constexpr size_t TINY64_FIXED_LDS = sizeof(double) * (PB * SPP_STAGE + PB * PB + 3 * PB * 17);

__global__ __launch_bounds__(256, 1) void k_lml_tiny64(TinyArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int c = blockIdx.x, tid = threadIdx.x, n = a.n;
  const int lane = tid & 63, w = tid >> 6;
  TSTAMP(a, 0);
  const TinyCand cand = reinterpret_cast<const TinyCand*>(a.blob)[c];
  const char* image = a.blob + cand.image;
  const int P = cand.P, n_parts = cand.n_parts;
  const size_t off_bw = (sizeof(PartDev) * n_parts + 15) & ~size_t(15);
  const size_t off_cols = off_bw + ((sizeof(double) * (P ? P : 1) + 15) & ~size_t(15));
  const PartDev* parts_g = reinterpret_cast<const PartDev*>(image);
  __shared__ PartDev parts[TINY_MAX_PARTS];
  for (int q = tid; q < n_parts * (int)(sizeof(PartDev) / sizeof(int)); q += 256)
    reinterpret_cast<int*>(parts)[q] = reinterpret_cast<const int*>(parts_g)[q];
  const double* bw = reinterpret_cast<const double*>(image + off_bw);
  const int* cols = reinterpret_cast<const int*>(image + off_cols);
  const double* y = reinterpret_cast<const double*>(a.blob + a.y_off);
  const double* pow10 = reinterpret_cast<const double*>(a.blob + a.pow_off);

  double* stage = lds;                               // [64][SPP_STAGE] the system, lower triangle
  double* ring = stage + PB * SPP_STAGE;             // [64][64] published columns
  double* tbuf0 = ring + PB * PB;                    // 3 x [64][17] layout buffers of waves 1..3
  double* Xp = tbuf0 + 3 * PB * 17;                  // [n][P]
  double* Np = Xp + n * P;                           // [n][n_parts]
  __shared__ int s_badv[4];
  __shared__ int s_ring_timeout;
  __shared__ double s_ld[4], s_zz[4];

  TSTAMP(a, 1);
  // identity below row n, zero above the diagonal; row n = y - m with a unit diagonal
  for (int idx = tid; idx < PB * PB; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    stage[i * SPP_STAGE + j] = (i == j && i >= n) ? 1.0 : 0.0;
  }
  for (int idx = tid; idx < n * P; idx += 256) {
    const int row = idx / P, pc = idx - row * P;
    const int col = cols[pc];
    Xp[idx] = col >= 0 ? a.X[(long)row * a.ldx + col] / bw[pc] : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < n * n_parts; idx += 256) {
    const int row = idx / n_parts, part = idx - row * n_parts;
    const PartDev& pd = parts[part];
    int nreal = 0;
    for (int q = 0; q < pd.kc; ++q) nreal += cols[pd.poff + q] >= 0;
    Np[idx] = np_sumsq(Xp + row * P + pd.poff, nreal);
  }
  for (int j = tid; j < n; j += 256) stage[n * SPP_STAGE + j] = y[j] - cand.mean;
  __syncthreads();

  TSTAMP(a, 2);
  double max_diag = 0.0;
  int power = -100;
  bool failed = false;
  double av[16];
  for (int attempt = 0; attempt < 17; ++attempt) {
    double jitter = 0.0;
    if (attempt > 0) {
      if (attempt == 1) {                            // np.diag(M).max() of the un-jittered matrix, still staged
        double m = -INFINITY;
        bool any_nan = false;
        for (int i = 0; i < n; ++i) { const double v = stage[i * SPP_STAGE + i]; any_nan |= (v != v); m = v > m ? v : m; }
        max_diag = any_nan ? NAN : m;
        __syncthreads();                             // every thread has read the old diagonal
      }
      power = attempt - 12;                          // -11 ... 4 (general_utils.py:183-203)
      jitter = pow10[attempt - 1] * max_diag;
    }
    // K + noise I (+ jitter I: M + diag_noise * np.eye(n), general_utils.py:190), lower triangle   (gp_core.py:843)
    tiny_gram_lower(cand, parts, n_parts, Xp, P, Np, n, a.ec, [&](int i, int j, double v) {
      if (i == j) {
        v += cand.noise;                             // gp_core.py:843
        if (attempt > 0) v += jitter;                // M + diag_noise * np.eye(n)
      }
      stage[i * SPP_STAGE + j] = v;
    });
    if (tid < PB) ring[tid * PB] = 0.0;              // row-0 entries double as the "published" flags
    if (tid == 0) s_ring_timeout = 0;
    __syncthreads();
    TSTAMP(a, 3);
    double* tbuf = tbuf0 + (w > 0 ? (w - 1) : 0) * PB * 17;
    const int bad = tiny64_factor(av, lane, w, stage, tbuf, ring, n - 1, &s_ring_timeout);
    if (lane == 0) s_badv[w] = (bad >= 0 && bad < n) ? bad : -1;
    __syncthreads();
    TSTAMP(a, 4);
    failed = s_badv[0] >= 0 || s_badv[1] >= 0 || s_badv[2] >= 0 || s_badv[3] >= 0 || s_ring_timeout != 0;
    if (!failed) break;
    if (!a.allow_jitter || attempt == 16) { power = attempt == 16 ? 99 : 98; break; }
    __syncthreads();                                 // the verdict is read before the next attempt rewrites it
  }

  double* out = a.out + 4 * (long)c;
  if (failed) {
    if (tid == 0) tiny_publish(out, a.direct != 0, NAN, NAN, (double)power, power == 98 ? 1.0 : 2.0);
    return;
  }
  // sum log L_kk (lane k of the wave that owns column k) and z.z (row n = lane n)
  double lkk = 1.0, zz = 0.0;                        // (one logarithm per lane: log(1) = 0 in the lanes that own no column)
  if (16 * w <= n - 1) {
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) {
      const int k = 16 * w + kl;
      if (k < n) {
        lkk = (lane == k) ? av[kl] : lkk;
        if (lane == n) zz = fma(av[kl], av[kl], zz);
      }
    }
  }
  double ldv = log(lkk);
  for (int o = 32; o > 0; o >>= 1) { ldv += __shfl_down(ldv, o, 64); zz += __shfl_down(zz, o, 64); }
  if (lane == 0) { s_ld[w] = ldv; s_zz[w] = zz; }
  __syncthreads();
  TSTAMP(a, 5);
  if (tid == 0)
    tiny_publish(out, a.direct != 0, (s_ld[0] + s_ld[1]) + (s_ld[2] + s_ld[3]), (s_zz[0] + s_zz[1]) + (s_zz[2] + s_zz[3]),
                 (double)power, 0.0);
  TSTAMP(a, 6);
}

}  // namespace

bool lml_tiny_applies(const KernDev* kds, int count, int64_t n) {
  if (n > TINY_MAX_N) return false;
  for (int c = 0; c < count; ++c)
    if (kds[c].P > TINY_MAX_P || kds[c].n_parts > TINY_MAX_PARTS || kds[c].P < 1 || !kds[c].stationary) return false;
  return true;
}

// logdet_dot[2c], [2c+1] = sum(log(diag(L_c))), (y - m_c)^T (K_c + noise_c I)^-1 (y - m_c);
// powers[c] = jitter power used (INT32_MIN: none).  Returns DFH_ERR_NOT_PD / DFH_ERR_JITTER as the
// one-fit path would.
int tiny_blob_build(dfh_ctx* ctx, const KernDev* kds, int count, int64_t n, const double* y_host,
                    const double* noise_vars, const double* mean_consts, TinyBlob* tb) {
  std::vector<size_t> image_off((size_t)count);
  size_t at = ((sizeof(TinyCand) * (size_t)count) + 15) & ~size_t(15);
  int Pmax = 1, parts_max = 1;
  for (int c = 0; c < count; ++c) {
    image_off[c] = at;
    at += kerndev_blob_bytes(kds[c]);
    Pmax = std::max(Pmax, kds[c].P);
    parts_max = std::max(parts_max, kds[c].n_parts);
  }
  const size_t y_off = at;
  at += sizeof(double) * (size_t)n;
  const size_t pow_off = at;
  at += sizeof(double) * 16;
  // blob and results go through pinned staging memory: two asynchronous copies and one
  // synchronisation per call instead of two staged, blocking ones
  const size_t res_off = (at + 63) & ~size_t(63);
  void* pinned = nullptr;
  DFH_TRY(pinned_get(ctx, res_off + sizeof(double) * 4 * (size_t)count, &pinned));
  char* host_blob = static_cast<char*>(pinned);
  std::memset(host_blob, 0, at);
  TinyCand* cands = reinterpret_cast<TinyCand*>(host_blob);
  for (int c = 0; c < count; ++c) {
    cands[c].image = (long)image_off[c];
    cands[c].P = kds[c].P; cands[c].n_parts = kds[c].n_parts;
    cands[c].multi = kds[c].multi ? 1 : 0; cands[c].product = kds[c].product ? 1 : 0;
    cands[c].outer = kds[c].outer_scale;
    cands[c].noise = noise_vars[c];
    cands[c].mean = mean_consts ? mean_consts[c] : 0.0;
    blob_fill(kds[c], host_blob + image_off[c]);
  }
  std::memcpy(host_blob + y_off, y_host, sizeof(double) * (size_t)n);
  double* pw = reinterpret_cast<double*>(host_blob + pow_off);
  static const std::vector<double> pow10_table = []() {                 // 10 ** diag_noise_power, once per process
    std::vector<double> t(16);
    for (int p = -11; p < 5; ++p) t[(size_t)(p + 11)] = pow(10.0, (double)p);
    return t;
  }();
  std::memcpy(pw, pow10_table.data(), sizeof(double) * 16);
  tb->host = host_blob; tb->bytes = at; tb->y_off = y_off; tb->pow_off = pow_off;
  tb->res = reinterpret_cast<double*>(host_blob + res_off);
  tb->Pmax = Pmax; tb->parts_max = parts_max;
  return DFH_OK;
}

// Host side of a direct call's results: the kernel's status words (res[4 c + 3], -1.0 before the launch) polled in the
// pinned buffer; past the budget the stream is synchronised like any other call and a kernel that never wrote is an error.
int tiny_poll_results(dfh_ctx* ctx, volatile double* vres, int count, const char* what) {
  bool all_in = false;
  const auto t_start = std::chrono::steady_clock::now();
  for (long spin = 0; !all_in; ++spin) {
    all_in = true;
    for (int c = 0; c < count; ++c) all_in = all_in && vres[4 * c + 3] != -1.0;
    if (all_in) break;
    if ((spin & 1023) == 1023 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 0.25) break;
  }
  if (!all_in) {
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < count; ++c)
      if (vres[4 * c + 3] == -1.0) { dfh_set_error("%s: no result for candidate %d", what, c); return DFH_ERR_HIP; }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return DFH_OK;
}

int lml_tiny_batch(dfh_ctx* ctx, const KernDev* kds, int count, const double* dX, int64_t n, int64_t ldx,
                   const double* y_host, const double* noise_vars, const double* mean_consts,
                   bool allow_jitter, double* logdet_dot, int32_t* powers) {
  TinyBlob tb;
  DFH_TRY(tiny_blob_build(ctx, kds, count, n, y_host, noise_vars, mean_consts, &tb));
  char* host_blob = tb.host;
  const size_t at = tb.bytes, y_off = tb.y_off, pow_off = tb.pow_off;
  const int Pmax = tb.Pmax, parts_max = tb.parts_max;
  double* res = tb.res;
  // A handful of candidates (a slice sampler's or a tree search's call: gp_core.py:551-574 under sampling/slice.py,
  // utils/doo.py) is latency, not work: the kernel reads the descriptors straight from the pinned buffer (mapped into
  // the device: a few hundred bytes over PCIe) and writes its four numbers per candidate straight back into it, status
  // word last, while the host polls that word -- one launch, no copy, no stream synchronisation.  DFH_LML_DIRECT=0: off;
  // =N: groups of up to N candidates (default 16).
  static const int direct_max = []() { const char* e = getenv("DFH_LML_DIRECT"); return e ? atoi(e) : 16; }();
  const bool direct = count <= direct_max && at <= (size_t)32768 && !ctx->timing;
  void* d_blob = nullptr;
  double* d_out = nullptr;
  if (!direct) {
    DFH_TRY(scratch_get(ctx, SCR_AUG2, at, &d_blob));
    DFH_TRY(scratch_get(ctx, SCR_OUT2, sizeof(double) * 4 * (size_t)count, (void**)&d_out));
    DFH_HIP(hipMemcpyAsync(d_blob, host_blob, at, hipMemcpyHostToDevice, ctx->stream));
  }
  TinyArgs a;
  a.ec = kExpConsts;
  a.X = dX; a.ldx = ldx;
  a.blob = direct ? host_blob : static_cast<const char*>(d_blob);
  a.y_off = (long)y_off; a.pow_off = (long)pow_off;
  a.n = (int)n; a.count = count; a.allow_jitter = allow_jitter ? 1 : 0;
  a.direct = direct ? 1 : 0;
  a.out = direct ? res : d_out;
#ifdef DFH_DEBUG_HOOKS
  static long long* d_stamps = nullptr;
  static const bool want_stamps = getenv("DFH_TINY_STAMPS") != nullptr;
  if (want_stamps && !d_stamps) DFH_HIP(hipMalloc((void**)&d_stamps, 64 * 16 * 8));
  a.stamps = (want_stamps && count <= 64) ? d_stamps : nullptr;
#endif
  volatile double* vres = res;
  if (direct)
    for (int c = 0; c < count; ++c) vres[4 * c + 3] = -1.0;      // "not there yet": the kernel's status is 0, 1 or 2
  static bool attr_set[DFH_MAX_DEVICES] = {false};
  if (!attr_set[ctx->device]) {
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lml_tiny), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 4096));      // static LDS (kernel parts, flags) takes ~2 KB
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lml_tiny64), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 4096));
    attr_set[ctx->device] = true;
  }
  // n <= 63: the system is one 64 x 64 tile for the barrier-free factorisation (k_lml_tiny64); DFH_LML_TINY64=0: k_lml_tiny
  static const bool tiny64 = []() { const char* e = getenv("DFH_LML_TINY64"); return e ? atoi(e) != 0 : true; }();
  if (tiny64 && n <= TINY64_MAX_N) {
    const size_t lds_bytes = TINY64_FIXED_LDS + sizeof(double) * ((size_t)n * Pmax + (size_t)n * parts_max);
    hipLaunchKernelGGL(k_lml_tiny64, dim3((unsigned)count), dim3(256), lds_bytes, ctx->stream, a);
  } else {
    const size_t lds_bytes = sizeof(double) * ((size_t)(n + 1) * (n + 2) / 2 + (size_t)n * Pmax + (size_t)n * parts_max);
    hipLaunchKernelGGL(k_lml_tiny, dim3((unsigned)count), dim3(256), lds_bytes, ctx->stream, a);
  }
  DFH_LAUNCH_CHECK();
  if (direct) {
    // (a kernel of this size runs tens of microseconds; a ladder over seventeen attempts a millisecond)
    DFH_TRY(tiny_poll_results(ctx, vres, count, "k_lml_tiny"));
  } else {
    DFH_HIP(hipMemcpyAsync(res, d_out, sizeof(double) * 4 * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
  }
#ifdef DFH_DEBUG_HOOKS
  if (a.stamps) {
    static long long acc[8] = {0}; static long calls = 0;
    long long hs[16];
    DFH_HIP(hipMemcpy(hs, d_stamps, sizeof(hs), hipMemcpyDeviceToHost));
    for (int e = 0; e < 6; ++e) acc[e] += hs[e + 1] - hs[e];
    if (++calls % 1000 == 0) {
      fprintf(stderr, "[tiny64 stamps, mean of 1000, us] cand %.2f | descr+fill %.2f | pack+norms %.2f | gram %.2f | factor %.2f | reduce %.2f | publish %.2f\n",
              0.0, acc[0] / 1e5, acc[1] / 1e5, acc[2] / 1e5, acc[3] / 1e5, acc[4] / 1e5, acc[5] / 1e5);
      for (int e = 0; e < 8; ++e) acc[e] = 0;
    }
  }
#endif
  for (int c = 0; c < count; ++c) {
    const int status = (int)res[4 * c + 3];
    if (status == 1) {
      dfh_set_error("Matrix is not positive definite (candidate %d)", c);
      return DFH_ERR_NOT_PD;
    }
    if (status == 2) {
      dfh_set_error("Could not compute Cholesky decomposition despite adding jitter to the diagonal (candidate %d). "
                    "This is likely because the M is not positive semi-definite or has infinities/nans.", c);
      return DFH_ERR_JITTER;
    }
    logdet_dot[2 * c] = res[4 * c];
    logdet_dot[2 * c + 1] = res[4 * c + 1];
    const int pwr = (int)res[4 * c + 2];
    if (powers) powers[c] = pwr == -100 ? INT32_MIN : pwr;
  }
  return DFH_OK;
}

// One part from (kind, scale, nu, per-column parameters): SE / Matern bandwidths divide the inputs,
// polynomial scalings multiply them (stored negated, see k_pack_cols), exponential-decay powers go
// into the part and its inputs stay as they are.
static int make_part(KernDev* kd, int kind, double scale, double nu, const int* cols, const double* par, int ncols) {
  PartDev pd;
  DFH_TRY(fill_part(pd, kind, scale, nu));
  pd.fmode = 0; pd.fpad = 0; pd.fscale = 1.0;
  std::vector<double> bw((size_t)ncols);
  if (kind == DFH_KERNEL_POLY) {
    for (int c = 0; c < ncols; ++c) {
      if (!(par[c] > 0.0)) { dfh_set_error("polynomial kernel: dim_scalings must be positive"); return DFH_ERR_BAD_ARG; }
      bw[c] = -par[c];
    }
  } else if (kind == DFH_KERNEL_EXPDECAY) {
    if (ncols > EXPDECAY_MAX_DIM) {
      dfh_set_error("exponential-decay kernel: at most %d dimensions (got %d)", EXPDECAY_MAX_DIM, ncols);
      return DFH_ERR_BAD_ARG;
    }
    pd.p = ncols;
    for (int c = 0; c < ncols; ++c) { pd.coeff[c] = par[c]; bw[c] = 1.0; }
  } else {
    for (int c = 0; c < ncols; ++c) bw[c] = par[c];
  }
  add_part_cols(kd, pd, cols, bw.data(), ncols);
  kd->parts.push_back(pd);
  return DFH_OK;
}

static bool kind_is_stationary(int kind) { return kind == DFH_KERNEL_SE || kind == DFH_KERNEL_MATERN; }

int kerndev_build_host(const dfh_kernel_desc* k, KernDev* kd) {
  DFH_ARG(k != nullptr && kd != nullptr);
  DFH_ARG(k->dim >= 1);
  kd->kind = k->kind; kd->dim = k->dim; kd->P = 0;
  kd->parts.clear(); kd->cols.clear(); kd->lcols.clear(); kd->bw.clear();
  kd->stationary = true; kd->kxx = 0.0;
  if (k->kind == DFH_KERNEL_SE || k->kind == DFH_KERNEL_MATERN) {
    DFH_ARG(k->bw != nullptr);
    std::vector<int> ident(k->dim);
    for (int i = 0; i < k->dim; ++i) ident[i] = i;
    DFH_TRY(make_part(kd, k->kind, k->scale, k->nu, ident.data(), k->bw, k->dim));
    kd->multi = false; kd->product = false; kd->outer_scale = 1.0; kd->nested = false;
    kd->kxx = kd->parts[0].k0;
  } else if (k->kind == DFH_KERNEL_POLY || k->kind == DFH_KERNEL_EXPDECAY) {
    // a product with one factor and outer scale 1 (1.0 * k is exact): the generic multi-part
    // kernel-matrix kernel is the only one that knows these kinds
    DFH_ARG(k->bw != nullptr);
    std::vector<int> ident(k->dim);
    for (int i = 0; i < k->dim; ++i) ident[i] = i;
    DFH_TRY(make_part(kd, k->kind, k->scale, k->nu, ident.data(), k->bw, k->dim));
    kd->multi = true; kd->product = true; kd->outer_scale = 1.0; kd->nested = false;
    kd->stationary = false;
  } else if (k->kind == DFH_KERNEL_ADDITIVE || k->kind == DFH_KERNEL_PRODUCT) {
    DFH_ARG(k->n_groups >= 1 && k->group_off && k->group_dims && k->sub_kind && k->sub_scale && k->sub_bw);
    const bool product = (k->kind == DFH_KERNEL_PRODUCT);
    const bool nested = product && k->group_factor != nullptr;
    if (k->group_factor || k->factor_is_sum || k->factor_scale)
      DFH_ARG(product && k->group_factor && k->factor_is_sum && k->factor_scale);
    double acc = product ? k->scale : 0.0;
    double facc = 0.0;                              // k(x, x) of the additive factor under way
    for (int g = 0; g < k->n_groups; ++g) {
      const int lo = k->group_off[g], hi = k->group_off[g + 1];
      DFH_ARG(hi > lo);
      for (int c = lo; c < hi; ++c) DFH_ARG(k->group_dims[c] >= 0 && k->group_dims[c] < k->dim);
      const int sk = k->sub_kind[g];
      // polynomial groups also in an additive kernel (the reference's factory builds them: euclidean_gp.py:870-879)
      DFH_ARG(kind_is_stationary(sk) || sk == DFH_KERNEL_POLY || (product && sk == DFH_KERNEL_EXPDECAY));
      DFH_TRY(make_part(kd, sk, k->sub_scale[g], k->sub_nu ? k->sub_nu[g] : 0.0, k->group_dims + lo,
                        k->sub_bw + lo, hi - lo));
      if (!kind_is_stationary(sk)) kd->stationary = false;
      const double k0 = kd->parts.back().k0;
      PartDev& pd = kd->parts.back();
      pd.fmode = 0; pd.fpad = 0; pd.fscale = 1.0;
      if (nested) {
        const int f = k->group_factor[g];
        DFH_ARG(f >= 0 && f <= g && (g == 0 ? f == 0 : (f == k->group_factor[g - 1] || f == k->group_factor[g - 1] + 1)));
        const bool first = g == 0 || k->group_factor[g - 1] != f;
        const bool last = g + 1 == k->n_groups || k->group_factor[g + 1] != f;
        if (k->factor_is_sum[f]) {
          DFH_ARG(sk != DFH_KERNEL_EXPDECAY);       // an additive kernel's groups: SE / Matern / polynomial
          pd.fmode = FM_IN | (first ? FM_BEGIN : 0) | (last ? FM_END : 0);
          pd.fscale = k->factor_scale[f];
          facc = first ? 0.0 + k0 : facc + k0;
          if (last) acc *= pd.fscale * facc;
        } else {
          DFH_ARG(first && last);                   // a plain factor is one group
          acc *= k0;
        }
      } else if (product) {
        acc *= k0;                                    // K *= kernel(...)        kernel.py:588
      } else {
        acc += k0;                                    // result += kernel(...)   kernel.py:493
      }
    }
    kd->multi = true; kd->product = product; kd->outer_scale = k->scale; kd->nested = nested;
    kd->kxx = !kd->stationary ? 0.0 : (product ? acc : k->scale * acc);        // kernel.py:494
  } else {
    dfh_set_error("unknown kernel kind %d", k->kind);
    return DFH_ERR_BAD_ARG;
  }
  kd->n_parts = (int)kd->parts.size();
  return DFH_OK;
}

// k(x_i, x_i) for every packed point: what the diagonal of kernel(X, X) holds in the reference
// (gp_core.py:181 takes it from the full test Gram matrix).
__global__ void k_prior_diag(const PartDev* __restrict__ parts, int n_parts, int multi, int product, double outer,
                             const double* __restrict__ Xp, const double* __restrict__ Np, long m, int P,
                             double* __restrict__ out, int g_lo, int g_hi) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double res = (multi && product) ? outer : 0.0;
  double fsum = 0.0;
  for (int g = g_lo; g < g_hi; ++g) {
    const PartDev& pd = parts[g];
    double kv;
    if (pd.kind == DFH_KERNEL_POLY) kv = poly_eval(pd, Np[i * n_parts + g]);
    else if (pd.kind == DFH_KERNEL_EXPDECAY) kv = expdecay_eval(pd, Xp + i * P + pd.poff, Xp + i * P + pd.poff);
    else kv = pd.k0;
    if (!multi) res = kv;
    else if (!product) res = res + kv;
    else combine_nested(pd, kv, res, fsum);
  }
  if (multi && !product) res = outer * res;
  out[i] = res;
}

int prior_diag(dfh_ctx* ctx, const KernDev& kd, const double* Xp, const double* Np, int64_t m, double* out, int part_lo,
               int part_hi) {
  if (m <= 0) return DFH_OK;
  if (part_hi < 0) { part_lo = 0; part_hi = kd.n_parts; }
  hipLaunchKernelGGL(k_prior_diag, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, kd.d_parts,
                     kd.n_parts, kd.multi ? 1 : 0, kd.product ? 1 : 0, kd.outer_scale, Xp, Np, (long)m, kd.P, out, part_lo,
                     part_hi);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

int kerndev_build(dfh_ctx* ctx, const dfh_kernel_desc* k, KernDev* kd) {
  DFH_TRY(kerndev_build_host(k, kd));
  return upload(ctx, kd);
}

size_t kerndev_blob_bytes(const KernDev& kd) {
  size_t off[4], total;
  blob_layout(kd, off, &total);
  return total;
}

// The same without the copy: the images are laid out in `host` (the caller's pinned staging memory, copied up by the
// caller together with whatever else the launch needs) and the descriptors pointed at where they WILL be on the device.
int kerndev_stage_many(KernDev* kds, int count, char* host, void* d_blob, size_t blob_bytes) {
  std::memset(host, 0, blob_bytes);
  size_t at = 0;
  for (int c = 0; c < count; ++c) {
    const size_t sz = kerndev_blob_bytes(kds[c]);
    DFH_ARG(at + sz <= blob_bytes);
    blob_fill(kds[c], host + at);
    kds[c].d_blob = nullptr;                    // not owned
    blob_point(&kds[c], static_cast<char*>(d_blob) + at);
    at += sz;
  }
  return DFH_OK;
}

int kerndev_upload_many(dfh_ctx* ctx, KernDev* kds, int count, void* d_blob, size_t blob_bytes) {
  std::vector<char> host(blob_bytes, 0);
  size_t at = 0;
  for (int c = 0; c < count; ++c) {
    const size_t sz = kerndev_blob_bytes(kds[c]);
    DFH_ARG(at + sz <= blob_bytes);
    blob_fill(kds[c], host.data() + at);
    kds[c].d_blob = nullptr;                    // not owned
    blob_point(&kds[c], static_cast<char*>(d_blob) + at);
    at += sz;
  }
  DFH_HIP(hipMemcpyAsync(d_blob, host.data(), at, hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}


int kerndev_build_dist(dfh_ctx* ctx, int dim, KernDev* kd) {
  DFH_ARG(dim >= 1);
  kd->kind = DFH_KERNEL_DIST; kd->dim = dim; kd->P = 0;
  kd->parts.clear(); kd->cols.clear(); kd->lcols.clear(); kd->bw.clear();
  PartDev pd;
  DFH_TRY(fill_part(pd, DFH_KERNEL_DIST, 1.0, 0.0));
  std::vector<int> ident(dim);
  std::vector<double> ones(dim, 1.0);
  for (int i = 0; i < dim; ++i) ident[i] = i;
  add_part_cols(kd, pd, ident.data(), ones.data(), dim);
  kd->parts.push_back(pd);
  kd->multi = false; kd->outer_scale = 1.0; kd->kxx = 0.0;
  kd->n_parts = 1;
  return upload(ctx, kd);
}

int kerndev_clone(dfh_ctx* ctx, const KernDev& src, KernDev* out) {
  *out = src;
  out->d_blob = nullptr; out->d_parts = nullptr; out->d_cols = nullptr; out->d_lcols = nullptr; out->d_bw = nullptr;
  return upload(ctx, out);
}

void kerndev_free(KernDev* kd) {
  if (!kd) return;
  if (kd->d_blob) (void)hipFree(kd->d_blob);
  kd->d_blob = nullptr;
  kd->d_parts = nullptr; kd->d_cols = nullptr; kd->d_lcols = nullptr; kd->d_bw = nullptr;
}

double kerndev_part_kxx(const KernDev& kd, int part) { return part_value_at_zero(kd.parts[part]); }

int pack_scaled(dfh_ctx* ctx, const KernDev& kd, int part_lo, int part_hi, bool pre_gathered,
                const double* X, int64_t n, int64_t ldx, double* Xp, double* Np, int count,
                int64_t sBlob, int64_t sXp, int64_t sNp) {
  if (n <= 0 || count <= 0) return DFH_OK;
  DFH_ARG(part_lo >= 0 && part_hi <= kd.n_parts && part_lo < part_hi);
  DFH_ARG(!pre_gathered || part_hi == part_lo + 1);
  const int c_lo = kd.parts[part_lo].poff;
  const int c_hi = kd.parts[part_hi - 1].poff + kd.parts[part_hi - 1].kc;
  static const bool fused_pack = getenv("DFH_PACK_FUSED") ? atoi(getenv("DFH_PACK_FUSED")) != 0 : true;
  if (fused_pack && c_hi - c_lo <= 2048) {
    const int w = c_hi - c_lo;
    int R = 4096 / w;                          // <= 32 KB of LDS
    R = R < 1 ? 1 : (R > 64 ? 64 : R);
    hipLaunchKernelGGL(k_pack_fused, dim3((unsigned)((n + R - 1) / R), (unsigned)count), dim3(256), (size_t)R * w * 8,
                       ctx->stream, X, (long)n, (long)ldx, kd.P, c_lo, c_hi, pre_gathered ? kd.d_lcols : kd.d_cols,
                       kd.d_cols, kd.d_bw, kd.d_parts, part_lo, part_hi, kd.n_parts, R, Xp, Np, (long)sBlob, (long)sXp,
                       (long)sNp);
    DFH_LAUNCH_CHECK();
    return DFH_OK;
  }
  const int64_t total = n * (c_hi - c_lo);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  // pre-gathered input: local column index c - poff ; d_lcols holds that mapping
  hipLaunchKernelGGL(k_pack_cols, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, ctx->stream, X, (long)n,
                     (long)ldx, kd.P, c_lo, c_hi, pre_gathered ? kd.d_lcols : kd.d_cols, kd.d_bw, Xp,
                     (long)sBlob, (long)sXp);
  DFH_LAUNCH_CHECK();
  const int64_t tn = n * (part_hi - part_lo);
  hipLaunchKernelGGL(k_pack_norms, dim3((unsigned)((tn + 255) / 256), (unsigned)count), dim3(256), 0, ctx->stream,
                     Xp, (long)n, kd.P, kd.n_parts, kd.d_parts, kd.d_cols, part_lo, part_hi, Np,
                     (long)sBlob, (long)sXp, (long)sNp);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

// `count` symmetric Gram matrices of structurally identical single-part kernels in one launch
// (blockIdx.z): kernel images sBlob bytes apart starting at kd's, packed inputs sXp / sNp doubles
// apart, outputs sK doubles apart, diag_adds[count] on the device.  Needs an even ldk.
int kernmat_sym_batch(dfh_ctx* ctx, const KernDev& kd, int count, int64_t sBlob, const double* Xp,
                      int64_t sXp, const double* Np, int64_t sNp, int64_t n, const double* d_diag_adds,
                      double* K, int64_t sK, int64_t ldk) {
  if (n <= 0 || count <= 0) return DFH_OK;
  DFH_ARG(!kd.multi && kd.n_parts == 1 && (ldk & 1) == 0 && (sK & 1) == 0 &&
          (reinterpret_cast<uintptr_t>(K) & 15) == 0 && count <= 65535);
  KmArgs a;
  a.lower_only = 0;
  a.ec = kExpConsts;
  a.Xp1 = Xp; a.Np1 = Np; a.Xp2 = Xp; a.Np2 = Np;
  a.n1 = (int)n; a.n2 = (int)n; a.P = kd.P; a.n_parts_total = kd.n_parts;
  a.parts = kd.d_parts; a.part_lo = 0; a.part_hi = 1;
  a.outer = kd.outer_scale; a.apply_outer = 1; a.symmetric = 1; a.diag_add = 0.0; a.product = 0;
  a.nt_stores = 0;             // (lock-step batches of small matrices: they are factored right away, out of the caches)
  a.K = K; a.ldk = ldk;
  a.sXp = sXp; a.sNp = sNp; a.sK = sK; a.sBlob = sBlob; a.diag_adds = d_diag_adds;
  const int64_t T = (n + 63) / 64;
  const int smem = ((2 * 64 * (16 + 2) > 32 * (64 + 2) ? 2 * 64 * (16 + 2) : 32 * (64 + 2)) + 2 * 64) * 8;
  hipLaunchKernelGGL((kernmat_sym_kernel<64, 16, 32, 7, true>), dim3((unsigned)(T * (T + 1) / 2), 1, (unsigned)count),
                     dim3(256), smem, ctx->stream, a);
  DFH_LAUNCH_CHECK();
  return DFH_OK;
}

int kernmat_packed(dfh_ctx* ctx, const KernDev& kd, int part_lo, int part_hi, bool apply_outer,
                   const double* Xp1, const double* Np1, int64_t n1, const double* Xp2,
                   const double* Np2, int64_t n2, bool symmetric, double diag_add, double* K,
                   int64_t ldk, const double* mu_alpha, double* mu_out, bool* mu_done) {
  if (mu_done) *mu_done = false;
  if (n1 <= 0 || n2 <= 0) return DFH_OK;
  DFH_ARG(n1 < (1LL << 31) && n2 < (1LL << 31));
  KmArgs a;
  a.lower_only = 0;
  a.mu_alpha = nullptr; a.mu_part = nullptr; a.mu_out = nullptr; a.mu_nblk = 0;
  a.ec = kExpConsts;
  a.sXp = a.sNp = a.sK = a.sBlob = 0; a.diag_adds = nullptr;
  a.Xp1 = Xp1; a.Np1 = Np1; a.Xp2 = Xp2; a.Np2 = Np2;
  a.n1 = (int)n1; a.n2 = (int)n2; a.P = kd.P; a.n_parts_total = kd.n_parts;
  a.parts = kd.d_parts; a.part_lo = part_lo; a.part_hi = part_hi;
  a.outer = kd.outer_scale; a.apply_outer = apply_outer ? 1 : 0; a.product = kd.product ? 1 : 0;
  a.symmetric = symmetric ? 1 : 0; a.diag_add = diag_add;
  a.lower_only = (symmetric && ctx->km_lower_only) ? 1 : 0;
  a.K = K; a.ldk = ldk;
  {
    static const int nt_env = []() { const char* e = getenv("DFH_KM_NT"); return e ? atoi(e) : -1; }();
    a.nt_stores = nt_env >= 0 ? (nt_env != 0) : (kd.P >= 16 && n1 * n2 >= (int64_t)4096 * 4096);
  }
  const bool multi = kd.multi;
  static bool attr_set_dev[DFH_MAX_DEVICES] = {false};
  bool& attr_set = attr_set_dev[ctx->device];
  constexpr int SM4 = ((KM_BM + 128) * KM_KP + KM_BM + 128) * 8;
  constexpr int SM2 = ((KM_BM + 64) * KM_KP + KM_BM + 64) * 8;
  if (!attr_set) {
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernmat_kernel<4, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SM4));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernmat_kernel<2, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SM2));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernmat_kernel<2, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SM2));
    DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernmat_kernel<2, true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SM2));
    attr_set = true;
  }
  if (!multi && part_hi == part_lo + 1 && (ldk & 1) == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0 &&
      (n1 + 63) / 64 <= 65535) {
    // 64 x 64 tiles, 16-column operand chunks, 32-row staging: ~20 KB of LDS and 69 VGPRs per
    // workgroup -> 7-8 workgroups per CU whose load / MFMA / exp / store phases overlap.
    static const int sym_cfg = []() { const char* e = getenv("DFH_KM_CFG"); return e ? atoi(e) : 0; }();
    auto smem_bytes = [](int TS, int KC, int SR) {
      const int oper = 2 * TS * (KC + 2), stage = SR * (TS + 2);
      return ((oper > stage ? oper : stage) + 2 * TS) * 8;
    };
    if (symmetric) {
      if (sym_cfg == 2) {
        static bool attr_dev[DFH_MAX_DEVICES] = {false};
        bool& attr = attr_dev[ctx->device];
        if (!attr) { DFH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernmat_sym_kernel<128, 32, 64, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes(128, 32, 64))); attr = true; }
        const int64_t T = (n1 + 127) / 128;
        hipLaunchKernelGGL((kernmat_sym_kernel<128, 32, 64, 2, true>), dim3((unsigned)(T * (T + 1) / 2)), dim3(256), smem_bytes(128, 32, 64), ctx->stream, a);
      } else if (sym_cfg == 1) {
        const int64_t T = (n1 + 63) / 64;
        hipLaunchKernelGGL((kernmat_sym_kernel<64, 32, 64, 4, true>), dim3((unsigned)(T * (T + 1) / 2)), dim3(256), smem_bytes(64, 32, 64), ctx->stream, a);
      } else {
        const int64_t T = (n1 + 63) / 64;
        hipLaunchKernelGGL((kernmat_sym_kernel<64, 16, 32, 7, true>), dim3((unsigned)(T * (T + 1) / 2)), dim3(256), smem_bytes(64, 16, 32), ctx->stream, a);
      }
    } else {
      // strip kernel: SE / Matern (nu = 0.5, 1.5, 2.5), packed width 8 / 16 / 24 / 32, 32-bit in-strip offsets
      static const bool strip_on = []() { const char* e = getenv("DFH_KM_STRIP"); return e ? atoi(e) != 0 : true; }();
      const PartDev& hp = kd.parts[part_lo];
      const bool strip_ok = strip_on && (hp.kind == DFH_KERNEL_SE || (hp.kind == DFH_KERNEL_MATERN && hp.p <= 2)) &&
                            hp.kc >= 8 && hp.kc <= 32 && hp.kc % 8 == 0 && kd.P % 2 == 0 && hp.poff % 2 == 0 &&
                            32 * ldk + 64 < (1LL << 31) && (n1 + 127) / 128 <= 65535 &&
                            (reinterpret_cast<uintptr_t>(Xp1) & 15) == 0 && (reinterpret_cast<uintptr_t>(Xp2) & 15) == 0;
      static const bool mu_fused = []() { const char* e = getenv("DFH_KM_FUSED_MEAN"); return e ? atoi(e) != 0 : true; }();
      if (strip_ok && mu_fused && mu_alpha && mu_out && mu_done) {
        a.mu_nblk = (int)((n2 + KM_MU_BLOCK - 1) / KM_MU_BLOCK);
        DFH_TRY(scratch_get(ctx, SCR_MUPART, (size_t)n1 * a.mu_nblk * 8, (void**)&a.mu_part));
        a.mu_alpha = mu_alpha; a.mu_out = mu_out;
        *mu_done = true;
      }
      if (strip_ok) {
        if (hp.kind == DFH_KERNEL_SE) {
          for (int i = 0; i < 12; ++i) a.ec.c[i] *= hp.scale_c;      // scale folded into the exp polynomial
          return launch_strip<DFH_KERNEL_SE, 0>(ctx, a, hp.kc / 4);
        }
        if (hp.p == 0) return launch_strip<DFH_KERNEL_MATERN, 0>(ctx, a, hp.kc / 4);
        if (hp.p == 1) return launch_strip<DFH_KERNEL_MATERN, 1>(ctx, a, hp.kc / 4);
        return launch_strip<DFH_KERNEL_MATERN, 2>(ctx, a, hp.kc / 4);
      }
      dim3 grid((unsigned)((n2 + 63) / 64), (unsigned)((n1 + 63) / 64));
      hipLaunchKernelGGL((kernmat_sym_kernel<64, 16, 32, 7, false>), grid, dim3(256), smem_bytes(64, 16, 32), ctx->stream, a);
    }
    DFH_LAUNCH_CHECK();
    return DFH_OK;
  }
  if (multi && symmetric && kd.stationary && !kd.nested && (ldk & 1) == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0 &&
      (n1 + 63) / 64 <= 65535 && kd.P % 2 == 0 && (reinterpret_cast<uintptr_t>(Xp1) & 15) == 0) {
    // symmetric Gram of an additive / product kernel: lower-triangle tiles, parts adjacent and <= 16 columns wide
    static const bool symmulti_on = []() { const char* e = getenv("DFH_KM_SYMMULTI"); return e ? atoi(e) != 0 : true; }();
    bool ok = symmulti_on;
    for (int g = part_lo; g < part_hi && ok; ++g) {
      ok = kd.parts[g].kc <= 16 && kd.parts[g].poff % 2 == 0 &&
           (g == part_lo || kd.parts[g].poff == kd.parts[g - 1].poff + kd.parts[g - 1].kc);
    }
    if (ok) {
      constexpr int KCM = 16, SRM = 32;
      constexpr int oper = 2 * 64 * (KCM + 2), stage = SRM * 66;
      constexpr int smem = ((oper > stage ? oper : stage) + 2 * (KCM / 4) * 64) * 8;
      const int64_t T = (n1 + 63) / 64;
      hipLaunchKernelGGL((kernmat_symmulti_kernel<KCM, SRM, 5>), dim3((unsigned)(T * (T + 1) / 2)), dim3(256), smem,
                         ctx->stream, a);
      DFH_LAUNCH_CHECK();
      return DFH_OK;
    }
  }
  const int64_t rows_per_launch = 65535LL * KM_BM;
  for (int64_t r0 = 0; r0 < n1; r0 += rows_per_launch) {
    const int64_t rr = n1 - r0 < rows_per_launch ? n1 - r0 : rows_per_launch;
    KmArgs b = a;
    b.Xp1 = Xp1 + r0 * kd.P; b.Np1 = Np1 + r0 * kd.n_parts; b.n1 = (int)rr; b.K = K + r0 * ldk;
    if (r0 != 0) b.symmetric = 0;    // only reachable for n1 > 8M rows; diagonal handled in slab 0
    if (multi) {
      dim3 grid((unsigned)((n2 + 63) / 64), (unsigned)((rr + KM_BM - 1) / KM_BM));
      if (kd.nested) hipLaunchKernelGGL((kernmat_kernel<2, true, true, true>), grid, dim3(256), SM2, ctx->stream, b);
      else if (kd.stationary) hipLaunchKernelGGL((kernmat_kernel<2, true>), grid, dim3(256), SM2, ctx->stream, b);
      else hipLaunchKernelGGL((kernmat_kernel<2, true, true>), grid, dim3(256), SM2, ctx->stream, b);
    } else {
      dim3 grid((unsigned)((n2 + 127) / 128), (unsigned)((rr + KM_BM - 1) / KM_BM));
      hipLaunchKernelGGL((kernmat_kernel<4, false>), grid, dim3(256), SM4, ctx->stream, b);
    }
    DFH_LAUNCH_CHECK();
  }
  return DFH_OK;
}
