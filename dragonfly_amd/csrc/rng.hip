// Candidate generation on the device (reference row a16).
//
// The reference draws the candidates of every random-search acquisition with
//   np.random.random((max_evals, dim))                      dragonfly/utils/oper_utils.py:62
//   pts * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]      dragonfly/utils/general_utils.py:25-27
// from the GLOBAL NumPy state (MT19937).  Two streams are provided, both bit-identical to NumPy
// (checked word for word against numpy.random in tests/test_gpu_rng.py):
//
//  * MT19937 continues the caller's legacy state in place, so a seeded run draws exactly the
//    reference's candidates without generating them on the host or copying m x d doubles over
//    PCIe.  The recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) only exposes 227-way
//    parallelism, so ONE workgroup walks the state (three barrier-separated phases per 624-word
//    block, state in LDS) and streams the raw words to HBM; tempering, the 53-bit double
//    construction and the map to the box are a second, fully parallel, HBM-bound kernel.
//  * Philox4x64-10 (numpy.random.Philox) is counter based: every thread computes its own block
//    of four words, no sequential part at all.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_THREADS = 256;
constexpr int64_t MT_CHUNK_WORDS = int64_t(1) << 26;   // raw words per pass (256 MiB of scratch)

__device__ __forceinline__ uint32_t mt_mix(uint32_t cur, uint32_t nxt, uint32_t far) {
  const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// One workgroup.  state[624] (device) is the current block, `pos` the next unread word in it.
// Walks n_words raw (untempered) words of the stream, keeps those with index in [keep_lo, keep_hi)
// in raw[0 .. keep_hi - keep_lo), and leaves the last block in state[].
__global__ __launch_bounds__(MT_THREADS) void mt19937_stream_kernel(uint32_t* __restrict__ state, int pos,
                                                                      int64_t n_words, int64_t keep_lo,
                                                                      int64_t keep_hi,
                                                                      uint32_t* __restrict__ raw) {
  __shared__ uint32_t buf[2][MT_N];
  const int t = threadIdx.x;
  constexpr int LAG = MT_N - MT_M;   // 227
  uint32_t* cur = buf[0];
  uint32_t* nxt = buf[1];
  for (int k = t; k < MT_N; k += MT_THREADS) cur[k] = state[k];
  __syncthreads();
  int64_t done = 0;
  {  // what is left of the current block
    const int64_t left = MT_N - pos;
    const int64_t take = left < n_words ? left : n_words;
    for (int64_t k = t; k < take; k += MT_THREADS)
      if (k >= keep_lo && k < keep_hi) raw[k - keep_lo] = cur[pos + k];
    done = take;
  }
  while (done < n_words) {
    // words [lo, hi) of this block are kept (block-local indices; empty when lo >= hi)
    const int64_t room = n_words - done;
    const int64_t lo = keep_lo - done, hi = (keep_hi < n_words ? keep_hi : n_words) - done;
    const int64_t shift = done - keep_lo;   // raw index of this block's word 0 (may be negative)
    // The new block goes to the other LDS buffer, so a phase never overwrites what a neighbour
    // still has to read and one barrier per phase is enough:
    //   phase A: k in [0, 227)    old[k], old[k+1], old[k+397]
    //   phase B: k in [227, 454)  old[k], old[k+1], new[k-227]   (phase A)
    //   phase C: k in [454, 624)  old[k], old[k+1] (new[0] for k = 623), new[k-227]   (phase B)
    if (t < LAG) {
      const uint32_t v = mt_mix(cur[t], cur[t + 1], cur[t + MT_M]);
      nxt[t] = v;
      if (t >= lo && t < hi) raw[shift + t] = v;
    }
    __syncthreads();
    if (t < LAG) {
      const int k = LAG + t;
      const uint32_t v = mt_mix(cur[k], cur[k + 1], nxt[t]);
      nxt[k] = v;
      if (k >= lo && k < hi) raw[shift + k] = v;
    }
    __syncthreads();
    if (t < MT_N - 2 * LAG) {
      const int k = 2 * LAG + t;
      const uint32_t v = mt_mix(cur[k], k + 1 == MT_N ? nxt[0] : cur[k + 1], nxt[k - LAG]);
      nxt[k] = v;
      if (k >= lo && k < hi) raw[shift + k] = v;
    }
    __syncthreads();
    uint32_t* swap = cur; cur = nxt; nxt = swap;
    done += room < MT_N ? room : MT_N;
  }
  for (int k = t; k < MT_N; k += MT_THREADS) state[k] = cur[k];
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// out[i] = lo[j] + u_i * width[j], j = (first + i) % d ; box == nullptr: out[i] = u_i.
// box holds width[0..d) then lo[d..2d).  Multiply and add round separately (-ffp-contract=off),
// as the two NumPy operations of map_to_bounds do.
__device__ __forceinline__ double to_box(double u, const double* __restrict__ box, int64_t d, int64_t idx) {
  if (box == nullptr) return u;
  const int64_t j = idx % d;
  const double scaled = u * box[j];
  return scaled + box[d + j];
}

// legacy NumPy double: (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53 from two consecutive words
__global__ __launch_bounds__(256) void mt19937_uniform_kernel(const uint2* __restrict__ raw, int64_t count,
                                                                int64_t first, int64_t d,
                                                                const double* __restrict__ box,
                                                                double* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
    const uint2 w = raw[i];
    const double a = double(mt_temper(w.x) >> 5), b = double(mt_temper(w.y) >> 6);
    const double u = (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    out[i] = to_box(u, box, d, first + i);
  }
}

// ---------------------------------------------------------------------------------------
// Philox4x64-10
// ---------------------------------------------------------------------------------------
struct Philox4 { uint64_t v[4]; };
struct PhiloxKey { uint64_t k0, k1; };

__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return uint64_t((unsigned __int128)a * b >> 64);
#endif
}

__host__ __device__ inline Philox4 philox4x64_10(Philox4 c, PhiloxKey key) {
  const uint64_t M0 = 0xD2E7470EE14C6C93ull, M1 = 0xCA5A826395121157ull;
  const uint64_t W0 = 0x9E3779B97F4A7C15ull, W1 = 0xBB67AE8584CAA73Bull;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r) { key.k0 += W0; key.k1 += W1; }
    const uint64_t hi0 = mulhi64(M0, c.v[0]), lo0 = M0 * c.v[0];
    const uint64_t hi1 = mulhi64(M1, c.v[2]), lo1 = M1 * c.v[2];
    Philox4 n;
    n.v[0] = hi1 ^ c.v[1] ^ key.k0;
    n.v[1] = lo1;
    n.v[2] = hi0 ^ c.v[3] ^ key.k1;
    n.v[3] = lo0;
    c = n;
  }
  return c;
}

__host__ __device__ inline Philox4 counter_add(Philox4 c, uint64_t inc) {
  uint64_t carry = inc;
  for (int i = 0; i < 4 && carry; ++i) {
    const uint64_t s = c.v[i] + carry;
    carry = s < carry ? 1 : 0;
    c.v[i] = s;
  }
  return c;
}

// Thread b computes block b: counter + 1 + b (NumPy increments before it generates), whose four
// words are the doubles [lead + 4b, lead + 4b + 4) of the stream.  The first `lead` doubles come
// from the words NumPy still holds in its buffer.  Doubles [keep_lo, keep_hi) are written, to
// out[0 .. keep_hi - keep_lo); only blocks [b_lo, b_hi) overlap them.
__global__ __launch_bounds__(256) void philox_uniform_kernel(Philox4 counter, PhiloxKey key, Philox4 held,
                                                               int held_pos, int lead, int64_t b_lo,
                                                               int64_t b_hi, int64_t keep_lo, int64_t keep_hi,
                                                               int64_t d, const double* __restrict__ box,
                                                               double* __restrict__ out) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  const int64_t tid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (tid < lead && tid >= keep_lo && tid < keep_hi)
    out[tid - keep_lo] = to_box(double(held.v[held_pos + tid] >> 11) * (1.0 / 9007199254740992.0), box, d, tid);
  for (int64_t b = b_lo + tid; b < b_hi; b += stride) {
    const Philox4 w = philox4x64_10(counter_add(counter, uint64_t(b) + 1), key);
    const int64_t base = lead + 4 * b;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (base + q >= keep_lo && base + q < keep_hi)
        out[base + q - keep_lo] = to_box(double(w.v[q] >> 11) * (1.0 / 9007199254740992.0), box, d, base + q);
  }
}


// ---------------------------------------------------------------------------------------------
// np.random.normal from the legacy MT19937 state, on the device, bit for bit
// ---------------------------------------------------------------------------------------------
// draw_gaussian_samples (dragonfly/utils/general_utils.py:230) takes its standard normals from
// np.random.normal(size=(m, 1)): NumPy's legacy_gauss, the Marsaglia polar method with rejection,
//     do { x1 = 2 u - 1; x2 = 2 u' - 1; r2 = x1 x1 + x2 x2; } while (r2 >= 1 || r2 == 0);
//     f = sqrt(-2 log(r2) / r2);   return f x2 now, f x1 on the next call (has_gauss / gauss)
// on the MT19937 double stream.  Here: the uniforms of P attempted pairs come from the stream
// walker above; acceptance flags, an exclusive prefix sum and a compaction put the k-th accepted
// pair's two normals at out[base + 2k], out[base + 2k + 1]; the generator state handed back is the
// state after the last pair NumPy would have consumed (rejected attempts included), has_gauss and
// the cached second normal included.
//
// Every operation but one is an IEEE-exact +, *, /, sqrt on both sides.  The one is log(): glibc's
// (what NumPy calls) is within 0.519 ulp, i.e. it returns the correctly rounded value unless the
// true log lies within 0.019 ulp of a rounding boundary.  The device evaluates log(r2) in
// double-double (error ~1e-30), rounds once, and flags the ~4 % of pairs within 0.025 ulp of a
// boundary; for those -- and only those -- the host calls the C library's log() on the same r2 and
// patches the two normals.  Result: word for word np.random.normal (tests/test_gpu_rng.py), with
// O(0.04 m) doubles crossing PCIe instead of m.
struct dd { double hi, lo; };
__host__ __device__ inline dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
__host__ __device__ inline dd quick_two_sum(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
__host__ __device__ inline dd two_prod(double a, double b) { const double p = a * b; return {p, fma(a, b, -p)}; }
__host__ __device__ inline dd dd_add(dd a, dd b) {
  dd s = two_sum(a.hi, b.hi); const dd t = two_sum(a.lo, b.lo);
  s.lo += t.hi; s = quick_two_sum(s.hi, s.lo); s.lo += t.lo; return quick_two_sum(s.hi, s.lo);
}
__host__ __device__ inline dd dd_add_d(dd a, double b) { dd s = two_sum(a.hi, b); s.lo += a.lo; return quick_two_sum(s.hi, s.lo); }
__host__ __device__ inline dd dd_mul(dd a, dd b) { dd p = two_prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return quick_two_sum(p.hi, p.lo); }
__host__ __device__ inline dd dd_mul_d(dd a, double b) { dd p = two_prod(a.hi, b); p.lo += a.lo * b; return quick_two_sum(p.hi, p.lo); }
__host__ __device__ inline dd dd_div(dd a, dd b) {
  const double q1 = a.hi / b.hi;
  dd r = dd_add(a, dd_mul_d(b, -q1));
  const double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul_d(b, -q2));
  const double q3 = r.hi / b.hi;
  return dd_add_d(quick_two_sum(q1, q2), q3);
}

// log(x), 0 < x < 1 finite normal, in double-double: x = 2^e m, m in [sqrt(1/2), sqrt(2));
// log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716:
//   2 s (1 + s^2/3 + s^4/5 + s^6 (1/7 + s^2/9 + ...)) -- the first two terms of the bracket in
// double-double, the tail (<= 3.6e-6, needed to 3e-15 relative) in double.
__host__ __device__ inline dd log_dd(double x) {
  int e;
  double m = frexp(x, &e);                 // m in [0.5, 1)
  if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
  const dd num = two_sum(m, -1.0), den = two_sum(m, 1.0);
  const dd s = dd_div(num, den);
  const dd s2 = dd_mul(s, s);
  const double z = s2.hi;
  double tail = 1.0 / 43.0;
  for (int k = 41; k >= 7; k -= 2) tail = fma(tail, z, 1.0 / (double)k);      // 1/7 + z/9 + ... + z^18/43
  const dd third = {0.33333333333333331, 1.8503717077085941e-17};
  const dd fifth = {0.20000000000000001, -1.1102230246251566e-17};
  dd br = dd_mul(s2, dd_add(third, dd_mul(s2, fifth)));                       // s^2/3 + s^4/5
  br = dd_add_d(br, (z * z * z) * tail);                                      // + s^6 (...)
  br = dd_add_d(br, 1.0);
  dd r = dd_mul(dd_mul_d(s, 2.0), br);
  const dd ln2 = {0.69314718055994529, 2.3190468138462996e-17};
  return dd_add(dd_mul_d(ln2, (double)e), r);
}

struct NormalPair { double x1, x2, r2; int ok; };
__host__ __device__ inline NormalPair polar_pair(double u0, double u1) {
  NormalPair p;
  p.x1 = 2.0 * u0 - 1.0;
  p.x2 = 2.0 * u1 - 1.0;
  p.r2 = p.x1 * p.x1 + p.x2 * p.x2;
  p.ok = !(p.r2 >= 1.0 || p.r2 == 0.0);
  return p;
}

// flags[i] = pair i accepted (as uint32 for the scan)
__global__ __launch_bounds__(256) void normal_flags_kernel(const double2_t* __restrict__ U, int64_t P,
                                                            uint32_t* __restrict__ flags) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < P; i += stride) {
    const double2_t u = U[i];
    flags[i] = (uint32_t)polar_pair(u.x, u.y).ok;
  }
}

// three-kernel exclusive scan of uint32 flags into int64 positions: per-block sums, scan of the
// block sums by one block, per-block rescan with the offset
constexpr int SCAN_ITEMS = 2048;     // items per block (256 threads x 8)
__global__ __launch_bounds__(256) void scan_block_sums(const uint32_t* __restrict__ f, int64_t n, int64_t* __restrict__ sums) {
  __shared__ int64_t sm[256];
  const int64_t base = int64_t(blockIdx.x) * SCAN_ITEMS;
  int64_t s = 0;
  for (int k = 0; k < 8; ++k) { const int64_t i = base + threadIdx.x * 8 + k; if (i < n) s += f[i]; }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(256) void scan_sums_inplace(int64_t* __restrict__ sums, int64_t nb, int64_t* __restrict__ total) {
  // one block, sequential over chunks of 256 (nb is a few thousand at most)
  __shared__ int64_t sm[256];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < nb; c0 += 256) {
    const int64_t i = c0 + threadIdx.x;
    const int64_t v = i < nb ? sums[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int64_t t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) sums[i] = carry + sm[threadIdx.x] - v;      // exclusive
    __syncthreads();
    if (threadIdx.x == 255) carry += sm[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// Emits the normals of the accepted pairs with rank < pairs_needed.  out[base + 2k] = f x2,
// out[base + 2k + 1] = f x1 (if inside m, else it is the cached gaussian).  last_idx receives the
// attempt index of the pair with rank pairs_needed - 1.  Hard log() cases are appended to
// hard_k / hard_u (rank, the pair's two uniforms).
__global__ __launch_bounds__(256) void normal_emit_kernel(const double2_t* __restrict__ U, int64_t P,
                                                           const uint32_t* __restrict__ flags,
                                                           const int64_t* __restrict__ block_off,
                                                           int64_t pairs_needed, int64_t base, int64_t m,
                                                           double* __restrict__ out, double* __restrict__ cached,
                                                           int64_t* __restrict__ last_idx,
                                                           unsigned long long* __restrict__ hard_count,
                                                           int64_t hard_cap, int64_t* __restrict__ hard_k,
                                                           double2_t* __restrict__ hard_u) {
  __shared__ int64_t sm[256];
  const int64_t b0 = int64_t(blockIdx.x) * SCAN_ITEMS;
  // exclusive scan of this block's 2048 flags: per-thread 8 items
  uint32_t f[8];
  int64_t s = 0;
  for (int k = 0; k < 8; ++k) { const int64_t i = b0 + threadIdx.x * 8 + k; f[k] = i < P ? flags[i] : 0u; s += f[k]; }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int64_t t = (int)threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t rank = block_off[blockIdx.x] + sm[threadIdx.x] - s;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = b0 + threadIdx.x * 8 + k;
    if (i >= P || !f[k]) continue;
    const int64_t kk = rank++;
    if (kk >= pairs_needed) continue;
    if (kk == pairs_needed - 1) *last_idx = i;
    const double2_t u = U[i];
    const NormalPair p = polar_pair(u.x, u.y);
    const dd L = log_dd(p.r2);
    // L.hi is the correctly rounded log unless L sits within 0.025 ulp of the midpoint between two
    // doubles: |L.lo| close to half the spacing of doubles at L.hi
    const double ulp = ldexp(1.0, ilogb(L.hi) - 52);
    const double t = fabs(L.lo) / ulp;
    if (t > 0.475) {
      const unsigned long long slot = atomicAdd(hard_count, 1ull);
      if ((int64_t)slot < hard_cap) { hard_k[slot] = kk; hard_u[slot] = u; }
    }
    const double fct = sqrt(-2.0 * L.hi / p.r2);
    const int64_t o = base + 2 * kk;
    out[o] = fct * p.x2;
    if (o + 1 < m) out[o + 1] = fct * p.x1;
    else *cached = fct * p.x1;
  }
}

__global__ void normal_patch_kernel(const int64_t* __restrict__ k, const double2_t* __restrict__ z, int64_t count,
                                    int64_t base, int64_t m, double* __restrict__ out, double* __restrict__ cached) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t o = base + 2 * k[i];
  out[o] = z[i].x;
  if (o + 1 < m) out[o + 1] = z[i].y;
  else *cached = z[i].y;
}

// width[d] then lo[d] on the device (SCR_VEC3), from the caller's host bounds[d][2]
int upload_box(dfh_ctx* ctx, const double* bounds, int64_t d, const double** d_box) {
  *d_box = nullptr;
  if (bounds == nullptr) return DFH_OK;
  DFH_ARG(!is_device_ptr(bounds));
  std::vector<double> box(size_t(2 * d));
  for (int64_t j = 0; j < d; ++j) {
    box[size_t(j)] = bounds[2 * j + 1] - bounds[2 * j];
    box[size_t(d + j)] = bounds[2 * j];
  }
  void* p = nullptr;
  DFH_TRY(scratch_get(ctx, SCR_VEC3, box.size() * sizeof(double), &p));
  DFH_HIP(hipMemcpyAsync(p, box.data(), box.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));   // `box` is a stack-lifetime host buffer
  *d_box = static_cast<const double*>(p);
  return DFH_OK;
}

int grid_for(const dfh_ctx* ctx, int64_t items, int threads) {
  const int64_t want = (items + threads - 1) / threads;
  const int64_t cap = int64_t(ctx->n_cu) * 16;
  return int(want < 1 ? 1 : (want < cap ? want : cap));
}

}  // namespace

extern "C" int dfh_rand_mt19937_uniform(dfh_ctx* ctx, uint32_t* key, int32_t* pos, int64_t m, int64_t d,
                                        int64_t row_begin, int64_t row_count, const double* bounds,
                                        double* out) {
  DFH_ARG(ctx != nullptr && key != nullptr && pos != nullptr);
  DFH_ARG(m >= 0 && d >= 1 && *pos >= 0 && *pos <= MT_N);
  DFH_ARG(row_begin >= 0 && row_count >= 0 && row_begin + row_count <= m);
  DFH_ARG(out != nullptr || row_count == 0);
  DFH_ARG(!is_device_ptr(key) && !is_device_ptr(pos));
  DFH_HIP(hipSetDevice(ctx->device));
  if (m * d == 0) return DFH_OK;
  const int64_t count = row_count * d;        // doubles kept
  const double* d_box = nullptr;
  DFH_TRY(upload_box(ctx, bounds, d, &d_box));
  const bool out_on_device = count == 0 || is_device_ptr(out);
  void* p = nullptr;
  double* d_out = out;
  if (!out_on_device) {
    DFH_TRY(scratch_get(ctx, SCR_OUT, size_t(count) * sizeof(double), &p));
    d_out = static_cast<double*>(p);
  }
  const int64_t total_words = 2 * m * d;
  const int64_t keep_lo = 2 * row_begin * d, keep_hi = keep_lo + 2 * count;
  // The words before and after the shard are not walked when there are many of them: the state
  // jumps over them on the host (mtjump.hip; a few ms whatever the distance, the walk is 0.36 us per
  // 624 words).  DFH_MT_JUMP_MIN_WORDS moves the threshold (0: never jump).
  static const int64_t jump_min = getenv("DFH_MT_JUMP_MIN_WORDS") ? atoll(getenv("DFH_MT_JUMP_MIN_WORDS")) : (int64_t(1) << 23);
  const bool jump_front = jump_min > 0 && keep_lo >= jump_min;
  const bool jump_back = jump_min > 0 && total_words - keep_hi >= jump_min;
  const int64_t walk_lo = jump_front ? keep_lo : 0, walk_hi = jump_back ? keep_hi : total_words;
  uint32_t start_key[MT_N];
  memcpy(start_key, key, sizeof(start_key));
  int32_t start_pos = *pos;
  if (jump_front) DFH_TRY(mt19937_advance_host(start_key, &start_pos, keep_lo));
  if (walk_hi == walk_lo) {                       // nothing to generate here: the state only
    if (jump_back) DFH_TRY(mt19937_advance_host(start_key, &start_pos, total_words - keep_hi));
    memcpy(key, start_key, sizeof(start_key));
    *pos = start_pos;
    return DFH_OK;
  }
  DFH_TRY(scratch_get(ctx, SCR_VEC2, MT_N * sizeof(uint32_t), &p));
  uint32_t* d_state = static_cast<uint32_t*>(p);
  DFH_HIP(hipMemcpyAsync(d_state, start_key, MT_N * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));     // start_key lives on this frame
  const int64_t span_words = walk_hi - walk_lo;
  const int64_t chunk_words = span_words < MT_CHUNK_WORDS ? span_words : MT_CHUNK_WORDS;   // even
  const int64_t raw_words = 2 * count < chunk_words ? 2 * count : chunk_words;
  DFH_TRY(scratch_get(ctx, SCR_TMP, size_t(raw_words > 0 ? raw_words : 2) * sizeof(uint32_t), &p));
  uint32_t* d_raw = static_cast<uint32_t*>(p);
  int cur = start_pos;
  for (int64_t w0 = walk_lo; w0 < walk_hi; w0 += chunk_words) {
    const int64_t nw = walk_hi - w0 < chunk_words ? walk_hi - w0 : chunk_words;
    // the part of this chunk that belongs to the kept rows, chunk-local word indices
    const int64_t lo = (keep_lo > w0 ? keep_lo : w0) - w0;
    const int64_t hi = (keep_hi < w0 + nw ? keep_hi : w0 + nw) - w0;
    mt19937_stream_kernel<<<1, MT_THREADS, 0, ctx->stream>>>(d_state, cur, nw, lo, hi > lo ? hi : lo, d_raw);
    DFH_LAUNCH_CHECK();
    if (hi > lo) {
      const int64_t nd = (hi - lo) / 2, first = (w0 + lo) / 2;     // global index of the first double
      mt19937_uniform_kernel<<<grid_for(ctx, nd, 256), 256, 0, ctx->stream>>>(
          reinterpret_cast<const uint2*>(d_raw), nd, first, d, d_box, d_out + (first - row_begin * d));
      DFH_LAUNCH_CHECK();
    }
    const int64_t left = MT_N - cur;
    if (nw <= left) {
      cur += int(nw);
    } else {
      const int64_t rem = nw - left, blocks = (rem + MT_N - 1) / MT_N;
      cur = int(rem - (blocks - 1) * MT_N);
    }
  }
  DFH_HIP(hipMemcpyAsync(key, d_state, MT_N * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  *pos = cur;
  if (jump_back) DFH_TRY(mt19937_advance_host(key, pos, total_words - keep_hi));
  if (!out_on_device) DFH_TRY(from_device(ctx, out, d_out, size_t(count) * sizeof(double)));
  return DFH_OK;
}

extern "C" int dfh_rand_philox_uniform(dfh_ctx* ctx, const uint64_t* key, uint64_t* counter, uint64_t* buffer,
                                       int32_t* buffer_pos, int64_t m, int64_t d, int64_t row_begin,
                                       int64_t row_count, const double* bounds, double* out) {
  DFH_ARG(ctx != nullptr && key != nullptr && counter != nullptr && buffer != nullptr);
  DFH_ARG(buffer_pos != nullptr && *buffer_pos >= 0 && *buffer_pos <= 4 && m >= 0 && d >= 1);
  DFH_ARG(row_begin >= 0 && row_count >= 0 && row_begin + row_count <= m);
  DFH_ARG(out != nullptr || row_count == 0);
  DFH_ARG(!is_device_ptr(key) && !is_device_ptr(counter) && !is_device_ptr(buffer));
  DFH_HIP(hipSetDevice(ctx->device));
  const int64_t total = m * d;
  if (total == 0) return DFH_OK;
  const int64_t keep_lo = row_begin * d, keep = row_count * d;
  const double* d_box = nullptr;
  DFH_TRY(upload_box(ctx, bounds, d, &d_box));
  const bool out_on_device = keep == 0 || is_device_ptr(out);
  double* d_out = out;
  if (!out_on_device) {
    void* p = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_OUT, size_t(keep) * sizeof(double), &p));
    d_out = static_cast<double*>(p);
  }
  Philox4 ctr, held;
  for (int i = 0; i < 4; ++i) { ctr.v[i] = counter[i]; held.v[i] = buffer[i]; }
  const PhiloxKey pk{key[0], key[1]};
  const int held_pos = *buffer_pos;
  const int64_t avail = 4 - held_pos;
  const int lead = int(total < avail ? total : avail);
  const int64_t n_blocks = (total - lead + 3) / 4;
  if (keep > 0) {
    // counter based: only the blocks that overlap the kept doubles are computed
    const int64_t b_lo = keep_lo > lead ? (keep_lo - lead) / 4 : 0;
    const int64_t b_hi = keep_lo + keep > lead ? (keep_lo + keep - lead + 3) / 4 : 0;
    const int64_t items = (b_hi - b_lo) > lead ? (b_hi - b_lo) : lead;
    philox_uniform_kernel<<<grid_for(ctx, items, 256), 256, 0, ctx->stream>>>(
        ctr, pk, held, held_pos, lead, b_lo, b_hi, keep_lo, keep_lo + keep, d, d_box, d_out);
    DFH_LAUNCH_CHECK();
  }
  // the state NumPy would be left in: counter of the last block, its four words, words used
  if (n_blocks == 0) {
    *buffer_pos = held_pos + lead;
  } else {
    ctr = counter_add(ctr, uint64_t(n_blocks));
    const Philox4 last = philox4x64_10(ctr, pk);
    for (int i = 0; i < 4; ++i) { counter[i] = ctr.v[i]; buffer[i] = last.v[i]; }
    *buffer_pos = int(total - lead - 4 * (n_blocks - 1));
  }
  if (!out_on_device) DFH_TRY(from_device(ctx, out, d_out, size_t(keep) * sizeof(double)));
  else DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

extern "C" int dfh_rand_mt19937_normal(dfh_ctx* ctx, uint32_t* key, int32_t* pos, int32_t* has_gauss,
                                       double* gauss, int64_t m, double* out) {
  DFH_ARG(ctx != nullptr && key != nullptr && pos != nullptr && has_gauss != nullptr && gauss != nullptr);
  DFH_ARG(m >= 0 && *pos >= 0 && *pos <= MT_N && (out != nullptr || m == 0));
  DFH_ARG(!is_device_ptr(key) && !is_device_ptr(pos));
  DFH_HIP(hipSetDevice(ctx->device));
  if (m == 0) return DFH_OK;
  const bool out_on_device = is_device_ptr(out);
  void* p = nullptr;
  double* d_out = out;
  if (!out_on_device) {
    DFH_TRY(scratch_get(ctx, SCR_OUT2, size_t(m) * sizeof(double), &p));
    d_out = static_cast<double*>(p);
  }
  const int64_t base = *has_gauss ? 1 : 0;
  const int64_t need = m - base;                     // normals that come from new pairs
  const int64_t pairs_needed = (need + 1) / 2;
  if (base) DFH_HIP(hipMemcpyAsync(d_out, gauss, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (pairs_needed == 0) {                           // m == 1 served from the cache
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    *has_gauss = 0; *gauss = 0.0;
    if (!out_on_device) DFH_TRY(from_device(ctx, out, d_out, size_t(m) * sizeof(double)));
    return DFH_OK;
  }
  // small scalars on the device: [0] cached gaussian (double), [1] last_idx, [2] hard count, [3] accepted total
  DFH_TRY(scratch_get(ctx, SCR_RED, 256, &p));
  double* d_cached = static_cast<double*>(p);
  int64_t* d_last = reinterpret_cast<int64_t*>(d_cached + 1);
  unsigned long long* d_hard = reinterpret_cast<unsigned long long*>(d_cached + 2);
  int64_t* d_total = reinterpret_cast<int64_t*>(d_cached + 3);
  int64_t P = (int64_t)((double)pairs_needed / 0.75) + 64;      // acceptance probability pi/4 = 0.785
  for (int attempt = 0; attempt < 8; ++attempt, P = P * 3 / 2) {
    std::vector<uint32_t> key_try(key, key + MT_N);
    int32_t pos_try = *pos;
    double* dU = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_XS, size_t(P) * 16, (void**)&dU));
    DFH_TRY(dfh_rand_mt19937_uniform(ctx, key_try.data(), &pos_try, P, 2, 0, P, nullptr, dU));
    const int64_t nb = (P + SCAN_ITEMS - 1) / SCAN_ITEMS;
    DFH_TRY(scratch_get(ctx, SCR_XS2, size_t(P) * 4 + size_t(nb) * 8 + 64, &p));
    int64_t* d_sums = static_cast<int64_t*>(p);
    uint32_t* d_flags = reinterpret_cast<uint32_t*>(d_sums + nb + 1);
    normal_flags_kernel<<<grid_for(ctx, P, 256), 256, 0, ctx->stream>>>(reinterpret_cast<const double2_t*>(dU), P, d_flags);
    DFH_LAUNCH_CHECK();
    scan_block_sums<<<(unsigned)nb, 256, 0, ctx->stream>>>(d_flags, P, d_sums);
    DFH_LAUNCH_CHECK();
    scan_sums_inplace<<<1, 256, 0, ctx->stream>>>(d_sums, nb, d_total);
    DFH_LAUNCH_CHECK();
    int64_t accepted = 0;
    DFH_HIP(hipMemcpyAsync(&accepted, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    if (accepted < pairs_needed) continue;           // too few accepted pairs (practically never): more attempts
    const int64_t hard_cap = pairs_needed / 8 + 1024;
    char* hb = nullptr;
    DFH_TRY(scratch_get(ctx, SCR_AUG, size_t(hard_cap) * 24 + 64, (void**)&hb));
    int64_t* d_hk = reinterpret_cast<int64_t*>(hb);
    double2_t* d_hu = reinterpret_cast<double2_t*>(hb + ((size_t(hard_cap) * 8 + 15) / 16) * 16);
    DFH_HIP(hipMemsetAsync(d_hard, 0, 8, ctx->stream));
    normal_emit_kernel<<<(unsigned)nb, 256, 0, ctx->stream>>>(reinterpret_cast<const double2_t*>(dU), P, d_flags, d_sums,
                                                              pairs_needed, base, m, d_out, d_cached, d_last, d_hard,
                                                              hard_cap, d_hk, d_hu);
    DFH_LAUNCH_CHECK();
    struct { double cached; int64_t last; unsigned long long hard; } h;
    DFH_HIP(hipMemcpyAsync(&h, d_cached, 24, hipMemcpyDeviceToHost, ctx->stream));
    DFH_HIP(hipStreamSynchronize(ctx->stream));
    DFH_ARG((int64_t)h.hard <= hard_cap);            // 4 % expected against a 12.5 % reserve
    if (h.hard > 0) {
      // the C library's log() decides the pairs whose log is within 0.025 ulp of a rounding boundary
      std::vector<int64_t> hk(h.hard);
      std::vector<double2_t> hu(h.hard);
      DFH_HIP(hipMemcpyAsync(hk.data(), d_hk, h.hard * 8, hipMemcpyDeviceToHost, ctx->stream));
      DFH_HIP(hipMemcpyAsync(hu.data(), d_hu, h.hard * 16, hipMemcpyDeviceToHost, ctx->stream));
      DFH_HIP(hipStreamSynchronize(ctx->stream));
      for (size_t i = 0; i < hk.size(); ++i) {
        const NormalPair pr = polar_pair(hu[i].x, hu[i].y);
        volatile double lg = log(pr.r2);              // libm, as NumPy's legacy_gauss
        const double fct = sqrt(-2.0 * lg / pr.r2);
        hu[i].x = fct * pr.x2;
        hu[i].y = fct * pr.x1;
      }
      DFH_HIP(hipMemcpyAsync(d_hk, hk.data(), h.hard * 8, hipMemcpyHostToDevice, ctx->stream));
      DFH_HIP(hipMemcpyAsync(d_hu, hu.data(), h.hard * 16, hipMemcpyHostToDevice, ctx->stream));
      normal_patch_kernel<<<(unsigned)((h.hard + 255) / 256), 256, 0, ctx->stream>>>(d_hk, d_hu, (int64_t)h.hard, base, m,
                                                                                     d_out, d_cached);
      DFH_LAUNCH_CHECK();
      DFH_HIP(hipMemcpyAsync(&h.cached, d_cached, 8, hipMemcpyDeviceToHost, ctx->stream));
      DFH_HIP(hipStreamSynchronize(ctx->stream));
    }
    // the state NumPy is left in: 2 (last + 1) doubles consumed; the second normal of the last pair is
    // cached when an odd number was asked of the pairs
    DFH_TRY(dfh_rand_mt19937_uniform(ctx, key, pos, h.last + 1, 2, 0, 0, nullptr, nullptr));
    if (need % 2 == 1) { *has_gauss = 1; *gauss = h.cached; }
    else { *has_gauss = 0; *gauss = 0.0; }
    if (!out_on_device) DFH_TRY(from_device(ctx, out, d_out, size_t(m) * sizeof(double)));
    else DFH_HIP(hipStreamSynchronize(ctx->stream));
    return DFH_OK;
  }
  dfh_set_error("dfh_rand_mt19937_normal: could not collect %lld accepted pairs", (long long)pairs_needed);
  return DFH_ERR_HIP;
}
