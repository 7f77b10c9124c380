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
  DFH_TRY(scratch_get(ctx, SCR_VEC2, MT_N * sizeof(uint32_t), &p));
  uint32_t* d_state = static_cast<uint32_t*>(p);
  DFH_HIP(hipMemcpyAsync(d_state, key, MT_N * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  const int64_t total_words = 2 * m * d;
  const int64_t keep_lo = 2 * row_begin * d, keep_hi = keep_lo + 2 * count;
  const int64_t chunk_words = total_words < MT_CHUNK_WORDS ? total_words : MT_CHUNK_WORDS;   // even
  const int64_t raw_words = 2 * count < chunk_words ? 2 * count : chunk_words;
  DFH_TRY(scratch_get(ctx, SCR_TMP, size_t(raw_words > 0 ? raw_words : 2) * sizeof(uint32_t), &p));
  uint32_t* d_raw = static_cast<uint32_t*>(p);
  int cur = *pos;
  for (int64_t w0 = 0; w0 < total_words; w0 += chunk_words) {
    const int64_t nw = total_words - w0 < chunk_words ? total_words - w0 : chunk_words;
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
