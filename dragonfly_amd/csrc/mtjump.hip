// MT19937 jump-ahead: the generator state n words further down the stream without walking it.
//
// np.random.random((m, d)) (dragonfly/utils/oper_utils.py:62) is ONE stream; a rank that keeps rows
// [lo, hi) of the block needs the state at word 2*lo*d to start from and the state at word 2*m*d to
// hand back.  The words x_k satisfy a linear recurrence over GF(2) (x_{k+624} = x_{k+397} ^
// mix(x_k, x_{k+1})) whose characteristic polynomial phi has degree 19937, so with F the one-word
// shift of the 624-word window, F^J s = g(F) s for g(t) = t^J mod phi(t): thirty-odd polynomial
// squarings to get g, then one Horner pass of 19937 window shifts (Haramoto, Matsumoto, Nishimura,
// Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008).
// phi is not written down here: it is found once per process by Berlekamp-Massey on 2 x 19937
// bits of the sequence and checked (degree 19937).  Everything in this file is host code.
#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int N = 624, M = 397, DEG = 19937;
constexpr int PW = 313;              // 64-bit words of a polynomial of degree <= 19937
using Poly = std::array<uint64_t, PW>;

inline uint32_t mix(uint32_t cur, uint32_t nxt, uint32_t far) {
  const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// The window (x_k, ..., x_{k+623}) as a ring; step() is F.  F is linear on all 624 x 32 bits; the
// low 31 bits of x_k never reach the future.
struct Window {
  uint32_t x[N];
  int head = 0;
  void step() {
    const int i1 = head + 1 < N ? head + 1 : head + 1 - N;
    const int im = head + M < N ? head + M : head + M - N;
    x[head] = mix(x[head], x[i1], x[im]);
    head = i1;
  }
};

inline void xor_bits(uint64_t* dst, uint64_t v, int64_t pos) {       // dst ^= v << pos
  const int sh = int(pos & 63);
  dst[pos >> 6] ^= v << sh;
  if (sh) dst[(pos >> 6) + 1] ^= v >> (64 - sh);
}

inline uint64_t get_bits(const uint64_t* src, int64_t pos, int width) {   // width <= 64 bits from pos
  const int sh = int(pos & 63);
  uint64_t v = src[pos >> 6] >> sh;
  if (sh && sh + width > 64) v |= src[(pos >> 6) + 1] << (64 - sh);
  return width == 64 ? v : (v & ((uint64_t(1) << width) - 1));
}

struct CharPoly {
  bool ok = false;
  std::vector<int> taps;      // exponents j < DEG with phi_j = 1   (t^DEG = sum_j t^j mod phi)
  int chunk = 1;              // bits a reduction step may take at once: DEG - max(taps), at most 64
};

// Berlekamp-Massey over GF(2) on the lowest bit of x_0, x_1, ... of some state in general position.
CharPoly find_char_poly() {
  const int NB = 2 * DEG + 64;
  const int W = NB / 64 + 2;
  Window win;
  win.x[0] = 19650218u;
  for (int i = 1; i < N; ++i) win.x[i] = 1812433253u * (win.x[i - 1] ^ (win.x[i - 1] >> 30)) + uint32_t(i);
  win.step();      // the low bits of the very first word are not part of the recurrence
  std::vector<uint64_t> C(W, 0), B(W, 0), T(W, 0), R(W, 0);
  C[0] = B[0] = 1;
  int L = 0, m = 1;
  for (int n = 0; n < NB; ++n) {
    const uint64_t s = win.x[win.head] & 1u;
    win.step();
    const int used = n / 64 + 1;
    for (int w = used < W - 1 ? used : W - 1; w > 0; --w) R[w] = (R[w] << 1) | (R[w - 1] >> 63);   // R_i = s_{n-i}
    R[0] = (R[0] << 1) | s;
    uint64_t acc = 0;
    for (int w = 0; w <= L / 64; ++w) acc ^= C[w] & R[w];
    if ((__builtin_popcountll(acc) & 1) == 0) { ++m; continue; }
    const bool grow = 2 * L <= n;
    if (grow) T = C;
    const int wsh = m >> 6, bsh = m & 63;                          // C ^= B << m
    for (int w = W - 1 - wsh; w >= 0; --w) {
      if (!B[w]) continue;
      C[w + wsh] ^= B[w] << bsh;
      if (bsh && w + wsh + 1 < W) C[w + wsh + 1] ^= B[w] >> (64 - bsh);
    }
    if (grow) { L = n + 1 - L; B.swap(T); m = 1; } else { ++m; }
  }
  CharPoly cp;
  if (L != DEG) return cp;
  // connection polynomial C: sum_i c_i s_{n-i} = 0  ->  phi(t) = sum_i c_i t^(L-i)
  int top = -1;
  for (int i = 1; i <= L; ++i)
    if ((C[i >> 6] >> (i & 63)) & 1) { cp.taps.push_back(L - i); if (L - i > top) top = L - i; }
  if (cp.taps.empty() || !((C[0]) & 1)) return cp;
  cp.chunk = DEG - top < 64 ? DEG - top : 64;
  cp.ok = true;
  return cp;
}

const CharPoly& char_poly() {
  static std::once_flag once;
  static CharPoly cp;
  std::call_once(once, [] { cp = find_char_poly(); });
  return cp;
}

// tmp: a polynomial of degree < 2*DEG in 2*PW words -> reduced mod phi in place (low PW words)
void reduce(const CharPoly& cp, uint64_t* tmp, int64_t top_bit) {
  int64_t p = top_bit;
  while (p >= DEG) {
    const int64_t lo = p - cp.chunk + 1 > DEG ? p - cp.chunk + 1 : DEG;
    const int width = int(p - lo + 1);
    const uint64_t h = get_bits(tmp, lo, width);
    if (h) {
      xor_bits(tmp, h, lo);                                   // clear
      for (int j : cp.taps) xor_bits(tmp, h, lo - DEG + j);   // t^(lo+i) = sum_j t^(lo+i-DEG+j)
    }
    p = lo - 1;
  }
}

inline uint64_t spread32(uint32_t v) {      // bit i -> bit 2i
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000ffff0000ffffull;
  x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
  x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

// g(t) = t^J mod phi(t)
Poly power_of_t(const CharPoly& cp, int64_t J) {
  Poly g{};
  g[0] = 1;
  std::array<uint64_t, 2 * PW + 2> tmp;
  int nbits = 0;
  while ((J >> nbits) > 0) ++nbits;
  for (int b = nbits - 1; b >= 0; --b) {
    tmp.fill(0);
    for (int w = 0; w < PW; ++w) {                                  // square: spread the bits
      tmp[2 * w] = spread32(uint32_t(g[w]));
      tmp[2 * w + 1] = spread32(uint32_t(g[w] >> 32));
    }
    int64_t top = 2 * int64_t(DEG - 1);
    if ((J >> b) & 1) {                                              // times t
      for (int w = 2 * PW; w > 0; --w) tmp[w] = (tmp[w] << 1) | (tmp[w - 1] >> 63);
      tmp[0] <<= 1;
      ++top;
    }
    reduce(cp, tmp.data(), top);
    for (int w = 0; w < PW; ++w) g[w] = tmp[w];
  }
  return g;
}

const Poly& cached_power(const CharPoly& cp, int64_t J, Poly* local) {
  static std::mutex mu;
  static std::map<int64_t, Poly> cache;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(J);
    if (it != cache.end()) { *local = it->second; return *local; }
  }
  *local = power_of_t(cp, J);
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() >= 32) cache.clear();
  cache[J] = *local;
  return *local;
}

// s <- F^J s (up to the low 31 bits of the window's first word, which the next twist ignores)
void jump_words(const CharPoly& cp, uint32_t* key, int64_t J) {
  if (J <= 0) return;
  Poly gbuf;
  const Poly& g = cached_power(cp, J, &gbuf);
  int deg = DEG - 1;
  while (deg > 0 && !((g[deg >> 6] >> (deg & 63)) & 1)) --deg;
  Window r;
  for (int i = 0; i < N; ++i) r.x[i] = 0;
  for (int i = deg; i >= 0; --i) {                                   // Horner: r = F r + g_i s
    r.step();
    if ((g[i >> 6] >> (i & 63)) & 1) {
      const int h = r.head;
      for (int j = 0; j < N - h; ++j) r.x[h + j] ^= key[j];
      for (int j = N - h; j < N; ++j) r.x[h + j - N] ^= key[j];
    }
  }
  for (int j = 0; j < N; ++j) key[j] = r.x[r.head + j < N ? r.head + j : r.head + j - N];
}

void twist_block(uint32_t* key) {                                    // the next 624 words from these
  Window w;
  for (int i = 0; i < N; ++i) w.x[i] = key[i];
  for (int i = 0; i < N; ++i) w.step();
  for (int i = 0; i < N; ++i) key[i] = w.x[i];                       // head is back at 0
}

}  // namespace

// (key, pos) after n_words further 32-bit words of the stream have been consumed: what NumPy's
// legacy generator holds after drawing them (pos = 624: the block is used up).
int mt19937_advance_host(uint32_t* key, int32_t* pos, int64_t n_words) {
  DFH_ARG(key != nullptr && pos != nullptr && n_words >= 0 && *pos >= 0 && *pos <= N);
  const int64_t left = N - *pos;
  if (n_words <= left) { *pos += int32_t(n_words); return DFH_OK; }
  const int64_t rem = n_words - left, blocks = (rem + N - 1) / N;
  if (blocks > 1) {
    const CharPoly& cp = char_poly();
    if (!cp.ok) {
      dfh_set_error("MT19937 jump-ahead: characteristic polynomial not found");
      return DFH_ERR_HIP;
    }
    jump_words(cp, key, (blocks - 1) * N);
  }
  twist_block(key);        // from the block before the target: every word of the result is genuine
  *pos = int32_t(rem - (blocks - 1) * N);
  return DFH_OK;
}

extern "C" int dfh_mt19937_advance(uint32_t* key, int32_t* pos, int64_t n_words) {
  return mt19937_advance_host(key, pos, n_words);
}
