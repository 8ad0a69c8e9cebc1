// sampler.h — bit-exact restatement of the minibatch index sampler of
// ReplayMemory.getMinibatch (src/replay_memory.py:54-68) on CPython 3.10's MT19937 stream.
// Pure host code; the 625-word state is interchangeable with random.getstate()[1] so the
// agent's own random.random()/randrange draws (src/agent.py:32,50-51) stay interleaved.
#pragma once
#include <stdint.h>

namespace sdqn {

// 32-bit outputs drawn by this library since it was loaded (host-side, one thread): lets a caller that handed over a COPY of its generator
// state advance its own generator by exactly the words a call consumed (random.getrandbits(32 * words)) instead of re-importing 625 words
inline uint64_t& mt_words_drawn() { static uint64_t w = 0; return w; }

struct MT {
  uint32_t* s;   // 624 words + position
  explicit MT(uint32_t* st) : s(st) {}

  static void init_genrand(uint32_t* mt, uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  }
  // random.seed(int): init_by_array over the 32-bit little-endian limbs of abs(seed)
  static void seed(uint32_t* st, uint64_t sd) {
    uint32_t key[2]; int klen = 1;
    key[0] = (uint32_t)(sd & 0xFFFFFFFFu); key[1] = (uint32_t)(sd >> 32);
    if (key[1]) klen = 2;
    uint32_t* mt = st;
    init_genrand(mt, 19650218u);
    int i = 1, j = 0;
    for (int k = 624 > klen ? 624 : klen; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      ++i; ++j;
      if (i >= 624) { mt[0] = mt[623]; i = 1; }
      if (j >= klen) j = 0;
    }
    for (int k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      ++i;
      if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
    st[624] = 624;
  }
  uint32_t genrand() {
    uint32_t* mt = s;
    if (s[624] >= 624) {
      int kk;
      for (kk = 0; kk < 624 - 397; ++kk) {
        uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7FFFFFFFu);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
      }
      for (; kk < 623; ++kk) {
        uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7FFFFFFFu);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
      }
      uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7FFFFFFFu);
      mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
      s[624] = 0;
    }
    uint32_t y = mt[s[624]++];
    ++mt_words_drawn();
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
  }
  // Lib/random.py _randbelow_with_getrandbits, n < 2^32 (ring sizes are far below that)
  uint64_t randbelow(uint64_t n) {
    int k = 0;
    for (uint64_t v = n; v; v >>= 1) ++k;             // n.bit_length()
    if (k <= 32) {
      uint32_t r = genrand() >> (32 - k);
      while (r >= n) r = genrand() >> (32 - k);
      return r;
    }
    // getrandbits(k > 32): words filled little-endian, the last one shifted (Modules/_randommodule.c)
    for (;;) {
      uint64_t lo = genrand();
      uint64_t hi = genrand() >> (64 - k);
      uint64_t r = (hi << 32) | lo;
      if (r < n) return r;
    }
  }
  int64_t randint(int64_t a, int64_t b) { return a + (int64_t)randbelow((uint64_t)(b - a + 1)); }
};

// replay_memory.py:54-68. Returns number of draws consumed.
inline int64_t sample_indices(uint32_t* mt_state, const uint8_t* terminals, int64_t count, int64_t current,
                              int hist, int batch, int64_t* out) {
  MT mt(mt_state);
  int64_t draws = 0;
  for (int n = 0; n < batch; ++n) {
    for (;;) {
      int64_t index = mt.randint(hist, count - 1);                     // :59
      ++draws;
      if (index >= current && index - hist < current) continue;        // :61
      bool any = false;                                                // :65 terminals[index-hist:index].any()
      for (int64_t i = index - hist; i < index; ++i) any |= terminals[i] != 0;
      if (any) continue;
      out[n] = index;
      break;
    }
  }
  return draws;
}

}  // namespace sdqn
