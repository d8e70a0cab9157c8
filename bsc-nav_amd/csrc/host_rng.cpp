// host_rng.cpp — host-side helper (no device code): np.random.shuffle of arange(n), bit for bit.
//
// obs2voxeltoken sub-samples the depth image through `np.random.shuffle(idx); idx[::rate]` on NumPy's global MT19937
// stream (memory_2.py:747-749); parity of everything downstream (point order, ids, rgb chain) needs exactly that
// permutation and leaves the stream exactly where NumPy would.  At 640x480 the shuffle is the per-frame bottleneck of
// the reference-semantics mode (2.4 ms in NumPy), so it is restated here: MT19937 (Matsumoto & Nishimura) advanced in
// place on the caller's copy of NumPy's state, the legacy Fisher-Yates loop of RandomState.shuffle
// (`for i in reversed(range(1, n)): j = random_interval(i); swap`), and NumPy's bounded draw random_interval() =
// rejection on the smallest all-ones mask >= i using one 32-bit output per try (i < 2^32).
// Round 5: the loop was bound by (i) the unpredictable rejection branch (a draw is accepted with probability 1/2 .. 1), (ii) the
// scalar MT19937 block functions.  Now: the block functions are compiled a second time for AVX2 (runtime dispatch); the rejection
// runs branch-free over whole blocks of outputs into a batch of accepted indices — with AVX2, eight draws at a time: under one
// mask a draw j <= i - 8 is accepted whatever the seven before it did and a draw j > i is rejected, so only a draw that falls
// within 8 of i (probability 8 / 2^k) sends its group of eight through the scalar loop; the accepted lanes are packed with a
// 256-entry permutation table —; the swap targets of a batch are prefetched before the batch's swaps run IN ORDER (the
// permutation and the stream position are NumPy's, bit for bit: tests/test_abi.py).
// Plain C++ (g++), no device code: csrc/build.sh compiles this file on its own.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#define BSC_X86 1
#else
#define BSC_X86 0
#endif
#include "bscnav.h"

namespace {
struct Mt {
    uint32_t *key;              // 624 words, NumPy's layout (untempered)
    uint32_t out[624 + 8];      // the same block tempered, produced 624 at a time (the loop vectorises)
    int pos;
};

#define BSC_MT_BLOCK_FNS(ATTR, SUFFIX)                                                                                          \
    ATTR void mt_temper_block##SUFFIX(Mt &m)                                                                                    \
    {                                                                                                                           \
        for (int k = 0; k < 624; ++k) {                                                                                         \
            uint32_t y = m.key[k];                                                                                              \
            y ^= y >> 11;                                                                                                       \
            y ^= (y << 7) & 0x9d2c5680u;                                                                                        \
            y ^= (y << 15) & 0xefc60000u;                                                                                       \
            y ^= y >> 18;                                                                                                       \
            m.out[k] = y;                                                                                                       \
        }                                                                                                                       \
    }                                                                                                                           \
    ATTR void mt_refill##SUFFIX(Mt &m)                                                                                          \
    {                                                                                                                           \
        uint32_t *mt = m.key;                                                                                                   \
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;                                          \
        int kk = 0;                                                                                                             \
        for (; kk < 624 - 397; ++kk) {                                                                                          \
            const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);                                                         \
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);                                                        \
        }                                                                                                                       \
        for (; kk < 623; ++kk) {                                                                                                \
            const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);                                                         \
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);                                                \
        }                                                                                                                       \
        const uint32_t y = (mt[623] & UPPER) | (mt[0] & LOWER);                                                                 \
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);                                                                \
        m.pos = 0;                                                                                                              \
        mt_temper_block##SUFFIX(m);                                                                                             \
    }
BSC_MT_BLOCK_FNS(static, _base)
#if BSC_X86
BSC_MT_BLOCK_FNS(__attribute__((target("avx2"))) static, _avx2)
#endif

bool have_avx2()
{
#if BSC_X86
    static const bool v = __builtin_cpu_supports("avx2") && !getenv("BSC_HOST_NO_AVX2");      // (the switch: for the parity test of the plain path)
    return v;
#else
    return false;
#endif
}

inline void mt_temper_block(Mt &m)
{
#if BSC_X86
    if (have_avx2()) return mt_temper_block_avx2(m);
#endif
    mt_temper_block_base(m);
}
inline void mt_refill(Mt &m)
{
#if BSC_X86
    if (have_avx2()) return mt_refill_avx2(m);
#endif
    mt_refill_base(m);
}
inline uint32_t mt_next(Mt &m)
{
    if (m.pos == 624) mt_refill(m);
    return m.out[m.pos++];
}

constexpr int SH_NB = 512;          // accepted draws per batch (their swap targets are prefetched together)

// Accepted draws for the indices i0, i0 - 1, ... (ii = the next index to serve), from the outputs out[p .. 624), under ONE mask
// (ii stays in [lo, mask]); stops when the block, the batch or the mask's range ends.  Branch-free: a rejected draw is
// overwritten by the next one.
inline int accept_scalar(const uint32_t *out, int p, int pend, uint32_t mask, uint32_t lo, uint32_t i0, uint32_t &ii, uint32_t *jb)
{
    while (p < pend && ii >= lo && i0 - ii < (uint32_t)SH_NB) {
        const uint32_t j = out[p++] & mask;
        jb[i0 - ii] = j;
        ii -= (j <= ii);
    }
    return p;
}

#if BSC_X86
struct PackLut { uint32_t idx[256][8]; };
const PackLut &pack_lut()
{
    static const PackLut lut = [] {
        PackLut l;
        for (int m = 0; m < 256; ++m) {
            int n = 0;
            for (int b = 0; b < 8; ++b)
                if (m >> b & 1) l.idx[m][n++] = (uint32_t)b;
            for (; n < 8; ++n) l.idx[m][n] = 0;
        }
        return l;
    }();
    return lut;
}

__attribute__((target("avx2,popcnt"))) int accept_avx2(const uint32_t *out, int p, int pend, uint32_t mask, uint32_t lo, uint32_t i0,
                                                       uint32_t &ii, uint32_t *jb)
{
    const PackLut &lut = pack_lut();
    const __m256i vmask = _mm256_set1_epi32((int)mask);
    // (all values are below 2^31: signed compares are exact)
    while (p + 8 <= pend && ii >= lo + 8 && i0 - ii + 8 <= (uint32_t)SH_NB) {
        const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(out + p)), vmask);
        const __m256i sure = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)(ii - 7)), v);        // j <= ii - 8
        const __m256i rej = _mm256_cmpgt_epi32(v, _mm256_set1_epi32((int)ii));               // j > ii
        const int ms = _mm256_movemask_ps(_mm256_castsi256_ps(sure)), mr = _mm256_movemask_ps(_mm256_castsi256_ps(rej));
        if ((ms | mr) != 0xff) {            // a draw within 8 of ii: its outcome depends on the draws before it — these eight go one by one
            p = accept_scalar(out, p, p + 8, mask, lo, i0, ii, jb);
            continue;
        }
        const __m256i packed = _mm256_permutevar8x32_epi32(v, _mm256_loadu_si256((const __m256i *)lut.idx[ms]));
        _mm256_storeu_si256((__m256i *)(jb + (i0 - ii)), packed);
        ii -= (uint32_t)__builtin_popcount((unsigned)ms);
        p += 8;
    }
    return p;
}
#endif
}  // namespace

// key624 / pos: NumPy's np.random.get_state()[1:3], advanced in place.  out[k] = shuffled_arange(n)[k * rate].
extern "C" bsc_status bsc_host_shuffled_sample(uint32_t *key624, int32_t *pos, int64_t n, int32_t rate, int32_t *scratch_n,
                                               int32_t *out)
{
    if (!key624 || !pos || !scratch_n || !out || n < 1 || n > 0x7fffffffll || rate < 1 || *pos < 0 || *pos > 624)
        return BSC_E_INVALID;
    Mt m;
    m.key = key624;
    m.pos = *pos;
    mt_temper_block(m);
    int32_t *x = scratch_n;
    for (int64_t i = 0; i < n; ++i) x[i] = (int32_t)i;
    uint32_t jb[SH_NB + 16];
    uint32_t ii = (uint32_t)n - 1;                      // the next index to serve (the legacy loop: i = n - 1 .. 1)
    while (ii >= 1) {
        const uint32_t i0 = ii;
        while (ii >= 1 && i0 - ii < (uint32_t)SH_NB) {
            if (m.pos == 624) mt_refill(m);
            const uint32_t mask = 0xffffffffu >> __builtin_clz(ii);      // smallest all-ones mask >= ii
            const uint32_t lo = (mask >> 1) + 1;                         // ii keeps this mask down to lo
            int p = m.pos;
#if BSC_X86
            if (have_avx2()) p = accept_avx2(m.out, p, 624, mask, lo, i0, ii, jb);
#endif
            // the tail of the block / batch / mask range, one draw at a time (at least one draw per round: progress)
            p = accept_scalar(m.out, p, p + 8 < 624 ? p + 8 : 624, mask, lo, i0, ii, jb);
            m.pos = p;
        }
        const uint32_t c = i0 - ii;
        for (uint32_t k = 0; k < c; ++k) __builtin_prefetch(&x[jb[k]], 1, 3);
        for (uint32_t k = 0; k < c; ++k) {
            const uint32_t i2 = i0 - k, j = jb[k];
            const int32_t t = x[j];
            x[j] = x[i2];
            x[i2] = t;
        }
    }
    for (int64_t k = 0, src = 0; src < n; ++k, src += rate) out[k] = x[src];
    *pos = m.pos;
    return BSC_OK;
}

// n draws of Python's `random.choice(range(n_choices))` (memory_2.py:352) on a copy of the `random` module's MT19937
// state (random.getstate()[1] = 624 key words + pos, advanced in place): CPython's _randbelow_with_getrandbits takes
// k = n_choices.bit_length() bits per try as genrand_uint32() >> (32 - k) and rejects values >= n_choices.
extern "C" bsc_status bsc_host_choice_draws(uint32_t *key624, int32_t *pos, uint32_t n_choices, uint32_t n, uint32_t *out)
{
    if (!key624 || !pos || (!out && n) || n_choices < 1 || *pos < 0 || *pos > 624) return BSC_E_INVALID;
    Mt m;
    m.key = key624;
    m.pos = *pos;
    mt_temper_block(m);
    const int k = 32 - __builtin_clz(n_choices);            // bit_length
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t r;
        while ((r = mt_next(m) >> (32 - k)) >= n_choices) {}
        out[i] = r;
    }
    *pos = m.pos;
    return BSC_OK;
}
