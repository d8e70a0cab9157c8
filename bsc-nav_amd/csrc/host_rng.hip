// host_rng.hip — host-side helper (no device code): np.random.shuffle of arange(n), bit for bit.
//
// obs2voxeltoken sub-samples the depth image through `np.random.shuffle(idx); idx[::rate]` on NumPy's global MT19937
// stream (memory_2.py:747-749); parity of everything downstream (point order, ids, rgb chain) needs exactly that
// permutation and leaves the stream exactly where NumPy would.  At 640x480 the shuffle is the per-frame bottleneck of
// the reference-semantics mode (2.4 ms in NumPy), so it is restated here: MT19937 (Matsumoto & Nishimura) advanced in
// place on the caller's copy of NumPy's state, the legacy Fisher-Yates loop of RandomState.shuffle
// (`for i in reversed(range(1, n)): j = random_interval(i); swap`), and NumPy's bounded draw random_interval() =
// rejection on the smallest all-ones mask >= i using one 32-bit output per try (i < 2^32).
#include "bsc_internal.h"

namespace {
struct Mt {
    uint32_t *key;              // 624 words, NumPy's layout (untempered)
    uint32_t out[624];          // the same block tempered, produced 624 at a time (the loop vectorises)
    int pos;
};

inline void mt_temper_block(Mt &m)
{
    for (int k = 0; k < 624; ++k) {
        uint32_t y = m.key[k];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        m.out[k] = y;
    }
}

inline void mt_refill(Mt &m)
{
    uint32_t *mt = m.key;
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
    int kk = 0;
    for (; kk < 624 - 397; ++kk) {
        const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
    }
    for (; kk < 623; ++kk) {
        const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
    }
    const uint32_t y = (mt[623] & UPPER) | (mt[0] & LOWER);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
    m.pos = 0;
    mt_temper_block(m);
}

inline uint32_t mt_next(Mt &m)
{
    if (m.pos == 624) mt_refill(m);
    return m.out[m.pos++];
}
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
    for (uint32_t i = (uint32_t)n - 1; i >= 1; --i) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz(i);      // smallest all-ones mask >= i
        uint32_t j;
        while ((j = (mt_next(m) & mask)) > i) {}
        const int32_t t = x[j];
        x[j] = x[i];
        x[i] = t;
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
