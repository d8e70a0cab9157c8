// radix.hip — the two sorts on the critical path of bsc_ingest, in-tree: a stable least-significant-digit radix sort of
// (u32 key, u32 value) pairs, one launch per digit ("onesweep": chained scan with decoupled look-back), written for the
// keys this library sorts — the 24-bit Morton cell codes of the (voxel, frame, patch) pairs and the 15-22-bit voxel ids
// of the point runs (memory_2.py:888-903: the order of a voxel's points, and of its tokens, defines its colour and its
// mean) — on gfx950's 64-wide wavefronts.
//
//   k_radix_hist   one read of the keys: digit histograms of ALL passes (per-workgroup partials, the last workgroup to
//                  finish adds them up and leaves the exclusive digit offsets; no atomics on global counters, nothing
//                  to clear between calls)
//   k_radix_pass   per pass: a workgroup takes the next tile of 8192 items (ticket = tile index, so every tile it waits
//                  for is already running), ranks its items digit by digit — a wavefront's 64 items of a round by
//                  ballots (the lanes with the same digit are the AND of the 8 bit ballots), rounds by a per-wavefront
//                  count in LDS, wavefronts by an exclusive prefix — publishes its digit counts, looks back over the
//                  tiles before it for their sum, and writes keys and values through LDS in digit order (consecutive
//                  lanes, consecutive addresses inside a digit's stretch).
// Status words carry {epoch, state, count}: the epoch is unique per (sort, pass), so the array is never cleared.
// Stable: tiles in index order, inside a tile wavefront chunks in order, inside a chunk rounds in order, inside a round
// lanes in order.  The input arrays are preserved; the result lands in (keys_out, vals_out) after any number of passes.
#include "bsc_internal.h"

#define RX_THREADS 512
#define RX_WAVES (RX_THREADS / 64)
#define RX_IPT 16                               // items per thread
#define RX_TILE (RX_THREADS * RX_IPT)           // 8192 items per tile
#define RX_BINS 256
#define RX_HIST_BLOCKS 1024
#define RX_ST_AGG 1ull                          // status: the tile's own digit count
#define RX_ST_INC 2ull                          // status: inclusive sum over the tiles up to this one

__device__ __forceinline__ void rx_wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t rx_wave_incl_sum(uint32_t v)
{
#define RX_SCAN_STEP(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false);
    RX_SCAN_STEP(0x111, 0xf)                    // row_shr:1
    RX_SCAN_STEP(0x112, 0xf)                    // row_shr:2
    RX_SCAN_STEP(0x114, 0xf)                    // row_shr:4
    RX_SCAN_STEP(0x118, 0xf)                    // row_shr:8
    RX_SCAN_STEP(0x142, 0xa)                    // row_bcast:15 into rows 1 and 3
    RX_SCAN_STEP(0x143, 0xc)                    // row_bcast:31 into rows 2 and 3
#undef RX_SCAN_STEP
    return v;
}

struct RadixPlan { int npass; int shift[4]; int bits[4]; };

// Partials travel between workgroups (possibly on different XCDs, each with its own L2) as agent-scope relaxed atomics — stores that
// write through, loads that do not hit a stale line — ordered against the ticket by waiting for the stores' acknowledgements.
// A __threadfence() here is a release fence at agent scope = write back EVERY dirty line of the XCD's L2: beside k_points / the pair
// tiles, which keep the L2s full of dirty lines, that made k_radix_hist take 0.2 ms for 0.05 ms of work.
__device__ __forceinline__ void rx_publish(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t rx_fetch(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rx_stores_done() { __builtin_amdgcn_s_waitcnt(0); }

// ---- histograms of all passes in one read ------------------------------------------------------------------------------------
// Up to RX_HIST_BLOCKS workgroups of 256 threads (small workgroups: beside the kernels of the other stream a 1024-thread form waited
// for whole CUs and took 0.2 ms for what it does alone in 0.05).  A workgroup counts into RX_HREP copies of every histogram (copy =
// lane mod RX_HREP, the copies of a digit in consecutive LDS banks): the sorted ids of a scene — a third of all runs in one voxel,
// the upper digit of every id nearly constant — otherwise hit ONE counter 64 lanes deep (2.9 ms for 1.2e8 keys with one copy;
// matching the lanes of a digit through ballots first, as k_radix_pass does for its ranks, costs ~80 vector instructions per key
// and pass: 1.4 ms).  Partials are added up in two levels, each by the last workgroup to arrive (ticket): the last of every group
// of 32 adds its group, the last of those adds the groups and writes the exclusive digit offsets of every pass — two short
// dependent steps instead of one workgroup reading a thousand partials.  All tickets are left at zero.
#define RX_HIST_THREADS 256
#define RX_HREP 4
#define RX_HGROUP 32
__global__ __launch_bounds__(RX_HIST_THREADS) void k_radix_hist(const uint32_t *__restrict__ keys, uint32_t n, RadixPlan pl,
                                                                uint32_t *__restrict__ partial, uint32_t *__restrict__ gpartial,
                                                                uint32_t *__restrict__ goff, uint32_t *tickets)
{
    __shared__ uint32_t h[4 * RX_BINS * RX_HREP];       // [pass][digit][copy]
    __shared__ uint32_t s_w[4];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    for (int i = tid; i < 4 * RX_BINS * RX_HREP; i += RX_HIST_THREADS) h[i] = 0u;
    __syncthreads();
    const uint32_t n4 = n >> 2;
    const uint32_t rep = (uint32_t)tid & (RX_HREP - 1);
    uint32_t sh[4], mk[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { sh[p] = (uint32_t)pl.shift[p]; mk[p] = p < pl.npass ? (1u << pl.bits[p]) - 1u : 0u; }
    for (uint32_t i = blockIdx.x * (uint32_t)RX_HIST_THREADS + tid; i < n4; i += gridDim.x * (uint32_t)RX_HIST_THREADS) {
        const uint4 k4 = ((const uint4 *)keys)[i];
        const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p < pl.npass) {
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicAdd(&h[((p * RX_BINS + ((kk[c] >> sh[p]) & mk[p])) * RX_HREP) + rep], 1u);
            }
        }
    }
    if (blockIdx.x == 0 && tid < (int)(n & 3u)) {
        const uint32_t k = keys[4u * n4 + tid];
        for (int p = 0; p < pl.npass; ++p) atomicAdd(&h[(p * RX_BINS + ((k >> sh[p]) & mk[p])) * RX_HREP], 1u);
    }
    __syncthreads();
    for (int p = 0; p < pl.npass; ++p) {        // thread = digit: the sum of its copies
        uint32_t sum = 0;
#pragma unroll
        for (int r = 0; r < RX_HREP; ++r) sum += h[(p * RX_BINS + tid) * RX_HREP + r];
        rx_publish(&partial[((size_t)blockIdx.x * 4 + p) * RX_BINS + tid], sum);
    }
    const uint32_t grp = blockIdx.x / RX_HGROUP, ngrp = (gridDim.x + RX_HGROUP - 1) / RX_HGROUP;
    const uint32_t g0 = grp * RX_HGROUP, g1 = g0 + RX_HGROUP < gridDim.x ? g0 + RX_HGROUP : gridDim.x;
    rx_stores_done();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&tickets[8 + grp], 1u) == g1 - g0 - 1;
    __syncthreads();
    if (!s_last) return;
    for (int p = 0; p < pl.npass; ++p) {
        uint32_t sum = 0;
        for (uint32_t b = g0; b < g1; b += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = rx_fetch(&partial[((size_t)(b + u < g1 ? b + u : g0) * 4 + p) * RX_BINS + tid]);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += b + u < g1 ? v[u] : 0u;
        }
        rx_publish(&gpartial[((size_t)grp * 4 + p) * RX_BINS + tid], sum);
    }
    rx_stores_done();
    __syncthreads();
    if (tid == 0) { rx_publish(&tickets[8 + grp], 0u); s_last = atomicAdd(&tickets[0], 1u) == ngrp - 1; }
    __syncthreads();
    if (!s_last) return;
    for (int p = 0; p < pl.npass; ++p) {
        uint32_t sum = 0;
        for (uint32_t g = 0; g < ngrp; g += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = rx_fetch(&gpartial[((size_t)(g + u < ngrp ? g + u : 0) * 4 + p) * RX_BINS + tid]);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += g + u < ngrp ? v[u] : 0u;
        }
        const uint32_t incl = rx_wave_incl_sum(sum);
        if ((tid & 63) == 63) s_w[tid >> 6] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_w[w];
        goff[p * RX_BINS + tid] = base + incl - sum;
        __syncthreads();
    }
    if (tid < 8) tickets[tid] = 0u;             // the tile tickets of the passes that follow, and this kernel's own
}

// ---- one digit ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RX_THREADS) void k_radix_pass(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                          uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, uint32_t n,
                                                          int shift, int bits, const uint32_t *__restrict__ goff, u64 *status,
                                                          uint32_t *ticket, uint32_t epoch)
{
    __shared__ uint32_t s_whist[RX_WAVES][RX_BINS];     // per wavefront and digit: items so far; later the wavefront's offset inside the digit
    __shared__ uint32_t s_buf[RX_TILE];
    __shared__ uint32_t s_texcl[RX_BINS];               // first position of the digit inside the tile's digit-ordered buffer
    __shared__ uint32_t s_gbase[RX_BINS];               // output index of that position, minus the position
    __shared__ uint32_t s_w[RX_WAVES];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < RX_WAVES * RX_BINS; i += RX_THREADS) (&s_whist[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * (uint32_t)RX_TILE;
    const uint32_t tile_n = n - base < (uint32_t)RX_TILE ? n - base : (uint32_t)RX_TILE;
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t cb = (uint32_t)(wv * 64 * RX_IPT);   // the wavefront's chunk of the tile: items [cb, cb + 64 * RX_IPT), in order
    uint32_t key[RX_IPT], val[RX_IPT], rk[RX_IPT];
#pragma unroll
    for (int r = 0; r < RX_IPT; ++r) {
        const uint32_t i = cb + r * 64 + lane;
        key[r] = kin[base + (i < tile_n ? i : 0u)];
    }
#pragma unroll
    for (int r = 0; r < RX_IPT; ++r) {
        const uint32_t i = cb + r * 64 + lane;
        val[r] = vin[base + (i < tile_n ? i : 0u)];
    }
    // ---- rank of every item among the tile's items of its digit ----
    const uint32_t lt_lo = lane < 32 ? (1u << lane) - 1u : 0xffffffffu, lt_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
#pragma unroll
    for (int r = 0; r < RX_IPT; ++r) {
        const bool valid = cb + r * 64 + lane < tile_n;
        const uint32_t d = (key[r] >> shift) & mask;
        const u64 vm = __ballot(valid);
        uint32_t p_lo = (uint32_t)vm, p_hi = (uint32_t)(vm >> 32);
#pragma unroll
        for (int b = 0; b < 8; ++b) {           // bits at or above `bits` are zero in every lane: their ballot changes nothing
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            const uint32_t flip = bit ? 0u : 0xffffffffu;
            p_lo &= (uint32_t)bal ^ flip;
            p_hi &= (uint32_t)(bal >> 32) ^ flip;
        }
        const uint32_t before = valid ? s_whist[wv][d] : 0u;
        rx_wave_lds_order();
        const uint32_t below = (uint32_t)__popc(p_lo & lt_lo) + (uint32_t)__popc(p_hi & lt_hi);
        if (valid && below == 0u) s_whist[wv][d] = before + (uint32_t)__popc(p_lo) + (uint32_t)__popc(p_hi);
        rx_wave_lds_order();
        rk[r] = (before + below) | (d << 16);   // a wavefront's chunk is 1024 items: the rank fits 16 bits
    }
    __syncthreads();
    // ---- digit counts of the tile, their place in the tile and (look-back) in the whole output ----
    uint32_t cnt = 0;
    if (tid < RX_BINS) {
#pragma unroll
        for (int w = 0; w < RX_WAVES; ++w) { const uint32_t t = s_whist[w][tid]; s_whist[w][tid] = cnt; cnt += t; }
    }
    const uint32_t incl = rx_wave_incl_sum(cnt);
    if (lane == 63) s_w[wv] = incl;
    if (tid < RX_BINS) {
        u64 *const st = status + (size_t)tile * RX_BINS + tid;
        const u64 tag = (u64)epoch << 32;
        uint32_t excl = 0;
        if (tile > 0) {
            __hip_atomic_store(st, tag | (RX_ST_AGG << 30) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 *q = st - RX_BINS;
            for (;;) {
                const u64 s = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t state = (uint32_t)(s >> 30) & 3u;
                if ((uint32_t)(s >> 32) != epoch || state == 0u) { __builtin_amdgcn_s_sleep(1); continue; }
                excl += (uint32_t)s & 0x3fffffffu;
                if (state == (uint32_t)RX_ST_INC) break;
                q -= RX_BINS;
            }
        }
        __hip_atomic_store(st, tag | (RX_ST_INC << 30) | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_gbase[tid] = goff[tid] + excl;        // finished below, once the digit's place in the tile is known
    }
    __syncthreads();
    if (tid < RX_BINS) {
        uint32_t pre = incl - cnt;
        for (int w = 0; w < wv; ++w) pre += s_w[w];
        s_texcl[tid] = pre;
        s_gbase[tid] -= pre;
    }
    __syncthreads();
    // ---- keys through LDS in digit order, out as stretches; then the values the same way ----
#pragma unroll
    for (int r = 0; r < RX_IPT; ++r) {
        const uint32_t d = rk[r] >> 16;
        const uint32_t pos = s_texcl[d] + s_whist[wv][d] + (rk[r] & 0xffffu);
        rk[r] = pos;
        if (cb + r * 64 + lane < tile_n) s_buf[pos] = key[r];
    }
    __syncthreads();
    uint32_t gp[RX_IPT];
#pragma unroll
    for (int i = 0; i < RX_IPT; ++i) {
        const uint32_t p = (uint32_t)(i * RX_THREADS + tid);
        gp[i] = 0xffffffffu;
        if (p < tile_n) {
            const uint32_t k = s_buf[p];
            gp[i] = s_gbase[(k >> shift) & mask] + p;
            kout[gp[i]] = k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RX_IPT; ++r)
        if (cb + r * 64 + lane < tile_n) s_buf[rk[r]] = val[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RX_IPT; ++i) {
        const uint32_t p = (uint32_t)(i * RX_THREADS + tid);
        if (p < tile_n) vout[gp[i]] = s_buf[p];
    }
}

bsc_status radix_sort_pairs_u32(bsc_ctx *x, RadixWs *ws, hipStream_t st, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                                uint32_t *vout, size_t n, int b0, int b1)
{
    if (n == 0) return BSC_OK;
    const int nbits = b1 - b0;
    if (nbits <= 0) {
        BSC_HIP(hipMemcpyAsync(kout, kin, 4 * n, hipMemcpyDeviceToDevice, st));
        BSC_HIP(hipMemcpyAsync(vout, vin, 4 * n, hipMemcpyDeviceToDevice, st));
        return BSC_OK;
    }
    if (n > ws->max_items || n >= (1ull << 30) || nbits > 32 || ((uintptr_t)kin & 15)) {
        bsc_set_error("radix_sort_pairs_u32: %zu items (workspace for %zu), bits [%d, %d)", n, ws->max_items, b0, b1);
        return BSC_E_INVALID;
    }
    RadixPlan pl;
    pl.npass = (nbits + 7) / 8;
    for (int p = 0, b = b0; p < 4; ++p) {
        pl.bits[p] = p < pl.npass ? nbits / pl.npass + (p < nbits % pl.npass ? 1 : 0) : 0;
        pl.shift[p] = b;
        b += pl.bits[p];
    }
    const uint32_t tiles = (uint32_t)((n + RX_TILE - 1) / RX_TILE);
    uint32_t hb = (uint32_t)((n / 4 + 4 * RX_HIST_THREADS - 1) / (4 * RX_HIST_THREADS));     // four 16-byte loads per thread or more
    hb = hb < 1 ? 1 : (hb > RX_HIST_BLOCKS ? RX_HIST_BLOCKS : hb);
    hipLaunchKernelGGL(k_radix_hist, dim3(hb), dim3(RX_HIST_THREADS), 0, st, kin, (uint32_t)n, pl, ws->partial, ws->gpartial, ws->goff, ws->tickets);
    const uint32_t *ki = kin, *vi = vin;
    for (int p = 0; p < pl.npass; ++p) {
        // destination of pass p: the output after an even number of remaining passes, the temporaries otherwise
        const bool to_out = ((pl.npass - 1 - p) & 1) == 0;
        uint32_t *ko = to_out ? kout : ws->tmp_k, *vo = to_out ? vout : ws->tmp_v;
        ws->epoch += 1;
        if (ws->epoch == 0) ws->epoch = 1;
        hipLaunchKernelGGL(k_radix_pass, dim3(tiles), dim3(RX_THREADS), 0, st, ki, vi, ko, vo, (uint32_t)n, pl.shift[p], pl.bits[p],
                           ws->goff + p * RX_BINS, ws->status, ws->tickets + 1 + p, ws->epoch);
        ki = ko; vi = vo;
    }
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

bsc_status radix_ws_create(RadixWs *ws, size_t max_items)
{
    memset(ws, 0, sizeof *ws);
    ws->max_items = max_items;
    const size_t tiles = (max_items + RX_TILE - 1) / RX_TILE + 1;
    BSC_HIP(hipMalloc((void **)&ws->status, tiles * RX_BINS * sizeof(u64)));
    BSC_HIP(hipMemset(ws->status, 0, tiles * RX_BINS * sizeof(u64)));
    BSC_HIP(hipMalloc((void **)&ws->partial, (size_t)RX_HIST_BLOCKS * 4 * RX_BINS * 4));
    BSC_HIP(hipMalloc((void **)&ws->goff, 4 * RX_BINS * 4));
    BSC_HIP(hipMalloc((void **)&ws->gpartial, (size_t)(RX_HIST_BLOCKS / RX_HGROUP) * 4 * RX_BINS * 4));
    BSC_HIP(hipMalloc((void **)&ws->tickets, 4 * (8 + RX_HIST_BLOCKS / RX_HGROUP)));
    BSC_HIP(hipMemset(ws->tickets, 0, 4 * (8 + RX_HIST_BLOCKS / RX_HGROUP)));
    BSC_HIP(hipMalloc((void **)&ws->tmp_k, 4 * max_items));
    BSC_HIP(hipMalloc((void **)&ws->tmp_v, 4 * max_items));
    ws->epoch = 0;
    return BSC_OK;
}

void radix_ws_destroy(RadixWs *ws)
{
    void *ptrs[] = {ws->status, ws->partial, ws->gpartial, ws->goff, ws->tickets, ws->tmp_k, ws->tmp_v};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    memset(ws, 0, sizeof *ws);
}
