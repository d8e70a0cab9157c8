// ingest.hip — obs2voxeltoken's per-point loop (memory_2.py:859-903) as a batch pipeline on gfx950.
//
// The reference walks the sampled points of a frame sequentially; ids, the rgb running mean, the
// top-down map and the token cache are all defined by that order.  Here every point j of a batch
// (frames in call order, points in the reference's shuffled order) carries its order implicitly:
//
//   k_points    2048 consecutive points per workgroup: geometry (fp64, bit-exact) -> cell / rgb (/ alpha); the block groups its
//               records by cell (stable, in LDS), emits ONE run per (block, cell) and makes one first-touch claim per cell: the
//               smallest point index wins a still-empty cell, the claimer that found it empty lists the cell as new
//   k_block_totals   the block scans (runs, passing points) + the batch scalars in one launch; the scalars reach the host through
//               a mailbox in pinned memory (round 6; k_totals + two rocPRIM scans + a copy before)
//   k_patch_pairs / k_keys_pairs (dense.hip) LDS aggregation of (cell, frame, patch) pairs — independent of the ids
//   k_new_keys + sort + k_new_assign   the new cells ranked by their winning point: id = max_id + rank, as the
//               sequential max_id++ hands them out (a few thousand cells, not a pass over the points)
//   k_run_keys  every run as key = voxel id | (length - 1) << id bits, value = its first record
//   radix sort  of the runs (radix.hip: stable, on the id bits only), k_run_blocksum + scan + k_expand: every voxel's points in
//               order j as run start bits + a checkpoint per 64 positions — no index per point
//   k_chain     per voxel: sequential truncating weighted rgb mean + top-down map atomicMax on (h, order of the voxel's latest
//               point), a quad of lanes per voxel (short segments, the first points of new voxels)
//   k_chain_long   long segments 64 points per round by checked prediction: hot segments in register tiles over a workgroup,
//               the one-wavefront segments as a launch of their own beside them (round 6)
//   k_hwin      the winning voxel of each map cell writes its colour
//   dense.hip   pair sort + k_dense_reduce: multiplicity x token rows -> one RMW of the D-float
//               accumulator row per voxel
//   k_append    exact mode: token rows into the cache in order                (1 wavefront / row)
#include "bsc_internal.h"
#include "geometry_dev.h"

#include <limits.h>
#include <sched.h>
#include <math.h>

#define TPB 256

static GeomConst make_geom_const(const bsc_ctx *x)
{
    GeomConst g;
    memcpy(g.K, x->c.K, sizeof g.K);
    memcpy(g.Kinv, x->c.Kinv, sizeof g.Kinv);
    memcpy(g.Kp, x->c.Kpatch, sizeof g.Kp);
    g.cs = x->c.cell_size;
    g.half_gs = (double)x->c.grid_size / 2.0;   // utils.py:202 `gs / 2` is float division
    g.min_depth = x->c.min_depth;
    g.max_depth = x->c.max_depth;
    g.H = x->c.height; g.W = x->c.width; g.gs = x->c.grid_size;
    g.min_h = x->c.min_h; g.max_h = x->c.max_h; g.nh = x->nh; g.g = x->c.patch_grid;
    g.fast = x->geom_fast ? 1 : 0;
    g.rcs = 1.0 / x->c.cell_size;
    g.proj_id = x->proj_id ? 1 : 0;
    g.gs_even = (x->c.grid_size & 1) == 0 ? 1 : 0;
    g.pat_x = x->pat_x; g.pat_y = x->pat_y;
    // bsc_exp (geometry_dev.h): 64 / ln 2; ln 2 / 64 as a 40-bit head (k * head is exact for |k| < 2^13) and its tail; 1/2 .. 1/120
    const long double l64 = 0.693147180559945309417232121458176568L / 64.0L;
    double l1 = (double)l64;
    uint64_t bits;
    memcpy(&bits, &l1, 8);
    bits &= ~((1ull << 13) - 1);
    memcpy(&l1, &bits, 8);
    g.exp_il = (double)(1.0L / l64);
    g.exp_l1 = l1;
    g.exp_l2 = (double)(l64 - (long double)l1);
    g.exp_c2 = 0.5; g.exp_c3 = 1.0 / 6.0; g.exp_c4 = 1.0 / 24.0; g.exp_c5 = 1.0 / 120.0;
    g.exp_tab = (const double2 *)x->exp_tab;
    // 8-byte point records (geometry_dev.h rec8_*): depth offsets from a float at or below min_depth, exact index divisions
    g.zbase = x->rec8_zbase;
    g.rec_lb = x->group_rpw == 16 ? 12 : (x->group_rpw == 8 ? 11 : 10);
    const auto magic = [](uint32_t d, u64_t &m, int32_t &sft) {
        sft = 0;
        while ((1ull << sft) < d) ++sft;
        m = (u64_t)((((unsigned __int128)1) << (32 + sft)) / d) + 1ull;
    };
    magic((uint32_t)(x->c.height * x->c.width), g.div_n_m, g.div_n_s);
    magic((uint32_t)x->c.width, g.div_w_m, g.div_w_s);
    return g;
}

__device__ __forceinline__ int frame_of(const int64_t *offsets, int n_frames, int64_t j)
{
    int lo = 0, hi = n_frames;   // largest f with offsets[f] <= j
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offsets[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// k_points: geometry of GB consecutive points per workgroup + BLOCK-LOCAL STABLE GROUPING of their records by cell.
//
// The rgb chain needs every voxel's points in order j.  Sorting the points is out of the question (1.2e8 per call), and a
// "run" of consecutive same-cell points is short where a surface lies on a voxel boundary (1.9 points in the bench scene).
// So the block groups its own points: the records of a cell are written next to each other (in order j) inside the block's
// slice of p_rec, and the block emits ONE run per distinct cell — 4.5x (GB = 1024) to 8x (GB = 2048) fewer runs to sort,
// expand and gather, and the chain reads its records as contiguous stretches instead of one 128-byte line per 12-byte record.
// Order inside a voxel stays j: blocks are consecutive in j, the grouping is stable, the run sort is stable.
//
// Stable rank of a point inside its group without sorting: wavefront wv owns the points [wv * RPW * 64, (wv + 1) * RPW * 64) of
// the block and walks them in rounds of 64 (in order).  A cell gets a slot e of a workgroup-wide LDS hash table; in a round
// every lane ORs its bit into the wavefront's 64-bit word of that slot — the word is the ballot of the cell — so
//     rank inside the wavefront = points of the cell in the wavefront's earlier rounds (s_cnt[wv][e]) + popc(word & lanes below),
// and after the barrier the exclusive prefix over slots (group base) and over wavefronts finishes the position.  OR is
// order-free, so the layout does not depend on how the LDS serialises the lanes.  A cell that finds no slot within
// GROUP_PROBES probes (more distinct cells than the table holds: one voxel per point) keeps its points as runs of one,
// placed after the groups in order j.  Which slot a cell gets depends on the race for it; the RESULT does not.
#define GW (TPB / 64)           // wavefronts per workgroup
#define GROUP_HS 512            // hash slots per workgroup
#define GROUP_HS_LOG2 9
#define GROUP_PROBES 8
#define GROUP_OVF 0xffffu

__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v)
{
#define BSC_USCAN_STEP(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false);
    BSC_USCAN_STEP(0x111, 0xf)                   // row_shr:1
    BSC_USCAN_STEP(0x112, 0xf)                   // row_shr:2
    BSC_USCAN_STEP(0x114, 0xf)                   // row_shr:4
    BSC_USCAN_STEP(0x118, 0xf)                   // row_shr:8
    BSC_USCAN_STEP(0x142, 0xa)                   // row_bcast:15 into rows 1 and 3
    BSC_USCAN_STEP(0x143, 0xc)                   // row_bcast:31 into rows 2 and 3
#undef BSC_USCAN_STEP
    return v;
}

// LDS accesses of ONE wavefront execute in program order; this keeps the compiler from reordering them.  The fence names the LDS
// address space: as a fence over ALL memory it was lowered to s_waitcnt vmcnt(0) — three times per round of 64 points k_points
// then sat out the round trips of its depth load, colour gather and cell store (4.7 k clocks per round for ~1 k of issue).
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

// first-touch claim of a cell by the point j (memory_2.py:888-894): the smallest j wins an empty cell; the claimer that found
// the cell EMPTY lists it — exactly one entry per new voxel.  Wave-aggregated append.
__device__ __forceinline__ void claim_cells(bool want, int32_t cell, int64_t j, int32_t *occ, int32_t *__restrict__ new_cells,
                                            int64_t *dscal, int lane)
{
    bool first = false;
    if (want) {
        const int32_t mine = INT_MIN + (int32_t)j, cur = occ[cell];
        if (cur < 0 && cur > mine) first = atomicMin(&occ[cell], mine) == -1;
    }
    const u64 fm = __ballot(first);
    if (fm) {
        const int leader = __ffsll((long long)fm) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd((unsigned long long *)&dscal[DS_B_NNEW], (unsigned long long)__popcll(fm));
        base = __shfl(base, leader);
        if (first) new_cells[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = cell;
    }
}

// FAST: geom_point_fast (pinhole intrinsics, patch tables; GeomConst.fast) — otherwise the generic fma chains.
// RPW: rounds of 64 points per wavefront; GB = GW * RPW * 64 points per workgroup.
// PLAIN: every pixel of every frame, patch from the pixel, device alpha, no token-cache columns, no point log (the dense
// build): idx, p_patf, p_r2f, alpha_in and g_cell are null and their code is compiled out.
// REC8 (with PLAIN): 8-byte records {rgb, index inside the block, depth offset} instead of {alpha f64, rgb} — alpha is evaluated
// by the rgb chain (geometry_dev.h rec8_alpha: the same instructions on the same operands), not here.
// Four workgroups (16 wavefronts) per CU up to 2048-point blocks: without the bound the 8-byte-record form settled at 131 registers
// — three wavefronts per SIMD — although 95 do without a spill (-DBSC_POINTS_MIN_BLOCKS=1: the compiler's own choice)
#ifndef BSC_POINTS_MIN_BLOCKS
#define BSC_POINTS_MIN_BLOCKS 4
#endif
#ifndef BSC_POINTS_MIN_BLOCKS_PLAIN8
#define BSC_POINTS_MIN_BLOCKS_PLAIN8 4      // (6 — 80 registers, 24.6 KB of LDS — fits a CU but spills five registers and is slower: 5.05 against 4.91 ms per call)
#endif
template <bool FAST, int RPW, bool PLAIN, bool REC8>
__global__ __launch_bounds__(TPB, RPW <= 8 ? (PLAIN && REC8 ? BSC_POINTS_MIN_BLOCKS_PLAIN8 : BSC_POINTS_MIN_BLOCKS) : 1) void k_points(GeomConst gc, const float *__restrict__ depth,
                                                const uint8_t *__restrict__ rgb, int rgb_ch,
                                                const int32_t *__restrict__ idx_, const int64_t *__restrict__ offsets,
                                                int n_frames, const double *__restrict__ transforms,
                                                const double *__restrict__ alpha_in_, int64_t P, float inv_w, int cap_log2,
                                                int32_t *occ, int32_t *__restrict__ p_cell, uint32_t *__restrict__ p_patf_,
                                                void *__restrict__ p_rec, float *__restrict__ p_r2f_,
                                                int32_t *__restrict__ new_cells, int64_t *dscal,
                                                int32_t *__restrict__ blk_runs, int32_t *__restrict__ blk_pass,
                                                uint32_t *__restrict__ stage_cell, uint32_t *__restrict__ stage_pos,
                                                int32_t *__restrict__ g_cell_)
{
    constexpr int GB = GW * RPW * 64;
    constexpr int EPT = GROUP_HS / TPB;         // slots per thread in the prefix pass
    constexpr int RECW = REC8 ? 2 : 3;          // 32-bit words per record
    constexpr int WORD_BYTES = GW * (GROUP_HS + 1) * 8, REC_BYTES = 4 * RECW * GB;
    static_assert(!REC8 || PLAIN, "8-byte records belong to the every-pixel dense build");
    const int32_t *__restrict__ idx = PLAIN ? nullptr : idx_;
    const double *__restrict__ alpha_in = PLAIN ? nullptr : alpha_in_;
    uint32_t *__restrict__ p_patf = PLAIN ? nullptr : p_patf_;
    float *__restrict__ p_r2f = PLAIN ? nullptr : p_r2f_;
    int32_t *__restrict__ g_cell = PLAIN ? nullptr : g_cell_;
    __shared__ uint32_t s_key[GROUP_HS];
    __shared__ uint32_t s_first[GROUP_HS + 1];           // first point (index inside the block) of the slot's cell
    // the per-wavefront ballot words live until the last round; the record staging starts after it: one buffer
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[WORD_BYTES > REC_BYTES ? WORD_BYTES : REC_BYTES];
    __shared__ uint16_t s_cnt[GW][GROUP_HS + 1];          // + a spare entry for the lanes without a slot (16 bits: at most GB points;
                                                          // with 32-bit counts six workgroups did not fit a CU's 160 KB)
    __shared__ uint32_t s_wsum[GW];
    __shared__ int32_t s_ovf[GW];
    __shared__ double2 s_exp[REC8 ? 1 : 64];              // 2^(j/64) as (hi, lo): bsc_exp's table (8-byte records: alpha is the chain's)
    if (!REC8 && threadIdx.x < 64) s_exp[threadIdx.x] = gc.exp_tab[threadIdx.x];
    u64(*s_word)[GROUP_HS + 1] = (u64(*)[GROUP_HS + 1])s_raw;
    uint32_t *s_rec = (uint32_t *)s_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BSC_POINTS_PROFILE
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = clock64();
#define PT_T(k) { const long long now_ = clock64(); tph[k] += now_ - tq; tq = now_; }
#else
#define PT_T(k)
#endif
    for (int i = tid; i < GROUP_HS; i += TPB) { s_key[i] = 0xffffffffu; s_first[i] = 0xffffffffu; }
    for (int i = tid; i < GW * (GROUP_HS + 1); i += TPB) { (&s_word[0][0])[i] = 0ull; (&s_cnt[0][0])[i] = (uint16_t)0; }
    __syncthreads();
    PT_T(6)

    const int32_t N = gc.H * gc.W;
    const int64_t blk_base = (int64_t)blockIdx.x * GB;
    // frame / pixel of the wavefront's first point (all-pixel ingest); P <= max_points < 2^31
    const uint32_t jw = (uint32_t)blk_base + (uint32_t)(wv * RPW * 64);
    const int fw = (int)(jw / (uint32_t)N);
    const int32_t iw = (int32_t)(jw - (uint32_t)fw * (uint32_t)N);
    // pc_transform of the lane's current frame, in vector registers: loaded once per wavefront, again by the lanes that
    // cross into the next frame
    double T[12];
    int fT = fw < n_frames ? fw : n_frames - 1;
    {
        const double *Tv = transforms + 16 * (int64_t)fT;
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = Tv[k];
    }
    int32_t cells[RPW];
    uint32_t sr[RPW], ralo[RPW], rahi[RPW];
    int ovf_cnt = 0;
    PT_T(7)
    // Frame, pixel and depth of the lane's point of EVERY round first: the rounds below then start from registers.  (Loaded round
    // by round, each round sat out three memory round trips in a row — its depth, the completion of its cell store ahead of the
    // colour gather, the gather itself: 4.7 k clocks per round for ~1 k clocks of instruction issue.)
    int fr[RPW];
    int32_t ir[RPW];
    float zr[RPW];
    uint32_t xy[RPW];                               // fast geometry: pixel (x | y << 16)
    uint32_t tpx[RPW], tpy[RPW];                    // patch column / row of the pixel (raw table bytes, one register each: combining or
                                                    // packing them here would wait for the loads)
    // The "outside every patch" default of a lane past the last point comes out of an opaque register: with the literal 255 the
    // compiler turns `phi(255, load) != 255` (the patch test of the geometry) into `phi(false, load != 255)` and evaluates the
    // comparison right behind the load — a full memory round trip per round, eight in a row (12 k of a wavefront's 49 k clocks).
    uint32_t pat_none = 255u;
    asm volatile("" : "+v"(pat_none));
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t j = blk_base + wv * RPW * 64 + r * 64 + lane;
        fr[r] = 0; ir[r] = 0; zr[r] = 0.f; xy[r] = 0u; tpx[r] = pat_none; tpy[r] = pat_none;
        if (j < P) {
            int f;
            int32_t i;
            if (idx) {
                f = frame_of(offsets, n_frames, j);
                i = idx[j];
            } else {
                // a wavefront's RPW * 64 consecutive points straddle at most one frame boundary when N >= GB (smaller frames
                // take the per-thread division)
                int fb = fw;
                int32_t ib = iw + r * 64 + lane;
                if (PLAIN) {
                    // (the host takes PLAIN only when a frame is a whole number of wavefront slices: no wavefront crosses a frame)
                } else if (N >= GB) {
                    if (ib >= N) { ib -= N; ++fb; }
                } else {
                    fb += ib / N;
                    ib = ib % N;
                }
                f = fb;
                i = ib;
            }
            fr[r] = f; ir[r] = i;
            zr[r] = depth[(int64_t)f * N + i];
            if (FAST && !PLAIN) {
                // y = i / W without an integer division: float estimate (exact operands below 2^24), corrected by one
                int32_t y = (int32_t)((float)i * inv_w);
                int32_t x = i - y * gc.W;
                if (x < 0) { --y; x += gc.W; } else if (x >= gc.W) { ++y; x -= gc.W; }
                xy[r] = (uint32_t)x | ((uint32_t)y << 16);
                if (!PLAIN) {
                    tpx[r] = gc.pat_x[x];
                    tpy[r] = gc.pat_y[y];
                }
            }
        }
    }
    uint32_t raw0[RPW];                             // colour gathers, consumed after the rounds
    // ONE 32-bit gather per point whatever the frame format: RGBA pixels are aligned words; the three bytes of an RGB pixel come
    // with the byte behind them (an unaligned word: global memory takes it), except at the very end of the buffer, where the word
    // starts one byte early.  With two gathers + a wait on the RGB path and one on the RGBA path the compiler could not count
    // the loads in flight at the head of the next round and waited for all of them — the colour gather's round trip, every round.
    const int64_t rgb_last = (int64_t)n_frames * gc.H * gc.W * rgb_ch - 4;
    PT_T(0)
    // PLAIN: the pixel of the lane's point walks along with the rounds (64 pixels on, at most one row down: the host takes PLAIN only
    // for frames at least 64 pixels wide) — two registers instead of one per round
    int32_t run_x = 0, run_y = 0;
    if (PLAIN) {
        const int32_t i0 = iw + lane;
        run_y = (int32_t)((float)i0 * inv_w);
        run_x = i0 - run_y * gc.W;
        if (run_x < 0) { --run_y; run_x += gc.W; } else if (run_x >= gc.W) { ++run_y; run_x -= gc.W; }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int p_local = wv * RPW * 64 + r * 64 + lane;
        const int64_t j = blk_base + p_local;
        int32_t cell = -2;
        ralo[r] = rahi[r] = 0u;
        int64_t pix_off = 0;                        // byte offset of the colour the point samples (0: none — a harmless read)
        uint32_t tail = 0u;
        if (j < P) {
            const int f = fr[r];
            const int32_t i = ir[r];
            const float z = zr[r];
            if (!PLAIN && __ballot(f != fT)) {      // wave-uniform: the transform is re-read only by a wavefront that crosses a frame
                if (f != fT) {
                    const double *Tv = transforms + 16 * (int64_t)f;
#pragma unroll
                    for (int k = 0; k < 12; ++k) T[k] = Tv[k];
                    fT = f;
                }
                // the reload completes HERE, in the rare block: left pending, the registers of T made the compiler wait for every
                // load in flight (vmcnt(0)) at the head of the next round — the colour gather's round trip, once per round
                __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0)
            }
            int32_t sx = 0, sy = 0;
            uint32_t patch = 0;
            double r2 = 0.0, alpha = 0.0;
            cell = -1;
            if (FAST) {
                GeomFastOut o;
                // PLAIN: the patch is not recorded and no pixel lies outside the patch grid (the host checked): no table look-ups
                geom_point_fast_t(gc, PLAIN ? run_x : (int32_t)(xy[r] & 0xffffu), PLAIN ? run_y : (int32_t)(xy[r] >> 16), z, T, o, !REC8 && alpha_in == nullptr,
                                  PLAIN ? 0u : (uint32_t)tpx[r], PLAIN ? 0u : (uint32_t)tpy[r], s_exp);
                cell = o.cell;
                sx = o.sx; sy = o.sy; patch = o.patch; r2 = o.r2; alpha = o.alpha;
            } else {
                GeomOut o;
                geom_point(gc, i, z, T, o, alpha_in == nullptr, s_exp);
                if (o.flags == 7u) {
                    const int32_t row = o.vox[0], col = o.vox[1], h = o.vox[2] - gc.min_h;   // memory_2.py:867
                    cell = (row * gc.gs + col) * gc.nh + h;
                    sx = o.pix[0]; sy = o.pix[1];            // memory_2.py:870 rgb[py, px]: negative indices wrap
                    if (sx < 0) sx += gc.W;
                    if (sy < 0) sy += gc.H;
                    sx = min(max(sx, 0), gc.W - 1);
                    sy = min(max(sy, 0), gc.H - 1);
                    patch = (uint32_t)(o.pat[1] * gc.g + o.pat[0]);                          // tokens[py, px]
                    r2 = o.r2; alpha = o.alpha;
                }
            }
            if (p_patf || p_r2f || alpha_in) {
                if (cell >= 0) {
                    if (p_patf) p_patf[j] = ((uint32_t)f << 16) | patch;
                    if (p_r2f) p_r2f[j] = (float)r2;      // memory_2.py:885 grid_feat_dis is float32 (token cache only)
                    if (alpha_in) alpha = alpha_in[j];
                }
            }
            // selects, not a branch: the record of a point outside the grid is never written, so its alpha bits may be anything
            pix_off = cell >= 0 ? ((int64_t)f * N + (int64_t)sy * gc.W + sx) * rgb_ch : 0;
            if (!REC8) {
                ralo[r] = (uint32_t)__double2loint(alpha);
                rahi[r] = (uint32_t)__double2hiint(alpha);
            }
        }
        // the colour gather of every lane, unconditionally and unprocessed: nothing below needs it before the records are written,
        // so its round trip runs under the following rounds
        {
            const int64_t off4 = pix_off < rgb_last ? pix_off : rgb_last;
            tail = pix_off > off4 ? 1u : 0u;             // (rides in bit 30 of sr[r]: the word holds the colour one byte up)
            uint32_t word;
            __builtin_memcpy(&word, rgb + off4, 4);
            raw0[r] = word;
        }
        cells[r] = cell;
        if (PLAIN) {
            run_x += 64;
            if (run_x >= gc.W) { run_x -= gc.W; ++run_y; }
        }
        PT_T(1)
        // ---- slot of the cell, rank of the point among the wavefront's points of that cell --------------------------------
        uint32_t e = GROUP_OVF;
        if (cell >= 0) {
            uint32_t h = ((uint32_t)cell * 2654435761u) >> (32 - GROUP_HS_LOG2);
#pragma unroll 1
            for (int t = 0; t < GROUP_PROBES; ++t) {
                uint32_t k = __hip_atomic_load(&s_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (k == 0xffffffffu) {
                    k = atomicCAS(&s_key[h], 0xffffffffu, (uint32_t)cell);
                    if (k == 0xffffffffu) k = (uint32_t)cell;
                }
                if (k == (uint32_t)cell) { e = h; break; }
                h = (h + 1) & (GROUP_HS - 1);
            }
        }
        const bool grouped = e != GROUP_OVF;
        const uint32_t slot = grouped ? e : (uint32_t)GROUP_HS;        // lanes without a slot meet in the spare entry (OR 0)
        atomicOr((unsigned long long *)&s_word[wv][slot], grouped ? 1ull << lane : 0ull);
        wave_lds_order();
        const u64 word = __hip_atomic_load(&s_word[wv][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const uint32_t before = (uint32_t)__hip_atomic_load(&s_cnt[wv][slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        wave_lds_order();
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(word >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)word, 0u));
        uint32_t lr = before + rank;
        if (grouped && rank == 0u) {             // the cell's first lane of the round: count the round, clear the word
            __hip_atomic_store(&s_cnt[wv][slot], (uint16_t)(before + (uint32_t)__popcll(word)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_store(&s_word[wv][slot], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (before == 0u) atomicMin(&s_first[slot], (uint32_t)p_local);     // the wavefront's first point of the cell
        }
        wave_lds_order();
        const u64 om = __ballot(cell >= 0 && !grouped);
        if (!grouped) lr = (uint32_t)ovf_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
        ovf_cnt += __popcll(om);
        sr[r] = cell >= 0 ? (e | (lr << 16) | (tail << 30)) : 0xffffffffu;       // lr <= GB < 2^13
        PT_T(2)
    }
    // the cells of all rounds leave together: a store inside the rounds shares the vmcnt counter with the loads still in flight
    // (returns are ordered among loads, not between loads and stores), and every counted wait of a later round degraded to
    // "wait for everything", the store's own round trip included
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t j = blk_base + wv * RPW * 64 + r * 64 + lane;
        if (j < P) p_cell[j] = cells[r] == -2 ? -1 : cells[r];
    }
    __syncthreads();
    PT_T(3)
    // ---- group sizes -> positions: exclusive prefix over the slots (points | runs << 16), then over the wavefronts ----------
    if (lane == 0) s_ovf[wv] = ovf_cnt;
    uint32_t v[EPT], tv = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid * EPT + k;
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < GW; ++w) tot += s_cnt[w][e];
        const uint32_t nr = (tot + (1u << cap_log2) - 1u) >> cap_log2;     // a run's length has to fit its key bits
        v[k] = tot | (nr << 16);
        tv += v[k];
    }
    // first-touch claims, ONE per cell of the block (its first point), instead of one per stretch of same-cell lanes
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid * EPT + k;
        const bool has = (v[k] & 0xffffu) != 0u;
        claim_cells(has, has ? (int32_t)s_key[e] : 0, blk_base + (has ? s_first[e] : 0u), occ, new_cells, dscal, lane);
    }
    const uint32_t incl = wave_incl_sum_u32(tv);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    uint32_t prefix = incl - tv, total = 0;
#pragma unroll
    for (int w = 0; w < GW; ++w) {
        const uint32_t ws = s_wsum[w];
        if (w < wv) prefix += ws;
        total += ws;
    }
    const int64_t stage_base = blk_base;         // the block's staging slice: at most one run per point
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid * EPT + k;
        const uint32_t base = prefix & 0xffffu, rbase = prefix >> 16, nr = v[k] >> 16;
        uint32_t acc = base;
#pragma unroll
        for (int w = 0; w < GW; ++w) { const uint32_t c = s_cnt[w][e]; s_cnt[w][e] = (uint16_t)acc; acc += c; }
        for (uint32_t i = 0; i < nr; ++i) {
            stage_cell[stage_base + rbase + i] = s_key[e];
            stage_pos[stage_base + rbase + i] = (uint32_t)blk_base + base + (i << cap_log2);
        }
        prefix += v[k];
    }
    const uint32_t n_grouped = total & 0xffffu, n_gruns = total >> 16;
    __syncthreads();
    uint32_t ovf_base = n_grouped, n_ovf = 0;
#pragma unroll
    for (int w = 0; w < GW; ++w) {
        const uint32_t c = (uint32_t)s_ovf[w];
        if (w < wv) ovf_base += c;
        n_ovf += c;
    }
    const uint32_t n_valid = n_grouped + n_ovf;
    PT_T(4)
    // ---- records into the block's slice, group by group (through LDS: the global stores are whole lines) -------------------
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const bool valid = sr[r] != 0xffffffffu;
        const uint32_t e = sr[r] & 0xffffu, lr = (sr[r] >> 16) & 0x3fffu;
        const bool ovf = valid && e == GROUP_OVF;
        uint32_t pos = 0;
        if (valid) pos = ovf ? ovf_base + lr : s_cnt[wv][e] + lr;
        if (n_ovf) {            // (uniform) points without a slot: runs of one, claimed point by point
            if (ovf) {
                stage_cell[stage_base + n_gruns + (pos - n_grouped)] = (uint32_t)cells[r];
                stage_pos[stage_base + n_gruns + (pos - n_grouped)] = (uint32_t)blk_base + pos;
            }
            claim_cells(ovf, cells[r], blk_base + wv * RPW * 64 + r * 64 + lane, occ, new_cells, dscal, lane);
        }
        if (valid) {
            const uint32_t rgbv = (raw0[r] >> ((sr[r] >> 27) & 8u)) & 0xffffffu;
            if (REC8) {
                uint32_t lo, hi;
                rec8_pack(gc, rgbv, (uint32_t)(wv * RPW * 64 + r * 64 + lane), zr[r], lo, hi);
                s_rec[2 * pos] = lo; s_rec[2 * pos + 1] = hi;
            } else {
                s_rec[3 * pos] = ralo[r]; s_rec[3 * pos + 1] = rahi[r]; s_rec[3 * pos + 2] = rgbv;
            }
            if (g_cell) g_cell[blk_base + pos] = cells[r];
        }
    }
    __syncthreads();
    {
        uint32_t *dst = (uint32_t *)p_rec + (int64_t)RECW * blk_base;       // 12 (8) * GB bytes per block: 16-byte aligned
        const uint32_t nd = (uint32_t)RECW * n_valid, nq = nd >> 2;
        for (uint32_t i = tid; i < nq; i += TPB) ((uint4 *)dst)[i] = ((const uint4 *)s_rec)[i];
        if (tid < (int)(nd & 3u)) dst[4 * nq + tid] = s_rec[4 * nq + tid];
        if (g_cell) {
            const int64_t end = P - blk_base < GB ? P - blk_base : GB;
            for (int64_t i = n_valid + tid; i < end; i += TPB) g_cell[blk_base + i] = -1;
        }
    }
    if (tid == 0) { blk_runs[blockIdx.x] = (int32_t)(n_gruns + n_ovf); blk_pass[blockIdx.x] = (int32_t)n_valid; }
    PT_T(5)
#ifdef BSC_POINTS_PROFILE
    if (blockIdx.x == 1000 && lane == 0 && (wv == 0 || wv == 3))
        printf("k_points wave %d: clear+barrier %lld transform %lld loads %lld geometry %lld slot/rank %lld barrier %lld prefix/claims %lld records %lld (clocks, %d rounds)\n", wv,
               tph[6], tph[7], tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], RPW);
#endif
#undef PT_T
}

// scalars of the batch, on the device: run / passing-point totals from the block scans, the new voxels' id range
__global__ void k_totals(int64_t P, int64_t nblk, const int32_t *blk_runs, const int32_t *blk_run_off, const int32_t *blk_pass,
                         const int32_t *blk_pass_off, int64_t *dscal, int vcap, int64_t *bscal)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t npass = (int64_t)blk_pass_off[nblk - 1] + blk_pass[nblk - 1];
    int64_t nnew = dscal[DS_B_NNEW];
    dscal[DS_B_NPASS] = npass;
    dscal[DS_MAX_ID_PREV] = dscal[DS_MAX_ID];
    if (dscal[DS_MAX_ID] + nnew > vcap) { nnew = vcap - dscal[DS_MAX_ID]; dscal[DS_ERROR] = 1; }
    dscal[DS_B_NFIRST] = nnew;
    dscal[DS_MAX_ID] += nnew;
    dscal[DS_NPASS_TOTAL] += npass;
    dscal[DS_NSEEN_TOTAL] += P;
    dscal[DS_B_NRUN] = (int64_t)blk_run_off[nblk - 1] + blk_runs[nblk - 1];
    dscal[DS_B_NPSEG] = 0;
    bscal[0] = 0;                       // voxel segments of the point order (k_expand)
    bscal[1] = dscal[DS_MAX_ID_PREV];
    bscal[2] = 0;                       // points in the per-voxel order
    bscal[3] = 0;                       // segment queue of the rgb chain (quads: short segments)
    bscal[4] = 0;                       // long segments (k_seg_order): the first bscal[4] of the length-ordered list
    bscal[5] = 0;                       // (unused)
    bscal[6] = 0;                       // hot segments (k_seg_order): the first bscal[6] of the long ones, a workgroup each
}

// ---- ids of the new voxels ---------------------------------------------------------------------------------------
// new cell -> (winning point j, cell); sorted by j the list is in first-touch order
// Round 6: the two block scans (runs, passing points) and k_totals as ONE launch of one workgroup — five launches in a row sat
// between k_points and everything behind it (two rocPRIM look-back scans of 1.2e5 counts, each an init + a scan kernel, then a
// one-thread kernel: 0.03 ms alone, 0.2-0.3 ms inside the encoder pipeline) — and the scalars the host sizes the back end from leave
// through a mailbox in pinned host memory (system-scope stores + a sequence number the host polls) instead of an event, a copy
// kernel on a third stream and a stream synchronize: the order stage used to start 0.2-0.3 ms behind k_totals.
#define BT_THREADS 1024
__global__ __launch_bounds__(BT_THREADS) void k_block_totals(int64_t P, int64_t nblk, const int32_t *__restrict__ blk_runs,
                                                           int32_t *__restrict__ blk_run_off, const int32_t *__restrict__ blk_pass,
                                                           int32_t *__restrict__ blk_pass_off, int64_t *dscal, int vcap, int64_t *bscal,
                                                           int64_t *mail, int64_t seq)
{
    // an iteration takes BT_THREADS x 16 counts as FOUR rounds of BT_THREADS x 4: a thread's 16-byte loads and stores of a round
    // are contiguous with its neighbours' (whole lines per instruction; with 16 consecutive counts per thread every instruction
    // touched 64 lines and the one CU's address path made the launch 0.2 ms long)
    __shared__ uint32_t s_w[2][4][BT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t carry_r = 0, carry_p = 0;
    const int64_t n4 = (nblk + 3) >> 2;                 // the arrays hold nblk_cap >= nblk + 3 entries (bsc_create pads them)
    for (int64_t base4 = 0; base4 < n4; base4 += 4 * BT_THREADS) {
        uint4 r[4], q[4];
        uint32_t sr[4], sq[4], ir[4], iq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t v = base4 + k * BT_THREADS + tid, vc = v < n4 ? v : n4 - 1;
            r[k] = ((const uint4 *)blk_runs)[vc];
            q[k] = ((const uint4 *)blk_pass)[vc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t e0 = 4 * (base4 + k * BT_THREADS + tid);
            r[k].x = e0 < nblk ? r[k].x : 0u; r[k].y = e0 + 1 < nblk ? r[k].y : 0u; r[k].z = e0 + 2 < nblk ? r[k].z : 0u; r[k].w = e0 + 3 < nblk ? r[k].w : 0u;
            q[k].x = e0 < nblk ? q[k].x : 0u; q[k].y = e0 + 1 < nblk ? q[k].y : 0u; q[k].z = e0 + 2 < nblk ? q[k].z : 0u; q[k].w = e0 + 3 < nblk ? q[k].w : 0u;
            sr[k] = r[k].x + r[k].y + r[k].z + r[k].w;
            sq[k] = q[k].x + q[k].y + q[k].z + q[k].w;
            ir[k] = wave_incl_sum_u32(sr[k]);
            iq[k] = wave_incl_sum_u32(sq[k]);
            if (lane == 63) { s_w[0][k][wv] = ir[k]; s_w[1][k][wv] = iq[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t br = carry_r + ir[k] - sr[k], bq = carry_p + iq[k] - sq[k], tr = 0, tq = 0;
#pragma unroll
            for (int w = 0; w < BT_THREADS / 64; ++w) {
                const uint32_t a = s_w[0][k][w], b = s_w[1][k][w];
                if (w < wv) { br += a; bq += b; }
                tr += a; tq += b;
            }
            carry_r += tr; carry_p += tq;
            const int64_t v = base4 + k * BT_THREADS + tid;
            if (v < n4) {
                uint4 o, u;
                o.x = br; o.y = br + r[k].x; o.z = o.y + r[k].y; o.w = o.z + r[k].z;
                u.x = bq; u.y = bq + q[k].x; u.z = u.y + q[k].y; u.w = u.z + q[k].z;
                ((uint4 *)blk_run_off)[v] = o;
                ((uint4 *)blk_pass_off)[v] = u;
            }
        }
        __syncthreads();
    }
    if (tid != 0) return;
    // what k_totals did (memory_2.py:888-894: the new voxels' id range), from the scan totals — on a register copy of the scalars:
    // every value is read once, up front, and leaves twice (device array, host mailbox) without a load in between (interleaved,
    // each mailbox store waited for the one before it to be acknowledged by the host: 18 x ~10 us)
    int64_t d[DS_COUNT];
#pragma unroll
    for (int k = 0; k < DS_COUNT; ++k) d[k] = dscal[k];
    const int64_t npass = carry_p;
    int64_t nnew = d[DS_B_NNEW];
    d[DS_B_NPASS] = npass;
    d[DS_MAX_ID_PREV] = d[DS_MAX_ID];
    if (d[DS_MAX_ID] + nnew > vcap) { nnew = vcap - d[DS_MAX_ID]; d[DS_ERROR] = 1; }
    d[DS_B_NFIRST] = nnew;
    d[DS_MAX_ID] += nnew;
    d[DS_NPASS_TOTAL] += npass;
    d[DS_NSEEN_TOTAL] += P;
    d[DS_B_NRUN] = carry_r;
    d[DS_B_NPSEG] = 0;
#pragma unroll
    for (int k = 0; k < DS_COUNT; ++k) dscal[k] = d[k];
    bscal[0] = 0;                       // voxel segments of the point order (k_expand)
    bscal[1] = d[DS_MAX_ID_PREV];
    bscal[2] = 0;                       // points in the per-voxel order
    bscal[3] = 0;                       // segment queue of the rgb chain (quads: short segments)
    bscal[4] = 0;                       // long segments (k_seg_order): the first bscal[4] of the length-ordered list
    bscal[5] = 0;                       // (unused)
    bscal[6] = 0;                       // hot segments (k_seg_order): the first bscal[6] of the long ones, a workgroup each
    if (mail) {
        // write-through 8-byte stores to the host, EVERY word tagged with the call's sequence number (value << 16 | seq: the scalars
        // are counts below 2^47): the host takes the mailbox once all words carry the tag, whatever order the fabric delivered them
        // in — no release fence (at system scope it writes back the whole L2), no ordering assumed between posted writes
#pragma unroll
        for (int k = 0; k < DS_COUNT; ++k)
            __hip_atomic_store(&mail[k], (int64_t)(((u64)d[k] << 16) | ((u64)seq & 0xffffull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(TPB) void k_new_keys(int64_t n, const int32_t *__restrict__ new_cells, const int32_t *__restrict__ occ,
                                                  uint32_t *__restrict__ key, uint32_t *__restrict__ val)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int32_t c = new_cells[i];
    key[i] = (uint32_t)(occ[c] - INT_MIN);
    val[i] = (uint32_t)c;
}

// memory_2.py:888-894: id = max_id++ in first-touch order; grid_rgb_pos[id] = [row, col, h]
__global__ __launch_bounds__(TPB) void k_new_assign(int64_t n, int64_t n_ok, const uint32_t *__restrict__ cells_sorted, int32_t *occ,
                                                    const int64_t *dscal, int gs, int nh, int32_t *__restrict__ rgb_pos)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int32_t c = (int32_t)cells_sorted[i];
    if (i >= n_ok) return;              // capacity: the cell keeps its provisional (negative) value
    const int64_t id = dscal[DS_MAX_ID_PREV] + i;
    occ[c] = (int32_t)id;
    const int32_t h = c % nh, rc = c / nh;
    rgb_pos[3 * id + 0] = rc / gs;
    rgb_pos[3 * id + 1] = rc % gs;
    rgb_pos[3 * id + 2] = h;
}

// ---- runs -----------------------------------------------------------------------------------------------------------
// The runs k_points staged per block (cell, first position in p_rec; a block's runs tile its slice [base, base + passing
// points) in staging order, so a run's length is the distance to the next one) -> sort key = voxel id | (length - 1) << vb
// (id field all ones: no voxel — over capacity), value = first position.  One wavefront per block of points; the ids exist
// by now (k_new_assign).
template <int GB>
__global__ __launch_bounds__(TPB) void k_run_keys(int64_t nblk, int vb, const uint32_t *__restrict__ stage_cell,
                                                  const uint32_t *__restrict__ stage_pos, const int32_t *__restrict__ occ,
                                                  const int32_t *__restrict__ blk_runs, const int32_t *__restrict__ blk_run_off,
                                                  const int32_t *__restrict__ blk_pass, uint32_t *__restrict__ rkey,
                                                  uint32_t *__restrict__ rval)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const int n = blk_runs[b];
    const int32_t off = blk_run_off[b];
    const uint32_t end = (uint32_t)(b * GB) + (uint32_t)blk_pass[b];
    const uint32_t vmask = vb >= 32 ? 0xffffffffu : ((1u << vb) - 1u);
    const uint32_t *sc = stage_cell + b * GB, *sp = stage_pos + b * GB;
    for (int i = lane; i < n; i += 64) {
        const uint32_t pos = sp[i], nxt = i + 1 < n ? sp[i + 1] : end;
        const int32_t v = occ[sc[i]];
        rkey[off + i] = v >= 0 ? ((uint32_t)v | ((nxt - pos - 1u) << vb)) : vmask;
        rval[off + i] = pos;
    }
}

// exact mode: the passing points of the batch, listed in order j (the token cache fills in that order, memory_2.py:878-886)
template <int GB>
__global__ __launch_bounds__(TPB) void k_pass_list(int64_t P, const int32_t *__restrict__ p_cell,
                                                   const int32_t *__restrict__ blk_pass_off, int32_t *__restrict__ pass_list)
{
    constexpr int PPT = GB / TPB;
    __shared__ u64 s_pbits[GB / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bool pass[PPT];
    // word w = r * 4 + wave covers the points [64 w, 64 w + 64) of the block
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int64_t j = (int64_t)blockIdx.x * GB + r * TPB + threadIdx.x;
        pass[r] = j < P && p_cell[j] >= 0;
        const u64 pb = __ballot(pass[r]);
        if (lane == 0) s_pbits[r * (TPB / 64) + wid] = pb;
    }
    __syncthreads();
    const int32_t pass_base = blk_pass_off[blockIdx.x];
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        if (!pass[r]) continue;
        const int w = r * (TPB / 64) + wid;
        int rank = __popcll(s_pbits[w] & ((1ull << lane) - 1ull));
        for (int k = 0; k < w; ++k) rank += __popcll(s_pbits[k]);
        pass_list[pass_base + rank] = (int32_t)((int64_t)blockIdx.x * GB + r * TPB + threadIdx.x);
    }
}

// memory_2.py:888-903 — the rgb chain c' = trunc((f32(c*w) + r*a) / (w + a)), w' = f32(w + a) is sequential by
// definition (truncation and f32 rounding at every step), so the only parallelism is across voxels and channels, and
// the call's duration is bounded below by its longest voxel run (tens of thousands of points for a wall seen in every
// frame) times the latency of one step.  What can be minimised is the issue bandwidth the chain takes from the kernels
// it runs beside (it lives on the side stream):
//   * a QUAD of lanes per voxel (R, G, B + one spare), 16 voxels per wavefront, so one instruction advances 16 chains;
//   * quads pull voxel segments from a queue (one atomic per wave and refill), so no quad idles while work remains;
//   * every quad streams its segment in chunks of 64 points, 16 per lane: the order indices of chunk n+2 and the point
//     records of chunk n+1 are in flight while chunk n is stepped out of registers through quad-broadcast DPP moves;
//   * the divide is the hardware's own f64 sequence (v_rcp_f64, two Newton steps, multiply, residual fma, final fma —
//     what `/` compiles to, without the scale/fixup ops that only act outside the normal range) written out so that
//     the reciprocal half, which depends on the weights only, stays off the colour's dependency chain.
// The same quad settles the top-down map: `h >= max_height` in sequential order == max over (h, order), and a voxel's
// latest point is the last element of its segment, so one atomicMax per voxel replaces one per point.
#define CQ 16                                   // records per lane per chunk (64 per quad)
template <int Q> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, Q * 0x55, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ uint32_t quad_perm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
}

struct ChainRegs {
    uint32_t rv[CQ];        // rgb bytes | bit 30 = slot holds a point of the segment | bit 31 = its step is still to run
    uint32_t alo[CQ], ahi[CQ];
};

// order indices of the chunk starting at position `pos` of the per-voxel point order (0xffffffff = past the segment's
// end k1).  The loads are unconditional (clamped addresses) so that all of them issue back to back.
__device__ __forceinline__ void chain_load_idx(uint32_t (&J)[CQ], bool on, int64_t pos, int q, int64_t k1,
                                               const uint32_t *__restrict__ sj)
{
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const int64_t k = pos + 4 * i + q;
        const bool in = on && k < k1;
        const uint32_t j = sj[in ? k : 0];
        J[i] = in ? j : 0xffffffffu;
    }
}

template <bool REC8>
__device__ __forceinline__ void chain_load_rec(ChainRegs &R, const uint32_t (&J)[CQ], const void *__restrict__ p_rec,
                                               uint32_t &last_j)
{
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const bool valid = J[i] != 0xffffffffu;
        if (REC8) {
            const uint2 raw = ((const uint2 *)p_rec)[valid ? J[i] : 0u];       // { rgb | index in block, index | depth offset }: 8 bytes,
            R.alo[i] = raw.x; R.ahi[i] = raw.y;                                // decoded by chain_decode once the load has landed
            R.rv[i] = valid ? 0xc0000000u : 0u;
        } else {
            const PointRec raw = ((const PointRec *)p_rec)[valid ? J[i] : 0u]; // { alpha lo, alpha hi, rgbv }: 12 bytes
            R.alo[i] = raw.alo; R.ahi[i] = raw.ahi;
            R.rv[i] = valid ? (raw.rgbv | 0xc0000000u) : 0u;
        }
        last_j = valid ? J[i] : last_j;                                        // order indices grow along a segment
    }
}

// 8-byte records -> {alpha, rgb} in place (geometry_dev.h rec8_alpha; J[i] is the record's position).  Slots without a point
// decode whatever record 0 holds: their step never runs.
template <bool REC8>
__device__ __forceinline__ void chain_decode(ChainRegs &R, const uint32_t (&J)[CQ], const GeomConst &gc, const double2 *exp_tab)
{
    if (!REC8) return;
#pragma unroll
    for (int i = 0; i < CQ; ++i) {
        const uint32_t lo = R.alo[i], hi = R.ahi[i];
        const double a = rec8_alpha(gc, lo, hi, J[i], exp_tab);
        R.rv[i] |= lo & 0xffffffu;
        R.alo[i] = (uint32_t)__double2loint(a); R.ahi[i] = (uint32_t)__double2hiint(a);
    }
}

#define CHAIN_STEP(i, qq)                                                                          \
    {                                                                                              \
        const uint32_t rvb = quad_bcast<qq>(R.rv[i]);                                              \
        const double a = __hiloint2double((int)quad_bcast<qq>(R.ahi[i]), (int)quad_bcast<qq>(R.alo[i])); \
        const bool act = (int32_t)rvb < 0;                                                         \
        const uint32_t r = (rvb >> sh) & 0xffu;                                                    \
        const double den = (double)w + a;                       /* :896 weight + alpha (f32 + f64) */ \
        double rd = __builtin_amdgcn_rcp(den);                                                     \
        double e = fma(-den, rd, 1.0); rd = fma(rd, e, rd);                                        \
        e = fma(-den, rd, 1.0); rd = fma(rd, e, rd);                                               \
        const double num = (double)((float)c * w) + (double)r * a;   /* u8*f32 -> f32 ; u8*f64 -> f64 */ \
        const double q0 = num * rd;                                                                \
        const double rr = fma(-den, q0, num);                                                      \
        const double v = fma(rr, rd, q0);                       /* == num / den, correctly rounded */ \
        c = act ? (uint32_t)v : c;                              /* truncating uint8 store */       \
        w = act ? (float)den : w;                               /* :899 */                         \
    }

#define LONG_MIN_LOG2 6                          // segments of >= 64 points
#define LONG_EARLY 512                           // points of a new voxel that the quad chain steps first
#define HOT_MIN_LOG2 15                          // segments of >= 32768 points are split over the wavefronts of a workgroup
#define CHAIN_WG 256             // 4 wavefronts per workgroup, one per SIMD of a CU
#define CHAIN_WAVES 512
template <bool REC8>
__global__ __launch_bounds__(CHAIN_WG) void k_chain(const GeomConst gc, const uint32_t *__restrict__ sj, int64_t *bscal,
                                              const int4 *__restrict__ seg_info,
                                              const void *__restrict__ p_rec,
                                              const int32_t *__restrict__ rgb_pos, uint8_t *__restrict__ rgb,
                                              float *__restrict__ weight, u64 *hmap, int32_t *__restrict__ seg_last,
                                              int gs, int64_t order_base)
{
    // default wave priority: raised priority (s_setprio 3) bought the chain nothing (its steps are latency-bound) and cost
    // the kernels beside it 0.7 ms per 384-frame step
    __shared__ double2 s_exp[REC8 ? 64 : 1];
    if (REC8) {
        if (threadIdx.x < 64) s_exp[threadIdx.x] = gc.exp_tab[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int q = lane & 3;
    const int ch = q < 3 ? q : 2;
    const int sh = 8 * ch;
    const u64 quads_below = (1ull << (lane & ~3)) - 1ull;
    const int64_t nseg = bscal[0];
    const int64_t max_id_prev = bscal[1];
    const int64_t nlong = bscal[4];             // the long segments belong to k_chain_long
    unsigned long long *queue = (unsigned long long *)(bscal + 3);

    bool have = false, exhausted = false;
    int64_t s = 0, pos = 0, k1 = 0;
    uint32_t vid = 0, c = 0, last_j = 0;
    float w = 0.f;
    ChainRegs R, Rn;
    uint32_t Jn[CQ], Jnn[CQ];
#pragma unroll
    for (int i = 0; i < CQ; ++i) { R.rv[i] = Rn.rv[i] = 0u; R.alo[i] = Rn.alo[i] = 0u; R.ahi[i] = Rn.ahi[i] = 0u; Jn[i] = Jnn[i] = 0xffffffffu; }

    for (;;) {
        // ---- refill idle quads from the segment queue -----------------------------------------------------------
        const bool need = !have && !exhausted;
        const u64 mneed = __ballot(need && q == 0);
        if (mneed) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(queue, (unsigned long long)__popcll(mneed));
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                   (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            bool fresh = false, is_new = false;
            if (need) {
                const int64_t turn = (int64_t)base + __popcll(mneed & quads_below);
                if (turn < nseg) {
                    s = seg_info[turn].w;
                    const int4 info = seg_info[s];
                    pos = info.x; k1 = info.y;
                    vid = (uint32_t)info.z;
                    is_new = (int64_t)vid >= max_id_prev;
                    // a long segment (the first nlong of the list) belongs to k_chain_long, except for the first LONG_EARLY
                    // points of a voxel created by this batch: while the weight is small the colour changes at almost every
                    // step, which is this kernel's case (16 voxels per instruction), not the speculating kernel's
                    if (turn < nlong) {
                        if (is_new) { fresh = true; k1 = k1 - pos > LONG_EARLY ? pos + LONG_EARLY : k1; }
                    } else fresh = true;
                    have = fresh;
                } else exhausted = true;
            }
            if (fresh) {
                w = 0.f; c = 0u; last_j = 0u;
                if (!is_new) {
                    w = weight[vid];
                    c = rgb[3 * (int64_t)vid + ch];
                }
            }
            uint32_t J0[CQ];
            chain_load_idx(J0, fresh, pos, q, k1, sj);
            ChainRegs R0;
            uint32_t lj = last_j;
            chain_load_rec<REC8>(R0, J0, p_rec, lj);
            uint32_t J1[CQ];
            chain_load_idx(J1, fresh, pos + 64, q, k1, sj);
            chain_decode<REC8>(R0, J0, gc, s_exp);
            // :890-894 a new id takes the colour of its first point and weight f32(0 + alpha); that point is then done
            const uint32_t rv0 = quad_bcast<0>(R0.rv[0]);
            const double a0 = __hiloint2double((int)quad_bcast<0>(R0.ahi[0]), (int)quad_bcast<0>(R0.alo[0]));
            if (fresh) {
                last_j = lj;
                if (is_new) {
                    c = (rv0 >> sh) & 0xffu;
                    w = (float)((double)0.f + a0);
                    if (q == 0) R0.rv[0] &= 0x7fffffffu;
                }
#pragma unroll
                for (int i = 0; i < CQ; ++i) { R.rv[i] = R0.rv[i]; R.alo[i] = R0.alo[i]; R.ahi[i] = R0.ahi[i]; Jn[i] = J1[i]; }
            }
        }
        if (!__any(have)) {
            if (__all(exhausted)) break;
            continue;                            // every quad drew a segment that is not this kernel's: draw again
        }

        // ---- prefetch: records of the next chunk, order indices of the one after ----------------------------------
        chain_load_rec<REC8>(Rn, Jn, p_rec, last_j);
        chain_load_idx(Jnn, have, pos + 128, q, k1, sj);

        // ---- 64 steps out of registers; a block of 16 is skipped once no quad of the wave has points left in it ----
        bool go = true;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            go = go && __any((quad_bcast<0>(R.rv[4 * b]) & 0x40000000u) != 0u);
            if (go) {
#pragma unroll
                for (int i = 4 * b; i < 4 * b + 4; ++i) {
                    CHAIN_STEP(i, 0)
                    CHAIN_STEP(i, 1)
                    CHAIN_STEP(i, 2)
                    CHAIN_STEP(i, 3)
                }
            }
        }
        // a chunk whose last slot is empty was the segment's last
        const bool done = have && (quad_bcast<3>(R.rv[CQ - 1]) & 0x40000000u) == 0u;
        if (done) {
            uint32_t lj = last_j;
            lj = max(lj, quad_perm<0xB1>(lj));          // lanes 1,0,3,2
            lj = max(lj, quad_perm<0x4E>(lj));          // lanes 2,3,0,1
            if (q < 3) rgb[3 * (int64_t)vid + q] = (uint8_t)c;
            if (q == 0) {
                weight[vid] = w;
                const int32_t row = rgb_pos[3 * (int64_t)vid], col = rgb_pos[3 * (int64_t)vid + 1], h = rgb_pos[3 * (int64_t)vid + 2];
                const u64 packed = ((u64)(h + 1) << 40) | (u64)(order_base + lj);
                atomicMax(&hmap[(int64_t)row * gs + col], packed);
                seg_last[s] = (int32_t)lj;
            }
            have = false;
        }
        pos += 64;
        chain_decode<REC8>(Rn, Jn, gc, s_exp);      // behind the 64 steps: the records have long arrived
#pragma unroll
        for (int i = 0; i < CQ; ++i) {
            R.rv[i] = have ? Rn.rv[i] : 0u; R.alo[i] = Rn.alo[i]; R.ahi[i] = Rn.ahi[i];
            Jn[i] = have ? Jnn[i] : 0xffffffffu;
        }
    }
}

// ---- long segments: one wavefront per voxel, 64 points per round, speculate-and-verify ----------------------------------
// The recurrence of a voxel is sequential, but both of its state variables can be PREDICTED for a whole round of 64
// consecutive points and the prediction CHECKED with the reference's own arithmetic, all lanes at once:
//   * weight: w' = f32(f64(w) + alpha).  While w stays inside one binade every alpha adds a whole number of ulps that does
//     not depend on w (ties aside), so lane l predicts its entry weight as w0 + sum_{i<l} (f32(f64(w0) + alpha_i) - w0)
//     (a DPP prefix sum; exact, the terms are multiples of one ulp).  The check is the recurrence itself: lane l-1
//     computes f32(f64(entry_{l-1}) + alpha_{l-1}) from ITS entry and that must equal lane l's predicted entry.  Lane 0's
//     entry is known, so by induction every lane up to the first mismatch holds the true value; the lanes from there on
//     predict again from the now known weight (binade crossings, ties: a few times in a voxel's life; the first rounds of
//     a new voxel, whose weight doubles every few points, take several passes).
//   * colour: c' = trunc((f32(c * w) + r * alpha) / (w + alpha)) is truncated to uint8 at every step, so once w exceeds
//     255 a step can only keep c or lower it by one, and c stops moving at about the smallest value the voxel has seen.
//     Every lane evaluates its step with the round's entry colour; the first lane whose result differs is the first
//     point that changes c (all lanes before it had the right input), its result is the new c, the lanes after it are
//     evaluated again.  Per channel: one evaluation per round plus one per change of c.
// Nothing is assumed about alpha, w or c: whatever the predictor gets wrong is caught by the check and redone, so the result
// is the sequential one by construction.  A voxel that collects 2e5 points in a call takes ~3000 rounds of ~100
// instructions instead of 2e5 dependent steps of ~28 (6 ms -> 0.5 ms), and the loads of a round are 64 independent
// gathers instead of 4.
#define LONG_WAVES 32768
#define HOT_RPT 8                                // rounds of 64 points per wavefront and tile of a hot segment (records held in registers)
__device__ __forceinline__ float wave_incl_sum_f32(float x)
{
#define BSC_SCAN_STEP(ctrl, rows) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rows, 0xf, false));
    BSC_SCAN_STEP(0x111, 0xf)                   // row_shr:1
    BSC_SCAN_STEP(0x112, 0xf)                   // row_shr:2
    BSC_SCAN_STEP(0x114, 0xf)                   // row_shr:4
    BSC_SCAN_STEP(0x118, 0xf)                   // row_shr:8
    BSC_SCAN_STEP(0x142, 0xa)                   // row_bcast:15 into rows 1 and 3
    BSC_SCAN_STEP(0x143, 0xc)                   // row_bcast:31 into rows 2 and 3
#undef BSC_SCAN_STEP
    return x;
}

__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int64_t rfl64(int64_t v)        // a wave-uniform 64-bit value into scalar registers
{
    return (int64_t)(((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)((u64)v >> 32)) << 32) | (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u64)v));
}

struct ChainState { float w; uint32_t c0, c1, c2; };

// a point record as the long chain holds it between its load and its use: 12-byte {alpha, rgb}, or 8 bytes + the position
template <bool REC8> struct RecT;
template <> struct RecT<false> { PointRec r; };
template <> struct RecT<true> { uint2 r; uint32_t pos; };
__device__ __forceinline__ void rec_load(RecT<false> &t, const void *__restrict__ p_rec, uint32_t pos) { t.r = ((const PointRec *)p_rec)[pos]; }
__device__ __forceinline__ void rec_load(RecT<true> &t, const void *__restrict__ p_rec, uint32_t pos) { t.r = ((const uint2 *)p_rec)[pos]; t.pos = pos; }
__device__ __forceinline__ double rec_alpha(const RecT<false> &t, const GeomConst &, const double2 *) { return __hiloint2double((int)t.r.ahi, (int)t.r.alo); }
__device__ __forceinline__ double rec_alpha(const RecT<true> &t, const GeomConst &gc, const double2 *tab) { return rec8_alpha(gc, t.r.x, t.r.y, t.pos, tab); }
__device__ __forceinline__ uint32_t rec_rgb(const RecT<false> &t) { return t.r.rgbv; }
__device__ __forceinline__ uint32_t rec_rgb(const RecT<true> &t) { return t.r.x & 0xffffffu; }

// ---- the per-voxel point order, by runs ---------------------------------------------------------------------------------------
// The records of a run are consecutive (j0, j0 + 1, ...), so the order needs no entry per POINT (round 4: 4 bytes written by
// k_expand and read by the chain for each of them, 1.9 GB per 768-frame call): position k belongs to the run whose start is the
// last set bit of `bits` at or below k, and its record is j0[run] + (k - start).  Word w of `bits` holds the run starts of the
// positions [64 w, 64 w + 64); ck_run[w] / ck_start[w] name the (sorted) run that covers position 64 w and where it starts, so a
// lookup is stateless: one word, one checkpoint, one gather of j0 — any position, any order, no running count.
struct RunOrder { const u64 *bits; const uint32_t *ck_run, *ck_start, *j0; };
struct RunLook { u64 b; uint32_t r, s; };             // what the word of a position holds (loaded ahead of its use)
__device__ __forceinline__ RunLook run_look(const RunOrder &o, int64_t k)
{
    const int64_t w = k >> 6;
    RunLook l;
    l.b = o.bits[w]; l.r = o.ck_run[w]; l.s = o.ck_start[w];
    return l;
}
// -> index of the run in the sorted list, offset of position k inside it
__device__ __forceinline__ void run_of(const RunLook &l, int64_t k, uint32_t &run, uint32_t &delta)
{
    const int b = (int)(k & 63);
    const u64 below = l.b & ((2ull << b) - 1ull);        // b = 63: 2 << 63 wraps to 0, - 1 = all ones
    run = l.r + (uint32_t)__popcll(below) - (uint32_t)(l.b & 1ull);
    const uint32_t start = below ? (uint32_t)(k & ~63ll) + 63u - (uint32_t)__clzll((long long)below) : l.s;
    delta = (uint32_t)k - start;
}
__device__ __forceinline__ uint32_t order_j(const RunOrder &o, int64_t k)
{
    uint32_t run, delta;
    run_of(run_look(o, k), k, run, delta);
    return o.j0[run] + delta;
}

// rounds of 64 points over the positions [k, k1) of the per-voxel point order, from state `st` (wave-uniform) to the state
// after the last point.  `kend` clamps the prefetch addresses (the end of the whole segment).
// One round: 64 consecutive points of a voxel (lane l = point l of the round; invalid lanes carry alpha 0 and do not count),
// state (w, c0, c1, c2) wave-uniform in, the state after the round out.
__device__ __forceinline__ void chain_round_step(const double alpha, const uint32_t rgbv, const bool valid, float &w, uint32_t &c0,
                                                 uint32_t &c1, uint32_t &c2, const int lane)
{
    const double a = valid ? alpha : 0.0;      // alpha 0 leaves w as it is
    // ---- weights: predict, check with the recurrence, redo from the first lane that fails ----------------------------
    float wbase = w, wp = w, wn;
    int start = 0;
    for (;;) {
        const float inc = lane >= start ? (float)((double)wbase + a) - wbase : 0.f;
        const float incl = wave_incl_sum_f32(inc);
        if (lane >= start) wp = wbase + (incl - inc);
        wn = (float)((double)wp + a);                                  // :896,:899 the weight this point leaves
        const float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(wn), 0x138, 0xf, 0xf, false));   // wave_shr:1
        const u64 bad = __ballot(lane > start && left != wp);
        if (!bad) break;
        start = __ffsll((unsigned long long)bad) - 1;
        wbase = readlane_f32(wn, start - 1);
    }
    w = readlane_f32(wn, 63);
    const double den = (double)wp + a;
    double rd = __builtin_amdgcn_rcp(den);
    double e = fma(-den, rd, 1.0); rd = fma(rd, e, rd);
    e = fma(-den, rd, 1.0); rd = fma(rd, e, rd);
    // ---- colours: every lane steps from the round's entry colour; the first lane that disagrees sets the new one -------
    const u64 vmask = __ballot(valid);
#define BSC_LONG_CHANNEL(cc, shift)                                                                     \
    {                                                                                                   \
        const double ra = (double)((rgbv >> shift) & 0xffu) * a;                                        \
        u64 pend = vmask;                                                                               \
        for (;;) {                                                                                      \
            const double num = (double)((float)cc * wp) + ra;                                           \
            const double q0 = num * rd;                                                                 \
            const double rr = fma(-den, q0, num);                                                       \
            const uint32_t t = (uint32_t)fma(rr, rd, q0);                                               \
            const u64 diff = __ballot(t != cc) & pend;                                                  \
            if (!diff) break;                                                                           \
            const int f = __ffsll((unsigned long long)diff) - 1;                                        \
            cc = (uint32_t)__builtin_amdgcn_readlane((int)t, f);                                        \
            pend &= ~((2ull << f) - 1ull);                                                              \
        }                                                                                               \
    }
    BSC_LONG_CHANNEL(c0, 0)
    BSC_LONG_CHANNEL(c1, 8)
    BSC_LONG_CHANNEL(c2, 16)
#undef BSC_LONG_CHANNEL
}

// rounds of 64 points over the positions [k, k1) of the per-voxel point order, from state `st` (wave-uniform) to the state
// after the last point.  `kend` clamps the prefetch addresses (the end of the whole segment).
template <bool REC8>
__device__ __forceinline__ void chain_rounds(const RunOrder &o, const void *__restrict__ p_rec, int64_t k,
                                             const int64_t k1, const int64_t kend, ChainState &st, const int lane,
                                             const GeomConst &gc, const double2 *exp_tab)
{
    if (k >= k1) return;
    float w = st.w;
    uint32_t c0 = st.c0, c1 = st.c1, c2 = st.c2;
    // in flight while round n is worked on: the records of round n+1, the run (j0 gather) of round n+2, the word and checkpoint of
    // round n+3
    const int64_t klast = kend - 1;
    RecT<REC8> rec, rec_nxt;
    uint32_t j0_nxt, d_nxt;
    RunLook lk;
    {
        const int64_t ka = k + lane < kend ? k + lane : klast, kb = k + 64 + lane < kend ? k + 64 + lane : klast;
        const int64_t kc = k + 128 + lane < kend ? k + 128 + lane : klast;
        const RunLook la = run_look(o, ka), lb = run_look(o, kb);
        lk = run_look(o, kc);
        uint32_t ra, da, rb;
        run_of(la, ka, ra, da);
        run_of(lb, kb, rb, d_nxt);
        rec_load(rec, p_rec, o.j0[ra] + da);
        j0_nxt = o.j0[rb];
    }
    for (; k < k1; k += 64) {
        rec_load(rec_nxt, p_rec, j0_nxt + d_nxt);
        {
            const int64_t kc = k + 128 + lane < kend ? k + 128 + lane : klast, kd = k + 192 + lane < kend ? k + 192 + lane : klast;
            uint32_t rc;
            run_of(lk, kc, rc, d_nxt);
            // the old word / checkpoint are used up before the next ones are requested: scheduled the other way round the two sets
            // overlap, the new one lands in spare registers and is COPIED into the loop's registers at the end of the iteration —
            // behind a wait for every load in flight, the newest included: one memory round trip per round of 64 points
            asm volatile("" : "+v"(d_nxt), "+v"(rc));       // (both evaluated here, not sunk to their uses behind the new loads)
            __builtin_amdgcn_sched_barrier(0);
            j0_nxt = o.j0[rc];
            lk = run_look(o, kd);
        }
        chain_round_step(rec_alpha(rec, gc, exp_tab), rec_rgb(rec), k + lane < k1, w, c0, c1, c2, lane);
        rec = rec_nxt;
    }
    st.w = w; st.c0 = c0; st.c1 = c1; st.c2 = c2;
}

__device__ __forceinline__ void chain_finish(const ChainState &st, const uint32_t vid, const int32_t s, const int64_t klast,
                                             const RunOrder &o, const int32_t *__restrict__ rgb_pos,
                                             uint8_t *__restrict__ rgb, float *__restrict__ weight, u64 *hmap,
                                             int32_t *__restrict__ seg_last, int gs, int64_t order_base)
{
    rgb[3 * (int64_t)vid] = (uint8_t)st.c0; rgb[3 * (int64_t)vid + 1] = (uint8_t)st.c1; rgb[3 * (int64_t)vid + 2] = (uint8_t)st.c2;
    weight[vid] = st.w;
    const uint32_t lj = order_j(o, klast);              // record indices grow along a segment
    const int32_t row = rgb_pos[3 * (int64_t)vid], col = rgb_pos[3 * (int64_t)vid + 1], h = rgb_pos[3 * (int64_t)vid + 2];
    atomicMax(&hmap[(int64_t)row * gs + col], ((u64)(h + 1) << 40) | (u64)(order_base + lj));
    seg_last[s] = (int32_t)lj;
}

// Hot segments (>= 2^HOT_MIN_LOG2 points: the first bscal[6] of the length-ordered list) are split over the wavefronts of a
// workgroup, again by prediction and check: (A) every wavefront sums what its chunk would add to the weight — whole ulps
// that do not depend on the weight inside a binade — so every chunk's entry weight is known after one cheap pass; (B) every
// wavefront runs its chunk's rounds from that entry weight and the segment's entry colour (a heavy voxel's colour has long
// settled); (C) chunk k+1's assumed entry must equal chunk k's exit: the first chunk that fails is run again from the true
// state, and so on down the line.  A binade crossing or a colour change inside the segment costs the rest of it a second
// run; otherwise a segment of n points takes n / (64 * wavefronts) rounds.
// PART (round 6): 0 = hot segments, then the other long ones, in one launch; 1 = the hot segments only; 2 = the others only.  The
// two halves want different launch shapes: a hot tile keeps 8 rounds of records in flight per wavefront (110 registers, 50 KB of
// LDS per workgroup: four wavefronts per SIMD), a one-wavefront segment is a chain of dependent steps per round of 64 points whose
// latency only other wavefronts on the SIMD can cover — alone (PART 2) it fits BSC_MID_MIN_BLOCKS workgroups of four wavefronts
// per CU.  launch_pending_chain runs the halves side by side on two streams.
#ifndef BSC_MID_MIN_BLOCKS
#define BSC_MID_MIN_BLOCKS 6
#endif
template <int NWV, bool REC8, int PART>      // wavefronts per workgroup: the width of the hot-segment split; record format
__global__ __launch_bounds__(NWV * 64, PART == 2 ? BSC_MID_MIN_BLOCKS : 1) void k_chain_long(const GeomConst gc, const RunOrder o, int64_t *bscal,
                                                        const int4 *__restrict__ seg_info,
                                                        const void *__restrict__ p_rec,
                                                        const int32_t *__restrict__ rgb_pos, uint8_t *__restrict__ rgb,
                                                        float *__restrict__ weight, u64 *hmap, int32_t *__restrict__ seg_last,
                                                        int gs, int64_t order_base)
{
    __shared__ float s_sum[2][NWV];
    __shared__ ChainState s_entry[2][NWV], s_exit[2][NWV];
#ifndef BSC_CHAIN_TILES_V1
    __shared__ double s_al[PART == 2 ? 1 : HOT_RPT][NWV * 64];      // the hot tile in flight: alpha and colour of every point (96 KB at 16 wavefronts)
    __shared__ uint32_t s_rg[PART == 2 ? 1 : HOT_RPT][NWV * 64];
#endif
    __shared__ double2 s_exp[REC8 ? 64 : 1];
    if (REC8) {
        if (threadIdx.x < 64) s_exp[threadIdx.x] = gc.exp_tab[threadIdx.x];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the hot tiles of one segment are a serial chain (the hottest voxel of a 768-frame call: 46 tiles): beside the 8192 resident
    // wavefronts of the one-wavefront segments every phase of a tile waited for issue slots; at a raised wave priority they go first
#ifndef BSC_HOT_PRIO
#define BSC_HOT_PRIO 3
#endif
    if (PART == 1) __builtin_amdgcn_s_setprio(BSC_HOT_PRIO);
    const int64_t nlong = bscal[4], nhot = bscal[6];
    const int64_t max_id_prev = bscal[1];
#ifdef BSC_CHAIN_PROFILE
    // per-phase clocks of wavefront 0 (-DBSC_CHAIN_PROFILE, scripts/variant_ab.py with an A/B build): load + alpha, pass A + barriers,
    // pass B, check; tiles, repeated passes; the one-wavefront segments after the hot ones; wall clock (100 MHz) at entry and exit
    long long cph[5] = {0, 0, 0, 0, 0}, cq = clock64();
    const long long wall0 = wall_clock64();
    int n_tiles = 0, n_again = 0, n_hotseg = 0;
#define CP_T(k) { const long long now_ = clock64(); cph[k] += now_ - cq; cq = now_; }
#else
#define CP_T(k)
#endif
    // ---- hot segments: one workgroup each, in TILES of NWV x HOT_RPT x 64 points whose records every wavefront loads ONCE into
    // registers: (A) the increments of its slice summed from registers, (B) its rounds stepped from registers from the predicted
    // entry state, (C) the slices' entry / exit states compared; a slice whose entry was wrong runs (A, B) again from registers.
    // (Until round 4 the chunks spanned the whole segment and pass A re-read every record of it: +12 B per hot point.)
    for (int64_t turn = blockIdx.x; PART != 2 && turn < nhot; turn += gridDim.x) {
        const int32_t s = seg_info[turn].w;
        const int4 info = seg_info[s];
        int64_t k = info.x;
        const int64_t k1 = info.y;
        const uint32_t vid = (uint32_t)info.z;
        if ((int64_t)vid >= max_id_prev) k += LONG_EARLY;
        ChainState st;
        st.w = readlane_f32(weight[vid], 0);
        st.c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb[3 * (int64_t)vid]);
        st.c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb[3 * (int64_t)vid + 1]);
        st.c2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb[3 * (int64_t)vid + 2]);
#ifdef BSC_CHAIN_PROFILE
        ++n_hotseg;
#endif
#ifndef BSC_CHAIN_TILES_V1
        // Tiles on the grid of the run order's 64-position WORDS (round 6).  A round of 64 lanes then lies in ONE word: its start
        // bits and checkpoint are the same for all lanes — three scalar loads instead of three vector loads per lane (the look-up
        // of a tile held 32 vector registers per wavefront) — and the phases of consecutive tiles overlap:
        //   * the run gathers (j0) of tile t + 1 are issued before pass A of tile t and arrive under it and its barrier,
        //   * the record gathers of tile t + 1 are issued behind that barrier and arrive under pass B of tile t,
        //   * sums / entry / exit states alternate between two sets of shared arrays, so a tile needs TWO barriers (after the sums,
        //     after the exits) instead of four: a wavefront that is through with its check goes on to the next tile's loads while
        //     the others still step their rounds.
        // Profile of the four-barrier form (wavefront 0, per tile of 8192 points, 33 k clocks): loads + alpha 6.5 k, pass A + barriers
        // 10.4 k, pass B 6.3 k, barrier + check 10.2 k — two thirds of a tile were spent waiting.
        k = rfl64(k);
        const int64_t k1u = rfl64(k1);
        const int64_t W0 = k >> 6, wlast = (k1u - 1) >> 6;
        const int64_t n_tiles_seg = (wlast - W0 + (int64_t)NWV * HOT_RPT) / ((int64_t)NWV * HOT_RPT);
        uint32_t posn[HOT_RPT];                 // tile in flight: run gathers, then record positions
        uint32_t dn[HOT_RPT];
        RecT<REC8> recn[HOT_RPT];
        // word of round r of this wavefront in tile t; the lanes of the round are the positions 64 w + lane
#define HOT_WORD(t, r) (W0 + ((t) * NWV + wv) * HOT_RPT + (r))
#define HOT_VALID(t, r) (HOT_WORD(t, r) * 64 + lane >= k && HOT_WORD(t, r) * 64 + lane < k1u)
#define HOT_ISSUE_RUNS(t)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < HOT_RPT; ++r) {                                                          \
        const int64_t w_ = HOT_WORD(t, r), wc_ = w_ < wlast ? w_ : wlast;        /* wave-uniform: scalar loads */   \
        RunLook l_;                                                                                                \
        l_.b = o.bits[wc_]; l_.r = o.ck_run[wc_]; l_.s = o.ck_start[wc_];                                          \
        uint32_t run_;                                                                                             \
        run_of(l_, wc_ * 64 + lane, run_, dn[r]);                                                                  \
        posn[r] = o.j0[run_];                                                                                      \
    }
#define HOT_ISSUE_RECS(t)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < HOT_RPT; ++r) rec_load(recn[r], p_rec, HOT_VALID(t, r) ? posn[r] + dn[r] : 0u);
        HOT_ISSUE_RUNS((int64_t)0)
        HOT_ISSUE_RECS((int64_t)0)
        int par = 0;
        for (int64_t t = 0; t < n_tiles_seg; ++t, par ^= 1) {
            CP_T(4)
            // the slice's weights and colours go to LDS (a private slot per thread and round): pass B reads them round by round, and
            // the registers they would hold across it (24) carry the next tile's records in flight instead — with both in registers
            // the 16-wavefront form spilled (it has 128 registers per lane)
            float mine0 = 0.f;
            {
                const double wb = (double)st.w;
#pragma unroll
                for (int r = 0; r < HOT_RPT; ++r) {
                    const double a = rec_alpha(recn[r], gc, s_exp);
                    s_al[r][threadIdx.x] = a;
                    s_rg[r][threadIdx.x] = rec_rgb(recn[r]);
                    mine0 += HOT_VALID(t, r) ? (float)(wb + a) - st.w : 0.f;
                }
            }
            const bool more = t + 1 < n_tiles_seg;
            if (more) { HOT_ISSUE_RUNS(t + 1) }
#ifdef BSC_CHAIN_PROFILE
            ++n_tiles;
#endif
            CP_T(0)
            // chunks < first are final; `st` is the true state at the start of chunk `first`
            for (int first = 0, again = 0;; ++again) {
                if (again) __syncthreads();                     // a repeated pass rewrites arrays the slower wavefronts may still be checking
                if (wv >= first) {
                    float mine = mine0;
                    if (again) {                                // from the entry weight now known
                        mine = 0.f;
                        const double wb = (double)st.w;
#pragma unroll
                        for (int r = 0; r < HOT_RPT; ++r) mine += HOT_VALID(t, r) ? (float)(wb + s_al[r][threadIdx.x]) - st.w : 0.f;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
                    if (lane == 0) s_sum[par][wv] = mine;
                }
                __syncthreads();
                if (more && again == 0) { HOT_ISSUE_RECS(t + 1) }           // the runs have arrived under pass A and the barrier
                CP_T(1)
                if (wv >= first) {
                    ChainState me = st;
                    for (int i = first; i < wv; ++i) me.w += s_sum[par][i];
                    if (lane == 0) s_entry[par][wv] = me;
#pragma unroll
                    for (int r = 0; r < HOT_RPT; ++r) {
                        if (HOT_WORD(t, r) > wlast) break;      // (uniform) past the segment's end
                        chain_round_step(s_al[r][threadIdx.x], s_rg[r][threadIdx.x], HOT_VALID(t, r), me.w, me.c0, me.c1, me.c2, lane);
                    }
                    if (lane == 0) s_exit[par][wv] = me;
                }
                CP_T(2)
                __syncthreads();
                // lane c compares the entry chunk c assumed with the exit of chunk c - 1
                const int cl = lane < NWV ? lane : 0;
                const ChainState have = s_entry[par][cl], real = s_exit[par][cl > 0 ? cl - 1 : 0];
                const u64 bm = __ballot(lane > first && lane < NWV &&
                                        !(have.w == real.w && have.c0 == real.c0 && have.c1 == real.c1 && have.c2 == real.c2));
                const int bad = bm ? __ffsll((unsigned long long)bm) - 1 : NWV;
                CP_T(3)
                if (bad == NWV) break;
                st = s_exit[par][bad - 1];                      // a binade crossing or a colour change upstream: predict again from here
                st.w = readlane_f32(st.w, 0);
                first = bad;
#ifdef BSC_CHAIN_PROFILE
                ++n_again;
#endif
            }
            st = s_exit[par][NWV - 1];                          // the state after the tile (every thread reads the same entry)
            st.w = readlane_f32(st.w, 0);
        }
        __syncthreads();                                        // the next segment starts on set 0 again
#undef HOT_WORD
#undef HOT_VALID
#undef HOT_ISSUE_RUNS
#undef HOT_ISSUE_RECS
#else
        const int64_t klast = k1 - 1;
        for (int64_t tile = k; tile < k1; tile += (int64_t)NWV * HOT_RPT * 64) {
            CP_T(4)
            const int64_t ka = tile + (int64_t)wv * HOT_RPT * 64;          // this wavefront's slice [ka, ka + HOT_RPT * 64) of the tile
            double al[HOT_RPT];                                            // the slice's weights and colours, in registers for the tile
            uint32_t rg[HOT_RPT];
            {
                RecT<REC8> rec[HOT_RPT];
                RunLook lk[HOT_RPT];
                uint32_t j0r[HOT_RPT], dr[HOT_RPT];
#pragma unroll
                for (int r = 0; r < HOT_RPT; ++r) {
                    const int64_t kk = ka + r * 64 + lane;
                    lk[r] = run_look(o, kk < k1 ? kk : klast);
                }
#pragma unroll
                for (int r = 0; r < HOT_RPT; ++r) {
                    const int64_t kk = ka + r * 64 + lane;
                    uint32_t run;
                    run_of(lk[r], kk < k1 ? kk : klast, run, dr[r]);
                    j0r[r] = o.j0[run];
                }
#pragma unroll
                for (int r = 0; r < HOT_RPT; ++r) rec_load(rec[r], p_rec, j0r[r] + dr[r]);
#pragma unroll
                for (int r = 0; r < HOT_RPT; ++r) { al[r] = rec_alpha(rec[r], gc, s_exp); rg[r] = rec_rgb(rec[r]); }
            }
#ifdef BSC_CHAIN_PROFILE
            { double sink = 0; for (int r = 0; r < HOT_RPT; ++r) sink += al[r]; asm volatile("" :: "v"(sink)); ++n_tiles; }
#endif
            CP_T(0)
            // chunks < first are final; `st` is the true state at the start of chunk `first`
            for (int first = 0;;) {
                float mine = 0.f;
                if (wv >= first) {
                    const double wb = (double)st.w;
#pragma unroll
                    for (int r = 0; r < HOT_RPT; ++r) {
                        const bool valid = ka + r * 64 + lane < k1;
                        mine += valid ? (float)(wb + al[r]) - st.w : 0.f;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
                    mine = readlane_f32(mine, 0);
                }
                __syncthreads();                                // the shared arrays are free (previous pass / tile / segment)
                if (lane == 0) s_sum[0][wv] = mine;
                __syncthreads();
                CP_T(1)
                if (wv >= first) {
                    ChainState me = st;
                    for (int i = first; i < wv; ++i) me.w += s_sum[0][i];
                    if (lane == 0) s_entry[0][wv] = me;
#pragma unroll
                    for (int r = 0; r < HOT_RPT; ++r) {
                        if (ka + r * 64 >= k1) break;           // (uniform) past the segment's end
                        chain_round_step(al[r], rg[r], ka + r * 64 + lane < k1, me.w, me.c0, me.c1, me.c2, lane);
                    }
                    if (lane == 0) s_exit[0][wv] = me;
                }
                CP_T(2)
                __syncthreads();
                int bad = NWV;
                for (int c = NWV - 1; c > first; --c) {
                    const ChainState have = s_entry[0][c], real = s_exit[0][c - 1];
                    if (!(have.w == real.w && have.c0 == real.c0 && have.c1 == real.c1 && have.c2 == real.c2)) bad = c;
                }
                CP_T(3)
                if (bad == NWV) break;
                st = s_exit[0][bad - 1];                           // a binade crossing or a colour change upstream: predict again from here
                first = bad;
#ifdef BSC_CHAIN_PROFILE
                ++n_again;
#endif
            }
            st = s_exit[0][NWV - 1];                               // the state after the tile (every thread reads the same entry)
            __syncthreads();                                    // before the next tile's pass overwrites the shared arrays
        }
#endif
        if (threadIdx.x == 0) chain_finish(st, vid, s, k1 - 1, o, rgb_pos, rgb, weight, hmap, seg_last, gs, order_base);
    }
    // ---- the other long segments: one wavefront each, static schedule over the length-ordered list, back and forth (wave g
    // takes g, 2n-1-g, 2n+g, ...): no queue — an `if (lane == 0) atomicAdd` at the head of a loop that ends in another
    // `if (lane == 0)` block gets jump-threaded around the readfirstlane between them, and the wavefront then never leaves
    // the loop
    const int64_t nwaves = (int64_t)gridDim.x * NWV;
    const int64_t wave = (int64_t)blockIdx.x * NWV + wv;
    const int64_t nrest = nlong - nhot;
    for (int64_t pass = 0; PART != 1; ++pass) {
        const int64_t turn = nhot + pass * nwaves + ((pass & 1) ? nwaves - 1 - wave : wave);
        if (pass * nwaves >= nrest) break;
        if (turn >= nlong) continue;
        const int32_t s = seg_info[turn].w;
        const int4 info = seg_info[s];
        int64_t k = info.x;
        const int64_t k1 = info.y;
        const uint32_t vid = (uint32_t)info.z;
        // a voxel created by this batch has had its first LONG_EARLY points stepped by k_chain (launched before this kernel
        // on the same stream): its state is in the arrays like that of an old voxel
        if ((int64_t)vid >= max_id_prev) k += LONG_EARLY;
        if (k >= k1) continue;
        ChainState st;
        st.w = readlane_f32(weight[vid], 0);
        st.c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb[3 * (int64_t)vid]);
        st.c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb[3 * (int64_t)vid + 1]);
        st.c2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rgb[3 * (int64_t)vid + 2]);
        chain_rounds<REC8>(o, p_rec, k, k1, k1, st, lane, gc, s_exp);
        if (lane == 0) chain_finish(st, vid, s, k1 - 1, o, rgb_pos, rgb, weight, hmap, seg_last, gs, order_base);
    }
#ifdef BSC_CHAIN_PROFILE
    CP_T(4)
    if (threadIdx.x == 0 && (blockIdx.x < 2 || (blockIdx.x & 127) == 100 || blockIdx.x == gridDim.x - 1))
        printf("k_chain_long wg %4d: hot segs %d of %lld, tiles %d (+%d repeated passes): load+alpha %lld  A+barriers %lld  B %lld  barrier+check %lld | other %lld"
               " (clocks) | wall %lld .. %lld (10 ns)\n", (int)blockIdx.x, n_hotseg, (long long)nhot, n_tiles, n_again, cph[0], cph[1], cph[2], cph[3], cph[4],
               wall0 % 100000000ll, wall_clock64() % 100000000ll);
#endif
#undef CP_T
}

// ---- runs -> per-voxel point order ---------------------------------------------------------------------------------
// After the stable sort every voxel's runs are contiguous and in order j.  The position of a run in the per-voxel point
// order is the exclusive prefix of the run lengths, the ordinal of a voxel segment the exclusive prefix of the "first run
// of its voxel" flags: both ride in one 64-bit value (length | head << 32), summed per block of EB runs (k_run_blocksum),
// scanned over the ~R / 1024 block sums, and finished inside the block by k_expand — no R-sized scan array.
#define EB 1024
#define EXPAND_WORDS 2048          // k_expand: 64-position words of start bits a block collects in LDS (16 KB)
// (length | head << 32) of sorted run i from its key and its predecessor's.  The two keys are loaded by the caller, unconditionally
// (clamped indices) and for all its runs at once: read inside this function behind `if (i >= R)` and `if (v == vmask)`, every run
// cost two dependent memory round trips, four runs per thread in a row.
__device__ __forceinline__ int64_t run_item_of(uint32_t key, uint32_t prev, int64_t i, int64_t R, uint32_t vmask, int vb)
{
    const uint32_t v = key & vmask;
    if (i >= R || v == vmask) return 0;         // no voxel: takes no room, starts no segment
    const int64_t head = (i == 0 || (prev & vmask) != v) ? 1 : 0;
    return (int64_t)(key >> vb) + 1 + (head << 32);
}

__global__ __launch_bounds__(TPB) void k_run_blocksum(int64_t R, int vb, const uint32_t *__restrict__ rkey_sorted,
                                                      int64_t *__restrict__ blk_sum)
{
    __shared__ int64_t s_w[TPB / 64];
    const uint32_t vmask = vb >= 32 ? 0xffffffffu : ((1u << vb) - 1u);
    int64_t v = 0;
    uint32_t kc[EB / TPB], kp[EB / TPB];
#pragma unroll
    for (int r = 0; r < EB / TPB; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EB + r * TPB + threadIdx.x, ic = i < R ? i : R - 1;
        kc[r] = rkey_sorted[ic];
        kp[r] = rkey_sorted[ic > 0 ? ic - 1 : 0];
    }
#pragma unroll
    for (int r = 0; r < EB / TPB; ++r) v += run_item_of(kc[r], kp[r], (int64_t)blockIdx.x * EB + r * TPB + threadIdx.x, R, vmask, vb);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// Position of every run in the per-voxel point order: its start bit, and the checkpoints of the 64-position words it covers
// (RunOrder above).  One thread per sorted run, four per thread; the first run of a voxel also records the segment:
// seg_k0[s] = off, seg_vid[s] = voxel id.
__global__ __launch_bounds__(TPB) void k_expand(int64_t R, int vb, const uint32_t *__restrict__ rkey_sorted,
                                                const int64_t *__restrict__ blk_base, u64 *__restrict__ bits,
                                                uint32_t *__restrict__ ck_run, uint32_t *__restrict__ ck_start,
                                                int32_t *__restrict__ seg_k0, int32_t *__restrict__ seg_vid, int64_t *bscal)
{
    __shared__ uint32_t s_grp[EB / 64];
    __shared__ u64 s_bits[EXPAND_WORDS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t vmask = vb >= 32 ? 0xffffffffu : ((1u << vb) - 1u);
    // exclusive prefix of (length, head) inside the block, both in ONE 32-bit word — a block's 1024 runs hold at most 2^20
    // points (bits 0..20) and 1024 heads (bits 21..31) — scanned with DPP row shifts instead of 64-bit shuffles
    uint32_t item[EB / TPB], incl[EB / TPB];
    uint32_t kc[EB / TPB], kp[EB / TPB];                         // key and predecessor's key of the thread's four runs
#pragma unroll
    for (int r = 0; r < EB / TPB; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EB + (r * (TPB / 64) + wid) * 64 + lane, ic = i < R ? i : R - 1;
        kc[r] = rkey_sorted[ic];
        kp[r] = rkey_sorted[ic > 0 ? ic - 1 : 0];
    }
#pragma unroll
    for (int r = 0; r < EB / TPB; ++r) {
        const int64_t i = (int64_t)blockIdx.x * EB + (r * (TPB / 64) + wid) * 64 + lane;      // group g = r * 4 + wid
        const int64_t it = run_item_of(kc[r], kp[r], i, R, vmask, vb);
        item[r] = (uint32_t)(it & 0x1fffff) | ((uint32_t)(it >> 32) << 21);
        uint32_t v = item[r];
#define BSC_ISCAN_STEP(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false);
        BSC_ISCAN_STEP(0x111, 0xf)                   // row_shr:1
        BSC_ISCAN_STEP(0x112, 0xf)                   // row_shr:2
        BSC_ISCAN_STEP(0x114, 0xf)                   // row_shr:4
        BSC_ISCAN_STEP(0x118, 0xf)                   // row_shr:8
        BSC_ISCAN_STEP(0x142, 0xa)                   // row_bcast:15 into rows 1 and 3
        BSC_ISCAN_STEP(0x143, 0xc)                   // row_bcast:31 into rows 2 and 3
#undef BSC_ISCAN_STEP
        incl[r] = v;
        if (lane == 63) s_grp[r * (TPB / 64) + wid] = v;
    }
    __syncthreads();
    const int64_t base = blk_base[blockIdx.x];
    // The block's runs tile the positions [p_lo, p_lo + n_pos): their start bits are collected in LDS and leave as whole words —
    // atomically only where a word is shared with the neighbouring blocks (one global atomic per run, 4 to a word on average, cost
    // the order stage 0.2 ms per call and slowed the kernels beside it).  A block of very long runs (> EXPAND_WORDS x 64 positions)
    // sets its bits in global memory directly.
    uint32_t tot32 = 0;
    for (int k = 0; k < EB / 64; ++k) tot32 += s_grp[k];
    const int64_t p_lo = base & 0xffffffffll, n_pos = (int64_t)(tot32 & 0x1fffffu);
    const int64_t w_lo = p_lo >> 6, n_words = n_pos ? ((p_lo + n_pos + 63) >> 6) - w_lo : 0;
    const bool in_lds = n_words <= EXPAND_WORDS;
    if (in_lds) {
        for (int w = threadIdx.x; w < n_words; w += TPB) s_bits[w] = 0ull;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < EB / TPB; ++r) {
        const int g = r * (TPB / 64) + wid;
        uint32_t pre32 = 0;
        for (int k = 0; k < g; ++k) pre32 += s_grp[k];
        pre32 += incl[r] - item[r];                                 // exclusive prefix of this run inside the block
        const int64_t sc = base + (int64_t)(pre32 & 0x1fffffu) + ((int64_t)(pre32 >> 21) << 32);
        const int64_t i = (int64_t)blockIdx.x * EB + g * 64 + lane;
        if (i >= R) continue;
        const uint32_t key = kc[r];
        const uint32_t v = key & vmask;
        const int64_t off = sc & 0xffffffffll;
        if (v != vmask) {
            const int64_t len = (int64_t)(key >> vb) + 1;
            const bool head = (item[r] >> 21) != 0;
            if (head) { seg_k0[sc >> 32] = (int32_t)off; seg_vid[sc >> 32] = (int32_t)v; }
            if (i == R - 1) { bscal[0] = (sc >> 32) + (head ? 1 : 0); bscal[2] = off + len; }
            if (in_lds) atomicOr(&s_bits[(off >> 6) - w_lo], 1ull << (off & 63));
            else atomicOr(&bits[off >> 6], 1ull << (off & 63));
            for (int64_t m = (off + 63) & ~63ll; m < off + len; m += 64) { ck_run[m >> 6] = (uint32_t)i; ck_start[m >> 6] = (uint32_t)off; }
        } else if (i == R - 1) {
            bscal[0] = sc >> 32; bscal[2] = off;            // runs without a voxel sort last and take no room
        }
    }
    if (in_lds) {
        __syncthreads();
        for (int w = threadIdx.x; w < n_words; w += TPB) {
            const u64 v = s_bits[w];
            if (!v) continue;                                       // the array is cleared before the kernel
            if (w == 0 || w == n_words - 1) atomicOr(&bits[w_lo + w], v);
            else bits[w_lo + w] = v;
        }
    }
}

// The quad chain (k_chain: short segments, and the first LONG_EARLY points of the long segments of NEW voxels) reads one order
// index per point: written here, for those positions only.  One wavefront per segment of the length-ordered list.
__global__ __launch_bounds__(TPB) void k_expand_short(const int64_t *bscal, const int4 *__restrict__ seg_info, const RunOrder o,
                                                      uint32_t *__restrict__ sj, int early)
{
    const int lane = threadIdx.x & 63;
    const int64_t nseg = bscal[0], max_id_prev = bscal[1], nlong = bscal[4];
    const int64_t nwaves = (int64_t)gridDim.x * (TPB / 64);
    for (int64_t turn = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6); turn < nseg; turn += nwaves) {
        const int4 info = seg_info[seg_info[turn].w];
        const int64_t k0 = info.x;
        int64_t k1 = info.y;
        if (turn < nlong) {
            if ((int64_t)(uint32_t)info.z < max_id_prev) continue;          // an old voxel's long segment: k_chain_long alone
            k1 = k1 - k0 > early ? k0 + early : k1;
        }
        for (int64_t k = k0 + lane; k < k1; k += 64) sj[k] = order_j(o, k);
    }
}

// voxel segments of the point order: {first k, end k, voxel id, -} + sort keys that order the segments by length class
// (longest first).  The chain hands segments to quads in that order, so the 16 quads of a wavefront walk segments of
// similar length (a wavefront issues for as long as its longest segment lasts) and the longest voxels start first.
__global__ __launch_bounds__(TPB) void k_seg_bounds(const int64_t *bscal, int64_t n_bound, const int32_t *__restrict__ seg_k0,
                                                    const int32_t *__restrict__ seg_vid, int4 *__restrict__ seg_info,
                                                    uint32_t *__restrict__ okey, uint32_t *__restrict__ oval)
{
    const int64_t nseg = bscal[0];
    for (int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x; s < n_bound; s += (int64_t)gridDim.x * TPB) {
        uint32_t key = 63u;
        if (s < nseg) {
            const int32_t k0 = seg_k0[s];
            const int32_t k1 = s + 1 < nseg ? seg_k0[s + 1] : (int32_t)bscal[2];
            seg_info[s] = make_int4(k0, k1, seg_vid[s], 0);
            key = (uint32_t)__clz(k1 - k0);          // 2^(31-key) <= length < 2^(32-key)
        }
        okey[s] = key;
        oval[s] = (uint32_t)s;
    }
}

__global__ __launch_bounds__(TPB) void k_seg_order(int64_t *bscal, const uint32_t *__restrict__ okey_sorted,
                                                   const uint32_t *__restrict__ oval_sorted, int4 *__restrict__ seg_info,
                                                   int long_chain,   // log2 of the shortest long segment, 0 = none
                                                   int hot_chain)    // log2 of the shortest hot segment, 0 = none
{
    const int64_t nseg = bscal[0];
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nseg; i += (int64_t)gridDim.x * TPB) {
        seg_info[i].w = (int32_t)oval_sorted[i];
        // key = clz(length): the segments of >= 2^LONG_MIN_LOG2 points come first
        const bool is_long = okey_sorted[i] <= 31u - long_chain;
        if (long_chain && is_long && (i + 1 == nseg || okey_sorted[i + 1] > 31u - long_chain)) bscal[4] = i + 1;
        // ... and those of >= 2^HOT_MIN_LOG2 points before them
        const bool is_hot = hot_chain && okey_sorted[i] <= 31u - hot_chain;
        if (long_chain && is_hot && (i + 1 == nseg || okey_sorted[i + 1] > 31u - hot_chain)) bscal[6] = i + 1;
    }
}

// top-down map colour: the voxel whose (h, order) won the cell writes the rgb of its latest point
template <bool REC8>
__global__ __launch_bounds__(TPB) void k_hwin(const int64_t *bscal, const int4 *__restrict__ seg_info,
                                              const int32_t *__restrict__ seg_last,
                                              const int32_t *__restrict__ rgb_pos, const u64 *__restrict__ hmap,
                                              const void *__restrict__ p_rec, uint8_t *__restrict__ cv_map, int gs,
                                              int64_t order_base)
{
    const int64_t nseg = bscal[0];
    for (int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * TPB) {
        const uint32_t vid = (uint32_t)seg_info[s].z;
        const uint32_t last_j = (uint32_t)seg_last[s];
        const int32_t row = rgb_pos[3 * (int64_t)vid], col = rgb_pos[3 * (int64_t)vid + 1], h = rgb_pos[3 * (int64_t)vid + 2];
        const int64_t rc = (int64_t)row * gs + col;
        const u64 packed = ((u64)(h + 1) << 40) | (u64)(order_base + last_j);
        if (hmap[rc] == packed) {
            const uint32_t v = REC8 ? ((const uint2 *)p_rec)[last_j].x : ((const PointRec *)p_rec)[last_j].rgbv;
            cv_map[3 * rc + 0] = (uint8_t)(v & 0xff);
            cv_map[3 * rc + 1] = (uint8_t)((v >> 8) & 0xff);
            cv_map[3 * rc + 2] = (uint8_t)((v >> 16) & 0xff);
        }
    }
}

// exact mode, memory_2.py:882-886: rows [row0, row0+n) of the token cache <- passing points [q0, q0+n)
template <typename TOK>
__global__ __launch_bounds__(TPB) void k_append(const int32_t *__restrict__ pass_list, int64_t q0, int64_t n,
                                                int64_t row0, const int32_t *__restrict__ p_cell,
                                                const uint32_t *__restrict__ p_patf, const float *__restrict__ p_r2f,
                                                const TOK *__restrict__ tokens, int g2, int D, int gs, int nh,
                                                float *__restrict__ cache_f, int32_t *__restrict__ cache_pos,
                                                float *__restrict__ cache_d)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (w >= n) return;
    const int32_t j = pass_list[q0 + w];
    const uint32_t cc = p_patf[j];
    const TOK *src = tokens + ((int64_t)(cc >> 16) * g2 + (cc & 0xffffu)) * D;
    float4 *dst = (float4 *)(cache_f + (row0 + w) * D);
    for (int v = lane; v < (D >> 2); v += 64) dst[v] = load_tok4(src, v);
    if (lane == 0) {
        const int32_t c = p_cell[j];
        const int32_t h = c % nh, rc = c / nh;
        cache_pos[3 * (row0 + w) + 0] = rc / gs;
        cache_pos[3 * (row0 + w) + 1] = rc % gs;
        cache_pos[3 * (row0 + w) + 2] = h;
        cache_d[row0 + w] = p_r2f[j];
    }
}

// debug / parity entry: full geometry of one frame's points
__global__ __launch_bounds__(TPB) void k_geometry_debug(GeomConst gc, const float *depth, const int32_t *idx,
                                                        const double *T, int64_t P, uint8_t *flags, double *pc,
                                                        double *pg, int32_t *vox, int32_t *pix, int32_t *pat, double *r2,
                                                        double *alpha)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= P) return;
    const int32_t i = idx ? idx[j] : (int32_t)j;
    GeomOut o;
    memset(&o, 0, sizeof o);
    geom_point(gc, i, depth[i], T, o, true, gc.exp_tab);
    flags[j] = (uint8_t)o.flags;
    for (int k = 0; k < 3; ++k) { pc[3 * j + k] = o.pc[k]; pg[3 * j + k] = o.pg[k]; vox[3 * j + k] = o.vox[k]; }
    for (int k = 0; k < 2; ++k) { pix[2 * j + k] = o.pix[k]; pat[2 * j + k] = o.pat[k]; }
    r2[j] = o.r2;
    alpha[j] = o.alpha;
}

bsc_status launch_geometry_debug(bsc_ctx *x, const float *depth, const int32_t *idx, int64_t P, uint8_t *flags,
                                 double *pc, double *pg, int32_t *vox, int32_t *pix, int32_t *pat, double *r2,
                                 double *alpha)
{
    GeomConst gc = make_geom_const(x);
    hipLaunchKernelGGL(k_geometry_debug, dim3((unsigned)((P + TPB - 1) / TPB)), dim3(TPB), 0, x->stream, gc, depth, idx,
                       x->d_transforms, P, flags, pc, pg, vox, pix, pat, r2, alpha);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// launches the rgb chain + top-down map kernels of the last ingest call, if they have not been launched yet
bsc_status launch_pending_chain(bsc_ctx *x)
{
    if (!x->chain_pending) return BSC_OK;
    x->chain_pending = false;
    const int set = x->chain_set;
    BSC_HIP(hipStreamWaitEvent(x->side, x->ev_ready[set], 0));
    // A call followed at once by bsc_sync (the isolated call of the bench, a frame-by-frame user): the chain would start beside the
    // pair sort of the same call, whose look-back tiles then wait for CUs the chain's resident workgroups hold (pair sort 0.3 ->
    // 0.7 ms with two 8-wavefront chain workgroups per CU).  Behind the pair sort it overlaps the dense reduce alone.  In a pipeline
    // the chain is launched at the next call and the event has long fired.
    static const bool chain_after_psort = getenv("BSC_CHAIN_BESIDE_PAIRSORT") == nullptr;
    if (chain_after_psort && x->ev_psort_valid) BSC_HIP(hipStreamWaitEvent(x->side, x->ev_psort, 0));
    // CHAIN_WAVES wavefronts x 16 quads pull segments from the queue, longest first, in workgroups of 4 (one wavefront per
    // SIMD of a CU): the queue hands the longest segments to the first workgroups, which keeps the chain's long tail on one
    // or two CUs — a CU with a resident chain wavefront (178 VGPRs) cannot take a 512-register GEMM wavefront of the
    // caller's encoder.  What the kernels beside the chain lose is proportional to how long it runs (384-frame pipeline:
    // +1.0 ms per step for the 2.6 ms chain of the "hall" scene, +2.5 ms for the 6.2 ms chain of the "room" scene whose
    // hottest voxel collects 2e5 points per call), so the wave count is the one that finishes a scene without such a voxel
    // soonest (hall: 128 / 256 / 512 waves = 6.7 / 4.6 / 2.6 ms); a tail-bound call is indifferent to it.  Two wavefronts
    // per SIMD (workgroups of 8) slow the tail itself: 6 -> 9 ms.
    stat_begin(x, BSC_STAT_CHAIN, x->side);
    static const int chain_waves = getenv("BSC_CHAIN_WAVES") ? atoi(getenv("BSC_CHAIN_WAVES")) : CHAIN_WAVES;
    const bool rec8 = x->rec8_s[set];           // the record format of the call whose chain this is
    const GeomConst gc = make_geom_const(x);
#define BSC_LAUNCH_CHAIN(R8)                                                                                                    \
    hipLaunchKernelGGL(k_chain<R8>, dim3(chain_waves * 64 / CHAIN_WG), dim3(CHAIN_WG), 0, x->side, gc, x->sval_b_s[set],         \
                       x->bscal_s[set], x->seg_info_s[set], (const void *)x->p_rec_s[set], x->rgb_pos, x->rgb, x->weight, x->hmap, \
                       x->seg_last_s[set], x->c.grid_size, x->chain_order_base)
    if (rec8) BSC_LAUNCH_CHAIN(true); else BSC_LAUNCH_CHAIN(false);
#undef BSC_LAUNCH_CHAIN
    static const int long_waves = getenv("BSC_LONG_WAVES") ? atoi(getenv("BSC_LONG_WAVES")) : LONG_WAVES;
    // a wavefront per ~4096 points of the batch, at most long_waves (a frame-by-frame call launches a handful)
    int64_t nw = x->chain_points / 4096;
    nw = nw < 64 ? 64 : (nw > long_waves ? long_waves : nw);
    // 8 wavefronts per hot segment since round 6 (two workgroups per CU: one steps its rounds while the other waits for records or
    // at a barrier): chain 2.55 -> 1.95 ms per 768-frame call against the 16-wavefront form
    static const int long_nwv = getenv("BSC_LONG_NWV") ? atoi(getenv("BSC_LONG_NWV")) : 8;
    if (x->long_chain) {
        const RunOrder ro = {x->run_bits_s[set], x->ck_run_s[set], x->ck_start_s[set], x->run_val_s[set]};
#define BSC_LAUNCH_LONG(NWVV, R8, PARTV, GRID, ST)                                                                              \
    hipLaunchKernelGGL((k_chain_long<NWVV, R8, PARTV>), dim3((unsigned)(GRID)), dim3(NWVV * 64), 0, ST, gc, ro,                  \
                       x->bscal_s[set], x->seg_info_s[set], (const void *)x->p_rec_s[set], x->rgb_pos, x->rgb, x->weight,        \
                       x->hmap, x->seg_last_s[set], x->c.grid_size, x->chain_order_base)
        // round 6: the one-wavefront segments (three quarters of the room scene's points) as a launch of their own, at the occupancy
        // a chain of dependent steps needs, on a second side stream beside the hot tiles (BSC_CHAIN_SPLIT=0: one launch for both)
        static const bool chain_split = getenv("BSC_CHAIN_SPLIT") == nullptr || atoi(getenv("BSC_CHAIN_SPLIT")) != 0;
        if (chain_split && long_nwv != 16) {
            BSC_HIP(hipEventRecord(x->ev_chain0, x->side));                 // behind k_chain: the first points of the new voxels
            BSC_HIP(hipStreamWaitEvent(x->side2, x->ev_chain0, 0));
            // a workgroup per hot segment (the device counts them; the host bounds them by the points of the call), launched first
            int64_t ghot = x->chain_points >> HOT_MIN_LOG2;
            ghot = ghot < 1 ? 1 : (ghot > 2048 ? 2048 : ghot);
            if (rec8) BSC_LAUNCH_LONG(8, true, 1, ghot, x->side); else BSC_LAUNCH_LONG(8, false, 1, ghot, x->side);
            const int64_t gmid = (nw + 3) / 4;
            if (rec8) BSC_LAUNCH_LONG(4, true, 2, gmid, x->side2); else BSC_LAUNCH_LONG(4, false, 2, gmid, x->side2);
            BSC_HIP(hipEventRecord(x->ev_mid, x->side2));
            BSC_HIP(hipStreamWaitEvent(x->side, x->ev_mid, 0));
        } else if (long_nwv == 16) { if (rec8) BSC_LAUNCH_LONG(16, true, 0, (nw + 15) / 16, x->side); else BSC_LAUNCH_LONG(16, false, 0, (nw + 15) / 16, x->side); }
        else { if (rec8) BSC_LAUNCH_LONG(8, true, 0, (nw + 7) / 8, x->side); else BSC_LAUNCH_LONG(8, false, 0, (nw + 7) / 8, x->side); }
#undef BSC_LAUNCH_LONG
    }
#define BSC_LAUNCH_HWIN(R8)                                                                                                     \
    hipLaunchKernelGGL(k_hwin<R8>, dim3(256), dim3(TPB), 0, x->side, x->bscal_s[set], x->seg_info_s[set], x->seg_last_s[set],    \
                       x->rgb_pos, x->hmap, (const void *)x->p_rec_s[set], x->cv_map, x->c.grid_size, x->chain_order_base)
    if (rec8) BSC_LAUNCH_HWIN(true); else BSC_LAUNCH_HWIN(false);
#undef BSC_LAUNCH_HWIN
    stat_end(x, BSC_STAT_CHAIN, 0.0, x->side);
    BSC_HIP(hipEventRecord(x->ev_done[set], x->side));
    x->ev_done_valid[set] = true;
    x->last_chain_set = set;
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

bsc_status ingest_batch(bsc_ctx *x, int32_t n_frames, const float *depth, const uint8_t *rgb, int32_t rgb_ch,
                        const void *tokens, int token_dtype, const int32_t *idx, const int64_t *offsets_host,
                        const double *alpha, bsc_draw_fn draw, void *user)
{
    const int64_t N = (int64_t)x->c.height * x->c.width;
    const int64_t P = idx ? offsets_host[n_frames] : (int64_t)n_frames * N;
    if (P > x->c.max_points) {
        bsc_set_error("bsc_ingest: %lld points exceed max_points=%d", (long long)P, x->c.max_points);
        return BSC_E_CAPACITY;
    }
    if (P == 0) return BSC_OK;
    if (x->log_cap && x->log_n + P > x->log_cap) {
        bsc_set_error("bsc_ingest: point log full (%lld + %lld > %lld points)", (long long)x->log_n, (long long)P, (long long)x->log_cap);
        return BSC_E_CAPACITY;
    }
    const dim3 block(TPB);
    hipStream_t s = x->stream;
    const bool exact = x->c.mode == BSC_MODE_EXACT;
    BSC_TRY(launch_pending_chain(x));          // the previous call's rgb chain runs beside this call's front end
    // scratch set of this call; the rgb chain of the call before last may still be reading it on the side stream
    const int set = x->cur_set;
    x->cur_set ^= 1;
    if (x->ev_done_valid[set]) BSC_HIP(hipStreamWaitEvent(s, x->ev_done[set], 0));
    if (x->ev_runs_valid) BSC_HIP(hipStreamWaitEvent(s, x->ev_runs, 0));      // the last call's k_runs (side stream) is done with p_cell / blk_off
    PointRec *p_rec = x->p_rec_s[set];
    uint32_t *skey_b = x->skey_b_s[set];
    if (idx)
        BSC_HIP(hipMemcpyAsync(x->d_offsets, offsets_host, sizeof(int64_t) * (n_frames + 1), hipMemcpyHostToDevice, s));
    BSC_HIP(hipMemsetAsync(x->dscal + DS_B_NNEW, 0, sizeof(int64_t), s));
    GeomConst gc = make_geom_const(x);
    // a run's length travels in the key bits beside the voxel id: the id field is sized for the voxel capacity, the
    // rest (at most 10 bits) holds length - 1; k_points cuts longer groups into several runs
    const int vb = ceil_log2_u64((uint64_t)x->c.voxel_capacity + 2);
    const int lb = 32 - vb < 10 ? 32 - vb : 10;
    const int GBP = x->group_rpw * 256;                 // points per k_points workgroup
    const int64_t nblk = (P + GBP - 1) / GBP;
    const dim3 fgrid((unsigned)nblk);
    const bool all_px_dense = !exact && idx == nullptr && x->geom_fast;     // the patch comes from the pixel: no p_patf
    uint32_t *patf = all_px_dense ? (uint32_t *)nullptr : x->p_patf;
    float *r2f = exact ? x->p_r2f : (float *)nullptr;
    const float inv_w = 1.0f / (float)x->c.width;
    int32_t *g_cell = x->log_cap ? x->log_cell + x->log_n : (int32_t *)nullptr;    // bsc_point_log_*: cells in record order
    stat_begin(x, BSC_STAT_INGEST);
    stat_begin(x, BSC_STAT_POINTS);
#define BSC_LAUNCH_POINTS(FASTV, RPWV, PLAINV, R8)                                                                             \
    hipLaunchKernelGGL((k_points<FASTV, RPWV, PLAINV, R8>), fgrid, block, 0, s, gc, depth, rgb, rgb_ch, idx, x->d_offsets, n_frames, \
                       x->d_transforms, alpha, P, inv_w, lb, x->occ, x->p_cell, patf, (void *)p_rec, r2f, x->new_cells, x->dscal, \
                       x->blk_cnt, x->blk_pass, x->stage_cell, x->stage_pos, g_cell)
    // (PLAIN also takes for granted that every pixel lies inside the patch grid — bsc_create checked the tables — and reads none)
    // and that a frame is a whole number of wavefront slices (RPW x 64 points: 640x480 = 600 x 512), so that a wavefront never
    // crosses into the next frame and its transform stays what it loaded at the start: with the conditional reload inside the
    // rounds the compiler waited for EVERY load in flight at the head of each round — the previous round's colour gather included
    const bool plain = !idx && !patf && !r2f && !alpha && !g_cell && x->pat_all_in && N % (x->group_rpw * 64) == 0 && x->c.width >= 64;
    // 8-byte records {rgb, index in block, depth offset} where the depth range allows (bsc_create), alpha left to the rgb chain
    const bool rec8 = gc.fast && plain && x->rec8_ok;
    x->rec8_s[set] = rec8;
#define BSC_LAUNCH_POINTS_R(FASTV, PLAINV, R8)                                                                                 \
    do {                                                                                                                       \
        if (x->group_rpw == 16) BSC_LAUNCH_POINTS(FASTV, 16, PLAINV, R8);                                                      \
        else if (x->group_rpw == 8) BSC_LAUNCH_POINTS(FASTV, 8, PLAINV, R8);                                                   \
        else BSC_LAUNCH_POINTS(FASTV, 4, PLAINV, R8);                                                                          \
    } while (0)
    if (rec8) BSC_LAUNCH_POINTS_R(true, true, true);
    else if (gc.fast && plain) BSC_LAUNCH_POINTS_R(true, true, false);
    else if (gc.fast) BSC_LAUNCH_POINTS_R(true, false, false);
    else BSC_LAUNCH_POINTS_R(false, false, false);
#undef BSC_LAUNCH_POINTS_R
#undef BSC_LAUNCH_POINTS
    stat_end(x, BSC_STAT_POINTS, 0.0);
    if (x->log_cap)             // the call's records, block-grouped like p_rec (every voxel's points still in order j)
        BSC_HIP(hipMemcpyAsync(x->log_rec + x->log_n, p_rec, sizeof(PointRec) * (size_t)P, hipMemcpyDeviceToDevice, s));
    static const bool fused_totals = getenv("BSC_TOTALS_UNFUSED") == nullptr;
    const bool early = x->order_on_side && !exact;
    const bool mailbox = fused_totals && early && x->mail != nullptr;
    if (fused_totals) {
        x->mail_seq += 1;
        hipLaunchKernelGGL(k_block_totals, dim3(1), dim3(BT_THREADS), 0, s, P, nblk, x->blk_cnt, x->blk_off, x->blk_pass, x->blk_pass_off,
                           x->dscal, x->c.voxel_capacity, x->bscal_s[set], mailbox ? x->mail_dev : (int64_t *)nullptr, x->mail_seq);
    } else {
        BSC_TRY(prim_exclusive_sum_i32(x, x->blk_cnt, x->blk_off, (size_t)nblk));
        BSC_TRY(prim_exclusive_sum_i32(x, x->blk_pass, x->blk_pass_off, (size_t)nblk));
        hipLaunchKernelGGL(k_totals, dim3(1), dim3(64), 0, s, P, nblk, x->blk_cnt, x->blk_off, x->blk_pass, x->blk_pass_off, x->dscal,
                           x->c.voxel_capacity, x->bscal_s[set]);
    }
    // early: (dense modes, order stage on the side stream) the counts k_totals wrote come back over a copy stream WHILE the main
    // stream runs the pair tiles, and the new-voxel ids + order stage are enqueued on the side stream during that time; the pair
    // count follows with a second, short readback.  With ONE readback after the pair tiles the host came back to an empty main
    // stream and spent ~0.26 ms enqueueing before it had work again, and the order stage (hence the rgb chain) started 0.45 ms late.
    if (early) BSC_HIP(hipEventRecord(x->ev_tot, s));
    stat_begin(x, BSC_STAT_PAIRS);
    BSC_TRY(launch_keys_pairs(x, P, n_frames, idx == nullptr, patf));
    stat_end(x, BSC_STAT_PAIRS, 0.0);
    // one small readback per call: new voxels, runs, pairs (dense modes), passing points (exact mode), capacity flag.
    // Everything enqueued so far is the call's front end; the back end is sized from these numbers.
    if (early) {
        // the pair count: copied behind the pair tiles right away (its own pinned slot), waited for after the side stream's launches
        BSC_HIP(hipMemcpyAsync(x->hscal + DS_COUNT, x->dscal + DS_B_NPAIR, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        bool got = false;
        if (mailbox) {
            // the scalars arrive in the mailbox behind k_block_totals; spin on its sequence number (bounded: a failed launch or a
            // lost device never writes it — after ~2 s fall back to the copy, which reports the error)
            volatile int64_t *const mb = x->mail;
            const u64 tag = (u64)x->mail_seq & 0xffffull;
            const auto all_tagged = [&]() {
                for (int k = 0; k < DS_COUNT; ++k)
                    if (((u64)__atomic_load_n(&mb[k], __ATOMIC_ACQUIRE) & 0xffffull) != tag) return false;
                return true;
            };
            for (int64_t spin = 0; spin < (1ll << 31); ++spin) {
                if (((u64)__atomic_load_n(&mb[DS_COUNT - 1], __ATOMIC_ACQUIRE) & 0xffffull) == tag && all_tagged()) { got = true; break; }
                if ((spin & 1023) == 1023) sched_yield();       // behind a caller's encoder pass the wait is long: let other threads run
                if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(s) != hipErrorNotReady) { got = all_tagged(); break; }
            }
            if (got) for (int k = 0; k < DS_COUNT; ++k) x->hscal[k] = (int64_t)mb[k] >> 16;       // (arithmetic: a value in +-2^47 survives)
        }
        if (!got) {
            BSC_HIP(hipStreamWaitEvent(x->copy, x->ev_tot, 0));
            BSC_HIP(hipMemcpyAsync(x->hscal, x->dscal, sizeof(int64_t) * DS_COUNT, hipMemcpyDeviceToHost, x->copy));
            BSC_HIP(hipStreamSynchronize(x->copy));
        }
    } else {
        BSC_TRY(read_scalars(x));
    }
    const int64_t n_new_listed = x->hscal[DS_B_NNEW], n_new = x->hscal[DS_B_NFIRST];
    // ids of the new voxels: rank of their winning point among the winners (memory_2.py:888-894)
    auto assign_ids = [&](hipStream_t st) -> bsc_status {
        if (n_new_listed > 0) {
            const dim3 ngrid((unsigned)((n_new_listed + TPB - 1) / TPB));
            hipLaunchKernelGGL(k_new_keys, ngrid, block, 0, st, n_new_listed, x->new_cells, x->occ, x->skey_a, x->sval_a);
            BSC_TRY(prim_sort_pairs_u32(x, x->skey_a, skey_b, x->sval_a, x->run_val_b, (size_t)n_new_listed, 0,
                                        ceil_log2_u64((uint64_t)P + 1)));
            hipLaunchKernelGGL(k_new_assign, ngrid, block, 0, st, n_new_listed, n_new, x->run_val_b, x->occ, x->dscal,
                               x->c.grid_size, x->nh, x->rgb_pos);
        }
        return BSC_OK;
    };
    if (!early) {
        // The buffers the back end shares with the order stage (run keys / values and their sort outputs) may still be in use by
        // the previous call's order stage on the side stream when calls follow each other without an encoder pass in between
        if (x->last_order_set >= 0) BSC_HIP(hipStreamWaitEvent(s, x->ev_ready[x->last_order_set], 0));
        BSC_TRY(assign_ids(s));
    }
    if (x->hscal[DS_ERROR]) {
        if (early) {
            // k_totals has already advanced max_id by the (clipped) count of new voxels: their ids are still assigned — on the side
            // stream, as the order stage would have done — so that the map the refused call leaves behind is consistent
            void *const keep_tmp = x->prim_tmp;
            BSC_HIP(hipStreamWaitEvent(x->side, x->ev_tot, 0));
            x->stream = x->side; x->prim_tmp = x->prim_tmp_side;
            const bsc_status ist = assign_ids(x->side);
            x->stream = s; x->prim_tmp = keep_tmp;
            BSC_TRY(ist);
            BSC_HIP(hipEventRecord(x->ev_ids, x->side));
            BSC_HIP(hipStreamWaitEvent(s, x->ev_ids, 0));
        }
        bsc_set_error("voxel capacity %d exceeded", x->c.voxel_capacity);
        return BSC_E_CAPACITY;
    }
    if (x->log_cap) x->log_n += P;             // the log keeps the call only once it can no longer fail
    // The per-voxel point order (k_runs .. k_seg_order) feeds only the rgb chain, which runs on the side stream anyway: it is
    // enqueued THERE, as soon as the ids exist, and runs beside the pair sort / dense reduce of this call on the main stream.
    // Both halves are chains of short memory-bound kernels with launch gaps between them; side by side each fills the other's
    // gaps (own rocPRIM workspace, disjoint buffers).  BSC_ORDER_MAIN=1: one stream, as in round 2.
    const bool side_order = x->order_on_side;
    hipStream_t so = side_order ? x->side : s;
    if (early) {
        BSC_HIP(hipStreamWaitEvent(so, x->ev_tot, 0));          // the side stream takes over from k_totals: ids, then the order stage
    } else if (side_order) {
        BSC_HIP(hipEventRecord(x->ev_ids, s));
        BSC_HIP(hipStreamWaitEvent(so, x->ev_ids, 0));
    } else if (!exact) {
        BSC_TRY(dense_reduce_batch(x, tokens, token_dtype, n_frames));
    }
    // order-pipeline launches go through the ctx's stream / workspace fields
    void *const prim_main = x->prim_tmp;
    if (side_order) { x->stream = so; x->prim_tmp = x->prim_tmp_side; }
    const bsc_status order_st = [&]() -> bsc_status {
    if (early) {
        BSC_TRY(assign_ids(so));
        BSC_HIP(hipEventRecord(x->ev_ids, so));
    }
    stat_begin(x, BSC_STAT_ORDER, so);
    // stable radix sort of the RUNS on the voxel id alone: runs enter in order j, so each voxel's runs stay in order;
    // their expansion is the per-voxel point order
    const int64_t R = x->hscal[DS_B_NRUN];
    uint32_t *sj = x->sval_b_s[set];
    {
        const dim3 rgrid((unsigned)((nblk + TPB / 64 - 1) / (TPB / 64)));
#define BSC_LAUNCH_RUN_KEYS(GBV)                                                                                               \
    do {                                                                                                                       \
        hipLaunchKernelGGL(k_run_keys<GBV>, rgrid, block, 0, so, nblk, vb, x->stage_cell, x->stage_pos, x->occ, x->blk_cnt,    \
                           x->blk_off, x->blk_pass, x->skey_a, x->sval_a);                                                     \
        if (exact) hipLaunchKernelGGL(k_pass_list<GBV>, fgrid, block, 0, so, P, x->p_cell, x->blk_pass_off, x->pass_list);     \
    } while (0)
        if (x->group_rpw == 16) BSC_LAUNCH_RUN_KEYS(4096);
        else if (x->group_rpw == 8) BSC_LAUNCH_RUN_KEYS(2048);
        else BSC_LAUNCH_RUN_KEYS(1024);
#undef BSC_LAUNCH_RUN_KEYS
    }
    if (side_order) { BSC_HIP(hipEventRecord(x->ev_runs, so)); x->ev_runs_valid = true; }
    // ids in use are < max_id; runs without a voxel carry an all-ones id field, which sorts last under the bit mask
    const int vid_bits = ceil_log2_u64((uint64_t)x->hscal[DS_MAX_ID] + 2);
    if (R > 0) {        // a batch without a single passing point has no runs (k_totals left the segment count at 0)
        if (x->radix_intree)
            BSC_TRY(radix_sort_pairs_u32(x, side_order ? &x->rx_side : &x->rx_main, so, x->skey_a, skey_b, x->sval_a, x->run_val_s[set], (size_t)R, 0,
                                         vid_bits < vb ? vid_bits : vb));
        else
            BSC_TRY(prim_sort_pairs_u32(x, x->skey_a, skey_b, x->sval_a, x->run_val_s[set], (size_t)R, 0, vid_bits < vb ? vid_bits : vb));
        const int64_t neb = (R + EB - 1) / EB;
        hipLaunchKernelGGL(k_run_blocksum, dim3((unsigned)neb), block, 0, so, R, vb, skey_b, x->run_scan);
        BSC_TRY(prim_exclusive_sum_i64(x, x->run_scan, x->run_scan + neb, (size_t)neb));
        BSC_HIP(hipMemsetAsync(x->run_bits_s[set], 0, sizeof(u64) * (size_t)((P >> 6) + 2), so));
        hipLaunchKernelGGL(k_expand, dim3((unsigned)neb), block, 0, so, R, vb, skey_b, x->run_scan + neb, x->run_bits_s[set],
                           x->ck_run_s[set], x->ck_start_s[set], x->seg_k0, x->seg_vid, x->bscal_s[set]);
    }
    const int64_t seg_cap = (x->c.max_points < x->c.voxel_capacity ? x->c.max_points : x->c.voxel_capacity) + 1;
    int64_t n_bound = R < x->hscal[DS_MAX_ID] ? R : x->hscal[DS_MAX_ID];     // segments <= runs, <= voxels
    if (n_bound > seg_cap) n_bound = seg_cap;
    hipLaunchKernelGGL(k_seg_bounds, dim3(64), block, 0, so, x->bscal_s[set], n_bound, x->seg_k0, x->seg_vid,
                       x->seg_info_s[set], x->skey_a, x->sval_a);
    if (n_bound > 0) {
        if (x->radix_intree)
            BSC_TRY(radix_sort_pairs_u32(x, side_order ? &x->rx_side : &x->rx_main, so, x->skey_a, (uint32_t *)x->seg_k0, x->sval_a, (uint32_t *)x->seg_vid,
                                         (size_t)n_bound, 0, 6));
        else
            BSC_TRY(prim_sort_pairs_u32(x, x->skey_a, (uint32_t *)x->seg_k0, x->sval_a, (uint32_t *)x->seg_vid, (size_t)n_bound, 0, 6));
    }
    static const int long_log2 = getenv("BSC_LONG_LOG2") ? atoi(getenv("BSC_LONG_LOG2")) : LONG_MIN_LOG2;
    static const int hot_log2 = getenv("BSC_NO_HOT_SPLIT") ? 0 : (getenv("BSC_HOT_LOG2") ? atoi(getenv("BSC_HOT_LOG2")) : HOT_MIN_LOG2);
    hipLaunchKernelGGL(k_seg_order, dim3(64), block, 0, so, x->bscal_s[set], (const uint32_t *)x->seg_k0, (const uint32_t *)x->seg_vid,
                       x->seg_info_s[set], x->long_chain ? long_log2 : 0, hot_log2);
    if (R > 0) {
        // order indices for the quad chain: everything when there is no long chain (bscal[4] stays 0)
        const RunOrder ro = {x->run_bits_s[set], x->ck_run_s[set], x->ck_start_s[set], x->run_val_s[set]};
        int64_t nw = P / 1024;
        nw = nw < 64 ? 64 : (nw > 4096 ? 4096 : nw);
        hipLaunchKernelGGL(k_expand_short, dim3((unsigned)((nw + TPB / 64 - 1) / (TPB / 64))), block, 0, so, x->bscal_s[set],
                           x->seg_info_s[set], ro, sj, LONG_EARLY);
    }
    stat_end(x, BSC_STAT_ORDER, 0.0, so);
    return BSC_OK;
    }();
    x->stream = s;
    x->prim_tmp = prim_main;
    BSC_TRY(order_st);
    if (early) {
        // the pair count (copied behind the pair tiles above), then the back end behind the ids
        BSC_HIP(hipStreamSynchronize(s));
        x->hscal[DS_B_NPAIR] = x->hscal[DS_COUNT];
        BSC_HIP(hipStreamWaitEvent(s, x->ev_ids, 0));
        BSC_TRY(dense_reduce_batch(x, tokens, token_dtype, n_frames));
    } else if (side_order && !exact) {
        BSC_TRY(dense_reduce_batch(x, tokens, token_dtype, n_frames));
    }
    if (side_order && exact) BSC_HIP(hipStreamWaitEvent(s, x->ev_runs, 0));       // k_append reads the pass list k_runs wrote
    // rgb chain + top-down map: sequential-latency bound, on the library's side stream — and DEFERRED: the
    // call only marks its point order ready; the kernels are launched at the start of the next bsc_ingest (or by whatever
    // needs their result first: exports, merges, resets — sync_all).  The chain's long tail (one voxel seen in every frame
    // is a single dependent sequence of ~10^5 steps) then overlaps the next call's memory-bound front end instead of the
    // caller's encoder: its few resident wavefronts are harmless beside streaming kernels, but a library GEMM that splits
    // its work statically over all 256 CUs (stream-K) runs up to twice as long while any CU is held by a chain wavefront
    // (measured: 26.4 -> 22.3 ms per 384-frame step without the chain beside the encoder).
    BSC_HIP(hipEventRecord(x->ev_ready[set], so));
    x->last_order_set = side_order ? set : -1;
    x->chain_pending = true;
    x->chain_set = set;
    x->chain_order_base = x->order_base;
    x->chain_points = P;
    stat_end(x, BSC_STAT_INGEST, 0.0);
    BSC_HIP(hipGetLastError());
    x->order_base += P;
    static const bool chain_eager = getenv("BSC_CHAIN_EAGER") != nullptr;
    if (chain_eager) BSC_TRY(launch_pending_chain(x));      // A/B: the chain right behind its order stage instead of at the next call
    if (x->c.mode == BSC_MODE_EXACT) {
        // memory_2.py:880-886: rows fill the cache in order; the point that finds it full triggers the
        // flush and loses its own token.
        int64_t remaining = x->hscal[DS_B_NPASS], q = 0;
        while (remaining > 0) {
            const int64_t room = x->c.iter_size - x->iter_id;
            const int64_t n = remaining < room ? remaining : room;
            if (n > 0) {
                const dim3 agrid((unsigned)((n * 64 + TPB - 1) / TPB));
                if (token_dtype == BSC_TOK_BF16)
                    hipLaunchKernelGGL(k_append<bf16_t>, agrid, block, 0, s, x->pass_list, q, n, x->iter_id, x->p_cell,
                                       x->p_patf, x->p_r2f, (const bf16_t *)tokens, x->g2, x->c.token_dim, x->c.grid_size,
                                       x->nh, x->cache_f, x->cache_pos, x->cache_d);
                else
                    hipLaunchKernelGGL(k_append<float>, agrid, block, 0, s, x->pass_list, q, n, x->iter_id, x->p_cell,
                                       x->p_patf, x->p_r2f, (const float *)tokens, x->g2, x->c.token_dim, x->c.grid_size,
                                       x->nh, x->cache_f, x->cache_pos, x->cache_d);
                BSC_HIP(hipGetLastError());
                x->iter_id += n; q += n; remaining -= n;
            }
            if (remaining > 0) {        // next passing point meets a full cache
                BSC_TRY(flush_cache(x, draw, user));
                q += 1; remaining -= 1;
            }
        }
    }
    return BSC_OK;
}
