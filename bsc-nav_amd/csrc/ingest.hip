// ingest.hip — obs2voxeltoken's per-point loop (memory_2.py:859-903) as a batch pipeline on gfx950.
//
// The reference walks the sampled points of a frame sequentially; ids, the rgb running mean, the
// top-down map and the token cache are all defined by that order.  Here every point j of a batch
// (frames in call order, points in the reference's shuffled order) carries its order implicitly:
//
//   k_points    geometry (fp64, bit-exact) -> cell / patch / rgb / r2 / alpha; atomicMin claims the
//               first toucher of every still-empty cell                         (1 thread / point)
//   k_flags + scan of block totals + k_assign  first-touch points get ids max_id + rank in order
//   k_keys_pairs (dense.hip) sort key = voxel id, value = j; LDS aggregation of (voxel, frame, patch) pairs
//   radix sort  (stable, on the voxel id bits only) groups the points of a voxel in order; the segment
//               starts are compacted deterministically (block counts + scan)
//   k_chain     per voxel: sequential truncating weighted rgb mean + top-down map atomicMax on
//               (h, order of the voxel's latest point)                        (1 wavefront / voxel)
//   k_hwin      the winning voxel of each map cell writes its colour
//   dense.hip   pair sort + k_dense_reduce: multiplicity x token rows -> one RMW of the D-float
//               accumulator row per voxel                                     (1 wavefront / voxel)
//   k_append    exact mode: token rows into the cache in order                (1 wavefront / row)
#include "bsc_internal.h"
#include "geometry_dev.h"

#include <limits.h>
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
__global__ __launch_bounds__(TPB) void k_points(GeomConst gc, const float *__restrict__ depth,
                                                const uint8_t *__restrict__ rgb, int rgb_ch,
                                                const int32_t *__restrict__ idx, const int64_t *__restrict__ offsets,
                                                int n_frames, const double *__restrict__ transforms,
                                                const double *__restrict__ alpha_in, int64_t P, int32_t *occ,
                                                int32_t *__restrict__ p_cell, uint32_t *__restrict__ p_patf,
                                                PointRec *__restrict__ p_rec, float *__restrict__ p_r2f)
{
    const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (j >= P) return;
    const int64_t N = (int64_t)gc.H * gc.W;
    int f;
    int32_t i;
    if (idx) {
        f = frame_of(offsets, n_frames, j);
        i = idx[j];
    } else {
        f = (int)(j / N);
        i = (int32_t)(j - (int64_t)f * N);
    }
    const float z = depth[(int64_t)f * N + i];
    GeomOut o;
    geom_point(gc, i, z, transforms + 16 * f, o, alpha_in == nullptr);
    if (o.flags != 7u) {
        p_cell[j] = -1;
        return;
    }
    const int32_t row = o.vox[0], col = o.vox[1], h = o.vox[2] - gc.min_h;   // memory_2.py:867
    const int32_t cell = (row * gc.gs + col) * gc.nh + h;
    int sx = o.pix[0], sy = o.pix[1];            // memory_2.py:870 rgb[py, px]: negative indices wrap
    if (sx < 0) sx += gc.W;
    if (sy < 0) sy += gc.H;
    sx = min(max(sx, 0), gc.W - 1);
    sy = min(max(sy, 0), gc.H - 1);
    const uint8_t *pv = rgb + ((int64_t)f * N + (int64_t)sy * gc.W + sx) * rgb_ch;
    p_cell[j] = cell;
    p_patf[j] = ((uint32_t)f << 16) | (uint32_t)(o.pat[1] * gc.g + o.pat[0]);   // tokens[py, px]
    if (p_r2f) p_r2f[j] = (float)o.r2;            // memory_2.py:885 grid_feat_dis is float32 (token cache only)
    PointRec rec;
    rec.alpha = alpha_in ? alpha_in[j] : o.alpha;
    rec.rgbv = (uint32_t)pv[0] | ((uint32_t)pv[1] << 8) | ((uint32_t)pv[2] << 16);
    rec.pad = 0;
    p_rec[j] = rec;
    // first-touch claim: the smallest j wins an empty cell (ids are handed out in k_assign)
    if (occ[cell] < 0) atomicMin(&occ[cell], INT_MIN + (int32_t)j);
}

// ---- first-touch ranks without a per-point scan -----------------------------------------------------------------
// Blocks of FB consecutive points count their passing / first-touch points (k_flags); an exclusive scan over the
// ~P/1024 block totals gives every block its base; k_assign recomputes the flags and ranks its points inside the
// block in order j with wave ballots.  Same ranks as a scan over all P points, at a third of the traffic.
#define FB 1024
__device__ __forceinline__ int point_flags(int64_t j, int64_t P, const int32_t *__restrict__ p_cell,
                                           const int32_t *__restrict__ occ, int32_t &cell)
{
    cell = -1;
    if (j >= P) return 0;
    cell = p_cell[j];
    if (cell < 0) return 0;
    return (occ[cell] == INT_MIN + (int32_t)j) ? 3 : 1;     // bit0 passes, bit1 first toucher of its voxel
}

__global__ __launch_bounds__(TPB) void k_flags(int64_t P, const int32_t *__restrict__ p_cell,
                                               const int32_t *__restrict__ occ, int64_t *__restrict__ blk_tot)
{
    __shared__ int s_pass, s_first;
    if (threadIdx.x == 0) { s_pass = 0; s_first = 0; }
    __syncthreads();
    int np = 0, nf = 0;
    for (int r = 0; r < FB / TPB; ++r) {
        int32_t c;
        const int f = point_flags((int64_t)blockIdx.x * FB + r * TPB + threadIdx.x, P, p_cell, occ, c);
        np += f & 1;
        nf += f >> 1;
    }
    for (int o = 32; o > 0; o >>= 1) { np += __shfl_xor(np, o); nf += __shfl_xor(nf, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_pass, np); atomicAdd(&s_first, nf); }
    __syncthreads();
    if (threadIdx.x == 0) blk_tot[blockIdx.x] = ((int64_t)s_first << 32) | (int64_t)s_pass;
}

__global__ __launch_bounds__(TPB) void k_assign(int64_t P, const int32_t *__restrict__ p_cell, int32_t *occ,
                                                const int64_t *__restrict__ blk_off, int64_t *dscal, int vcap, int gs,
                                                int nh, int32_t *__restrict__ rgb_pos, int32_t *__restrict__ pass_list)
{
    __shared__ int w_pass[TPB / 64], w_first[TPB / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t base = blk_off[blockIdx.x];
    int64_t pass_base = base & 0xffffffffll, first_base = base >> 32;
    const int64_t max_id = dscal[DS_MAX_ID];
    for (int r = 0; r < FB / TPB; ++r) {            // rounds keep the order j: round, then wave, then lane
        const int64_t j = (int64_t)blockIdx.x * FB + r * TPB + threadIdx.x;
        int32_t c;
        const int f = point_flags(j, P, p_cell, occ, c);
        const u64 mp = __ballot(f & 1), mf = __ballot(f & 2);
        if (lane == 0) { w_pass[wid] = __popcll(mp); w_first[wid] = __popcll(mf); }
        __syncthreads();
        int bp = 0, bf = 0, tp = 0, tf = 0;
        for (int w = 0; w < TPB / 64; ++w) {
            if (w < wid) { bp += w_pass[w]; bf += w_first[w]; }
            tp += w_pass[w]; tf += w_first[w];
        }
        const u64 lt = (1ull << lane) - 1ull;
        if (f & 1) {
            if (pass_list) pass_list[pass_base + bp + __popcll(mp & lt)] = (int32_t)j;
            if (f & 2) {
                const int64_t id = max_id + first_base + bf + __popcll(mf & lt);
                if (id >= vcap) {
                    dscal[DS_ERROR] = 1;        // capacity: the cell keeps its provisional (negative) value
                } else {
                    occ[c] = (int32_t)id;           // memory_2.py:890
                    const int32_t h = c % nh, rc = c / nh;
                    rgb_pos[3 * id + 0] = rc / gs;  // memory_2.py:893
                    rgb_pos[3 * id + 1] = rc % gs;
                    rgb_pos[3 * id + 2] = h;
                }
            }
        }
        pass_base += tp;
        first_base += tf;
        __syncthreads();
    }
}

__global__ void k_totals(int64_t P, int64_t nblk, const int64_t *blk_tot, const int64_t *blk_off, int64_t *dscal, int vcap,
                         int64_t *bscal)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t tot = blk_off[nblk - 1] + blk_tot[nblk - 1];
    const int64_t npass = tot & 0xffffffffll, nfirst = tot >> 32;
    dscal[DS_B_NPASS] = npass;
    dscal[DS_B_NFIRST] = nfirst;
    dscal[DS_MAX_ID_PREV] = dscal[DS_MAX_ID];
    int64_t m = dscal[DS_MAX_ID] + nfirst;
    if (m > vcap) { m = vcap; dscal[DS_ERROR] = 1; }
    dscal[DS_MAX_ID] = m;
    dscal[DS_NPASS_TOTAL] += npass;
    dscal[DS_NSEEN_TOTAL] += P;
    dscal[DS_B_NSEG] = 0;
    bscal[0] = 0;
    bscal[1] = dscal[DS_MAX_ID_PREV];
    dscal[DS_B_NPAIR] = 0;
    dscal[DS_B_NPSEG] = 0;
}

// memory_2.py:888-903 — one WAVEFRONT per voxel walks that voxel's points of the batch in order.
// The chain c' = trunc((f32(c*w) + r*a) / (w + a)), w' = f32(w + a) is sequential by definition (truncation
// and f32 rounding at every step), so parallelism is across voxels; within the wave the 64 lanes prefetch 64
// points at a time (coalesced keys, gathered rgb / alpha) and the steps run out of registers through
// wave shuffles, lanes 0..2 carrying the R, G, B channels.  The same wave settles the top-down map:
// `h >= max_height` in sequential order == max over (h, order), and a voxel's latest point is the last
// element of its segment, so one atomicMax per voxel replaces one per point.
__global__ __launch_bounds__(TPB) void k_chain(int64_t P, const uint32_t *__restrict__ skey,
                                               const uint32_t *__restrict__ sval, const int64_t *bscal,
                                               const int32_t *__restrict__ seg_start,
                                               const PointRec *__restrict__ p_rec,
                                               const int32_t *__restrict__ rgb_pos, uint8_t *__restrict__ rgb,
                                               float *__restrict__ weight, u64 *hmap, int32_t *__restrict__ seg_last,
                                               int gs, int64_t order_base)
{
    const int lane = threadIdx.x & 63;
    const int ch = lane < 3 ? lane : 2;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    const int64_t nseg = bscal[0];
    const int64_t max_id_prev = bscal[1];
    for (int64_t s = wave; s < nseg; s += nwaves) {
        const int64_t i0 = seg_start[s];
        const uint32_t vid = skey[i0];
        const bool is_new = (int64_t)vid >= max_id_prev;
        float w = 0.f;
        uint32_t c = 0;
        if (!is_new) {
            w = weight[vid];
            c = rgb[3 * (int64_t)vid + ch];
        }
        bool first = is_new;
        uint32_t last_j = 0;
        for (int64_t base = i0;; base += 64) {
            const int64_t k = base + lane;
            const bool inseg = (k < P) && (skey[k] == vid);
            const uint32_t j = inseg ? sval[k] : 0u;
            PointRec rec;
            rec.alpha = 0.0; rec.rgbv = 0u;
            if (inseg) rec = p_rec[j];
            const uint32_t rv = rec.rgbv;
            const double al = rec.alpha;
            const int n = __popcll(__ballot(inseg));
            for (int t = 0; t < n; ++t) {
                const double a = __shfl(al, t);
                const uint32_t r = (__shfl(rv, t) >> (8 * ch)) & 0xffu;
                if (first) {                    // :890-894 new id: rgb = rgb_v, weight = f32(0 + alpha)
                    c = r;
                    w = (float)((double)w + a);
                    first = false;
                } else {                        // :896-899 u8*f32 -> f32 ; u8*f64 -> f64 ; truncating store
                    const double den = (double)w + a;
                    const double v = ((double)((float)c * w) + (double)r * a) / den;
                    c = (uint32_t)(uint8_t)v;
                    w = (float)den;
                }
            }
            if (n > 0) last_j = __shfl(j, n - 1);
            if (n < 64) break;
        }
        if (lane < 3) rgb[3 * (int64_t)vid + lane] = (uint8_t)c;
        if (lane == 0) {
            weight[vid] = w;
            const int32_t row = rgb_pos[3 * (int64_t)vid], col = rgb_pos[3 * (int64_t)vid + 1], h = rgb_pos[3 * (int64_t)vid + 2];
            const u64 packed = ((u64)(h + 1) << 40) | (u64)(order_base + last_j);
            atomicMax(&hmap[(int64_t)row * gs + col], packed);
            seg_last[s] = (int32_t)last_j;
        }
    }
}

// top-down map colour: the voxel whose (h, order) won the cell writes the rgb of its latest point
__global__ __launch_bounds__(TPB) void k_hwin(const int64_t *bscal, const uint32_t *__restrict__ skey,
                                              const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_last,
                                              const int32_t *__restrict__ rgb_pos, const u64 *__restrict__ hmap,
                                              const PointRec *__restrict__ p_rec, uint8_t *__restrict__ cv_map, int gs,
                                              int64_t order_base)
{
    const int64_t nseg = bscal[0];
    for (int64_t s = (int64_t)blockIdx.x * TPB + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * TPB) {
        const uint32_t vid = skey[seg_start[s]];
        const uint32_t last_j = (uint32_t)seg_last[s];
        const int32_t row = rgb_pos[3 * (int64_t)vid], col = rgb_pos[3 * (int64_t)vid + 1], h = rgb_pos[3 * (int64_t)vid + 2];
        const int64_t rc = (int64_t)row * gs + col;
        const u64 packed = ((u64)(h + 1) << 40) | (u64)(order_base + last_j);
        if (hmap[rc] == packed) {
            const uint32_t v = p_rec[last_j].rgbv;
            cv_map[3 * rc + 0] = (uint8_t)(v & 0xff);
            cv_map[3 * rc + 1] = (uint8_t)((v >> 8) & 0xff);
            cv_map[3 * rc + 2] = (uint8_t)((v >> 16) & 0xff);
        }
    }
}

// exact mode, memory_2.py:882-886: rows [row0, row0+n) of the token cache <- passing points [q0, q0+n)
__global__ __launch_bounds__(TPB) void k_append(const int32_t *__restrict__ pass_list, int64_t q0, int64_t n,
                                                int64_t row0, const int32_t *__restrict__ p_cell,
                                                const uint32_t *__restrict__ p_patf, const float *__restrict__ p_r2f,
                                                const float *__restrict__ tokens, int g2, int D, int gs, int nh,
                                                float *__restrict__ cache_f, int32_t *__restrict__ cache_pos,
                                                float *__restrict__ cache_d)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (w >= n) return;
    const int32_t j = pass_list[q0 + w];
    const uint32_t cc = p_patf[j];
    const float4 *src = (const float4 *)(tokens + ((int64_t)(cc >> 16) * g2 + (cc & 0xffffu)) * D);
    float4 *dst = (float4 *)(cache_f + (row0 + w) * D);
    for (int v = lane; v < (D >> 2); v += 64) dst[v] = src[v];
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
    geom_point(gc, i, depth[i], T, o, true);
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

bsc_status ingest_batch(bsc_ctx *x, int32_t n_frames, const float *depth, const uint8_t *rgb, int32_t rgb_ch,
                        const float *tokens, const int32_t *idx, const int64_t *offsets_host, const double *alpha,
                        bsc_draw_fn draw, void *user)
{
    const int64_t N = (int64_t)x->c.height * x->c.width;
    const int64_t P = idx ? offsets_host[n_frames] : (int64_t)n_frames * N;
    if (P > x->c.max_points) {
        bsc_set_error("bsc_ingest: %lld points exceed max_points=%d", (long long)P, x->c.max_points);
        return BSC_E_CAPACITY;
    }
    if (P == 0) return BSC_OK;
    const dim3 block(TPB), grid((unsigned)((P + TPB - 1) / TPB));
    hipStream_t s = x->stream;
    // scratch set of this call; the rgb chain of the call before last may still be reading it on the side stream
    const int set = x->cur_set;
    x->cur_set ^= 1;
    if (x->ev_done_valid[set]) BSC_HIP(hipStreamWaitEvent(s, x->ev_done[set], 0));
    PointRec *p_rec = x->p_rec_s[set];
    uint32_t *skey_b = x->skey_b_s[set], *sval_b = x->sval_b_s[set];
    if (idx)
        BSC_HIP(hipMemcpyAsync(x->d_offsets, offsets_host, sizeof(int64_t) * (n_frames + 1), hipMemcpyHostToDevice, s));
    GeomConst gc = make_geom_const(x);
    hipLaunchKernelGGL(k_points, grid, block, 0, s, gc, depth, rgb, rgb_ch, idx, x->d_offsets, n_frames, x->d_transforms,
                       alpha, P, x->occ, x->p_cell, x->p_patf, p_rec, x->c.mode == BSC_MODE_EXACT ? x->p_r2f : (float *)nullptr);
    const int64_t nblk = (P + FB - 1) / FB;
    const dim3 fgrid((unsigned)nblk);
    hipLaunchKernelGGL(k_flags, fgrid, block, 0, s, P, x->p_cell, x->occ, x->p_scan_in);
    BSC_TRY(prim_exclusive_sum_i64(x, x->p_scan_in, x->p_scan_out, (size_t)nblk));
    // k_assign reads occ while other blocks overwrite claimed cells with ids: a claim INT_MIN + j can only be
    // replaced by the id of that same point j, so the flags of every other point are unaffected
    hipLaunchKernelGGL(k_assign, fgrid, block, 0, s, P, x->p_cell, x->occ, x->p_scan_out, x->dscal, x->c.voxel_capacity,
                       x->c.grid_size, x->nh, x->rgb_pos, x->c.mode == BSC_MODE_EXACT ? x->pass_list : (int32_t *)nullptr);
    hipLaunchKernelGGL(k_totals, dim3(1), dim3(64), 0, s, P, nblk, x->p_scan_in, x->p_scan_out, x->dscal,
                       x->c.voxel_capacity, x->bscal_s[set]);
    BSC_TRY(launch_keys_pairs(x, P, n_frames, idx == nullptr));
    // one small readback per call: voxel count (sort width), pair count (dense modes), passing points (exact mode),
    // capacity flag.  Everything enqueued so far is the call's front end; the back end is sized from these numbers.
    BSC_TRY(read_scalars(x));
    if (x->hscal[DS_ERROR]) {
        bsc_set_error("voxel capacity %d exceeded", x->c.voxel_capacity);
        return BSC_E_CAPACITY;
    }
    // ids in use are < max_id; invalid points carry 0xffffffff, which must still sort last under the bit mask
    const int vid_bits = ceil_log2_u64((uint64_t)x->hscal[DS_MAX_ID] + 2);
    // stable radix sort on the voxel id alone: points enter in order j, so each voxel's run stays in order
    BSC_TRY(prim_sort_pairs_u32(x, x->skey_a, skey_b, x->sval_a, sval_b, (size_t)P, 0, vid_bits));
    BSC_TRY(compact_heads_u32(x, skey_b, P, x->seg_start_s[set], x->bscal_s[set]));
    // rgb chain + top-down map on the side stream: sequential-latency bound, so it overlaps the
    // HBM-bound dense reduce of this call and whatever the caller enqueues next (the next batch's encoder)
    BSC_HIP(hipEventRecord(x->ev_ready[set], s));
    BSC_HIP(hipStreamWaitEvent(x->side, x->ev_ready[set], 0));
    const dim3 wgrid(256 * 8);
    hipLaunchKernelGGL(k_chain, wgrid, block, 0, x->side, P, skey_b, sval_b, x->bscal_s[set], x->seg_start_s[set], p_rec,
                       x->rgb_pos, x->rgb, x->weight, x->hmap, x->seg_last_s[set], x->c.grid_size, x->order_base);
    hipLaunchKernelGGL(k_hwin, dim3(256), block, 0, x->side, x->bscal_s[set], skey_b, x->seg_start_s[set],
                       x->seg_last_s[set], x->rgb_pos, x->hmap, p_rec, x->cv_map, x->c.grid_size, x->order_base);
    BSC_HIP(hipEventRecord(x->ev_done[set], x->side));
    x->ev_done_valid[set] = true;
    if (x->c.mode != BSC_MODE_EXACT) BSC_TRY(dense_reduce_batch(x, tokens, n_frames));
    BSC_HIP(hipGetLastError());
    x->order_base += P;
    if (x->c.mode == BSC_MODE_EXACT) {
        // memory_2.py:880-886: rows fill the cache in order; the point that finds it full triggers the
        // flush and loses its own token.
        int64_t remaining = x->hscal[DS_B_NPASS], q = 0;
        while (remaining > 0) {
            const int64_t room = x->c.iter_size - x->iter_id;
            const int64_t n = remaining < room ? remaining : room;
            if (n > 0) {
                hipLaunchKernelGGL(k_append, dim3((unsigned)((n * 64 + TPB - 1) / TPB)), block, 0, s, x->pass_list, q, n,
                                   x->iter_id, x->p_cell, x->p_patf, x->p_r2f, tokens, x->g2, x->c.token_dim,
                                   x->c.grid_size, x->nh, x->cache_f, x->cache_pos, x->cache_d);
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
