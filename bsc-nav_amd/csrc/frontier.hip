// frontier.hip — FrontierExplorer helpers of VoxelTokenMemory (memory_2.py:1147-1311) on the resident top-down map.
//
// The reference walks the (gs,gs) colour map with Python loops: known/unknown cells, frontier cells (known + navigable
// + an unknown 4-neighbour), 4-connected frontier clusters by BFS, cluster centres, information gain (unknown cells
// around the centre) and the best cluster.  cv_map already lives in HBM (bsc_ingest maintains it), so the whole step
// runs there:
//
//   k_fr_mask        1 thread / cell: bit0 known, bit1 frontier
//   k_cc_init/union/flatten   connected components with an atomicMin union-find over the right / down edges; a
//                    component's root is its smallest cell index = the cell the reference's BFS starts from, so
//                    clusters come out in the reference's order without replaying the BFS
//   k_cc_stats       integer size / coordinate sums per root (exact, order-free)
//   k_fr_compact     ordered compaction of the roots that reach min_cluster_size (one workgroup, ballot ranks)
//   k_fr_eval        1 wavefront / cluster: centre = sums / size in f64, Python round() = round-half-even = rint(),
//                    unknown cells of the clipped (2r+1)^2 window
//   k_fr_best        first cluster with the strictly largest gain > 0
#include "bsc_internal.h"

#include <math.h>

#define TPB 256

__device__ __forceinline__ bool fr_unknown(const uint8_t *__restrict__ cv, int64_t i)
{
    return (int)cv[3 * i] + (int)cv[3 * i + 1] + (int)cv[3 * i + 2] == 0;     // memory_2.py:1165
}

__global__ __launch_bounds__(TPB) void k_fr_mask(const uint8_t *__restrict__ cv, const uint8_t *__restrict__ nav, int gs,
                                                 uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (int64_t)gs * gs) return;
    const int x = (int)(i / gs), y = (int)(i % gs);
    const bool known = !fr_unknown(cv, i);
    uint8_t m = known ? 1 : 0;
    if (known && (!nav || nav[i])) {                    // :1185-1207
        bool u = false;
        if (x + 1 < gs) u = u || fr_unknown(cv, i + gs);
        if (x - 1 >= 0) u = u || fr_unknown(cv, i - gs);
        if (y + 1 < gs) u = u || fr_unknown(cv, i + 1);
        if (y - 1 >= 0) u = u || fr_unknown(cv, i - 1);
        if (u) m |= 2;
    }
    mask[i] = m;
}

__global__ __launch_bounds__(TPB) void k_cc_init(const uint8_t *__restrict__ fr, int bit, int64_t n, int32_t *parent,
                                                 int32_t *size, unsigned long long *sumx, unsigned long long *sumy,
                                                 int32_t *ord)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    parent[i] = (fr[i] & bit) ? (int32_t)i : -1;
    size[i] = 0; sumx[i] = 0ull; sumy[i] = 0ull; ord[i] = -1;
}

__device__ __forceinline__ int32_t cc_find(const int32_t *parent, int32_t a)
{
    for (;;) {
        const int32_t p = ((volatile const int32_t *)parent)[a];
        if (p == a) return a;
        a = p;
    }
}

// links are always larger root -> smaller root, so parents only decrease and the final root is the component minimum
__device__ __forceinline__ void cc_union(int32_t *parent, int32_t a, int32_t b)
{
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) { const int32_t t = a; a = b; b = t; }
        const int32_t old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(TPB) void k_cc_union(int gs, int32_t *parent)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (int64_t)gs * gs || parent[i] < 0) return;
    const int x = (int)(i / gs), y = (int)(i % gs);
    if (x + 1 < gs && parent[i + gs] >= 0) cc_union(parent, (int32_t)i, (int32_t)(i + gs));
    if (y + 1 < gs && parent[i + 1] >= 0) cc_union(parent, (int32_t)i, (int32_t)(i + 1));
}

__global__ __launch_bounds__(TPB) void k_cc_stats(int gs, int32_t *parent, int32_t *size, unsigned long long *sumx,
                                                  unsigned long long *sumy)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= (int64_t)gs * gs || parent[i] < 0) return;
    const int32_t r = cc_find(parent, (int32_t)i);
    parent[i] = r;                                       // flatten (a root keeps itself)
    atomicAdd(&size[r], 1);
    atomicAdd(&sumx[r], (unsigned long long)(i / gs));
    atomicAdd(&sumy[r], (unsigned long long)(i % gs));
}

// roots with size >= min_size, ascending cell index (= order of the reference's cluster list, :1222-1247)
__global__ __launch_bounds__(1024) void k_fr_compact(int64_t n, const int32_t *__restrict__ parent,
                                                     const int32_t *__restrict__ size, int min_size, int max_out,
                                                     int32_t *__restrict__ roots, int32_t *__restrict__ ord, int32_t *count)
{
    __shared__ int wcnt[16];
    __shared__ int base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += 1024) {
        const int64_t i = i0 + threadIdx.x;
        const bool f = i < n && parent[i] == (int32_t)i && size[i] >= min_size;
        const u64 bal = __ballot(f);
        if (lane == 0) wcnt[wid] = __popcll(bal);
        __syncthreads();
        int before = base, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wid) before += wcnt[w];
            total += wcnt[w];
        }
        if (f) {
            const int k = before + __popcll(bal & ((1ull << lane) - 1ull));
            if (k < max_out) { roots[k] = (int32_t)i; ord[i] = k; }
        }
        __syncthreads();
        if (threadIdx.x == 0) base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base;
}

__global__ __launch_bounds__(TPB) void k_fr_eval(int n_clusters, const int32_t *__restrict__ roots,
                                                 const int32_t *__restrict__ size,
                                                 const unsigned long long *__restrict__ sumx,
                                                 const unsigned long long *__restrict__ sumy,
                                                 const uint8_t *__restrict__ cv, int gs, int radius,
                                                 int32_t *__restrict__ first, int32_t *__restrict__ sizes,
                                                 double *__restrict__ centers, double *__restrict__ gains)
{
    const int lane = threadIdx.x & 63;
    const int k = (int)(((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6);
    if (k >= n_clusters) return;
    const int32_t r = roots[k];
    const double cx = (double)sumx[r] / (double)size[r], cy = (double)sumy[r] / (double)size[r];   // :1256-1257
    const int rx = (int)rint(cx), ry = (int)rint(cy);   // :1265 int(round(c)): Python rounds halves to even
    const int w = 2 * radius + 1;
    int cnt = 0;
    for (int t = lane; t < w * w; t += 64) {
        const int nx = rx - radius + t / w, ny = ry - radius + t % w;
        if (nx >= 0 && nx < gs && ny >= 0 && ny < gs && fr_unknown(cv, (int64_t)nx * gs + ny)) ++cnt;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) {
        first[2 * k] = r / gs; first[2 * k + 1] = r % gs;
        sizes[k] = size[r];
        centers[2 * k] = cx; centers[2 * k + 1] = cy;
        gains[k] = (double)cnt;
    }
}

__global__ void k_fr_best(int n_clusters, const double *gains, int32_t *best)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double best_ig = 0.0;
    int b = -1;
    for (int k = 0; k < n_clusters; ++k)
        if (gains[k] > best_ig) { best_ig = gains[k]; b = k; }          // :1301 strict: the first maximum wins
    *best = b;
}

__global__ __launch_bounds__(TPB) void k_fr_labels(int64_t n, const int32_t *__restrict__ parent,
                                                   const int32_t *__restrict__ ord, int32_t *__restrict__ labels)
{
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int32_t p = parent[i];
    labels[i] = p >= 0 ? ord[p] : -1;
}

static bsc_status fr_alloc(bsc_ctx *x)
{
    if (x->fr_mask) return BSC_OK;
    const int64_t n = (int64_t)x->c.grid_size * x->c.grid_size;
#define FR_ALLOC(p, count) BSC_HIP(hipMalloc((void **)&(p), sizeof(*(p)) * (size_t)(count)))
    FR_ALLOC(x->fr_mask, n); FR_ALLOC(x->fr_in, n); FR_ALLOC(x->fr_parent, n); FR_ALLOC(x->fr_size, n);
    FR_ALLOC(x->fr_sumx, n); FR_ALLOC(x->fr_sumy, n); FR_ALLOC(x->fr_ord, n); FR_ALLOC(x->fr_roots, n);
    FR_ALLOC(x->fr_labels, n); FR_ALLOC(x->fr_first, 2 * n); FR_ALLOC(x->fr_sizes, n);
    FR_ALLOC(x->fr_centers, 2 * n); FR_ALLOC(x->fr_gains, n); FR_ALLOC(x->fr_scal, 2);
#undef FR_ALLOC
    BSC_HIP(hipMemsetAsync(x->fr_mask, 0, (size_t)n, x->stream));
    return BSC_OK;
}

bsc_status frontier_mask_impl(bsc_ctx *x, const uint8_t *navigable_host, uint8_t *mask_host)
{
    BSC_TRY(fr_alloc(x));
    const int gs = x->c.grid_size;
    const int64_t n = (int64_t)gs * gs;
    hipStream_t s = x->stream;
    if (navigable_host) BSC_HIP(hipMemcpyAsync(x->fr_in, navigable_host, (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_fr_mask, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, s, x->cv_map,
                       navigable_host ? x->fr_in : (const uint8_t *)nullptr, gs, x->fr_mask);
    BSC_HIP(hipGetLastError());
    if (mask_host) BSC_HIP(hipMemcpyAsync(mask_host, x->fr_mask, (size_t)n, hipMemcpyDeviceToHost, s));
    BSC_HIP(hipStreamSynchronize(s));
    return BSC_OK;
}

bsc_status frontier_clusters_impl(bsc_ctx *x, const uint8_t *frontier_host, int32_t min_cluster_size, int32_t ig_radius,
                                  int32_t max_clusters, int32_t *n_clusters_host, int32_t *labels_host,
                                  int32_t *first_host, int32_t *sizes_host, double *centers_host, double *gains_host,
                                  int32_t *best_host)
{
    BSC_TRY(fr_alloc(x));
    const int gs = x->c.grid_size;
    const int64_t n = (int64_t)gs * gs;
    hipStream_t s = x->stream;
    const dim3 grid((unsigned)((n + TPB - 1) / TPB)), block(TPB);
    const uint8_t *fr = x->fr_mask;
    int bit = 2;                                        // frontier cells of the last bsc_frontier_mask
    if (frontier_host) {
        BSC_HIP(hipMemcpyAsync(x->fr_in, frontier_host, (size_t)n, hipMemcpyHostToDevice, s));
        fr = x->fr_in;
        bit = 0xff;
    }
    const int cap = (int)(max_clusters < n ? max_clusters : n);
    hipLaunchKernelGGL(k_cc_init, grid, block, 0, s, fr, bit, n, x->fr_parent, x->fr_size, x->fr_sumx, x->fr_sumy, x->fr_ord);
    hipLaunchKernelGGL(k_cc_union, grid, block, 0, s, gs, x->fr_parent);
    hipLaunchKernelGGL(k_cc_stats, grid, block, 0, s, gs, x->fr_parent, x->fr_size, x->fr_sumx, x->fr_sumy);
    hipLaunchKernelGGL(k_fr_compact, dim3(1), dim3(1024), 0, s, n, x->fr_parent, x->fr_size, min_cluster_size, cap,
                       x->fr_roots, x->fr_ord, x->fr_scal);
    BSC_HIP(hipGetLastError());
    int32_t total = 0;
    BSC_HIP(hipMemcpyAsync(&total, x->fr_scal, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BSC_HIP(hipStreamSynchronize(s));
    const int kept = total < cap ? total : cap;
    if (kept > 0)
        hipLaunchKernelGGL(k_fr_eval, dim3((unsigned)(((int64_t)kept * 64 + TPB - 1) / TPB)), block, 0, s, kept, x->fr_roots,
                           x->fr_size, x->fr_sumx, x->fr_sumy, x->cv_map, gs, ig_radius, x->fr_first, x->fr_sizes,
                           x->fr_centers, x->fr_gains);
    hipLaunchKernelGGL(k_fr_best, dim3(1), dim3(64), 0, s, kept, x->fr_gains, x->fr_scal + 1);
    if (labels_host) {
        hipLaunchKernelGGL(k_fr_labels, grid, block, 0, s, n, x->fr_parent, x->fr_ord, x->fr_labels);
        BSC_HIP(hipMemcpyAsync(labels_host, x->fr_labels, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s));
    }
    BSC_HIP(hipGetLastError());
    if (kept > 0) {
        if (first_host) BSC_HIP(hipMemcpyAsync(first_host, x->fr_first, sizeof(int32_t) * 2 * kept, hipMemcpyDeviceToHost, s));
        if (sizes_host) BSC_HIP(hipMemcpyAsync(sizes_host, x->fr_sizes, sizeof(int32_t) * kept, hipMemcpyDeviceToHost, s));
        if (centers_host) BSC_HIP(hipMemcpyAsync(centers_host, x->fr_centers, sizeof(double) * 2 * kept, hipMemcpyDeviceToHost, s));
        if (gains_host) BSC_HIP(hipMemcpyAsync(gains_host, x->fr_gains, sizeof(double) * kept, hipMemcpyDeviceToHost, s));
    }
    if (best_host) BSC_HIP(hipMemcpyAsync(best_host, x->fr_scal + 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BSC_HIP(hipStreamSynchronize(s));
    *n_clusters_host = total;
    return BSC_OK;
}
