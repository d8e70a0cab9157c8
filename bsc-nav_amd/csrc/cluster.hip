// cluster.hip — the step right after localize: GESObjectNavRobot.weighted_cluster_centers (BSCAgent.py:479-497).
//
// DBSCAN(eps, min_samples) over the K top-ranked voxel positions, similarity-weighted cluster centres, clusters
// ordered by mean similarity.  K is ~100, so one workgroup does it out of LDS and the "query -> goal position" chain
// never leaves the GPU.  scikit-learn grows clusters depth-first from unlabelled core points in index order; that
// order-defined result has a closed form which is what the kernel evaluates in parallel:
//   core(i)    = |{ j : d(i,j) <= eps }| >= min_samples            (the point counts itself)
//   component  = connected components of the core points under d <= eps (min-label propagation)
//   label      = rank of the component by its smallest core index  (= order in which sklearn seeds clusters)
//   border     = non-core point with a core neighbour: the smallest label among them (the cluster expanded first
//                claims it); no core neighbour -> noise (-1)
#include "bsc_internal.h"

#include <math.h>

#define CL_TPB 256
#define CL_MAXK 1024

__global__ __launch_bounds__(CL_TPB) void k_cluster_centers(const int32_t *__restrict__ pos, const float *__restrict__ sim,
                                                            int K, double eps2, int min_samples, double *__restrict__ centers,
                                                            int32_t *__restrict__ labels_out, int32_t *__restrict__ sizes,
                                                            int32_t *__restrict__ n_clusters)
{
    __shared__ int32_t px[CL_MAXK], py[CL_MAXK], pz[CL_MAXK];
    __shared__ int32_t comp[CL_MAXK], lab[CL_MAXK];
    __shared__ uint8_t core[CL_MAXK];
    __shared__ double avg[CL_MAXK];
    __shared__ int changed, n_lab;
    const int tid = threadIdx.x;
    for (int i = tid; i < K; i += CL_TPB) { px[i] = pos[3 * i]; py[i] = pos[3 * i + 1]; pz[i] = pos[3 * i + 2]; }
    __syncthreads();
    auto near = [&](int i, int j) {
        const double dx = px[i] - px[j], dy = py[i] - py[j], dz = pz[i] - pz[j];
        return dx * dx + dy * dy + dz * dz <= eps2;
    };
    for (int i = tid; i < K; i += CL_TPB) {
        int cnt = 0;
        for (int j = 0; j < K; ++j) cnt += near(i, j) ? 1 : 0;
        core[i] = cnt >= min_samples;
        comp[i] = core[i] ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int it = 0; it < K; ++it) {            // min-label propagation over the core graph
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int i = tid; i < K; i += CL_TPB) {
            if (!core[i]) continue;
            int m = comp[i];
            for (int j = 0; j < K; ++j)
                if (core[j] && near(i, j)) m = min(m, comp[j]);
            if (m < comp[i]) { comp[i] = m; changed = 1; }
        }
        __syncthreads();
        const int c = changed;
        __syncthreads();
        if (!c) break;
    }
    // component roots (comp[i] == i) ranked by index = sklearn's cluster numbering
    for (int i = tid; i < K; i += CL_TPB) {
        int r = -1;
        if (core[i] && comp[i] == i) {
            r = 0;
            for (int j = 0; j < i; ++j) r += (core[j] && comp[j] == j) ? 1 : 0;
        }
        lab[i] = r;                              // label of a root, -1 otherwise (filled in below)
    }
    if (tid == 0) n_lab = 0;
    __syncthreads();
    for (int i = tid; i < K; i += CL_TPB)
        if (core[i] && comp[i] == i) atomicAdd(&n_lab, 1);
    __syncthreads();
    for (int i = tid; i < K; i += CL_TPB) {
        int l = -1;
        if (core[i]) {
            l = lab[comp[i]];                    // roots keep theirs; other cores read their root's label
        } else {
            for (int j = 0; j < K; ++j)
                if (core[j] && near(i, j)) {
                    const int lj = lab[comp[j]];
                    l = (l < 0 || lj < l) ? lj : l;
                }
        }
        labels_out[i] = l;
    }
    __syncthreads();                             // labels_out is re-read below (same workgroup, global memory)
    __threadfence_block();
    const int nl = n_lab;
    // BSCAgent.py:484-491 — per cluster: np.average(points, weights=sim), np.mean(sim), size (index order sums)
    for (int l = tid; l < nl; l += CL_TPB) {
        double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
        int n = 0;
        for (int i = 0; i < K; ++i)
            if (labels_out[i] == l) {
                const double w = (double)sim[i];
                sw += w; sx += px[i] * w; sy += py[i] * w; sz += pz[i] * w;
                ++n;
            }
        avg[l] = sw / n;
        comp[l] = n;                              // comp is free now: cluster sizes
        // stash centres in the output at the UNSORTED slot nl + l .. (output holds 2*K rows of scratch)
        centers[3 * (K + l)] = sx / sw; centers[3 * (K + l) + 1] = sy / sw; centers[3 * (K + l) + 2] = sz / sw;
    }
    __syncthreads();
    __threadfence_block();
    for (int l = tid; l < nl; l += CL_TPB) {      // :493 stable descending sort by mean similarity
        int rank = 0;
        for (int m = 0; m < nl; ++m) rank += (avg[m] > avg[l] || (avg[m] == avg[l] && m < l)) ? 1 : 0;
        centers[3 * rank] = centers[3 * (K + l)];
        centers[3 * rank + 1] = centers[3 * (K + l) + 1];
        centers[3 * rank + 2] = centers[3 * (K + l) + 2];
        sizes[rank] = comp[l];
    }
    if (tid == 0) *n_clusters = nl;
}

extern "C" bsc_status bsc_cluster_centers(bsc_ctx *x, int32_t query_index, int32_t K, const int32_t *pos_host,
                                          const float *sim_host, double eps, int32_t min_samples, double *centers_host,
                                          int32_t *labels_host, int32_t *sizes_host, int32_t *n_clusters_host)
{
    if (!x || !centers_host || !labels_host || !sizes_host || !n_clusters_host || K < 1 || K > CL_MAXK || min_samples < 1) {
        bsc_set_error("bsc_cluster_centers: invalid argument (1 <= K <= %d)", CL_MAXK);
        return BSC_E_INVALID;
    }
    BSC_HIP(hipSetDevice(x->device));
    hipStream_t s = x->stream;
    const int32_t *d_pos;
    const float *d_sim;
    int32_t *tmp_pos = nullptr;
    float *tmp_sim = nullptr;
    if (pos_host) {
        if (!sim_host) { bsc_set_error("bsc_cluster_centers: sim_host missing"); return BSC_E_INVALID; }
        BSC_HIP(hipMalloc((void **)&tmp_pos, sizeof(int32_t) * 3 * K));
        BSC_HIP(hipMalloc((void **)&tmp_sim, sizeof(float) * K));
        BSC_HIP(hipMemcpyAsync(tmp_pos, pos_host, sizeof(int32_t) * 3 * K, hipMemcpyHostToDevice, s));
        BSC_HIP(hipMemcpyAsync(tmp_sim, sim_host, sizeof(float) * K, hipMemcpyHostToDevice, s));
        d_pos = tmp_pos; d_sim = tmp_sim;
    } else {                                    // top-K of the last bsc_localize call, still resident in HBM
        if (query_index < 0 || query_index >= x->last_nq || K > x->last_counts[query_index]) {
            bsc_set_error("bsc_cluster_centers: query %d with %d results is not in the last bsc_localize call", query_index, K);
            return BSC_E_INVALID;
        }
        d_pos = x->l_out_pos + (int64_t)query_index * x->last_K * 3;
        d_sim = x->l_out_sim + (int64_t)query_index * x->last_K;
    }
    double *d_centers; int32_t *d_lab;
    BSC_HIP(hipMalloc((void **)&d_centers, sizeof(double) * 3 * 2 * K));
    BSC_HIP(hipMalloc((void **)&d_lab, sizeof(int32_t) * (2 * K + 1)));
    hipLaunchKernelGGL(k_cluster_centers, dim3(1), dim3(CL_TPB), 0, s, d_pos, d_sim, K, eps * eps, min_samples, d_centers,
                       d_lab, d_lab + K, d_lab + 2 * K);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(centers_host, d_centers, sizeof(double) * 3 * K, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(labels_host, d_lab, sizeof(int32_t) * K, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(sizes_host, d_lab + K, sizeof(int32_t) * K, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(n_clusters_host, d_lab + 2 * K, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_centers); (void)hipFree(d_lab);
    if (tmp_pos) (void)hipFree(tmp_pos);
    if (tmp_sim) (void)hipFree(tmp_sim);
    BSC_HIP(e);
    return BSC_OK;
}
