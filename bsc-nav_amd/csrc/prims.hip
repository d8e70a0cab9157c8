// prims.hip — device-wide radix sort and scans (rocPRIM) used by the ingest / flush / localize pipelines.
// These are the only library primitives in libbscnav; every other kernel is hand-written for gfx950.
#include "bsc_internal.h"

#include <rocprim/rocprim.hpp>

namespace {
// small inputs (the pair list, ~1e5) would otherwise take rocPRIM's merge-sort path (~0.3 ms); onesweep is faster
using onesweep_always = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 2048>;
// sorted run key -> length | (first run of its voxel) << 32; runs without a voxel (id field all ones) count nothing
struct run_scan_in {
    const uint32_t *k;
    uint32_t vmask;
    int vb;
    __host__ __device__ int64_t operator()(int64_t i) const
    {
        const uint32_t key = k[i], v = key & vmask;
        if (v == vmask) return 0;
        const int64_t head = (i == 0 || (k[i - 1] & vmask) != v) ? 1 : 0;
        return (int64_t)(key >> vb) + 1 + (head << 32);
    }
};
struct max_i32 {
    __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; }
};
}  // namespace

size_t prim_workspace_bytes(size_t n)
{
    size_t best = 0, b = 0;
    rocprim::radix_sort_keys(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, n, 0, 64, (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::radix_sort_pairs(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::radix_sort_pairs<onesweep_always>(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, (const uint32_t *)nullptr,
                                               (uint32_t *)nullptr, n, 0, 64, (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::radix_sort_pairs(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, n, 0, 32, (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::radix_sort_pairs<onesweep_always>(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                               (uint32_t *)nullptr, n, 0, 32, (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::exclusive_scan(nullptr, b, (const int64_t *)nullptr, (int64_t *)nullptr, (int64_t)0, n,
                            rocprim::plus<int64_t>(), (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::exclusive_scan(nullptr, b, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, n,
                            rocprim::plus<int32_t>(), (hipStream_t)0);
    best = b > best ? b : best;
    rocprim::inclusive_scan(nullptr, b, (const int32_t *)nullptr, (int32_t *)nullptr, n, max_i32(), (hipStream_t)0);
    best = b > best ? b : best;
    {
        auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), run_scan_in{nullptr, 0u, 0});
        rocprim::exclusive_scan(nullptr, b, it, (int64_t *)nullptr, (int64_t)0, n, rocprim::plus<int64_t>(), (hipStream_t)0);
        best = b > best ? b : best;
    }
    return best + 256;
}

#define PRIM_CALL(call)                                             \
    do {                                                            \
        size_t bytes = x->prim_tmp_bytes;                           \
        hipError_t e = (call);                                      \
        if (e != hipSuccess) {                                      \
            bsc_set_error("rocprim: %s", hipGetErrorString(e));     \
            return BSC_E_HIP;                                       \
        }                                                           \
    } while (0)

bsc_status prim_sort_keys(bsc_ctx *x, const u64 *in, u64 *out, size_t n, int b0, int b1)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::radix_sort_keys(x->prim_tmp, bytes, in, out, n, b0, b1, x->stream));
    return BSC_OK;
}

bsc_status prim_sort_pairs(bsc_ctx *x, const u64 *kin, u64 *kout, const uint32_t *vin, uint32_t *vout, size_t n, int b0,
                           int b1)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::radix_sort_pairs(x->prim_tmp, bytes, kin, kout, vin, vout, n, b0, b1, x->stream));
    return BSC_OK;
}

bsc_status prim_sort_pairs_onesweep(bsc_ctx *x, const u64 *kin, u64 *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                                    int b0, int b1)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::radix_sort_pairs<onesweep_always>(x->prim_tmp, bytes, kin, kout, vin, vout, n, b0, b1, x->stream));
    return BSC_OK;
}

bsc_status prim_sort_pairs_u32(bsc_ctx *x, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                               size_t n, int b0, int b1)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::radix_sort_pairs(x->prim_tmp, bytes, kin, kout, vin, vout, n, b0, b1, x->stream));
    return BSC_OK;
}

bsc_status prim_sort_pairs_u32_onesweep(bsc_ctx *x, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                                        size_t n, int b0, int b1)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::radix_sort_pairs<onesweep_always>(x->prim_tmp, bytes, kin, kout, vin, vout, n, b0, b1, x->stream));
    return BSC_OK;
}

bsc_status prim_exclusive_sum_i64(bsc_ctx *x, const int64_t *in, int64_t *out, size_t n)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::exclusive_scan(x->prim_tmp, bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), x->stream));
    return BSC_OK;
}

bsc_status prim_exclusive_sum_i32(bsc_ctx *x, const int32_t *in, int32_t *out, size_t n)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::exclusive_scan(x->prim_tmp, bytes, in, out, (int32_t)0, n, rocprim::plus<int32_t>(), x->stream));
    return BSC_OK;
}

bsc_status prim_inclusive_max_i32(bsc_ctx *x, const int32_t *in, int32_t *out, size_t n)
{
    if (n == 0) return BSC_OK;
    PRIM_CALL(rocprim::inclusive_scan(x->prim_tmp, bytes, in, out, n, max_i32(), x->stream));
    return BSC_OK;
}

bsc_status prim_scan_runs(bsc_ctx *x, const uint32_t *keys_sorted, int vb, int64_t *out, size_t n)
{
    if (n == 0) return BSC_OK;
    const uint32_t vmask = vb >= 32 ? 0xffffffffu : ((1u << vb) - 1u);
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), run_scan_in{keys_sorted, vmask, vb});
    PRIM_CALL(rocprim::exclusive_scan(x->prim_tmp, bytes, it, out, (int64_t)0, n, rocprim::plus<int64_t>(), x->stream));
    return BSC_OK;
}
