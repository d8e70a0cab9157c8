// prims.hip — device-wide radix sort and scans (rocPRIM) used by the ingest / flush / localize pipelines.
// These are the only library primitives in libbscnav; every other kernel is hand-written for gfx950.
#include "bsc_internal.h"

#include <rocprim/rocprim.hpp>

namespace {
// small inputs (the pair list, ~1e5) would otherwise take rocPRIM's merge-sort path (~0.3 ms); onesweep is faster
using onesweep_always = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                   rocprim::default_config, 2048>;
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

