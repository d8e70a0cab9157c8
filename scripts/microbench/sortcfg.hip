// rocPRIM onesweep configurations for the run sort of the ingest: 40M (u32 key, u32 value) pairs, 15 key bits.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/sortcfg.hip -o scripts/microbench/sortcfg.bin
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdint.h>
#include <vector>
template <class Cfg>
static void run(const char *name, uint32_t *ka, uint32_t *kb, uint32_t *va, uint32_t *vb, size_t n, int bits)
{
    size_t bytes = 0;
    rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, ka, kb, va, vb, n, 0, bits, (hipStream_t)0);
    void *tmp; hipMalloc(&tmp, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) rocprim::radix_sort_pairs<Cfg>(tmp, bytes, ka, kb, va, vb, n, 0, bits, (hipStream_t)0);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) rocprim::radix_sort_pairs<Cfg>(tmp, bytes, ka, kb, va, vb, n, 0, bits, (hipStream_t)0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s n=%zu bits=%d  %.3f ms\n", name, n, bits, ms / 5);
    hipFree(tmp);
}
template <unsigned BS, unsigned IPT, unsigned RB, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::default_algorithm>
using os = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, RB, ALG>, 2048>;
int main()
{
    for (size_t n : {(size_t)40000000, (size_t)2700000}) {
        const int bits = n > 10000000 ? 15 : 24;
        std::vector<uint32_t> h(n);
        uint32_t s = 12345;
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) & ((1u << bits) - 1); }
        uint32_t *ka, *kb, *va, *vb;
        hipMalloc(&ka, n * 4); hipMalloc(&kb, n * 4); hipMalloc(&va, n * 4); hipMalloc(&vb, n * 4);
        hipMemcpy(ka, h.data(), n * 4, hipMemcpyHostToDevice);
        hipMemset(va, 0, n * 4);
        run<rocprim::default_config>("default", ka, kb, va, vb, n, bits);
        run<os<256, 8, 8>>("onesweep 256x8 r8", ka, kb, va, vb, n, bits);
        run<os<256, 12, 8>>("onesweep 256x12 r8", ka, kb, va, vb, n, bits);
        run<os<256, 16, 8>>("onesweep 256x16 r8", ka, kb, va, vb, n, bits);
        run<os<256, 12, 8, rocprim::block_radix_rank_algorithm::match>>("onesweep 256x12 r8 match", ka, kb, va, vb, n, bits);
        run<os<512, 12, 8, rocprim::block_radix_rank_algorithm::match>>("onesweep 512x12 r8 match", ka, kb, va, vb, n, bits);
        run<os<512, 8, 8, rocprim::block_radix_rank_algorithm::match>>("onesweep 512x8 r8 match", ka, kb, va, vb, n, bits);
        run<os<256, 16, 6>>("onesweep 256x16 r6", ka, kb, va, vb, n, bits);
        hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb);
    }
    return 0;
}
