// encoder_ops.hip — fused element-wise pieces of the ViT patch-feature provider (memory_2.py:732-742).
// The GEMMs and attention of the encoder run on MFMA through the ROCm libraries; what is left between them
// (residual add, LayerNorm) is pure HBM traffic, fused here into one pass per residual update:
//     s = x + delta (bf16) ; y = LayerNorm(s) * gamma + beta            one wavefront per token row
// Stateless entry point, launched on the caller's stream (captured into the encoder's HIP graph).
#include "bsc_internal.h"

#define TPB 256

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                               // round to nearest even
    return (uint16_t)(u >> 16);
}

// NG = groups of 4 elements per lane (width = 256 * NG, e.g. 768 -> 3, 1024 -> 4)
template <int NG, bool HAS_DELTA>
__global__ __launch_bounds__(TPB) void k_add_layernorm(const ushort4 *__restrict__ x, const ushort4 *__restrict__ delta,
                                                       const ushort4 *__restrict__ gamma, const ushort4 *__restrict__ beta,
                                                       ushort4 *__restrict__ xout, ushort4 *__restrict__ y, int64_t rows,
                                                       int width, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (row >= rows) return;
    const int w4 = width >> 2;
    const ushort4 *xr = x + row * w4;
    float v[NG][4];
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        ushort4 a = xr[c];
        float f0 = bf2f(a.x), f1 = bf2f(a.y), f2 = bf2f(a.z), f3 = bf2f(a.w);
        if (HAS_DELTA) {
            const ushort4 d = delta[row * w4 + c];
            // the residual stream is stored in bf16: round the sum first, normalise the rounded values
            a.x = f2bf(f0 + bf2f(d.x)); a.y = f2bf(f1 + bf2f(d.y)); a.z = f2bf(f2 + bf2f(d.z)); a.w = f2bf(f3 + bf2f(d.w));
            xout[row * w4 + c] = a;
            f0 = bf2f(a.x); f1 = bf2f(a.y); f2 = bf2f(a.z); f3 = bf2f(a.w);
        }
        v[g][0] = f0; v[g][1] = f1; v[g][2] = f2; v[g][3] = f3;
        sum += (f0 + f1) + (f2 + f3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)width;
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = v[g][k] - mean;
            var += d * d;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = rsqrtf(var / (float)width + eps);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        const ushort4 ga = gamma[c], be = beta[c];
        ushort4 o;
        o.x = f2bf((v[g][0] - mean) * rstd * bf2f(ga.x) + bf2f(be.x));
        o.y = f2bf((v[g][1] - mean) * rstd * bf2f(ga.y) + bf2f(be.y));
        o.z = f2bf((v[g][2] - mean) * rstd * bf2f(ga.z) + bf2f(be.z));
        o.w = f2bf((v[g][3] - mean) * rstd * bf2f(ga.w) + bf2f(be.w));
        y[row * w4 + c] = o;
    }
}

extern "C" bsc_status bsc_enc_add_layernorm(const void *x, const void *delta, const void *gamma, const void *beta,
                                            void *xout, void *y, int64_t rows, int32_t width, float eps, void *hip_stream)
{
    if (!x || !gamma || !beta || !y || rows < 1 || (delta && !xout)) { bsc_set_error("bsc_enc_add_layernorm: null argument"); return BSC_E_INVALID; }
    if (width % 256 != 0 || width < 256 || width > 2048) {
        bsc_set_error("bsc_enc_add_layernorm: width %d (need a multiple of 256 up to 2048)", width);
        return BSC_E_INVALID;
    }
    const dim3 grid((unsigned)((rows * 64 + TPB - 1) / TPB)), block(TPB);
    hipStream_t s = (hipStream_t)hip_stream;
    const int ng = width / 256;
#define LN(NG)                                                                                                       \
    do {                                                                                                             \
        if (delta)                                                                                                   \
            hipLaunchKernelGGL((k_add_layernorm<NG, true>), grid, block, 0, s, (const ushort4 *)x, (const ushort4 *)delta, \
                               (const ushort4 *)gamma, (const ushort4 *)beta, (ushort4 *)xout, (ushort4 *)y, rows, width, eps); \
        else                                                                                                         \
            hipLaunchKernelGGL((k_add_layernorm<NG, false>), grid, block, 0, s, (const ushort4 *)x, (const ushort4 *)nullptr, \
                               (const ushort4 *)gamma, (const ushort4 *)beta, (ushort4 *)nullptr, (ushort4 *)y, rows, width, eps); \
    } while (0)
    switch (ng) {
        case 1: LN(1); break;
        case 2: LN(2); break;
        case 3: LN(3); break;
        case 4: LN(4); break;
        case 5: LN(5); break;
        case 6: LN(6); break;
        case 7: LN(7); break;
        default: LN(8); break;
    }
#undef LN
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// ---- the two ends of the transformer stack, fused the same way ---------------------------------------------------------
// wave-level LayerNorm statistics of a row held as NG x 4 values per lane
template <int NG>
__device__ __forceinline__ void ln_stats(const float (&v)[NG][4], int width, float eps, float &mean, float &rstd)
{
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) sum += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    mean = sum / (float)width;
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = v[g][k] - mean;
            var += d * d;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    rstd = rsqrtf(var / (float)width + eps);
}

// token assembly + first LayerNorm: row 0 = cls + pos[0]; rows 1..R = register tokens; the others = patch embedding +
// pos (bf16 sums, as `cat([cls, x]) + pos` and the register insertion produce them); writes the residual stream and
// LayerNorm(row)
template <int NG>
__global__ __launch_bounds__(TPB) void k_embed_layernorm(const ushort4 *__restrict__ patch, const ushort4 *__restrict__ cls,
                                                         const ushort4 *__restrict__ reg, const ushort4 *__restrict__ pos,
                                                         const ushort4 *__restrict__ gamma, const ushort4 *__restrict__ beta,
                                                         ushort4 *__restrict__ xout, ushort4 *__restrict__ y, int64_t rows,
                                                         int T, int R, int width, float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (row >= rows) return;
    const int w4 = width >> 2;
    const int64_t b = row / T;
    const int r = (int)(row - b * T);
    const int np = T - 1 - R;                               // patch tokens per image
    float v[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        ushort4 a;
        if (r >= 1 && r <= R) {
            a = reg[(int64_t)(r - 1) * w4 + c];
        } else {
            const ushort4 s0 = r == 0 ? cls[c] : patch[(b * np + (r - 1 - R)) * w4 + c];
            const ushort4 pp = pos[(int64_t)(r == 0 ? 0 : r - R) * w4 + c];
            a.x = f2bf(bf2f(s0.x) + bf2f(pp.x)); a.y = f2bf(bf2f(s0.y) + bf2f(pp.y));
            a.z = f2bf(bf2f(s0.z) + bf2f(pp.z)); a.w = f2bf(bf2f(s0.w) + bf2f(pp.w));
        }
        xout[row * w4 + c] = a;
        v[g][0] = bf2f(a.x); v[g][1] = bf2f(a.y); v[g][2] = bf2f(a.z); v[g][3] = bf2f(a.w);
    }
    float mean, rstd;
    ln_stats<NG>(v, width, eps, mean, rstd);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        const ushort4 ga = gamma[c], be = beta[c];
        ushort4 o;
        o.x = f2bf((v[g][0] - mean) * rstd * bf2f(ga.x) + bf2f(be.x));
        o.y = f2bf((v[g][1] - mean) * rstd * bf2f(ga.y) + bf2f(be.y));
        o.z = f2bf((v[g][2] - mean) * rstd * bf2f(ga.z) + bf2f(be.z));
        o.w = f2bf((v[g][3] - mean) * rstd * bf2f(ga.w) + bf2f(be.w));
        y[row * w4 + c] = o;
    }
}

// last residual add + final LayerNorm, patch rows only, straight into the (B, g*g, width) token tensor the memory ingests:
// f32 (the bf16-rounded value widened, what `.float()` of the bf16 result gives) or bf16
template <int NG, bool F32OUT>
__global__ __launch_bounds__(TPB) void k_final_layernorm(const ushort4 *__restrict__ x, const ushort4 *__restrict__ delta,
                                                         const ushort4 *__restrict__ gamma, const ushort4 *__restrict__ beta,
                                                         void *__restrict__ out, int64_t rows_out, int T, int skip, int width,
                                                         float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t orow = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    if (orow >= rows_out) return;
    const int w4 = width >> 2;
    const int np = T - skip;
    const int64_t b = orow / np;
    const int64_t row = b * T + skip + (orow - b * np);
    float v[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        const ushort4 a = x[row * w4 + c], d = delta[row * w4 + c];
        v[g][0] = bf2f(f2bf(bf2f(a.x) + bf2f(d.x))); v[g][1] = bf2f(f2bf(bf2f(a.y) + bf2f(d.y)));
        v[g][2] = bf2f(f2bf(bf2f(a.z) + bf2f(d.z))); v[g][3] = bf2f(f2bf(bf2f(a.w) + bf2f(d.w)));
    }
    float mean, rstd;
    ln_stats<NG>(v, width, eps, mean, rstd);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int c = lane + 64 * g;
        const ushort4 ga = gamma[c], be = beta[c];
        ushort4 o;
        o.x = f2bf((v[g][0] - mean) * rstd * bf2f(ga.x) + bf2f(be.x));
        o.y = f2bf((v[g][1] - mean) * rstd * bf2f(ga.y) + bf2f(be.y));
        o.z = f2bf((v[g][2] - mean) * rstd * bf2f(ga.z) + bf2f(be.z));
        o.w = f2bf((v[g][3] - mean) * rstd * bf2f(ga.w) + bf2f(be.w));
        if (F32OUT) ((float4 *)out)[orow * w4 + c] = make_float4(bf2f(o.x), bf2f(o.y), bf2f(o.z), bf2f(o.w));
        else ((ushort4 *)out)[orow * w4 + c] = o;
    }
}

#define ENC_NG_SWITCH(ng, CALL)                                                                          \
    switch (ng) {                                                                                        \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;  \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; default: CALL(8); break; \
    }

extern "C" bsc_status bsc_enc_embed_layernorm(const void *patch_dev, const void *cls_dev, const void *reg_dev,
                                              const void *pos_dev, const void *gamma_dev, const void *beta_dev,
                                              void *xout_dev, void *y_dev, int32_t B, int32_t T, int32_t registers,
                                              int32_t width, float eps, void *hip_stream)
{
    if (!patch_dev || !cls_dev || !pos_dev || !gamma_dev || !beta_dev || !xout_dev || !y_dev || B < 1 || registers < 0 ||
        T < 2 + registers || (registers > 0 && !reg_dev) || width % 256 != 0 || width < 256 || width > 2048) {
        bsc_set_error("bsc_enc_embed_layernorm: invalid argument");
        return BSC_E_INVALID;
    }
    const int64_t rows = (int64_t)B * T;
    const dim3 grid((unsigned)((rows * 64 + TPB - 1) / TPB)), block(TPB);
#define EL(NG) hipLaunchKernelGGL((k_embed_layernorm<NG>), grid, block, 0, (hipStream_t)hip_stream, (const ushort4 *)patch_dev, \
                                  (const ushort4 *)cls_dev, (const ushort4 *)reg_dev, (const ushort4 *)pos_dev,              \
                                  (const ushort4 *)gamma_dev, (const ushort4 *)beta_dev, (ushort4 *)xout_dev,                \
                                  (ushort4 *)y_dev, rows, T, registers, width, eps)
    ENC_NG_SWITCH(width / 256, EL)
#undef EL
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_final_layernorm(const void *x_dev, const void *delta_dev, const void *gamma_dev,
                                              const void *beta_dev, void *out_dev, int32_t out_f32, int32_t B, int32_t T,
                                              int32_t skip, int32_t width, float eps, void *hip_stream)
{
    if (!x_dev || !delta_dev || !gamma_dev || !beta_dev || !out_dev || B < 1 || skip < 0 || T <= skip || width % 256 != 0 ||
        width < 256 || width > 2048) {
        bsc_set_error("bsc_enc_final_layernorm: invalid argument");
        return BSC_E_INVALID;
    }
    const int64_t rows = (int64_t)B * (T - skip);
    const dim3 grid((unsigned)((rows * 64 + TPB - 1) / TPB)), block(TPB);
#define FL(NG)                                                                                                                  \
    do {                                                                                                                        \
        if (out_f32)                                                                                                            \
            hipLaunchKernelGGL((k_final_layernorm<NG, true>), grid, block, 0, (hipStream_t)hip_stream, (const ushort4 *)x_dev,  \
                               (const ushort4 *)delta_dev, (const ushort4 *)gamma_dev, (const ushort4 *)beta_dev, out_dev, rows, \
                               T, skip, width, eps);                                                                            \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_final_layernorm<NG, false>), grid, block, 0, (hipStream_t)hip_stream, (const ushort4 *)x_dev, \
                               (const ushort4 *)delta_dev, (const ushort4 *)gamma_dev, (const ushort4 *)beta_dev, out_dev, rows, \
                               T, skip, width, eps);                                                                            \
    } while (0)
    ENC_NG_SWITCH(width / 256, FL)
#undef FL
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// ---- bias-lagged residual stream -----------------------------------------------------------------------------------------
// With the projection and fc2 GEMMs accumulating straight into the residual stream (D = A*W + C, beta = 1, no bias), what
// the stream lacks is the sum of the biases of the residual updates so far — a constant vector per position in the stack
// (f32, computed once from the weights).  LayerNorm adds it on the fly: y = LayerNorm(u + bias_sum).  One read of the
// stream and one write of y per LayerNorm instead of two reads and two writes (k_add_layernorm), and the residual add
// itself happens in the GEMM's f32 accumulator (one rounding to bf16 instead of two).
// SKIP > 0: only rows [skip, T) of every image are normalised and written densely (the final LayerNorm of the patch rows).
// A wavefront takes BLN_RPW consecutive rows: gamma, beta and the bias sums (5 x the bytes of a bf16 row) are loaded once per
// wavefront instead of once per row — they come out of the L1, whose bandwidth they otherwise share with the rows — and the rows'
// loads are all requested before the first reduction.
#ifndef BLN_RPW
#define BLN_RPW 4
#endif
template <int NG, int OUT>      // OUT 0: bf16, 1: f32 (the bf16-rounded value widened)
__global__ __launch_bounds__(TPB) void k_bias_layernorm(const ushort4 *__restrict__ u, const float4 *__restrict__ bias_sum,
                                                        const ushort4 *__restrict__ gamma, const ushort4 *__restrict__ beta,
                                                        void *__restrict__ y, int64_t rows_out, int T, int skip, int width,
                                                        float eps)
{
    const int lane = threadIdx.x & 63;
    const int64_t orow0 = (((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6) * BLN_RPW;
    if (orow0 >= rows_out) return;
    const int w4 = width >> 2;
    ushort4 raw[BLN_RPW][NG];
#pragma unroll
    for (int r = 0; r < BLN_RPW; ++r) {
        const int64_t orow = orow0 + r < rows_out ? orow0 + r : rows_out - 1;
        int64_t row = orow;
        if (skip > 0) {
            const int np = T - skip;
            const int64_t b = orow / np;
            row = b * T + skip + (orow - b * np);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) raw[r][g] = u[row * w4 + lane + 64 * g];
    }
    float4 bs[NG];
    ushort4 ga[NG], be[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { bs[g] = bias_sum[lane + 64 * g]; ga[g] = gamma[lane + 64 * g]; be[g] = beta[lane + 64 * g]; }
#pragma unroll
    for (int r = 0; r < BLN_RPW; ++r) {
        const int64_t orow = orow0 + r;
        if (orow >= rows_out) break;
        float v[NG][4];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const ushort4 a = raw[r][g];
            v[g][0] = bf2f(a.x) + bs[g].x; v[g][1] = bf2f(a.y) + bs[g].y; v[g][2] = bf2f(a.z) + bs[g].z; v[g][3] = bf2f(a.w) + bs[g].w;
        }
        float mean, rstd;
        ln_stats<NG>(v, width, eps, mean, rstd);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = lane + 64 * g;
            ushort4 o;
            o.x = f2bf((v[g][0] - mean) * rstd * bf2f(ga[g].x) + bf2f(be[g].x));
            o.y = f2bf((v[g][1] - mean) * rstd * bf2f(ga[g].y) + bf2f(be[g].y));
            o.z = f2bf((v[g][2] - mean) * rstd * bf2f(ga[g].z) + bf2f(be[g].z));
            o.w = f2bf((v[g][3] - mean) * rstd * bf2f(ga[g].w) + bf2f(be[g].w));
            if (OUT) ((float4 *)y)[orow * w4 + c] = make_float4(bf2f(o.x), bf2f(o.y), bf2f(o.z), bf2f(o.w));
            else ((ushort4 *)y)[orow * w4 + c] = o;
        }
    }
}

extern "C" bsc_status bsc_enc_bias_layernorm(const void *u_dev, const void *bias_sum_f32_dev, const void *gamma_dev,
                                             const void *beta_dev, void *y_dev, int32_t out_f32, int32_t B, int32_t T,
                                             int32_t skip, int32_t width, float eps, void *hip_stream)
{
    if (!u_dev || !bias_sum_f32_dev || !gamma_dev || !beta_dev || !y_dev || B < 1 || skip < 0 || T <= skip ||
        width % 256 != 0 || width < 256 || width > 2048) {
        bsc_set_error("bsc_enc_bias_layernorm: invalid argument");
        return BSC_E_INVALID;
    }
    const int64_t rows = (int64_t)B * (T - skip);
    const dim3 grid((unsigned)(((rows + BLN_RPW - 1) / BLN_RPW * 64 + TPB - 1) / TPB)), block(TPB);
#define BL(NG)                                                                                                                  \
    do {                                                                                                                        \
        if (out_f32)                                                                                                            \
            hipLaunchKernelGGL((k_bias_layernorm<NG, 1>), grid, block, 0, (hipStream_t)hip_stream, (const ushort4 *)u_dev,      \
                               (const float4 *)bias_sum_f32_dev, (const ushort4 *)gamma_dev, (const ushort4 *)beta_dev, y_dev,  \
                               rows, T, skip, width, eps);                                                                      \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_bias_layernorm<NG, 0>), grid, block, 0, (hipStream_t)hip_stream, (const ushort4 *)u_dev,      \
                               (const float4 *)bias_sum_f32_dev, (const ushort4 *)gamma_dev, (const ushort4 *)beta_dev, y_dev,  \
                               rows, T, skip, width, eps);                                                                      \
    } while (0)
    ENC_NG_SWITCH(width / 256, BL)
#undef BL
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Frame pre-processing of the patch-feature provider (memory_2.py:733-736 + transform_ :71-74) in one pass:
//   u8 (B,H,W,C) -> /255 -> antialiased bilinear resize to (S,S) -> ImageNet normalise -> bf16 patch matrix
//   (B, g*g, 3*p*p), i.e. already unfolded for the patch-embedding GEMM.
// Resize = the separable triangle filter PyTorch's F.interpolate(mode="bilinear", antialias=True) uses:
// support = scale (when down-sampling), taps [int(c - support + .5), int(c + support + .5)), weights
// max(0, 1 - |i + .5 - c| / scale) normalised to 1.
#define PP_TAPS 12
__device__ __forceinline__ void aa_taps(int o, float scale, int in_size, int &lo, int &n, float (&w)[PP_TAPS])
{
    const float support = scale >= 1.f ? scale : 1.f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    const float center = scale * ((float)o + 0.5f);
    lo = max((int)(center - support + 0.5f), 0);
    const int hi = min((int)(center + support + 0.5f), in_size);
    n = min(hi - lo, PP_TAPS);
    float tot = 0.f;
    for (int i = 0; i < PP_TAPS; ++i) {
        float v = 0.f;
        if (i < n) {
            const float x = ((float)(i + lo) - center + 0.5f) * invscale;
            v = fmaxf(0.f, 1.f - fabsf(x));
        }
        w[i] = v;
        tot += v;
    }
    const float inv = tot != 0.f ? 1.f / tot : 0.f;
    for (int i = 0; i < PP_TAPS; ++i) w[i] *= inv;
}

// first tap and tap count of output index o alone (the same arithmetic as aa_taps)
__device__ __forceinline__ void aa_bounds(int o, float scale, int in_size, int &lo, int &n)
{
    const float support = scale >= 1.f ? scale : 1.f;
    const float center = scale * ((float)o + 0.5f);
    lo = max((int)(center - support + 0.5f), 0);
    const int hi = min((int)(center + support + 0.5f), in_size);
    n = min(hi - lo, PP_TAPS);
}

// one element of the unfolded patch matrix (row, k of K): mode 0 bf16, 1 f32, 2 fp16 pieces in the chunk-interleaved layout of the
// split-operand GEMM (encoder_gemm.hip: row = K / 32 chunks of [32 h | 32 l]; K % 32 == 0)
__device__ __forceinline__ void pp_store(void *out, int mode, int64_t row, int K, int k, float v)
{
    if (mode == 0) ((uint16_t *)out)[row * K + k] = f2bf(v);
    else if (mode == 1) ((float *)out)[row * K + k] = v;
    else {
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        uint16_t *o = (uint16_t *)out + row * 2 * K + (int64_t)(k >> 5) * 64 + (k & 31);
        o[0] = *(const uint16_t *)&h;
        o[32] = *(const uint16_t *)&l;
    }
}

// mode 2 with 3 p^2 not a multiple of 32 (patch 14: 588): the piece rows are padded with zeros to the next multiple (Kp = 608), the
// thread of patch pixel t < Kp - 3 p^2 writes pad column 3 p^2 + t
__device__ __forceinline__ void pp_store3(void *out, int mode, int64_t row, int p, int py, int px, float v0, float v1, float v2)
{
    const int K = 3 * p * p, Kp = mode == 2 ? (K + 31) / 32 * 32 : K, t = py * p + px;
    pp_store(out, mode, row, Kp, t, v0);
    pp_store(out, mode, row, Kp, p * p + t, v1);
    pp_store(out, mode, row, Kp, 2 * p * p + t, v2);
    if (t < Kp - K) pp_store(out, mode, row, Kp, K + t, 0.f);
}

__global__ __launch_bounds__(TPB) void k_preprocess_patches(const uint8_t *__restrict__ rgb, int B, int H, int W, int C,
                                                            int S, int p, void *__restrict__ out, int mode, float m0, float m1,
                                                            float m2, float s0, float s1, float s2)
{
    const int64_t idx = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (idx >= (int64_t)B * S * S) return;
    const int x = (int)(idx % S), y = (int)((idx / S) % S), b = (int)(idx / ((int64_t)S * S));
    float wy[PP_TAPS], wx[PP_TAPS];
    int ylo, yn, xlo, xn;
    aa_taps(y, (float)H / (float)S, H, ylo, yn, wy);
    aa_taps(x, (float)W / (float)S, W, xlo, xn, wx);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const uint8_t *img = rgb + (int64_t)b * H * W * C;
    if (C == 4) {           // RGBA frames: one 32-bit load per tap, all taps of a row in flight together
        for (int i = 0; i < yn; ++i) {
            const uint32_t *row = (const uint32_t *)(img + ((int64_t)(ylo + i) * W + xlo) * 4);
            uint32_t pxl[PP_TAPS];
#pragma unroll
            for (int k = 0; k < PP_TAPS; ++k) pxl[k] = row[k < xn ? k : 0];
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int k = 0; k < PP_TAPS; ++k) {             // weights beyond xn are zero
                r0 += wx[k] * (float)(pxl[k] & 0xffu);
                r1 += wx[k] * (float)((pxl[k] >> 8) & 0xffu);
                r2 += wx[k] * (float)((pxl[k] >> 16) & 0xffu);
            }
            a0 += wy[i] * r0; a1 += wy[i] * r1; a2 += wy[i] * r2;
        }
    } else {
        for (int i = 0; i < yn; ++i) {
            const uint8_t *row = img + ((int64_t)(ylo + i) * W + xlo) * C;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
            for (int k = 0; k < xn; ++k) {
                r0 += wx[k] * (float)row[k * C + 0];
                r1 += wx[k] * (float)row[k * C + 1];
                r2 += wx[k] * (float)row[k * C + 2];
            }
            a0 += wy[i] * r0; a1 += wy[i] * r1; a2 += wy[i] * r2;
        }
    }
    const int g = S / p, gy = y / p, gx = x / p, py = y - gy * p, px = x - gx * p;
    const int64_t row = (int64_t)b * g * g + (int64_t)gy * g + gx;
    pp_store3(out, mode, row, p, py, px, (a0 * (1.f / 255.f) - m0) / s0, (a1 * (1.f / 255.f) - m1) / s1, (a2 * (1.f / 255.f) - m2) / s2);
}

// RGBA frames, one workgroup per output patch.  The taps of every output row and column (first input index, count, 12
// normalised weights) are tabulated once per call by k_pp_taps; the input window of the patch (p * scale + 2 * support
// pixels each way) is staged in LDS with coalesced 32-bit loads, and every thread evaluates its output pixel from LDS:
// eight input pixels fetched per output pixel instead of ~35 cached loads, no per-thread tap arithmetic, and the
// patch's three channel planes leave as contiguous runs.  Same taps, weights and summation order as k_preprocess_patches.
#define PPT_MAXW 64
#define PPT_MAXH 48
struct PpTap {
    int lo, n;
    float w[PP_TAPS];
};

__global__ void k_pp_taps(int S, int H, int W, PpTap *__restrict__ ty, PpTap *__restrict__ tx)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= 2 * S) return;
    const bool isx = o >= S;
    const int i = isx ? o - S : o;
    PpTap t;
    aa_taps(i, isx ? (float)W / (float)S : (float)H / (float)S, isx ? W : H, t.lo, t.n, t.w);
    (isx ? tx : ty)[i] = t;
}

template <int TXN>      // column taps evaluated per row (>= the largest tap count of any output column)
__global__ __launch_bounds__(TPB) void k_preprocess_patches_tiled(const uint32_t *__restrict__ rgba, int B, int H, int W, int S,
                                                                  int p, const PpTap *__restrict__ ty,
                                                                  const PpTap *__restrict__ tx, void *__restrict__ out, int mode,
                                                                  float m0, float m1, float m2, float s0, float s1, float s2)
{
    __shared__ uint32_t tile[PPT_MAXH][PPT_MAXW + 1];
    const int g = S / p;
    const int patch = blockIdx.x % (g * g), b = blockIdx.x / (g * g);
    const int gy = patch / g, gx = patch % g;
    // window = union of the taps of the patch's first and last output row / column (taps are monotone in the output index);
    // its bounds are computed, not read from the tap table: the window loads and the thread's tap weights then leave in ONE round
    // trip (read from the table first, the bounds cost a dependent trip of their own ahead of the window's)
    int y_lo, x_lo, lo_l, n_l;
    aa_bounds(gy * p, (float)H / (float)S, H, y_lo, n_l);
    aa_bounds(gy * p + p - 1, (float)H / (float)S, H, lo_l, n_l);
    const int win_h = lo_l + n_l - y_lo;
    aa_bounds(gx * p, (float)W / (float)S, W, x_lo, n_l);
    aa_bounds(gx * p + p - 1, (float)W / (float)S, W, lo_l, n_l);
    const int win_w = lo_l + n_l - x_lo;
    const uint32_t *img = rgba + (int64_t)b * H * W;
    // TPB % p == 0: all (window row, output column) items of a thread share the output column, i.e. one set of column weights
    const PpTap wx_mine = tx[gx * p + (int)threadIdx.x % p];
    const PpTap wy_mine = ty[gy * p + ((int)threadIdx.x < p * p ? (int)threadIdx.x / p : 0)];
    {
        // one wavefront per window row, the lane is the column (PPT_MAXW = 64): every load of the window is issued before the first
        // LDS store (as a loop of load -> store the window cost one memory round trip per 256 pixels, 8-9 in a row: the kernel's time)
        const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
        uint32_t v[PPT_MAXH / (TPB / 64)];
#pragma unroll
        for (int j = 0; j < PPT_MAXH / (TPB / 64); ++j) {
            const int r = r0 + (TPB / 64) * j;
            const bool in = r < win_h && c < win_w;
            v[j] = in ? img[(int64_t)(y_lo + r) * W + x_lo + c] : 0u;
        }
#pragma unroll
        for (int j = 0; j < PPT_MAXH / (TPB / 64); ++j) {
            const int r = r0 + (TPB / 64) * j;
            if (r < win_h && c < win_w) tile[r][c] = v[j];
        }
    }
    __syncthreads();
    // The row sums r(input row, output column) = sum_k wx[k] * pixel do not depend on the output ROW: every (window row, output
    // column) pair is evaluated once (win_h * p of them, ~3 per thread) and shared through LDS by the p output rows that weigh it —
    // the same products and the same summation order as one thread per output pixel doing all its yn x TXN taps (bit-identical
    // results), at 40 % of the vector instructions: the kernel was bound by them (0.365 ms for 0.6 GB of traffic).
    __shared__ float hs[PPT_MAXH][3][16 + 1];
    const bool shared_rows = p <= 16 && TPB % p == 0;
    if (shared_rows) {
        for (int i = threadIdx.x; i < win_h * p; i += TPB) {
            const int r = i / p, px = i - r * p;
            const uint32_t *row = &tile[r][wx_mine.lo - x_lo];
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int k = 0; k < TXN; ++k) {                 // weights beyond n are zero
                const uint32_t v = row[k < wx_mine.n ? k : 0];
                r0 += wx_mine.w[k] * (float)(v & 0xffu);
                r1 += wx_mine.w[k] * (float)((v >> 8) & 0xffu);
                r2 += wx_mine.w[k] * (float)((v >> 16) & 0xffu);
            }
            hs[r][0][px] = r0; hs[r][1][px] = r1; hs[r][2][px] = r2;
        }
        __syncthreads();
    }
    if ((int)threadIdx.x >= p * p) return;
    const int py = threadIdx.x / p, px = threadIdx.x - py * p;
    const PpTap *wyp = &ty[gy * p + py];            // the generic path reads its weights in the row loop (dynamic index)
    const int yn = wy_mine.n, ylo = wy_mine.lo;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (shared_rows) {
#pragma unroll
        for (int i = 0; i < PP_TAPS; ++i) {
            if (i < yn) {
                const float wyi = wy_mine.w[i];
                const int r = ylo - y_lo + i;
                a0 += wyi * hs[r][0][px]; a1 += wyi * hs[r][1][px]; a2 += wyi * hs[r][2][px];
            }
        }
    } else {
        const PpTap wx = tx[gx * p + px];
        for (int i = 0; i < yn; ++i) {
            const uint32_t *row = &tile[ylo - y_lo + i][wx.lo - x_lo];
            float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int k = 0; k < TXN; ++k) {                 // weights beyond n are zero
                const uint32_t v = row[k < wx.n ? k : 0];
                r0 += wx.w[k] * (float)(v & 0xffu);
                r1 += wx.w[k] * (float)((v >> 8) & 0xffu);
                r2 += wx.w[k] * (float)((v >> 16) & 0xffu);
            }
            const float wyi = wyp->w[i];
            a0 += wyi * r0; a1 += wyi * r1; a2 += wyi * r2;
        }
    }
    const int64_t row = (int64_t)b * g * g + patch;
    pp_store3(out, mode, row, p, py, px, (a0 * (1.f / 255.f) - m0) / s0, (a1 * (1.f / 255.f) - m1) / s1, (a2 * (1.f / 255.f) - m2) / s2);
}

extern "C" bsc_status bsc_enc_preprocess_patches(const void *rgb_dev, int32_t B, int32_t H, int32_t W, int32_t C,
                                                 int32_t S, int32_t patch, void *out_dev, const float *mean3_host,
                                                 const float *std3_host, void *hip_stream)
{
    return bsc_enc_preprocess_patches_typed(rgb_dev, B, H, W, C, S, patch, out_dev, 0, mean3_host, std3_host, hip_stream);
}

extern "C" bsc_status bsc_enc_preprocess_patches_typed(const void *rgb_dev, int32_t B, int32_t H, int32_t W, int32_t C,
                                                       int32_t S, int32_t patch, void *out_dev, int32_t out_mode,
                                                       const float *mean3_host, const float *std3_host, void *hip_stream)
{
    if (!rgb_dev || !out_dev || !mean3_host || !std3_host || B < 1 || C < 3 || patch < 1 || S % patch != 0 || out_mode < 0 ||
        out_mode > 2 || (out_mode == 2 && patch * patch < 32)) {
        bsc_set_error("bsc_enc_preprocess_patches: invalid argument");
        return BSC_E_INVALID;
    }
    if ((float)H / S > (PP_TAPS - 1) / 2.0f || (float)W / S > (PP_TAPS - 1) / 2.0f) {
        bsc_set_error("bsc_enc_preprocess_patches: down-scale factor above %g not supported", (PP_TAPS - 1) / 2.0);
        return BSC_E_INVALID;
    }
    const float fy = (float)H / S, fx = (float)W / S;
    const float suph = fy >= 1.f ? fy : 1.f, supw = fx >= 1.f ? fx : 1.f;
    if (C == 4 && patch * patch <= TPB && patch * fy + 2.f * suph + 3.f <= PPT_MAXH && patch * fx + 2.f * supw + 3.f <= PPT_MAXW) {
        const int g = S / patch;
        // per-device tap table (2 x S entries), rewritten by every call on the call's stream
        static PpTap *tab[16] = {nullptr};
        static int tab_S[16] = {0};
        int dev = 0;
        BSC_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16) return BSC_E_INVALID;
        if (tab_S[dev] < S) {
            if (tab[dev]) BSC_HIP(hipFree(tab[dev]));
            BSC_HIP(hipMalloc((void **)&tab[dev], sizeof(PpTap) * 2 * S));
            tab_S[dev] = S;
        }
        hipLaunchKernelGGL(k_pp_taps, dim3((unsigned)((2 * S + 127) / 128)), dim3(128), 0, (hipStream_t)hip_stream, S, H, W,
                           tab[dev], tab[dev] + S);
#define PPT_LAUNCH(TXN)                                                                                               \
        hipLaunchKernelGGL((k_preprocess_patches_tiled<TXN>), dim3((unsigned)((int64_t)B * g * g)), dim3(TPB), 0,         \
                           (hipStream_t)hip_stream, (const uint32_t *)rgb_dev, B, H, W, S, patch, tab[dev], tab[dev] + S, \
                           out_dev, out_mode, mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1],    \
                           std3_host[2])
        if (2.f * supw + 2.f <= 8.f) PPT_LAUNCH(8);     // a column has at most floor(2 * support) + 2 taps
        else PPT_LAUNCH(PP_TAPS);
#undef PPT_LAUNCH
        BSC_HIP(hipGetLastError());
        return BSC_OK;
    }
    const int64_t n = (int64_t)B * S * S;
    hipLaunchKernelGGL(k_preprocess_patches, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)rgb_dev, B, H, W, C, S, patch, out_dev, out_mode, mean3_host[0], mean3_host[1],
                       mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

// ---- self-attention of the ViT patch-feature provider: short sequences (T = 197 / 261), head dim 64 -------------------
// One (image, head) item at a time per workgroup.  Q, K and V of the head live in LDS (Q, K row-major, V transposed
// so that the P.V operand is a contiguous 8-byte read); every wavefront takes strips of 16 queries:
//   S^T = K . Q^T     A = 16 keys x 32 features, B = 16 queries x 32 features, both from LDS
//                     -> accumulator tile t holds S^T[key 16t + 4g + i][query lane & 15]  (g = lane >> 4)
//   softmax           along the keys: a lane owns 4 keys of every tile for ONE query, the other keys of that query sit
//                     in the lanes 16, 32, 48 apart (two xor-shuffles); exp2 with log2(e)/sqrt(d) folded in
//   O = P . V         the S^T accumulators already ARE the A operand of this product (row = query = lane & 15,
//                     k = 8 keys per lane) if the contraction index of a 32-key step is taken in the order the
//                     accumulators hold it (4 keys of tile 2s, 4 keys of tile 2s+1); V^T is read in that same order
// v_mfma_f32_16x16x32_bf16 throughout; f32 accumulation and softmax, P rounded to bf16 like the library kernels do.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)      // v_cvt_pk_bf16_f32 (round to nearest even)
{
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *(const uint32_t *)&r;
}

// The staging registers of the next item are first-class vector values (not arrays: an array that is live across the
// item loop is left in memory by the compiler, and the loads would be waited for on the spot).
template <int N> struct u32vec { typedef uint32_t type __attribute__((ext_vector_type(N))); };

// every global load of an (image, head) item is issued together (clamped addresses, no branches)
template <int NLD, int NTHR>
__device__ __forceinline__ void att_issue_loads(typename u32vec<4 * NLD>::type &k8, typename u32vec<4 * NLD>::type &v8,
                                                typename u32vec<4 * NLD>::type &q8, const uint16_t *__restrict__ qkv,
                                                int item, int T, int H, int tid)
{
    const int64_t tok_stride = (int64_t)3 * H * 64;
    const int b = item / H, h = item % H;
    const uint16_t *Qp = qkv + (int64_t)b * T * tok_stride + (int64_t)h * 64;
    const uint16_t *Kp = Qp + (int64_t)H * 64, *Vp = Qp + (int64_t)2 * H * 64;
#pragma unroll
    for (int r = 0; r < NLD; ++r) {
        const int idx = tid + NTHR * r;
        const int t = idx >> 3, ch = idx & 7;
        const int tc = t < T ? t : T - 1;
        const uint4 k = *(const uint4 *)(Kp + (int64_t)tc * tok_stride + ch * 8);
        const uint4 v = *(const uint4 *)(Vp + (int64_t)tc * tok_stride + ch * 8);
        const uint4 q = *(const uint4 *)(Qp + (int64_t)tc * tok_stride + ch * 8);
        k8[4 * r] = k.x; k8[4 * r + 1] = k.y; k8[4 * r + 2] = k.z; k8[4 * r + 3] = k.w;
        v8[4 * r] = v.x; v8[4 * r + 1] = v.y; v8[4 * r + 2] = v.z; v8[4 * r + 3] = v.w;
        q8[4 * r] = q.x; q8[4 * r + 1] = q.y; q8[4 * r + 2] = q.z; q8[4 * r + 3] = q.w;
    }
}

// NT key tiles of 16 (even): sequence length <= 16 * NT.  NW wavefronts, each takes the query strips w, w + NW, ...
// Persistent workgroups: a workgroup walks (image, head) items and issues the global loads of its NEXT item (K, V and
// its wavefronts' query strips, into registers) before it computes the current one out of LDS, so the HBM-bound load
// phase and the issue-bound compute phase overlap instead of alternating chip-wide.  Items are handed out by a ticket
// counter (work[0]; work[1] counts finished workgroups, the last one re-arms both), not by a static stride: a workgroup
// whose CU is shared with another stream's kernel simply takes fewer items.  With the static stride one slow or late
// workgroup stretched the whole launch — the rgb chain of libbscnav's own side stream, resident on one or two CUs for
// milliseconds, cost the 12 attention launches of a ViT-B forward 2.4 ms (bench pipeline 24.7 -> 22.x ms per step).
// work == nullptr keeps the static stride.
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention(const uint16_t *__restrict__ qkv, int T, int H, int items,
                                                       uint16_t *__restrict__ out, int *work)
{
    __shared__ int s_ticket;
    constexpr int NTHR = 64 * NW;
    constexpr int TP = NT * 16;
    constexpr int KP = 64 + 8;                              // K row pitch (bf16 elements)
    constexpr int VP = 4 * (((TP / 4 - 1) | 7) + 1) + 8;    // V^T row pitch: granules of 4 keys (xor-swizzled) + pad
    constexpr int NLD = (TP * 8 + NTHR - 1) / NTHR;         // 16-byte pieces of K (and of V) per thread
    constexpr int NSTRIP = (NT + NW - 1) / NW;              // strips of 16 queries per wavefront
    __shared__ __attribute__((aligned(16))) uint16_t sK[TP * KP];
    __shared__ __attribute__((aligned(16))) uint16_t sQ[TP * KP];
    __shared__ __attribute__((aligned(16))) uint16_t sVt[64 * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int64_t tok_stride = (int64_t)3 * H * 64;                 // elements between consecutive tokens
    const float c = 0.125f * 1.44269504088896340736f;               // 1/sqrt(64) * log2(e)
    const int nstrip = (T + 15) >> 4;
    typename u32vec<4 * NLD>::type k8, v8, q8;

    // software pipeline over the workgroup's items; the prefetch past the last item re-reads the last one (harmless)
    int item = blockIdx.x;
    if (work) {
        if (tid == 0) s_ticket = atomicAdd(&work[0], 1);
        __syncthreads();
        item = s_ticket;
    }
    att_issue_loads<NLD, NTHR>(k8, v8, q8, qkv, item < items ? item : items - 1, T, H, tid);
    int nxt = item;
    for (; item < items; item = nxt) {
        const int b = item / H, h = item % H;
        __syncthreads();                                            // the previous item's strips are done with LDS
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int idx = tid + NTHR * r;
            const int t = idx >> 3, ch = idx & 7;
            if (idx < TP * 8) {
                // padded keys: zero rows, by a mask (a select between two uint4 values is compiled through scratch memory:
                // 16 scratch stores + loads per staged row until round 4)
                const uint32_t keep = t < T ? 0xffffffffu : 0u;
                *(uint4 *)&sK[t * KP + ch * 8] = make_uint4(k8[4 * r] & keep, k8[4 * r + 1] & keep, k8[4 * r + 2] & keep, k8[4 * r + 3] & keep);
                *(uint4 *)&sQ[t * KP + ch * 8] = make_uint4(q8[4 * r], q8[4 * r + 1], q8[4 * r + 2], q8[4 * r + 3]);
                const uint32_t vv[4] = {v8[4 * r] & keep, v8[4 * r + 1] & keep, v8[4 * r + 2] & keep, v8[4 * r + 3] & keep};
                // V^T[d][t] lives at granule (t >> 2) ^ (d >> 3) of row d: the 8 rows a lane writes per element pair
                // and the 8 lanes that share a token then fall into different banks
                const int col = 4 * ((t >> 2) ^ ch) + (t & 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sVt[(ch * 8 + 2 * e) * VP + col] = (uint16_t)(vv[e] & 0xffffu);
                    sVt[(ch * 8 + 2 * e + 1) * VP + col] = (uint16_t)(vv[e] >> 16);
                }
            }
        }
        if (work && tid == 0) s_ticket = atomicAdd(&work[0], 1);    // everyone has read the previous ticket (barrier above)
        __syncthreads();
        nxt = work ? s_ticket : item + (int)gridDim.x;
        att_issue_loads<NLD, NTHR>(k8, v8, q8, qkv, nxt < items ? nxt : items - 1, T, H, tid);      // in flight during the strips below
#pragma unroll
        for (int si = 0; si < NSTRIP; ++si) {
            const int strip = wave + NW * si;
            if (strip >= nstrip) break;
            const int q0 = strip * 16;
            bf16x8_t bq[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bq[kk] = *(const bf16x8_t *)&sQ[(q0 + n) * KP + kk * 32 + g * 8];
            f32x4_t acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8_t a = *(const bf16x8_t *)&sK[(t * 16 + n) * KP + kk * 32 + g * 8];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[kk], acc[t], 0, 0, 0);
                }
                // keep the K fragment loads of later key tiles behind these MFMAs: left alone the scheduler hoists them all to the
                // top of the strip (register pressure; T = 261: 146 -> 109 us per launch together with the masked staging)
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // padded keys leave the softmax with -inf; for the usual lengths only the last two tiles have any.  The lane's key
            // limit is made opaque per strip: left loop-invariant, the compiler evaluates all 4 NT comparisons once per kernel
            // into scalar mask pairs, runs out of SGPRs and spills them into VGPR lanes (72 v_readlane per strip)
            int lim = T - g * 4;
            asm volatile("" : "+v"(lim));
            if (T > 16 * (NT - 2)) {
#pragma unroll
                for (int t = NT - 2; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (t * 16 + i >= lim) acc[t][i] = -INFINITY;
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (t * 16 + i >= lim) acc[t][i] = -INFINITY;
            }
            f32x4_t mv = acc[0];
#pragma unroll
            for (int t = 1; t < NT; ++t) mv = __builtin_elementwise_max(mv, acc[t]);
            float m = fmaxf(fmaxf(mv[0], mv[1]), fmaxf(mv[2], mv[3]));
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            const f32x4_t cv = {c, c, c, c}, nmc = {-m * c, -m * c, -m * c, -m * c};
            f32x4_t sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4_t e = __builtin_elementwise_fma(acc[t], cv, nmc);     // exp2((s - m) * c), v_pk_fma_f32
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
                acc[t] = e;
                sv += e;
            }
            float sum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            f32x4_t o[4];
            // (8 ks + g) ^ x == 8 ks + (g ^ x) for x < 8: the swizzle folds into two lane constants per d tile and the
            // key step becomes an immediate offset
            const uint16_t *v0p[4], *v1p[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                const int d = dt * 16 + n;
                v0p[dt] = &sVt[d * VP + 4 * (g ^ (d >> 3))];
                v1p[dt] = &sVt[d * VP + 4 * ((4 + g) ^ (d >> 3))];
            }
#pragma unroll
            for (int ks = 0; ks < NT / 2; ++ks) {
                uint4 pa;
                pa.x = pack_bf16(acc[2 * ks][0], acc[2 * ks][1]);
                pa.y = pack_bf16(acc[2 * ks][2], acc[2 * ks][3]);
                pa.z = pack_bf16(acc[2 * ks + 1][0], acc[2 * ks + 1][1]);
                pa.w = pack_bf16(acc[2 * ks + 1][2], acc[2 * ks + 1][3]);
                const bf16x8_t a = *(bf16x8_t *)&pa;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const uint2 v0 = *(const uint2 *)(v0p[dt] + 32 * ks), v1 = *(const uint2 *)(v1p[dt] + 32 * ks);
                    uint4 vb = make_uint4(v0.x, v0.y, v1.x, v1.y);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(bf16x8_t *)&vb, a, o[dt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // V^T is the FIRST operand: o[dt][i] = O^T[d = 16 dt + 4 g + i][query q0 + n] — the lane that holds a query's row sum holds
            // its outputs, four consecutive features per accumulator: 8-byte stores, no shuffles (with the queries along the
            // registers the epilogue was 16 two-byte stores + 4 shuffles per lane and strip: 40 % of the kernel's time)
            const float inv = 1.f / sum;
            const int q = q0 + n;
            if (q < T) {
                uint16_t *dst = out + ((int64_t)b * T + q) * H * 64 + (int64_t)h * 64 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *(uint2 *)(dst + dt * 16) = make_uint2(pack_bf16(o[dt][0] * inv, o[dt][1] * inv), pack_bf16(o[dt][2] * inv, o[dt][3] * inv));
            }
        }
    }
    // the last workgroup to leave re-arms the counters for the next launch on this stream
    if (work && tid == 0) {
        __threadfence();
        if (atomicAdd(&work[1], 1) == (int)gridDim.x - 1) {
            work[0] = 0;
            work[1] = 0;
            __threadfence();
        }
    }
}

// ---- the same attention on 32 x 32 x 16 MFMA tiles (sequences up to 224 tokens) ----------------------------------------------
// k_attention reads every K and V^T fragment from LDS once per 16 queries: 56 KB of LDS reads per strip, ~730 KB per (image, head)
// item, and the LDS pipe — with the exp2 of the softmax on the vector unit — is what the kernel waits for (MFMA busy 11 %).  With
// v_mfma_f32_32x32x16_bf16 a wavefront takes 32 queries: every fragment read feeds twice the matrix work (half the LDS bytes per
// item), seven strips of 32 cover 197 tokens with one strip per wavefront (13 strips of 16 over 7 wavefronts left one in seven idle
// half the time), and the two halves of a wavefront hold a query's keys, so max and sum need ONE cross-lane step instead of two.
//   S^T tile t (32 keys x 32 queries) = K_t . Q^T     A = K rows (lane & 31 = key, 8 features at 8 (lane >> 5) of the 16-wide k step)
//                                                     B = Q rows (lane & 31 = query), four k steps
//        -> acc[t][r] = S^T[key 32 t + 8 (r >> 2) + 4 hh + (r & 3)][query lane & 31],  hh = lane >> 5
//   O^T (64 x 32 queries) = V^T . P^T                  the accumulators ARE the B operand if a 16-key step takes its contraction
//        index in accumulator order: step (t, jp) = registers 8 jp .. 8 jp + 7 = keys 32 t + 16 jp + {0..3, 8..11} + 4 hh;
//        V^T is read in that order (two 8-byte reads per lane and step)
// Staging: a thread owns FOUR consecutive tokens of one 16-byte piece, so V^T (row = feature, column = key) is written four keys at
// a time (8-byte stores); its row pitch is 4 x odd elements: the 32 feature rows a fragment read touches fall into 32 different
// bank pairs.  Q goes from global memory straight into the B-operand registers of its wavefront (no LDS copy).
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// One of the 12 global loads of a thread for an (image, head) item: J = 0..3 the K rows of its token group (4 consecutive tokens,
// one 16-byte piece each), 4..7 the V rows, 8..11 the four k-steps of its wavefront's Q fragment (straight in MFMA operand layout:
// Q never visits LDS).  Separate calls so that they can be placed between the MFMAs of Q.K^T.
template <int J>
__device__ __forceinline__ void att32_load_one(u32vec<16>::type &k8, u32vec<16>::type &v8, u32vec<16>::type &bqn,
                                               const uint16_t *__restrict__ qkv, int item, int T, int H, int tid)
{
    const int64_t tok_stride = (int64_t)3 * H * 64;
    const int b = item / H, h = item % H;
    const uint16_t *Qp = qkv + (int64_t)b * T * tok_stride + (int64_t)h * 64;
    if (J < 8) {
        const int tg = tid >> 3, ch = tid & 7, t = 4 * tg + (J & 3), tc = t < T ? t : T - 1;
        const uint16_t *src = Qp + (int64_t)(J < 4 ? 1 : 2) * H * 64 + (int64_t)tc * tok_stride + ch * 8;
        const uint4 v = *(const uint4 *)src;
        const int o = 4 * (J & 3);
        if (J < 4) { k8[o] = v.x; k8[o + 1] = v.y; k8[o + 2] = v.z; k8[o + 3] = v.w; }
        else { v8[o] = v.x; v8[o + 1] = v.y; v8[o + 2] = v.z; v8[o + 3] = v.w; }
    } else {
        const int lane = tid & 63, q = (tid >> 6) * 32 + (lane & 31), qc = q < T ? q : T - 1, kk = J - 8;
        const uint4 v = *(const uint4 *)(Qp + (int64_t)qc * tok_stride + kk * 16 + (lane >> 5) * 8);
        bqn[4 * kk] = v.x; bqn[4 * kk + 1] = v.y; bqn[4 * kk + 2] = v.z; bqn[4 * kk + 3] = v.w;
    }
}
template <int J0, int J1>
__device__ __forceinline__ void att32_load_range(u32vec<16>::type &k8, u32vec<16>::type &v8, u32vec<16>::type &bqn,
                                                 const uint16_t *__restrict__ qkv, int item, int T, int H, int tid)
{
    if constexpr (J0 < J1) {
        att32_load_one<J0>(k8, v8, bqn, qkv, item, T, H, tid);
        att32_load_range<J0 + 1, J1>(k8, v8, bqn, qkv, item, T, H, tid);
    }
}

template <int NT2, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention32(const uint16_t *__restrict__ qkv, int T, int H, int items,
                                                         uint16_t *__restrict__ out, int *work)
{
    __shared__ int s_ticket;
    constexpr int NTHR = 64 * NW;
    constexpr int TP = NT2 * 32;
    static_assert(TP * 2 == NTHR && NT2 >= 6, "one (4-token group, 16-byte piece) item per thread; six key tiles carry the next item's loads");
    constexpr int KP = 64 + 8;                              // K row pitch (bf16 elements)
    constexpr int VP = TP + 4;                              // V^T row pitch: 4 x odd
    __shared__ __attribute__((aligned(16))) uint16_t sK[TP * KP];
    __shared__ __attribute__((aligned(16))) uint16_t sVt[64 * VP + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 31, hh = lane >> 5;
    const float c = 0.125f * 1.44269504088896340736f;               // 1/sqrt(64) * log2(e)
    u32vec<16>::type k8, v8, bqn;

    int item = blockIdx.x;
    if (work) {
        if (tid == 0) s_ticket = atomicAdd(&work[0], 1);
        __syncthreads();
        item = s_ticket;
    }
    att32_load_range<0, 12>(k8, v8, bqn, qkv, item < items ? item : items - 1, T, H, tid);
    int nxt = item;
#ifdef BSC_ATT_PROFILE
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define ATT_T(k) { const long long now_ = clock64(); tph[k] += now_ - tq; tq = now_; }
    tq = clock64();
#else
#define ATT_T(k)
#endif
    for (; item < items; item = nxt) {
        const int b = item / H, h = item % H;
        __syncthreads();                                            // the previous item's strips are done with LDS
        ATT_T(0)
        {
            // the thread's token group: K rows as they are, V transposed four tokens at a time (8-byte stores)
            const int tg = tid >> 3, ch = tid & 7, t0 = 4 * tg;
            uint32_t keep[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) keep[u] = t0 + u < T ? 0xffffffffu : 0u;          // padded keys: zero rows
#pragma unroll
            for (int u = 0; u < 4; ++u)
                *(uint4 *)&sK[(t0 + u) * KP + ch * 8] = make_uint4(k8[4 * u] & keep[u], k8[4 * u + 1] & keep[u], k8[4 * u + 2] & keep[u], k8[4 * u + 3] & keep[u]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {                           // features 2e, 2e+1 of the piece
                const uint32_t w0 = v8[e] & keep[0], w1 = v8[4 + e] & keep[1], w2 = v8[8 + e] & keep[2], w3 = v8[12 + e] & keep[3];
                *(uint2 *)&sVt[(ch * 8 + 2 * e) * VP + t0] = make_uint2((w0 & 0xffffu) | (w1 << 16), (w2 & 0xffffu) | (w3 << 16));
                *(uint2 *)&sVt[(ch * 8 + 2 * e + 1) * VP + t0] = make_uint2((w0 >> 16) | (w1 & 0xffff0000u), (w2 >> 16) | (w3 & 0xffff0000u));
            }
        }
        if (work && tid == 0) s_ticket = atomicAdd(&work[0], 1);    // everyone has read the previous ticket (barrier above)
        __syncthreads();
        nxt = work ? s_ticket : item + (int)gridDim.x;
        const int nxc = nxt < items ? nxt : items - 1;
        ATT_T(1)
        const int q0 = wave * 32;
        bf16x8_t bq[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 v = make_uint4(bqn[4 * kk], bqn[4 * kk + 1], bqn[4 * kk + 2], bqn[4 * kk + 3]);
            bq[kk] = *(const bf16x8_t *)&v;
        }
        if (q0 >= T) {
            att32_load_range<0, 12>(k8, v8, bqn, qkv, nxc, T, H, tid);
            continue;
        }
        // Q.K^T with the next item's 12 loads spread between its MFMAs (requested in one burst by all wavefronts they queued at the
        // texture path: ~2 k clocks per item on the slowest wavefront)
        f32x16_t acc[NT2];
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8_t a = *(const bf16x8_t *)&sK[(t * 32 + nl) * KP + kk * 16 + hh * 8];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[kk], acc[t], 0, 0, 0);
            }
            if (t == 0) att32_load_range<0, 2>(k8, v8, bqn, qkv, nxc, T, H, tid);
            if (t == 1) att32_load_range<2, 4>(k8, v8, bqn, qkv, nxc, T, H, tid);
            if (t == 2) att32_load_range<4, 6>(k8, v8, bqn, qkv, nxc, T, H, tid);
            if (t == 3) att32_load_range<6, 8>(k8, v8, bqn, qkv, nxc, T, H, tid);
            if (t == 4) att32_load_range<8, 10>(k8, v8, bqn, qkv, nxc, T, H, tid);
            if (t == 5) att32_load_range<10, 12>(k8, v8, bqn, qkv, nxc, T, H, tid);
            __builtin_amdgcn_sched_barrier(0);                      // keep the K fragment loads of later tiles behind these MFMAs
        }
        ATT_T(3)
        ATT_T(2)
        // padded keys leave the softmax with -inf: only the last two tiles can hold any (the lane's limit is made opaque: left
        // loop-invariant the comparisons turn into scalar mask pairs that do not fit the SGPR file)
        int lim = T - hh * 4;
        asm volatile("" : "+v"(lim));
        if (T > 32 * (NT2 - 2)) {
#pragma unroll
            for (int t = (NT2 > 2 ? NT2 - 2 : 0); t < NT2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 32 + 8 * (r >> 2) + (r & 3) >= lim) acc[t][r] = -INFINITY;
        } else {
#pragma unroll
            for (int t = 0; t < NT2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 32 + 8 * (r >> 2) + (r & 3) >= lim) acc[t][r] = -INFINITY;
        }
        // row maximum: v_max3_f32 trees over the lane's 16 NT2 scores, then the other half of the wavefront
        float m;
        {
            float mt[NT2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const float a0 = fmaxf(fmaxf(acc[t][0], acc[t][1]), acc[t][2]), a1 = fmaxf(fmaxf(acc[t][3], acc[t][4]), acc[t][5]);
                const float a2 = fmaxf(fmaxf(acc[t][6], acc[t][7]), acc[t][8]), a3 = fmaxf(fmaxf(acc[t][9], acc[t][10]), acc[t][11]);
                const float a4 = fmaxf(fmaxf(acc[t][12], acc[t][13]), acc[t][14]);
                mt[t] = fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), acc[t][15]));
            }
            m = mt[0];
#pragma unroll
            for (int t = 1; t < NT2; ++t) m = fmaxf(m, mt[t]);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        const float nmc = -m * c;
        ATT_T(4)
        // exp2 and P.V tile by tile: the V^T fragments of a tile are requested first, its 16 scores go through exp2 / sum / bf16
        // packing on the vector unit while the previous tile's four MFMAs run, then its own MFMAs are issued
        f32x16_t o[2];
        const uint16_t *vp[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            vp[dt] = &sVt[(dt * 32 + nl) * VP + 4 * hh];
        }
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            uint4 vb[2][2];
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const uint2 v0 = *(const uint2 *)(vp[dt] + t * 32 + 16 * jp), v1 = *(const uint2 *)(vp[dt] + t * 32 + 16 * jp + 8);
                    vb[jp][dt] = make_uint4(v0.x, v0.y, v1.x, v1.y);
                }
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(fmaf(acc[t][r], c, nmc));       // exp2((s - m) * c)
            sum += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7])) + (((e[8] + e[9]) + (e[10] + e[11])) + ((e[12] + e[13]) + (e[14] + e[15])));
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                uint4 pa;
                pa.x = pack_bf16(e[8 * jp], e[8 * jp + 1]);
                pa.y = pack_bf16(e[8 * jp + 2], e[8 * jp + 3]);
                pa.z = pack_bf16(e[8 * jp + 4], e[8 * jp + 5]);
                pa.w = pack_bf16(e[8 * jp + 6], e[8 * jp + 7]);
                const bf16x8_t a = *(bf16x8_t *)&pa;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8_t *)&vb[jp][dt], a, o[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sum += __shfl_xor(sum, 32);
        ATT_T(5)
        // V^T is the FIRST operand: o[dt][r] = O^T[feature 32 dt + 8 (r >> 2) + 4 hh + (r & 3)][query q0 + nl] — the lane that holds a
        // query's row sum also holds its outputs, four consecutive features per register quad: 8-byte stores, no shuffles
        const float inv = 1.f / sum;
        const int q = q0 + nl;
        if (q < T) {
            uint16_t *dst = out + ((int64_t)b * T + q) * H * 64 + (int64_t)h * 64 + 4 * hh;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(uint2 *)(dst + 32 * dt + 8 * j) = make_uint2(pack_bf16(o[dt][4 * j] * inv, o[dt][4 * j + 1] * inv),
                                                                   pack_bf16(o[dt][4 * j + 2] * inv, o[dt][4 * j + 3] * inv));
        }
        ATT_T(6)
    }
#ifdef BSC_ATT_PROFILE
    if (blockIdx.x == 3 && lane == 0 && (wave == 0 || wave == 6))
        printf("att32 wave %d: wait-barrier %lld stage %lld issue %lld qk %lld softmax %lld pv %lld store %lld (clock64 ticks)\n", wave,
               tph[0], tph[1], tph[2], tph[3], tph[4], tph[5], tph[6]);
#endif
#undef ATT_T
    // the last workgroup to leave re-arms the counters for the next launch on this stream
    if (work && tid == 0) {
        __threadfence();
        if (atomicAdd(&work[1], 1) == (int)gridDim.x - 1) {
            work[0] = 0;
            work[1] = 0;
            __threadfence();
        }
    }
}

extern "C" bsc_status bsc_enc_attention_dyn(const void *qkv_dev, int32_t B, int32_t T, int32_t heads, int32_t head_dim,
                                            void *out_dev, int32_t *work2_dev, void *hip_stream)
{
    if (!qkv_dev || !out_dev || B < 1 || T < 1 || heads < 1) return BSC_E_INVALID;
    if (head_dim != 64 || T > 288) {
        bsc_set_error("bsc_enc_attention: head_dim 64 and T <= 288 only (got %d, %d)", head_dim, T);
        return BSC_E_INVALID;
    }
    hipStream_t s = (hipStream_t)hip_stream;
    static int n_cu = 0;                // one workgroup per CU (its LDS footprint allows no second one)
    if (!n_cu) {
        int dev = 0;
        BSC_HIP(hipGetDevice(&dev));
        BSC_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int64_t items = (int64_t)B * heads;
    const dim3 grid((unsigned)(items < n_cu ? items : n_cu));
    static const bool tiles32 = !(getenv("BSC_ATT_TILE") && atoi(getenv("BSC_ATT_TILE")) == 16);
    if (T <= 224 && tiles32)
        hipLaunchKernelGGL((k_attention32<7, 7>), grid, dim3(64 * 7), 0, s, (const uint16_t *)qkv_dev, T, heads, (int)items,
                           (uint16_t *)out_dev, (int *)work2_dev);
    else if (T <= 224)
        hipLaunchKernelGGL((k_attention<14, 7>), grid, dim3(64 * 7), 0, s, (const uint16_t *)qkv_dev, T, heads, (int)items,
                           (uint16_t *)out_dev, (int *)work2_dev);
    else
        hipLaunchKernelGGL((k_attention<18, 8>), grid, dim3(64 * 8), 0, s, (const uint16_t *)qkv_dev, T, heads, (int)items,
                           (uint16_t *)out_dev, (int *)work2_dev);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}

extern "C" bsc_status bsc_enc_attention(const void *qkv_dev, int32_t B, int32_t T, int32_t heads, int32_t head_dim,
                                        void *out_dev, void *hip_stream)
{
    return bsc_enc_attention_dyn(qkv_dev, B, T, heads, head_dim, out_dev, nullptr, hip_stream);
}
