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

__global__ __launch_bounds__(TPB) void k_preprocess_patches(const uint8_t *__restrict__ rgb, int B, int H, int W, int C,
                                                            int S, int p, uint16_t *__restrict__ out, float m0, float m1,
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
    const int g = S / p, gy = y / p, gx = x / p, py = y - gy * p, px = x - gx * p;
    uint16_t *dst = out + ((int64_t)b * g * g + (int64_t)gy * g + gx) * (3 * p * p) + py * p + px;
    dst[0] = f2bf((a0 * (1.f / 255.f) - m0) / s0);
    dst[p * p] = f2bf((a1 * (1.f / 255.f) - m1) / s1);
    dst[2 * p * p] = f2bf((a2 * (1.f / 255.f) - m2) / s2);
}

extern "C" bsc_status bsc_enc_preprocess_patches(const void *rgb_dev, int32_t B, int32_t H, int32_t W, int32_t C,
                                                 int32_t S, int32_t patch, void *out_dev, const float *mean3_host,
                                                 const float *std3_host, void *hip_stream)
{
    if (!rgb_dev || !out_dev || !mean3_host || !std3_host || B < 1 || C < 3 || patch < 1 || S % patch != 0) {
        bsc_set_error("bsc_enc_preprocess_patches: invalid argument");
        return BSC_E_INVALID;
    }
    if ((float)H / S > (PP_TAPS - 1) / 2.0f || (float)W / S > (PP_TAPS - 1) / 2.0f) {
        bsc_set_error("bsc_enc_preprocess_patches: down-scale factor above %g not supported", (PP_TAPS - 1) / 2.0);
        return BSC_E_INVALID;
    }
    const int64_t n = (int64_t)B * S * S;
    hipLaunchKernelGGL(k_preprocess_patches, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)hip_stream,
                       (const uint8_t *)rgb_dev, B, H, W, C, S, patch, (uint16_t *)out_dev, mean3_host[0], mean3_host[1],
                       mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
    BSC_HIP(hipGetLastError());
    return BSC_OK;
}
