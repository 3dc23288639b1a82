// KV token compression, 'conv' sampling (AttentionKVCompress.downsample_2d, PixArt_blocks.py:84-89,97-121):
// depthwise Conv2d(C, C, groups=C, kernel=sr, stride=sr) over the (H, W) token grid, then affine LayerNorm(C, eps=1e-5).
// HBM-bound: one 32-lane half-wave produces one compressed token (C/128 float4 per lane), reading the sr*sr source
// tokens straight from the strided k / v slice of the qkv GEMM output and writing bf16 for the attention kernel.
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

template <int NV>
__global__ __launch_bounds__(256) void kv_compress_fwd_kernel(const bf16_t* __restrict__ in, long in_bs, long in_ts, const float* __restrict__ cw,
                                                              const float* __restrict__ cb, const float* __restrict__ lw, const float* __restrict__ lb,
                                                              bf16_t* __restrict__ out, int B, int H, int W, int C, int sr, float eps) {
  const int hl = threadIdx.x & 31;
  const int nH = H / sr, nW = W / sr;
  const long row = blockIdx.x * 8L + (threadIdx.x >> 5), total = (long)B * nH * nW;
  if (row >= total) return;
  const int b = row / (nH * nW), rem = row - (long)b * nH * nW, r = rem / nW, c = rem - r * nW;
  float4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) v[j] = *reinterpret_cast<const float4*>(cb + (hl + 32 * j) * 4);
  for (int i = 0; i < sr; i++)
    for (int jj = 0; jj < sr; jj++) {
      const bf16_t* src = in + b * in_bs + ((long)(r * sr + i) * W + (c * sr + jj)) * in_ts;
      const int tap = i * sr + jj, taps = sr * sr;
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int ch = (hl + 32 * j) * 4;
        const uint2 xx = *reinterpret_cast<const uint2*>(src + ch);
        float x0, x1, x2, x3;
        unpack_bf16x2(xx.x, x0, x1); unpack_bf16x2(xx.y, x2, x3);
        v[j].x += cw[(ch + 0) * taps + tap] * x0; v[j].y += cw[(ch + 1) * taps + tap] * x1;
        v[j].z += cw[(ch + 2) * taps + tap] * x2; v[j].w += cw[(ch + 3) * taps + tap] * x3;
      }
    }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = half_wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    float a = v[j].x - mean, bb = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
    q += (a * a + bb * bb) + (cc * cc + d * d);
  }
  const float rstd = rsqrtf(half_wave_sum(q) / C + eps);
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int ch = (hl + 32 * j) * 4;
    const float4 w4 = *reinterpret_cast<const float4*>(lw + ch), b4 = *reinterpret_cast<const float4*>(lb + ch);
    *reinterpret_cast<uint2*>(out + row * C + ch) = pack_bf16x4((v[j].x - mean) * rstd * w4.x + b4.x, (v[j].y - mean) * rstd * w4.y + b4.y,
                                                                (v[j].z - mean) * rstd * w4.z + b4.z, (v[j].w - mean) * rstd * w4.w + b4.w);
  }
}
}  // namespace

extern "C" int pxa_kv_compress_fwd(const void* in_bf16, long in_bs, long in_ts, const float* conv_w, const float* conv_b,
                                   const float* ln_w, const float* ln_b, void* out_bf16, int B, int H, int W, int C, int sr, float eps,
                                   hipStream_t stream) {
  PXA_CHECK(in_bf16 && conv_w && conv_b && ln_w && ln_b && out_bf16, "pxa_kv_compress_fwd: null pointer");
  PXA_CHECK(B > 0 && sr >= 1 && H >= sr && W >= sr && in_ts % 4 == 0 && in_bs % 4 == 0, "pxa_kv_compress_fwd: bad shape");
  const long total = (long)B * (H / sr) * (W / sr);
  dim3 grid((total + 7) / 8);
  switch (C / 128) {
    case 9:
      PXA_CHECK(C == 1152, "pxa_kv_compress_fwd: C must be 1152");
      hipLaunchKernelGGL(kv_compress_fwd_kernel<9>, grid, dim3(256), 0, stream, (const bf16_t*)in_bf16, in_bs, in_ts, conv_w, conv_b, ln_w, ln_b,
                         (bf16_t*)out_bf16, B, H, W, C, sr, eps);
      break;
    default:
      pxa_set_error("pxa_kv_compress_fwd: unsupported C=%d", C);
      return -1;
  }
  PXA_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the 'conv' compression and the token-pick compressions.
namespace {
using namespace pxa;

constexpr int KB_ROWS = 64;   // compressed tokens per block in the backward kernel

// One half-wave per compressed token: recompute conv output + LN statistics from the sr*sr source tokens, LayerNorm backward,
// scatter the source-token gradient, and accumulate the five parameter gradients in LDS (ds_add_f32), flushed once per block.
template <int NV>
__global__ __launch_bounds__(256) void kv_compress_bwd_kernel(const bf16_t* __restrict__ dyc, const bf16_t* __restrict__ in, long in_bs, long in_ts,
                                                              const float* __restrict__ cw, const float* __restrict__ cb, const float* __restrict__ lw,
                                                              bf16_t* __restrict__ din, long din_bs, long din_ts, float* __restrict__ d_cw,
                                                              float* __restrict__ d_cb, float* __restrict__ d_lw, float* __restrict__ d_lb,
                                                              int B, int H, int W, int C, int sr, float eps) {
  extern __shared__ float acc[];                          // [ (3 + taps) ][C] : d_lb, d_lw, d_cb, d_cw[tap]
  const int taps = sr * sr;
  for (int i = threadIdx.x; i < (3 + taps) * C; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int nH = H / sr, nW = W / sr;
  const long total = (long)B * nH * nW;
  for (int it = 0; it < KB_ROWS / 8; it++) {
    const long row = (long)blockIdx.x * KB_ROWS + it * 8 + hw;
    if (row >= total) break;
    const int b = row / (nH * nW), rem = row - (long)b * nH * nW, r = rem / nW, c = rem - r * nW;
    float4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = *reinterpret_cast<const float4*>(cb + (hl + 32 * j) * 4);
    for (int tap = 0; tap < taps; tap++) {
      const bf16_t* src = in + b * in_bs + ((long)(r * sr + tap / sr) * W + (c * sr + tap % sr)) * in_ts;
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int ch = (hl + 32 * j) * 4;
        const uint2 xx = *reinterpret_cast<const uint2*>(src + ch);
        float x0, x1, x2, x3;
        unpack_bf16x2(xx.x, x0, x1); unpack_bf16x2(xx.y, x2, x3);
        v[j].x += cw[(ch + 0) * taps + tap] * x0; v[j].y += cw[(ch + 1) * taps + tap] * x1;
        v[j].z += cw[(ch + 2) * taps + tap] * x2; v[j].w += cw[(ch + 3) * taps + tap] * x3;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = half_wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      float a = v[j].x - mean, bb = v[j].y - mean, cc = v[j].z - mean, d = v[j].w - mean;
      q += (a * a + bb * bb) + (cc * cc + d * d);
    }
    const float rstd = rsqrtf(half_wave_sum(q) / C + eps);
    float4 g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int ch = (hl + 32 * j) * 4;
      const uint2 dd = *reinterpret_cast<const uint2*>(dyc + row * C + ch);
      float d0, d1, d2, d3;
      unpack_bf16x2(dd.x, d0, d1); unpack_bf16x2(dd.y, d2, d3);
      const float4 w4 = *reinterpret_cast<const float4*>(lw + ch);
      v[j] = make_float4((v[j].x - mean) * rstd, (v[j].y - mean) * rstd, (v[j].z - mean) * rstd, (v[j].w - mean) * rstd);   // xhat
      atomicAdd(&acc[0 * C + ch + 0], d0); atomicAdd(&acc[0 * C + ch + 1], d1); atomicAdd(&acc[0 * C + ch + 2], d2); atomicAdd(&acc[0 * C + ch + 3], d3);
      atomicAdd(&acc[1 * C + ch + 0], d0 * v[j].x); atomicAdd(&acc[1 * C + ch + 1], d1 * v[j].y);
      atomicAdd(&acc[1 * C + ch + 2], d2 * v[j].z); atomicAdd(&acc[1 * C + ch + 3], d3 * v[j].w);
      g[j] = make_float4(d0 * w4.x, d1 * w4.y, d2 * w4.z, d3 * w4.w);
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * v[j].x + g[j].y * v[j].y) + (g[j].z * v[j].z + g[j].w * v[j].w);
    }
    const float c1 = half_wave_sum(s1) / C, c2 = half_wave_sum(s2) / C;
#pragma unroll
    for (int j = 0; j < NV; j++) {   // g <- gradient of the conv output
      const int ch = (hl + 32 * j) * 4;
      g[j] = make_float4(rstd * (g[j].x - c1 - v[j].x * c2), rstd * (g[j].y - c1 - v[j].y * c2),
                         rstd * (g[j].z - c1 - v[j].z * c2), rstd * (g[j].w - c1 - v[j].w * c2));
      atomicAdd(&acc[2 * C + ch + 0], g[j].x); atomicAdd(&acc[2 * C + ch + 1], g[j].y); atomicAdd(&acc[2 * C + ch + 2], g[j].z); atomicAdd(&acc[2 * C + ch + 3], g[j].w);
    }
    for (int tap = 0; tap < taps; tap++) {
      const long tok = (long)(r * sr + tap / sr) * W + (c * sr + tap % sr);
      const bf16_t* src = in + b * in_bs + tok * in_ts;
      bf16_t* dst = din + b * din_bs + tok * din_ts;
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int ch = (hl + 32 * j) * 4;
        const uint2 xx = *reinterpret_cast<const uint2*>(src + ch);
        float x0, x1, x2, x3;
        unpack_bf16x2(xx.x, x0, x1); unpack_bf16x2(xx.y, x2, x3);
        float* aw = &acc[(3 + tap) * C + ch];
        atomicAdd(aw + 0, g[j].x * x0); atomicAdd(aw + 1, g[j].y * x1); atomicAdd(aw + 2, g[j].z * x2); atomicAdd(aw + 3, g[j].w * x3);
        *reinterpret_cast<uint2*>(dst + ch) = pack_bf16x4(g[j].x * cw[(ch + 0) * taps + tap], g[j].y * cw[(ch + 1) * taps + tap],
                                                          g[j].z * cw[(ch + 2) * taps + tap], g[j].w * cw[(ch + 3) * taps + tap]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    atomicAdd(d_lb + i, acc[i]);
    atomicAdd(d_lw + i, acc[C + i]);
    atomicAdd(d_cb + i, acc[2 * C + i]);
    for (int tap = 0; tap < taps; tap++) atomicAdd(d_cw + i * taps + tap, acc[(3 + tap) * C + i]);
  }
}

// 'uniform' / 'ave' (nearest) compression = pick token (r*sr, c*sr); FWD copies rows, BWD scatters them back (others zeroed by the caller)
template <bool FWD>
__global__ __launch_bounds__(256) void kv_pick_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long full_bs, long full_ts,
                                                      int B, int H, int W, int C, int sr) {
  const int nH = H / sr, nW = W / sr;   // int(H / sr) as in the reference
  const long row = blockIdx.x, total = (long)B * nH * nW;
  if (row >= total) return;
  const int b = row / (nH * nW), rem = row - (long)b * nH * nW, r = rem / nW, c = rem - r * nW;
  const long full_off = b * full_bs + ((long)(r * sr) * W + c * sr) * full_ts;
  for (int ch = threadIdx.x * 8; ch < C; ch += 2048) {
    if (FWD) *reinterpret_cast<uint4*>(dst + row * C + ch) = *reinterpret_cast<const uint4*>(src + full_off + ch);
    else *reinterpret_cast<uint4*>(dst + full_off + ch) = *reinterpret_cast<const uint4*>(src + row * C + ch);
  }
}
}  // namespace

extern "C" int pxa_kv_compress_bwd(const void* dyc_bf16, const void* in_bf16, long in_bs, long in_ts, const float* conv_w, const float* conv_b,
                                   const float* ln_w, void* din_bf16, long din_bs, long din_ts, float* d_conv_w, float* d_conv_b, float* d_ln_w,
                                   float* d_ln_b, int B, int H, int W, int C, int sr, float eps, hipStream_t stream) {
  PXA_CHECK(dyc_bf16 && in_bf16 && conv_w && conv_b && ln_w && din_bf16 && d_conv_w && d_conv_b && d_ln_w && d_ln_b, "pxa_kv_compress_bwd: null pointer");
  PXA_CHECK(C == 1152 && sr >= 1 && sr <= 4 && H >= sr && W >= sr, "pxa_kv_compress_bwd: unsupported shape (C must be 1152, sr <= 4)");
  const long total = (long)B * (H / sr) * (W / sr);
  const int lds = (3 + sr * sr) * C * 4;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kv_compress_bwd_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 19 * 1152 * 4);
    if (e != hipSuccess) { pxa_set_error("hipFuncSetAttribute(kv_compress_bwd): %s", hipGetErrorString(e)); return -3; }
    attr = true;
  }
  hipLaunchKernelGGL(kv_compress_bwd_kernel<9>, dim3((total + KB_ROWS - 1) / KB_ROWS), dim3(256), lds, stream, (const bf16_t*)dyc_bf16, (const bf16_t*)in_bf16,
                     in_bs, in_ts, conv_w, conv_b, ln_w, (bf16_t*)din_bf16, din_bs, din_ts, d_conv_w, d_conv_b, d_ln_w, d_ln_b, B, H, W, C, sr, eps);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_kv_pick(int backward, const void* src_bf16, void* dst_bf16, long full_bs, long full_ts, int B, int H, int W, int C, int sr,
                           hipStream_t stream) {
  PXA_CHECK(src_bf16 && dst_bf16 && C % 8 == 0 && sr >= 1 && H >= sr && W >= sr && full_ts % 8 == 0 && full_bs % 8 == 0, "pxa_kv_pick: bad args");
  const long total = (long)B * (H / sr) * (W / sr);
  if (backward) hipLaunchKernelGGL(kv_pick_kernel<false>, dim3(total), dim3(256), 0, stream, (const bf16_t*)src_bf16, (bf16_t*)dst_bf16, full_bs, full_ts, B, H, W, C, sr);
  else hipLaunchKernelGGL(kv_pick_kernel<true>, dim3(total), dim3(256), 0, stream, (const bf16_t*)src_bf16, (bf16_t*)dst_bf16, full_bs, full_ts, B, H, W, C, sr);
  PXA_LAUNCH_CHECK();
  return 0;
}
