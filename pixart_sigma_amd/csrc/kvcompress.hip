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
