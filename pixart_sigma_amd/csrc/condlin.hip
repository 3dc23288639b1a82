// fp32 linear layers of the conditioning path: t_embedder.mlp (256 -> D -> D), t_block (D -> 6 D), csize / ar embedders - a handful of (batch x 256 / 1152 /
// 6912) products per step whose outputs are the adaLN shift / scale / gate vectors of EVERY token of EVERY block, so they stay fp32 end to end
// (reference: TimestepEmbedder / SizeEmbedder / t_block, PixArt_blocks.py:267-344, PixArtMS.py:134-137,193).  Rounds 1-4 left them to torch.nn.Linear,
// i.e. eight vendor-library GEMM launches per step in a product path that claims none: these three kernels replace them.
// Rows = samples (M <= a few dozen), so nothing here is a matrix-pipe problem: each kernel streams the weight once, coalesced, at HBM / L2 speed.
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;
constexpr int MC = 16;      // rows per pass (batch 16 = one pass)

// y[m][n] = b[n] + sum_k x[m][k] w[n][k]: one wave per output column, lanes stride k four at a time
__global__ __launch_bounds__(256) void condlin_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                                          int M, int N, int K) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6), m0 = blockIdx.y * MC;
  if (n >= N) return;
  float acc[MC];
#pragma unroll
  for (int i = 0; i < MC; i++) acc[i] = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)n * K + k);
#pragma unroll
    for (int i = 0; i < MC; i++) {
      if (m0 + i < M) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)(m0 + i) * K + k);
        acc[i] += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
      }
    }
  }
  const float bias = b ? b[n] : 0.f;
#pragma unroll
  for (int i = 0; i < MC; i++) {
    const float s = wave_sum(acc[i]);
    if (lane == 0 && m0 + i < M) y[(size_t)(m0 + i) * N + n] = s + bias;
  }
}

// dx[m][k] += sum_{n in slice} dy[m][n] w[n][k]: a thread owns 4 consecutive k (coalesced weight rows), a block one slice of 128 n; dx is caller-zeroed
__global__ __launch_bounds__(256) void condlin_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int M, int N, int K) {
  const int k = (blockIdx.x * 256 + threadIdx.x) * 4, n0 = blockIdx.y * 128, n1 = min(N, n0 + 128), m0 = blockIdx.z * MC;
  if (k >= K) return;
  float4 acc[MC];
#pragma unroll
  for (int i = 0; i < MC; i++) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = n0; n < n1; n++) {
    const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)n * K + k);
#pragma unroll
    for (int i = 0; i < MC; i++) {
      if (m0 + i < M) {
        const float g = dy[(size_t)(m0 + i) * N + n];            // wave-uniform address: a scalar load
        acc[i].x += g * wv.x; acc[i].y += g * wv.y; acc[i].z += g * wv.z; acc[i].w += g * wv.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MC; i++) {
    if (m0 + i < M) {
      float* d = dx + (size_t)(m0 + i) * K + k;
      atomicAdd(d, acc[i].x); atomicAdd(d + 1, acc[i].y); atomicAdd(d + 2, acc[i].z); atomicAdd(d + 3, acc[i].w);
    }
  }
}

// dw[n][k] = sum_m dy[m][n] x[m][k] (a thread owns 4 consecutive k of one n); db[n] = sum_m dy[m][n] (the threads with k == 0)
__global__ __launch_bounds__(256) void condlin_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
                                                         int M, int N, int K) {
  const int k = (blockIdx.x * 256 + threadIdx.x) * 4, n = blockIdx.y;
  if (k >= K) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  for (int m = 0; m < M; m++) {
    const float g = dy[(size_t)m * N + n];
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * K + k);
    acc.x += g * xv.x; acc.y += g * xv.y; acc.z += g * xv.z; acc.w += g * xv.w;
    sb += g;
  }
  *reinterpret_cast<float4*>(dw + (size_t)n * K + k) = acc;
  if (db && k == 0) db[n] = sb;
}
}  // namespace

extern "C" int pxa_linear_f32_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, hipStream_t stream) {
  PXA_CHECK(x && w && y && M > 0 && N > 0 && K > 0 && K % 4 == 0, "pxa_linear_f32_fwd: null pointer / bad shape (K must be a multiple of 4)");
  hipLaunchKernelGGL(condlin_fwd_kernel, dim3((N + 3) / 4, (M + MC - 1) / MC), dim3(256), 0, stream, x, w, bias, y, M, N, K);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_linear_f32_bwd(const float* dy, const float* x, const float* w, float* dx_zeroed, float* dw, float* db, int M, int N, int K, hipStream_t stream) {
  PXA_CHECK(dy && x && w && M > 0 && N > 0 && K > 0 && K % 4 == 0, "pxa_linear_f32_bwd: null pointer / bad shape (K must be a multiple of 4)");
  if (dx_zeroed) {
    hipLaunchKernelGGL(condlin_dx_kernel, dim3((K / 4 + 255) / 256, (N + 127) / 128, (M + MC - 1) / MC), dim3(256), 0, stream, dy, w, dx_zeroed, M, N, K);
    PXA_LAUNCH_CHECK();
  }
  if (dw) {
    hipLaunchKernelGGL(condlin_dw_kernel, dim3((K / 4 + 255) / 256, N), dim3(256), 0, stream, dy, x, dw, db, M, N, K);
    PXA_LAUNCH_CHECK();
  }
  return 0;
}
