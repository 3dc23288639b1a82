// Token-boundary kernels of PixArtMS.forward (PixArtMS.py:165-211): everything that converts between the latent
// image layout (B,C,Hl,Wl) and the token layout (B,N,D), plus the caption row gather.
//   patch_embed_fwd : x_tok = Conv2d(4->D, k=2, s=2)(x) + bias + pos_embed        (PixArtMS.py:38-44,184) fp32, K=16 (VALU)
//   patch_embed_bwd : dW[d][16], db[d] from the residual-stream gradient
//   unpatchify_fwd  : (B,N,p*p*C) -> (B,C,Hl,Wl)  'nhwpqc->nchpwq'                  (PixArtMS.py:236-248)
//   patchify_bwd    : inverse permutation of the output gradient, emitted as bf16 GEMM operand
//   gather_rows     : packed caption rows (masked_select, PixArtMS.py:196-204) with the train-time token drop
//                     (CaptionEmbedder.token_drop, PixArt_blocks.py:389-398), fp32 -> bf16
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

constexpr int PE_TOK = 32;  // tokens per block in patch_embed_fwd

// thread = 4 consecutive output channels; weights for them (4x16) live in registers; patches staged in LDS
__global__ __launch_bounds__(512) void patch_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ pos, float* __restrict__ out,
                                                              int B, int C, int Hl, int Wl, int D) {
  __shared__ float patch[PE_TOK][16];
  const int h = Hl / 2, wd = Wl / 2, N = h * wd, K = C * 4;
  const long tok0 = (long)blockIdx.x * PE_TOK, T = (long)B * N;
  for (int i = threadIdx.x; i < PE_TOK * K; i += blockDim.x) {
    const int t = i / K, k = i - t * K;
    const long tok = tok0 + t;
    float v = 0.f;
    if (tok < T) {
      const int b = tok / N, n = tok - (long)b * N, r = n / wd, cc = n - r * wd;
      const int ch = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
      v = x[(((long)b * C + ch) * Hl + 2 * r + dy) * Wl + 2 * cc + dx];
    }
    patch[t][k] = v;
  }
  __syncthreads();
  const int d0 = threadIdx.x * 4;
  if (d0 >= D) return;
  float wr[4][16];
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int k = 0; k < 16; k++) wr[e][k] = w[(d0 + e) * 16 + k];
  const float4 bb = *reinterpret_cast<const float4*>(bias + d0);
  for (int t = 0; t < PE_TOK; t++) {
    const long tok = tok0 + t;
    if (tok >= T) break;
    const int n = tok % N;
    float a[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const float pv = patch[t][k];
#pragma unroll
      for (int e = 0; e < 4; e++) a[e] += wr[e][k] * pv;
    }
    const float4 pp = *reinterpret_cast<const float4*>(pos + (long)n * D + d0);
    *reinterpret_cast<float4*>(out + tok * D + d0) = make_float4(a[0] + pp.x, a[1] + pp.y, a[2] + pp.z, a[3] + pp.w);
  }
}

constexpr int PB_TOK = 256;  // tokens per block in patch_embed_bwd
__global__ __launch_bounds__(512) void patch_embed_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dtok, float* __restrict__ dw,
                                                              float* __restrict__ dbias, int B, int C, int Hl, int Wl, int D) {
  __shared__ float patch[32][16];
  const int h = Hl / 2, wd = Wl / 2, N = h * wd, K = C * 4;
  const long T = (long)B * N, tokb = (long)blockIdx.x * PB_TOK;
  const int d0 = threadIdx.x * 4;
  float aw[4][16], ab[4] = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; e++)
#pragma unroll
    for (int k = 0; k < 16; k++) aw[e][k] = 0.f;
  for (int t0 = 0; t0 < PB_TOK; t0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * K; i += blockDim.x) {
      const int t = i / K, k = i - t * K;
      const long tok = tokb + t0 + t;
      float v = 0.f;
      if (tok < T) {
        const int b = tok / N, n = tok - (long)b * N, r = n / wd, cc = n - r * wd;
        const int ch = k >> 2, dy = (k >> 1) & 1, dx = k & 1;
        v = x[(((long)b * C + ch) * Hl + 2 * r + dy) * Wl + 2 * cc + dx];
      }
      patch[t][k] = v;
    }
    __syncthreads();
    if (d0 < D) {
      // eight tokens' gradient rows are requested before the first is used: one row per trip was a chain of 256 dependent HBM latencies per
      // thread (460 us for a 302 MB read); rows beyond T read as zero instead of ending the loop
      for (int t8 = 0; t8 < 32; t8 += 8) {
        float4 g8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const long tok = tokb + t0 + t8 + u;
          g8[u] = tok < T ? *reinterpret_cast<const float4*>(dtok + tok * D + d0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const float gv[4] = {g8[u].x, g8[u].y, g8[u].z, g8[u].w};
#pragma unroll
          for (int e = 0; e < 4; e++) ab[e] += gv[e];
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const float pv = patch[t8 + u][k];
#pragma unroll
            for (int e = 0; e < 4; e++) aw[e][k] += gv[e] * pv;
          }
        }
      }
    }
  }
  if (d0 < D) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      atomicAdd(dbias + d0 + e, ab[e]);
#pragma unroll
      for (int k = 0; k < 16; k++) atomicAdd(dw + (d0 + e) * 16 + k, aw[e][k]);
    }
  }
}

// lin [B][N][p*p*Co] (p=2) -> img [B][Co][2h][2w]
__global__ void unpatchify_kernel(const float* __restrict__ lin, float* __restrict__ img, int B, int h, int w, int Co) {
  const long idx = blockIdx.x * 256L + threadIdx.x, total = (long)B * Co * 4 * h * w;
  if (idx >= total) return;
  const int W2 = 2 * w, H2 = 2 * h;
  const int xx = idx % W2;
  long t = idx / W2;
  const int yy = t % H2; t /= H2;
  const int c = t % Co, b = t / Co;
  const int r = yy >> 1, p = yy & 1, cc = xx >> 1, q = xx & 1;
  img[idx] = lin[(((long)b * h + r) * w + cc) * (4 * Co) + (p * 2 + q) * Co + c];
}
// dimg [B][Co][2h][2w] -> dlin bf16 [B][N][4*Co]
__global__ void patchify_bwd_kernel(const float* __restrict__ dimg, bf16_t* __restrict__ dlin, int B, int h, int w, int Co) {
  const long idx = blockIdx.x * 256L + threadIdx.x, total = (long)B * h * w * 4 * Co;
  if (idx >= total) return;
  const int j = idx % (4 * Co);
  long t = idx / (4 * Co);
  const int cc = t % w; t /= w;
  const int r = t % h, b = t / h;
  const int c = j % Co, pq = j / Co, p = pq >> 1, q = pq & 1;
  dlin[idx] = (bf16_t)dimg[(((long)b * Co + c) * (2 * h) + 2 * r + p) * (2 * w) + 2 * cc + q];
}

// out[i][:] = bf16( drop[b(i)] ? alt[l(i)][:] : src[row_idx[i]][:] ),  row_idx[i] = b*L + l
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const float* __restrict__ alt, const int* __restrict__ row_idx,
                                                          const int* __restrict__ drop, bf16_t* __restrict__ out, int rows, int L, int Cw) {
  const int i = blockIdx.x;
  if (i >= rows) return;
  const int ri = row_idx[i], b = ri / L, l = ri - b * L;
  const float* s = (drop && drop[b] && alt) ? alt + (long)l * Cw : src + (long)ri * Cw;
  for (int c = threadIdx.x * 4; c < Cw; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(s + c);
    *reinterpret_cast<uint2*>(out + (long)i * Cw + c) = pack_bf16x4(v.x, v.y, v.z, v.w);
  }
}
}  // namespace

extern "C" int pxa_patch_embed_fwd(const float* x, const float* w, const float* bias, const float* pos, float* out,
                                   int B, int C, int Hl, int Wl, int D, hipStream_t stream) {
  PXA_CHECK(x && w && bias && pos && out, "pxa_patch_embed_fwd: null pointer");
  PXA_CHECK(C == 4 && Hl % 2 == 0 && Wl % 2 == 0 && D % 4 == 0 && D <= 2048, "pxa_patch_embed_fwd: needs C=4, patch 2, D<=2048");
  const long T = (long)B * (Hl / 2) * (Wl / 2);
  hipLaunchKernelGGL(patch_embed_fwd_kernel, dim3((T + PE_TOK - 1) / PE_TOK), dim3(512), 0, stream, x, w, bias, pos, out, B, C, Hl, Wl, D);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_patch_embed_bwd(const float* x, const float* dtok, float* dw, float* dbias, int B, int C, int Hl, int Wl, int D, hipStream_t stream) {
  PXA_CHECK(x && dtok && dw && dbias, "pxa_patch_embed_bwd: null pointer");
  PXA_CHECK(C == 4 && Hl % 2 == 0 && Wl % 2 == 0 && D % 4 == 0 && D <= 2048, "pxa_patch_embed_bwd: needs C=4, patch 2, D<=2048");
  const long T = (long)B * (Hl / 2) * (Wl / 2);
  hipLaunchKernelGGL(patch_embed_bwd_kernel, dim3((T + PB_TOK - 1) / PB_TOK), dim3(512), 0, stream, x, dtok, dw, dbias, B, C, Hl, Wl, D);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_unpatchify_fwd(const float* lin, float* img, int B, int h, int w, int Co, hipStream_t stream) {
  PXA_CHECK(lin && img && B > 0 && h > 0 && w > 0 && Co > 0, "pxa_unpatchify_fwd: bad args");
  const long total = (long)B * Co * 4 * h * w;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, lin, img, B, h, w, Co);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_patchify_bwd(const float* dimg, void* dlin_bf16, int B, int h, int w, int Co, hipStream_t stream) {
  PXA_CHECK(dimg && dlin_bf16 && B > 0 && h > 0 && w > 0 && Co > 0, "pxa_patchify_bwd: bad args");
  const long total = (long)B * Co * 4 * h * w;
  hipLaunchKernelGGL(patchify_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, dimg, (bf16_t*)dlin_bf16, B, h, w, Co);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_gather_rows_bf16(const float* src, const float* alt, const int* row_idx, const int* drop, void* out_bf16,
                                    int rows, int L, int Cw, hipStream_t stream) {
  PXA_CHECK(src && row_idx && out_bf16 && rows > 0 && L > 0 && Cw % 4 == 0, "pxa_gather_rows_bf16: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, stream, src, alt, row_idx, drop, (bf16_t*)out_bf16, rows, L, Cw);
  PXA_LAUNCH_CHECK();
  return 0;
}
