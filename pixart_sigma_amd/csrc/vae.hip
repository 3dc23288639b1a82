// SDXL-VAE / SD-VAE (AutoencoderKL) conv stack: the kernels around the GEMMs.
// Call sites in the reference: vae.encode(...).latent_dist (train_scripts/train.py:149-153), vae.decode(...).sample
// (scripts/inference.py:136, train_scripts/train.py:88).  The network itself lives in an un-vendored dependency (diffusers
// AutoencoderKL, unpinned: requirements.txt:2); the architecture is restated from its public config in oracle/vae_ref.py.
//
// Data layout: activations are bf16 NHWC "pixel grids" (pxa_grid): a pixel is a C-vector, so every convolution is a GEMM over
// pixels.  A 3x3 stride-1 convolution reads a ZERO-PADDED grid (H+2) x (W+2): the three taps of one kernel row are 3*C contiguous
// elements there, so the convolution is pxa_gemm with the segmented-K A operand (k_seg = 3*C, a_seg_stride = (W+2)*C, lda = C,
// one output row per PADDED pixel; the border rows are never read by anybody).  No im2col matrix exists for those layers.
// GroupNorm + SiLU (+ nearest 2x upsampling) are applied by the kernel that writes the padded grid: one read of the producer's
// output, one write.  Stride-2 (encoder downsampling) and the 3/4-channel stem convolutions gather an explicit patch matrix
// (im2col) - their outputs are 4x smaller / their K is 72, so that traffic is small.
// All kernels here are HBM-bound; statistics and arithmetic are fp32 (group sums are combined in fp64).
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

struct Grid {
  bf16_t* ptr; int B, H, W, C, row_pitch; long img_pitch, origin;
  __device__ __forceinline__ bf16_t* at(int b, int y, int x) const { return ptr + ((long)b * img_pitch + (long)y * row_pitch + x + origin) * C; }
};
static inline Grid to_grid(const pxa_grid* g) { return Grid{(bf16_t*)g->ptr, g->B, g->H, g->W, g->C, g->row_pitch, g->img_pitch, g->origin}; }

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  const uint2 a = pack_bf16x4(f[0], f[1], f[2], f[3]), b = pack_bf16x4(f[4], f[5], f[6], f[7]);
  return make_uint4(a.x, a.y, b.x, b.y);
}

// ---- GroupNorm statistics: per (sample, group) sum and sum of squares over H*W*(C/groups) elements.
// A thread owns one 8-channel chunk of a pixel and keeps two partial pairs (channels 0-3 / 4-7: groups are >= 4 channels wide);
// a block folds its partials per group in LDS and adds them to the fp64 accumulators with one atomic per group.
__global__ __launch_bounds__(256) void gn_stats_kernel(Grid x, int cpg, double* __restrict__ ws) {
  __shared__ float red[2 * 256];                       // [group][sum, sumsq], groups <= 256
  const int CV = x.C / 8, PPB = 256 / CV, G = x.C / cpg, b = blockIdx.y;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  for (int i = threadIdx.x; i < 2 * G; i += 256) red[i] = 0.f;
  __syncthreads();
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  for (int y = blockIdx.x; y < x.H; y += gridDim.x) {   // a block walks whole image rows: no index divisions in the loop
    const bf16_t* row = x.at(b, y, 0) + cv * 8;
    for (int xx = pl; xx < x.W; xx += PPB) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(row + (long)xx * x.C), f);
      s0 += (f[0] + f[1]) + (f[2] + f[3]); q0 += (f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3]);
      s1 += (f[4] + f[5]) + (f[6] + f[7]); q1 += (f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]);
    }
  }
  const int g0 = (cv * 8) / cpg, g1 = (cv * 8 + 4) / cpg;
  atomicAdd(&red[2 * g0], s0); atomicAdd(&red[2 * g0 + 1], q0);
  atomicAdd(&red[2 * g1], s1); atomicAdd(&red[2 * g1 + 1], q1);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * G; i += 256) atomicAdd(&ws[(long)b * 2 * G + i], (double)red[i]);
}
__global__ void gn_finalize_kernel(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ rstd, int n_groups, double count, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups) return;
  const double m = ws[2 * i] / count, var = fmax(ws[2 * i + 1] / count - m * m, 0.0);
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// statistics that a convolution's epilogue already produced (pxa_gemm_args.gn_part): part[slot][b][C/4][2] (sum, sum of squares per quad of
// adjacent channels) -> mean / rstd per (b, group)
__global__ void gn_finalize_part_kernel(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd, int B, int C, int cpg,
                                        double count, float eps) {
  const int G = C / cpg, i = blockIdx.x * blockDim.x + threadIdx.x;      // i = b * G + g
  if (i >= B * G) return;
  const int b = i / G, g = i - b * G;
  double s = 0.0, q = 0.0;
  for (int slot = 0; slot < PXA_COLSUM_SLOTS; slot++) {
    const float2* row = reinterpret_cast<const float2*>(part) + ((size_t)slot * B + b) * (C / 4) + g * (cpg / 4);
    for (int c = 0; c < cpg / 4; c++) { s += (double)row[c].x; q += (double)row[c].y; }
  }
  const double m = s / count, var = fmax(q / count - m * m, 0.0);
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// the value a consumer sees: optional GroupNorm (affine), optional SiLU
struct Norm {
  const float* mean; const float* rstd; const float* gamma; const float* beta; int cpg, G, silu;
  int cpg_shift;                                       // log2(cpg) when it is a power of two (every SD / SDXL VAE layer), else -1
  __device__ __forceinline__ void apply(float (&f)[8], int b, int c0) const {
    if (mean) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int g = cpg_shift >= 0 ? (c0 + 4 * h) >> cpg_shift : (c0 + 4 * h) / cpg;
        const float m = mean[b * G + g], r = rstd[b * G + g];
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c0 + 4 * h), be = *reinterpret_cast<const float4*>(beta + c0 + 4 * h);
        f[4 * h] = (f[4 * h] - m) * r * ga.x + be.x; f[4 * h + 1] = (f[4 * h + 1] - m) * r * ga.y + be.y;
        f[4 * h + 2] = (f[4 * h + 2] - m) * r * ga.z + be.z; f[4 * h + 3] = (f[4 * h + 3] - m) * r * ga.w + be.w;
      }
    }
    if (silu) {
#pragma unroll
      for (int e = 0; e < 8; e++) f[e] = pxa_silu(f[e]);
    }
  }
  static __device__ __forceinline__ float pxa_silu(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }   // rcp: 1 ulp
};

// ---- y[b, yo, xo] = act(norm(x[b, yo / up, xo / up])): writes the interior of y (the zero border of a padded grid is the caller's)
// A thread handles one 16-byte chunk column of GA_ROWS consecutive output rows: the loads are issued before any arithmetic, so a
// wave keeps GA_ROWS KiB in flight (one chunk per thread left this kernel latency-bound at ~2.4 TB/s).
#ifndef PXA_GA_ROWS
#define PXA_GA_ROWS 4       // rows per thread (A/B builds: tools/build_variant.py ... -DPXA_GA_ROWS=8)
#endif
#ifndef PXA_GA_NT
#define PXA_GA_NT 2         // 0 = plain, 1 = non-temporal loads of x (read once), 2 = also non-temporal stores.  Two alternating rounds of the 64 x 512px decode
                            // (profiles/r6_09_run.txt): 0: 169.3 / 168.1 ms, 1: 168.7 / 168.1, 2: 167.9 / 167.2 - small, but of one sign: the default since round 6
#endif
constexpr int GA_ROWS = PXA_GA_ROWS;
__global__ __launch_bounds__(256) void gn_apply_kernel(Grid x, Norm nm, int up, Grid y) {
  const int CV = x.C / 8, i = blockIdx.x * 256 + threadIdx.x;   // grid: (chunks of one output row, group of output rows, sample)
  if (i >= y.W * CV) return;
  const int xo = i / CV, cv = i - xo * CV, y0 = blockIdx.y * GA_ROWS, b = blockIdx.z;
  uint4 v[GA_ROWS];
#pragma unroll
  for (int r = 0; r < GA_ROWS; r++) {
    const int yo = min(y0 + r, y.H - 1);
    const uint4* src = reinterpret_cast<const uint4*>(x.at(b, yo >> (up - 1), xo >> (up - 1)) + cv * 8);
    if (PXA_GA_NT) { const nt_u4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(src)); v[r] = make_uint4(t[0], t[1], t[2], t[3]); }
    else v[r] = *src;
  }
#pragma unroll
  for (int r = 0; r < GA_ROWS; r++) {
    if (y0 + r >= y.H) break;
    float f[8];
    unpack_bf16x8(v[r], f);
    nm.apply(f, b, cv * 8);
    const uint4 o4 = pack8(f);
    if (PXA_GA_NT == 2) __builtin_nontemporal_store(nt_u4{o4.x, o4.y, o4.z, o4.w}, reinterpret_cast<nt_u4*>(y.at(b, y0 + r, xo) + cv * 8));
    else *reinterpret_cast<uint4*>(y.at(b, y0 + r, xo) + cv * 8) = o4;
  }
}

// ---- explicit patch matrix: col[(b, yo, xo)][tap * C + c] = act(norm(x[b, yo*stride + ky - pad, xo*stride + kx - pad])) or 0 outside
__global__ __launch_bounds__(256) void im2col3x3_kernel(Grid x, Norm nm, int stride, int pad, int Ho, int Wo, bf16_t* __restrict__ col) {
  const int CV = x.C / 8, i = blockIdx.x * 256 + threadIdx.x;   // grid: (chunks of one output row's patches, output row, sample)
  if (i >= Wo * 9 * CV) return;
  const int t = i / CV, cv = i - t * CV, xo = t / 9, tap = t - xo * 9, yo = blockIdx.y, b = blockIdx.z;
  const int yi = yo * stride + tap / 3 - pad, xi = xo * stride + tap % 3 - pad;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (yi >= 0 && yi < x.H && xi >= 0 && xi < x.W) {
    float f[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x.at(b, yi, xi) + cv * 8), f);
    nm.apply(f, b, cv * 8);
    v = pack8(f);
  }
  // i enumerates the 16-byte chunks of this output row's patches in memory order
  *reinterpret_cast<uint4*>(col + (((long)b * Ho + yo) * Wo * 9 * CV + i) * 8) = v;
}

// ---- 3x3 stride-1 pad-1 convolution of act(norm(x)) down to a FEW output channels (decoder conv_out: 128 -> 3), fp32 NCHW image out.
// As an implicit GEMM this layer is a 256 x 128 tile with 3 live columns in front of a full-size GroupNorm-apply pass and behind a crop / permute:
// 6.65 ms of the 206 ms decode at 0.08 of the HBM roofline (profiles/r6_01_vae_layer_table.txt).  Here it is what it is - one HBM read of the input:
// a workgroup owns an 8 x 32 pixel tile; per 64-channel chunk it stages the (8+2) x (32+2) halo of act(norm(x)), rounded to the operand type exactly as the
// padded grid of the GEMM path holds it, in LDS (pixel stride 144 B: the 16 lanes of a ds_read_b128 group land on 16 distinct 4-bank groups), and every
// thread accumulates its pixel's CO outputs with packed dot products (v_dot2_f32_*; the weights are wave-uniform -> scalar loads through the constant cache).
// Outside the image the staged value is ZERO (the convolution pads the activated tensor), not act(norm(0)).
constexpr int SO_TH = 8, SO_TW = 32, SO_CC = 64, SO_PIX = (SO_CC + 8) * 2;      // 144 bytes per staged pixel
template <int CO>
__global__ __launch_bounds__(256) void conv3x3_small_out_kernel(Grid x, Norm nm, const uint32_t* __restrict__ wts, const float* __restrict__ bias,
                                                                float* __restrict__ img) {
  __shared__ __attribute__((aligned(16))) char tile[(SO_TH + 2) * (SO_TW + 2) * SO_PIX];
  const int tid = threadIdx.x, px = tid % SO_TW, py = tid / SO_TW;
  const int x0 = blockIdx.x * SO_TW, y0 = blockIdx.y * SO_TH, b = blockIdx.z;
  const int C2 = x.C / 2;                              // weights: [tap][co][C / 2] packed channel pairs
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; co++) acc[co] = 0.f;
  for (int c0 = 0; c0 < x.C; c0 += SO_CC) {
    if (c0) __syncthreads();                           // everybody is done with the previous chunk's tile
    constexpr int PIECES = (SO_TH + 2) * (SO_TW + 2) * (SO_CC / 8);
    for (int pi = tid; pi < PIECES; pi += 256) {
      const int pix = pi / (SO_CC / 8), ch = pi % (SO_CC / 8), ty = pix / (SO_TW + 2), tx = pix % (SO_TW + 2);
      const int yy = y0 + ty - 1, xx = x0 + tx - 1;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (yy >= 0 && yy < x.H && xx >= 0 && xx < x.W) {
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(x.at(b, yy, xx) + c0 + ch * 8), f);
        nm.apply(f, b, c0 + ch * 8);
        v = pack8(f);
      }
      *reinterpret_cast<uint4*>(tile + pix * SO_PIX + ch * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const char* src = tile + ((py + tap / 3) * (SO_TW + 2) + px + tap % 3) * SO_PIX;
      const uint32_t* w = wts + (size_t)tap * CO * C2 + c0 / 2;
#pragma unroll
      for (int j = 0; j < SO_CC / 8; j++) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + j * 16);
#pragma unroll
        for (int co = 0; co < CO; co++) {
          const uint4 wv = *reinterpret_cast<const uint4*>(w + co * C2 + j * 4);      // wave-uniform address
          acc[co] = dot2_acc(v.w, wv.w, dot2_acc(v.z, wv.z, dot2_acc(v.y, wv.y, dot2_acc(v.x, wv.x, acc[co]))));
        }
      }
    }
  }
  const int yo = y0 + py, xo = x0 + px;
  if (yo < x.H && xo < x.W) {
#pragma unroll
    for (int co = 0; co < CO; co++) img[(((long)b * CO + co) * x.H + yo) * x.W + xo] = acc[co] + (bias ? bias[co] : 0.f);
  }
}

__global__ __launch_bounds__(256) void add_kernel(Grid a, Grid bb, Grid o) {
  const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;   // grid: (chunks of one row, row, sample)
  if (i >= a.W * (a.C / 8)) return;
  float f[8], g[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(a.at(b, y, 0) + i * 8), f);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(bb.at(b, y, 0) + i * 8), g);
#pragma unroll
  for (int e = 0; e < 8; e++) f[e] += g[e];
  *reinterpret_cast<uint4*>(o.at(b, y, 0) + i * 8) = pack8(f);
}

// ---- P = softmax(scale * S) per row, S fp32 (the single 512-wide head of the mid-block attention: the scores stay fp32 until here)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long ld, bf16_t* __restrict__ p, long ldp, int cols, float scale) {
  __shared__ float red[4];
  const float* row = s + (long)blockIdx.x * ld;
  bf16_t* out = p + (long)blockIdx.x * ldp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_reduce = [&](float v, bool is_max) -> float {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float w = __shfl_xor(v, o); v = is_max ? fmaxf(v, w) : v + w; }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
  };
  float mx = -INFINITY;
  for (int c = threadIdx.x * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  mx = block_reduce(mx, true);
  float sum = 0.f;
  for (int c = threadIdx.x * 4; c < cols; c += 1024) {   // second and third pass hit the L2 (a row is <= 64 KiB at 2K latents)
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    sum += (__expf((v.x - mx) * scale) + __expf((v.y - mx) * scale)) + (__expf((v.z - mx) * scale) + __expf((v.w - mx) * scale));
  }
  sum = block_reduce(sum, false);
  const float inv = 1.f / sum;
  for (int c = threadIdx.x * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    *reinterpret_cast<uint2*>(out + c) = pack_bf16x4(__expf((v.x - mx) * scale) * inv, __expf((v.y - mx) * scale) * inv,
                                                     __expf((v.z - mx) * scale) * inv, __expf((v.w - mx) * scale) * inv);
  }
}

// ---- fp32 NCHW image / latent <-> bf16 grid (channels padded with zeros up to the grid's C)
__global__ __launch_bounds__(256) void nchw_to_grid_kernel(const float* __restrict__ img, int C, float mul, Grid y) {
  const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y, b = blockIdx.z;
  if (xx >= y.W) return;
  const long HW = (long)y.H * y.W, p = (long)yy * y.W + xx;
  bf16_t* dst = y.at(b, yy, xx);
  for (int c0 = 0; c0 < y.C; c0 += 8) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (c0 + e < C) ? img[((long)b * C + c0 + e) * HW + p] * mul : 0.f;
    *reinterpret_cast<uint4*>(dst + c0) = pack8(f);
  }
}
__global__ __launch_bounds__(256) void grid_to_nchw_kernel(Grid x, int C, float* __restrict__ img) {
  const int xx = blockIdx.x * 256 + threadIdx.x, yy = blockIdx.y, b = blockIdx.z;
  if (xx >= x.W) return;
  const long HW = (long)x.H * x.W, p = (long)yy * x.W + xx;
  const bf16_t* src = x.at(b, yy, xx);
  for (int c0 = 0; c0 < C; c0 += 8) {
    float f[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(src + c0), f);
#pragma unroll
    for (int e = 0; e < 8; e++)
      if (c0 + e < C) img[((long)b * C + c0 + e) * HW + p] = f[e];
  }
}

static int check_grid(const pxa_grid* g, const char* what) {
  PXA_CHECK(g && g->ptr, "%s: null grid", what);
  PXA_CHECK(g->B > 0 && g->H > 0 && g->W > 0 && g->C > 0 && g->C % 8 == 0, "%s: bad grid %d x %d x %d x %d (C must be a multiple of 8)", what, g->B, g->H, g->W, g->C);
  PXA_CHECK(g->row_pitch >= g->W && g->img_pitch >= (long)(g->H - 1) * g->row_pitch + g->W, "%s: bad pitches", what);
  return 0;
}
static int make_norm(Norm& nm, const float* mean, const float* rstd, const float* gamma, const float* beta, int C, int groups, int silu, const char* what) {
  nm = Norm{mean, rstd, gamma, beta, 1, 1, silu, 0};
  if (mean) {
    PXA_CHECK(rstd && gamma && beta, "%s: GroupNorm needs mean, rstd, gamma and beta", what);
    PXA_CHECK(groups > 0 && C % groups == 0 && (C / groups) % 4 == 0, "%s: C=%d / groups=%d must be a multiple of 4", what, C, groups);
    nm.cpg = C / groups; nm.G = groups;
    nm.cpg_shift = (nm.cpg & (nm.cpg - 1)) == 0 ? __builtin_ctz(nm.cpg) : -1;
  }
  return 0;
}
}  // namespace

extern "C" int pxa_vae_gn_stats(const pxa_grid* x, int groups, float eps, double* ws, float* mean, float* rstd, hipStream_t stream) {
  if (int rc = check_grid(x, "pxa_vae_gn_stats")) return rc;
  PXA_CHECK(ws && mean && rstd, "pxa_vae_gn_stats: null output");
  const int C = x->C, CV = C / 8;
  PXA_CHECK(groups > 0 && groups <= 256 && C % groups == 0 && (C / groups) % 4 == 0, "pxa_vae_gn_stats: C=%d / groups=%d must be a multiple of 4", C, groups);
  PXA_CHECK(CV <= 256 && 256 % CV == 0, "pxa_vae_gn_stats: C=%d must be 8 * a power of two <= 2048", C);
  hipError_t e = hipMemsetAsync(ws, 0, sizeof(double) * 2 * groups * x->B, stream);
  PXA_CHECK(e == hipSuccess, "pxa_vae_gn_stats: memset failed: %s", hipGetErrorString(e));
  const long HW = (long)x->H * x->W;
  const int ppb = 256 / CV, per_row = (x->W + ppb - 1) / ppb;      // loads per thread and image row
  int nb = per_row >= 8 ? x->H : (x->H * per_row + 7) / 8;         // ~8 loads per thread, at most one block per row
  nb = nb < 1 ? 1 : (nb > x->H ? x->H : nb);
  hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)nb, x->B), dim3(256), 0, stream, to_grid(x), C / groups, ws);
  PXA_LAUNCH_CHECK();
  const int n = groups * x->B;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ws, mean, rstd, n, (double)HW * (C / groups), eps);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_gn_finalize(const float* part, int B, int C, int groups, long pixels, float eps, float* mean, float* rstd, hipStream_t stream) {
  PXA_CHECK(part && mean && rstd && B > 0 && pixels > 0, "pxa_vae_gn_finalize: bad arguments");
  PXA_CHECK(groups > 0 && C % groups == 0 && (C / groups) % 4 == 0, "pxa_vae_gn_finalize: C=%d / groups=%d must be a multiple of 4", C, groups);
  const int n = B * groups;
  hipLaunchKernelGGL(gn_finalize_part_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, part, mean, rstd, B, C, C / groups, (double)pixels * (C / groups), eps);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_gn_apply(const pxa_grid* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups,
                                int silu, int upsample, const pxa_grid* y, hipStream_t stream) {
  if (int rc = check_grid(x, "pxa_vae_gn_apply(x)")) return rc;
  if (int rc = check_grid(y, "pxa_vae_gn_apply(y)")) return rc;
  PXA_CHECK(upsample == 1 || upsample == 2, "pxa_vae_gn_apply: upsample must be 1 or 2");
  PXA_CHECK(y->B == x->B && y->C == x->C && y->H == x->H * upsample && y->W == x->W * upsample, "pxa_vae_gn_apply: output grid does not match");
  Norm nm;
  if (int rc = make_norm(nm, mean, rstd, gamma, beta, x->C, groups, silu, "pxa_vae_gn_apply")) return rc;
  PXA_CHECK(y->H <= 65535 && y->B <= 65535, "pxa_vae_gn_apply: grid too large");
  hipLaunchKernelGGL(gn_apply_kernel, dim3((y->W * (y->C / 8) + 255) / 256, (y->H + GA_ROWS - 1) / GA_ROWS, y->B), dim3(256), 0, stream, to_grid(x), nm, upsample, to_grid(y));
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_im2col3x3(const pxa_grid* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups,
                                 int silu, int stride, int pad, int Ho, int Wo, void* col_bf16, hipStream_t stream) {
  if (int rc = check_grid(x, "pxa_vae_im2col3x3")) return rc;
  PXA_CHECK(col_bf16 && (stride == 1 || stride == 2) && (pad == 0 || pad == 1) && Ho > 0 && Wo > 0, "pxa_vae_im2col3x3: bad arguments");
  Norm nm;
  if (int rc = make_norm(nm, mean, rstd, gamma, beta, x->C, groups, silu, "pxa_vae_im2col3x3")) return rc;
  PXA_CHECK(Ho <= 65535 && x->B <= 65535, "pxa_vae_im2col3x3: grid too large");
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((Wo * 9 * (x->C / 8) + 255) / 256, Ho, x->B), dim3(256), 0, stream, to_grid(x), nm, stride, pad, Ho, Wo,
                     (bf16_t*)col_bf16);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_add(const pxa_grid* a, const pxa_grid* b, const pxa_grid* out, hipStream_t stream) {
  if (int rc = check_grid(a, "pxa_vae_add(a)")) return rc;
  if (int rc = check_grid(b, "pxa_vae_add(b)")) return rc;
  if (int rc = check_grid(out, "pxa_vae_add(out)")) return rc;
  PXA_CHECK(a->B == b->B && a->H == b->H && a->W == b->W && a->C == b->C && a->B == out->B && a->H == out->H && a->W == out->W && a->C == out->C,
            "pxa_vae_add: grids differ");
  PXA_CHECK(a->H <= 65535 && a->B <= 65535, "pxa_vae_add: grid too large");
  hipLaunchKernelGGL(add_kernel, dim3((a->W * (a->C / 8) + 255) / 256, a->H, a->B), dim3(256), 0, stream, to_grid(a), to_grid(b), to_grid(out));
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_softmax_rows(const float* s, long ld, void* p_bf16, long ldp, int rows, int cols, float scale, hipStream_t stream) {
  PXA_CHECK(s && p_bf16 && rows > 0 && cols > 0, "pxa_vae_softmax_rows: bad arguments");
  PXA_CHECK(cols % 4 == 0 && ld % 4 == 0 && ldp % 4 == 0, "pxa_vae_softmax_rows: cols / ld / ldp must be multiples of 4");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, s, ld, (bf16_t*)p_bf16, ldp, cols, scale);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_nchw_to_grid(const float* img, int C, float mul, const pxa_grid* y, hipStream_t stream) {
  if (int rc = check_grid(y, "pxa_vae_nchw_to_grid")) return rc;
  PXA_CHECK(img && C > 0 && C <= y->C, "pxa_vae_nchw_to_grid: bad channel count %d (grid has %d)", C, y->C);
  PXA_CHECK(y->H <= 65535 && y->B <= 65535, "pxa_vae_nchw_to_grid: grid too large");
  hipLaunchKernelGGL(nchw_to_grid_kernel, dim3((y->W + 255) / 256, y->H, y->B), dim3(256), 0, stream, img, C, mul, to_grid(y));
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_grid_to_nchw(const pxa_grid* x, int C, float* img, hipStream_t stream) {
  if (int rc = check_grid(x, "pxa_vae_grid_to_nchw")) return rc;
  PXA_CHECK(img && C > 0 && C <= x->C, "pxa_vae_grid_to_nchw: bad channel count %d (grid has %d)", C, x->C);
  PXA_CHECK(x->H <= 65535 && x->B <= 65535, "pxa_vae_grid_to_nchw: grid too large");
  hipLaunchKernelGGL(grid_to_nchw_kernel, dim3((x->W + 255) / 256, x->H, x->B), dim3(256), 0, stream, to_grid(x), C, img);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_vae_conv3x3_small_out(const pxa_grid* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int groups, int silu,
                                         const void* w_taps, const float* bias, int Cout, float* img, hipStream_t stream) {
  if (int rc = check_grid(x, "pxa_vae_conv3x3_small_out")) return rc;
  PXA_CHECK(w_taps && img && Cout >= 1 && Cout <= 4, "pxa_vae_conv3x3_small_out: Cout=%d must be 1..4", Cout);
  PXA_CHECK(x->C % SO_CC == 0, "pxa_vae_conv3x3_small_out: C=%d must be a multiple of %d", x->C, SO_CC);
  Norm nm;
  if (int rc = make_norm(nm, mean, rstd, gamma, beta, x->C, groups, silu, "pxa_vae_conv3x3_small_out")) return rc;
  PXA_CHECK(x->H <= 65535 * SO_TH && x->B <= 65535, "pxa_vae_conv3x3_small_out: grid too large");
  const dim3 grid((x->W + SO_TW - 1) / SO_TW, (x->H + SO_TH - 1) / SO_TH, x->B);
  const uint32_t* w = (const uint32_t*)w_taps;
  switch (Cout) {
    case 1: hipLaunchKernelGGL(conv3x3_small_out_kernel<1>, grid, dim3(256), 0, stream, to_grid(x), nm, w, bias, img); break;
    case 2: hipLaunchKernelGGL(conv3x3_small_out_kernel<2>, grid, dim3(256), 0, stream, to_grid(x), nm, w, bias, img); break;
    case 3: hipLaunchKernelGGL(conv3x3_small_out_kernel<3>, grid, dim3(256), 0, stream, to_grid(x), nm, w, bias, img); break;
    default: hipLaunchKernelGGL(conv3x3_small_out_kernel<4>, grid, dim3(256), 0, stream, to_grid(x), nm, w, bias, img); break;
  }
  PXA_LAUNCH_CHECK();
  return 0;
}
