// CAME optimizer step (Luo et al., "CAME: Confidence-guided Adaptive Memory Efficient Optimization", ACL 2023) over the flat
// parameter store.  The PixArt-Sigma configs train with it: optimizer = dict(type='CAMEWrapper', lr=2e-5, weight_decay=0.0,
// betas=(0.9, 0.999, 0.9999), eps=(1e-30, 1e-16)) (configs/pixart_sigma_config/PixArt_sigma_xl2_img1024_internalms.py:29);
// CAMEWrapper subclasses came_pytorch.CAME unchanged (diffusion/utils/optimizer.py:15,242-246).  came_pytorch is an un-vendored,
// unpinned dependency: the arithmetic below follows its published step() (restated with citations in oracle/came_ref.py).
//
// Every tensor with >= 2 dims is "factored": viewed as [batch][R][C] (R, C = its last two dims), second moments are kept as row
// means [batch][R] and column means [batch][C] only.  One step is a fixed sequence of launches over ALL tensors at once (a tile
// table maps a workgroup to a run of whole rows of one tensor), with the dependent reductions meeting at the launch boundaries:
//   A  v = g^2 + eps0: row means -> EMA into sq_row (a tile owns whole rows: single writer), column partials -> scratch,
//      row-state means -> scratch; 1-D tensors: EMA into their full second moment, sum(u^2) of u = g * rsqrt(nf_sq)
//   B  sq_col EMA from the column partials
//   C  factored: sum(u^2) of u = g * rsqrt(sq_row / mean_r(sq_row)) * rsqrt(sq_col)
//   D  u /= max(1, rms(u) / clip); m = b1 m + (1 - b1) u; res = (u - m)^2 + eps1: row / column / row-state means as in A;
//      1-D tensors finish here: p = p (1 - lr wd) - lr m
//   E  res_col EMA
//   F  factored: p = p (1 - lr wd) - lr m * rsqrt(res_row / mean_r(res_row)) * rsqrt(res_col); bf16 shadow refresh
// HBM-bound: g is read three times, m twice, p once (+ writes): ~9 passes over the 2.44 GB of fp32 state per step.
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

struct CameParams {
  float* p; const float* g; float* m; bf16_t* shadow;
  float* sq_row; float* sq_col; float* res_row; float* res_col; float* nf_sq;
  float* col_sum1; float* col_sum2; float* rm1; float* rm2; float* usq;   // scratch, zeroed per step
  const pxa_came_tensor* tensors; const pxa_came_tile* tiles;
  const float* col_inv_r;
  float lr, b1, b2, b3, omb1, omb2, omb3, eps0, eps1, clip, decay;   // omb = 1 - beta and decay = 1 - lr * wd, formed in fp64 on the host
  const float* gscale;
  const float* scaler;     // optional loss-scaler record (optim.hip): [2] != 0 -> the whole step is skipped
};

constexpr int MAXC = 256 * 18;                        // widest row whose column partials are combined in registers / LDS

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// update scale of a tensor once sum(u^2) is complete: 1 / max(1, rms(u) / clip)
__device__ __forceinline__ float u_scale(const CameParams& a, int t, double numel) {
  const float rms = sqrtf((float)((double)a.usq[t] / numel));
  return 1.f / fmaxf(1.f, rms / a.clip);
}

// Rows [first, first + count) of a factored tensor, one wave per row at a time, 16 bytes per lane and array.  MAXJ > 0: C is a
// multiple of 4 and at most 256 * MAXJ, and each lane keeps the column partials of ITS columns (4 * lane + 256 j) in registers for
// the whole tile - the four waves meet in LDS once per tile, the tile meets the others with one global atomic per column.
// MAXJ == 0: any C / batched matrices: scalar loads, column partials straight to global atomics (tiny or odd tensors only).
template <int PASS, int MAXJ>
__device__ __forceinline__ void came_rows(const CameParams& a, const pxa_came_tensor& T, const pxa_came_tile& tile, int t_id, float gs, float* colsum, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = T.R, C = T.C;
  constexpr bool STATS = (PASS == 0 || PASS == 3);     // passes that take row / column means
  const float sc = PASS == 3 ? u_scale(a, t_id, (double)T.batch * R * C) : 1.f;
  float* csum = PASS == 0 ? a.col_sum1 : a.col_sum2;
  const float* cstat = PASS == 5 ? a.res_col : a.sq_col;
  float acc = 0.f, rmacc = 0.f;
  float cacc[MAXJ > 0 ? MAXJ * 4 : 1];
#pragma unroll
  for (int i = 0; i < (MAXJ > 0 ? MAXJ * 4 : 1); i++) cacc[i] = 0.f;
  for (int gr = tile.first + wave; gr < tile.first + tile.count; gr += 4) {
    const int b = gr / R;
    const long base = T.off + (long)gr * C;
    const long cbase = T.col_off + (long)b * C;
    float rf = 0.f;
    if (PASS == 2 || PASS == 3) rf = rsqrtf(a.sq_row[T.row_off + gr] / a.rm1[T.rm_off + b]);
    if (PASS == 5) rf = rsqrtf(a.res_row[T.row_off + gr] / a.rm2[T.rm_off + b]);
    float rowacc = 0.f;
    // one element: returns the quantity whose row / column means this pass takes (passes A and D); updates m / p in place
    auto element = [&](float gin, float& mv, float& pv, float colstat) -> float {
      if (PASS == 5) {
        pv = pv * a.decay - a.lr * (mv * rf * rsqrtf(colstat));
        return 0.f;
      }
      const float g = gin * gs;
      if (PASS == 0) return g * g + a.eps0;
      float u = g * rf * rsqrtf(colstat);
      if (PASS == 2) { acc += u * u; return 0.f; }
      u *= sc;
      mv = a.b1 * mv + a.omb1 * u;
      return (u - mv) * (u - mv) + a.eps1;
    };
    if (MAXJ > 0) {
#pragma unroll
      for (int j = 0; j < MAXJ; j++) {
        const int c = lane * 4 + 256 * j;
        if (c >= C) continue;                          // (no early exit: the unrolled copies index cacc statically)
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = g4, p4 = g4, s4 = g4;
        if (PASS != 5) g4 = *reinterpret_cast<const float4*>(a.g + base + c);
        if (PASS == 3 || PASS == 5) m4 = *reinterpret_cast<const float4*>(a.m + base + c);
        if (PASS == 5) p4 = *reinterpret_cast<const float4*>(a.p + base + c);
        if (PASS != 0) s4 = *reinterpret_cast<const float4*>(cstat + cbase + c);
        float gv[4] = {g4.x, g4.y, g4.z, g4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float v = element(gv[e], mv[e], pv[e], sv[e]);
          if (STATS) { rowacc += v; cacc[j * 4 + e] += v; }
        }
        if (PASS == 3) *reinterpret_cast<float4*>(a.m + base + c) = make_float4(mv[0], mv[1], mv[2], mv[3]);
        if (PASS == 5) {
          *reinterpret_cast<float4*>(a.p + base + c) = make_float4(pv[0], pv[1], pv[2], pv[3]);
          *reinterpret_cast<uint2*>(a.shadow + base + c) = pack_bf16x4(pv[0], pv[1], pv[2], pv[3]);
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        float mv = (PASS == 3 || PASS == 5) ? a.m[base + c] : 0.f, pv = PASS == 5 ? a.p[base + c] : 0.f;
        const float v = element(PASS != 5 ? a.g[base + c] : 0.f, mv, pv, PASS != 0 ? cstat[cbase + c] : 0.f);
        if (STATS) { rowacc += v; atomicAdd(&csum[cbase + c], v); }
        if (PASS == 3) a.m[base + c] = mv;
        if (PASS == 5) { a.p[base + c] = pv; a.shadow[base + c] = f2bf(pv); }
      }
    }
    if (STATS) {
      rowacc = wave_sum(rowacc);
      if (lane == 0) {
        float* st = (PASS == 0 ? a.sq_row : a.res_row) + T.row_off + gr;
        const float beta = PASS == 0 ? a.b2 : a.b3, omb = PASS == 0 ? a.omb2 : a.omb3;
        const float nv = beta * *st + omb * (rowacc / C);
        *st = nv;
        if (T.batch == 1) rmacc += nv / R;             // one matrix: the tile's rows share one row-state mean -> one atomic per tile
        else atomicAdd(&(PASS == 0 ? a.rm1 : a.rm2)[T.rm_off + b], nv / R);
      }
    }
  }
  if (PASS == 2) {
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&a.usq[t_id], acc);
  }
  if (STATS && T.batch == 1) {
    rmacc = block_sum(rmacc, red);                     // (non-zero on lane 0 of each wave only)
    if (threadIdx.x == 0) atomicAdd(&(PASS == 0 ? a.rm1 : a.rm2)[T.rm_off], rmacc);
  }
  if (STATS && MAXJ > 0) {                             // column partials: registers -> LDS (4 waves) -> one global atomic per column
    for (int c = threadIdx.x; c < C; c += 256) colsum[c] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXJ; j++) {
      const int c = lane * 4 + 256 * j;
      if (c >= C) continue;
#pragma unroll
      for (int e = 0; e < 4; e++) atomicAdd(&colsum[e * (C / 4) + (c >> 2)], cacc[j * 4 + e]);   // transposed: consecutive lanes, consecutive banks
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(&csum[T.col_off + c], colsum[(c & 3) * (C / 4) + (c >> 2)]);
  }
}

template <int PASS>   // 0 = A, 2 = C, 3 = D, 5 = F
__global__ __launch_bounds__(256) void came_tile_kernel(CameParams a) {
  __shared__ float colsum[MAXC];
  __shared__ float red[4];
  if (a.scaler && a.scaler[2] != 0.f) return;          // loss-scaled training: inf/nan in the gradients -> the step is skipped
  const pxa_came_tile tile = a.tiles[blockIdx.x];
  const pxa_came_tensor T = a.tensors[tile.tensor];
  const float gs = a.gscale ? *a.gscale : 1.f;
  if (!T.factored) {                                   // 1-D tensor: tile = [first, first + count) elements
    if (PASS != 0 && PASS != 3) return;
    const float sc = PASS == 3 ? u_scale(a, tile.tensor, (double)T.C) : 1.f;
    float acc = 0.f;
    for (int i = tile.first + threadIdx.x; i < tile.first + tile.count; i += 256) {
      const float g = a.g[T.off + i] * gs;
      if (PASS == 0) {
        const float nf = a.b2 * a.nf_sq[T.nf_off + i] + a.omb2 * (g * g + a.eps0);
        a.nf_sq[T.nf_off + i] = nf;
        const float u = g * rsqrtf(nf);
        acc += u * u;
      } else {
        const float u = g * rsqrtf(a.nf_sq[T.nf_off + i]) * sc;
        const float m = a.b1 * a.m[T.off + i] + a.omb1 * u;
        a.m[T.off + i] = m;
        const float pn = a.p[T.off + i] * a.decay - a.lr * m;
        a.p[T.off + i] = pn;
        a.shadow[T.off + i] = f2bf(pn);
      }
    }
    if (PASS == 0) {
      acc = block_sum(acc, red);
      if (threadIdx.x == 0) atomicAdd(&a.usq[tile.tensor], acc);
    }
    return;
  }
  const bool vec = T.batch == 1 && (T.C & 3) == 0;
  if (vec && T.C <= 256 * 5) came_rows<PASS, 5>(a, T, tile, tile.tensor, gs, colsum, red);          // C = 1152: most of the model
  else if (vec && T.C <= 256 * 18) came_rows<PASS, 18>(a, T, tile, tile.tensor, gs, colsum, red);   // C = 4096 / 4608
  else came_rows<PASS, 0>(a, T, tile, tile.tensor, gs, colsum, red);
}

__global__ __launch_bounds__(256) void came_col_kernel(float* __restrict__ state, const float* __restrict__ sums, const float* __restrict__ inv_r, long n, float beta, float omb,
                                                       const float* __restrict__ scaler) {
  if (scaler && scaler[2] != 0.f) return;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i < n) state[i] = beta * state[i] + omb * (sums[i] * inv_r[i]);
}
}  // namespace

extern "C" long pxa_came_scratch_elems(long n_cols_total, long n_rm_total, int n_tensors) { return 2 * n_cols_total + 2 * n_rm_total + n_tensors; }

extern "C" int pxa_came_step(const pxa_came_args* a, hipStream_t stream) {
  PXA_CHECK(a && a->p && a->g && a->exp_avg && a->p_bf16 && a->scratch && a->tensors && a->tiles, "pxa_came_step: null argument");
  PXA_CHECK(a->n_tensors > 0 && a->n_tiles > 0 && a->n_cols_total >= 0 && a->n_rm_total >= 0, "pxa_came_step: bad table sizes");
  PXA_CHECK(a->n_cols_total == 0 || (a->sq_row && a->sq_col && a->res_row && a->res_col && a->col_inv_r), "pxa_came_step: factored state missing");
  PXA_CHECK(a->clip_threshold > 0.f, "pxa_came_step: clip_threshold must be positive");
  CameParams k;
  k.p = a->p; k.g = a->g; k.m = a->exp_avg; k.shadow = (bf16_t*)a->p_bf16;
  k.sq_row = a->sq_row; k.sq_col = a->sq_col; k.res_row = a->res_row; k.res_col = a->res_col; k.nf_sq = a->nf_sq;
  k.col_sum1 = a->scratch; k.col_sum2 = k.col_sum1 + a->n_cols_total; k.rm1 = k.col_sum2 + a->n_cols_total; k.rm2 = k.rm1 + a->n_rm_total;
  k.usq = k.rm2 + a->n_rm_total;
  k.tensors = a->tensors; k.tiles = a->tiles; k.col_inv_r = a->col_inv_r;
  k.lr = (float)a->lr; k.b1 = (float)a->beta1; k.b2 = (float)a->beta2; k.b3 = (float)a->beta3;
  k.omb1 = (float)(1.0 - a->beta1); k.omb2 = (float)(1.0 - a->beta2); k.omb3 = (float)(1.0 - a->beta3);
  k.eps0 = (float)a->eps0; k.eps1 = (float)a->eps1; k.clip = (float)a->clip_threshold; k.decay = (float)(1.0 - a->lr * a->weight_decay);
  k.gscale = a->gscale; k.scaler = a->scaler;
  hipError_t e = hipMemsetAsync(a->scratch, 0, sizeof(float) * pxa_came_scratch_elems(a->n_cols_total, a->n_rm_total, a->n_tensors), stream);
  PXA_CHECK(e == hipSuccess, "pxa_came_step: memset failed: %s", hipGetErrorString(e));
  const dim3 grid(a->n_tiles), blk(256), cgrid((unsigned)((a->n_cols_total + 255) / 256));
  hipLaunchKernelGGL(came_tile_kernel<0>, grid, blk, 0, stream, k);
  PXA_LAUNCH_CHECK();
  if (a->n_cols_total) {
    hipLaunchKernelGGL(came_col_kernel, cgrid, blk, 0, stream, k.sq_col, k.col_sum1, k.col_inv_r, a->n_cols_total, k.b2, k.omb2, k.scaler);
    PXA_LAUNCH_CHECK();
    hipLaunchKernelGGL(came_tile_kernel<2>, grid, blk, 0, stream, k);
    PXA_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(came_tile_kernel<3>, grid, blk, 0, stream, k);
  PXA_LAUNCH_CHECK();
  if (a->n_cols_total) {
    hipLaunchKernelGGL(came_col_kernel, cgrid, blk, 0, stream, k.res_col, k.col_sum2, k.col_inv_r, a->n_cols_total, k.b3, k.omb3, k.scaler);
    PXA_LAUNCH_CHECK();
    hipLaunchKernelGGL(came_tile_kernel<5>, grid, blk, 0, stream, k);
    PXA_LAUNCH_CHECK();
  }
  return 0;
}
