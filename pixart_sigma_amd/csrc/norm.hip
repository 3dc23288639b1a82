// HBM-bound row kernels of the adaLN-single block (PixArtMS.py:71-79, PixArt_blocks.py:24-25):
//   ln_mod_fwd : x' = x (+ gate*u) ; xn = LayerNorm_noaffine(x', eps) * (1+scale) + shift   [fused residual + LN + modulate]
//   ln_mod_bwd : dx = dx_in + LN^T( dy*(1+scale) ) ; dshift += sum_rows dy ; dscale += sum_rows dy*xhat
//   gate_bwd   : g = dx (+ add) ; du = gate*g ; dgate += sum_rows g*u
//   colsum     : bias gradients  db[n] += sum_rows dY[r][n]
// One 32-lane half-wave owns a row: D/128 float4 per lane, 512-byte coalesced segments, fp32 statistics
// (two-pass mean / centered variance in registers), one HBM pass per tensor.
#include <cstdlib>
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

template <int NV>
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(
    const float* x, const bf16_t* __restrict__ u, const float* __restrict__ gate,
    const float* __restrict__ shift, const float* __restrict__ scale, int mod_stride, int gate_stride,
    float* x_out, bf16_t* __restrict__ xn, bf16_t* __restrict__ xb,
    float* __restrict__ mean_out, float* __restrict__ rstd_out, int R, int D, int rows_per_batch, float eps) {
  const int hl = threadIdx.x & 31;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= R) return;
  const int b = row / rows_per_batch;
  const size_t base = (size_t)row * D;
  float4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) v[j] = ld_f4(x + base + (hl + 32 * j) * 4);
  if (u) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      const uint2 uu = ld_u2(u + base + c);
      float u0, u1, u2, u3;
      unpack_bf16x2(uu.x, u0, u1); unpack_bf16x2(uu.y, u2, u3);
      if (gate) {
        const float4 g = *reinterpret_cast<const float4*>(gate + (size_t)b * gate_stride + c);
        v[j].x += g.x * u0; v[j].y += g.y * u1; v[j].z += g.z * u2; v[j].w += g.w * u3;
      } else {
        v[j].x += u0; v[j].y += u1; v[j].z += u2; v[j].w += u3;
      }
    }
  }
  if (x_out) {
#pragma unroll
    for (int j = 0; j < NV; j++) st_f4(x_out + base + (hl + 32 * j) * 4, v[j]);
  }
  if (xb) {
#pragma unroll
    for (int j = 0; j < NV; j++) st_u2(xb + base + (hl + 32 * j) * 4, pack_bf16x4(v[j].x, v[j].y, v[j].z, v[j].w));
  }
  if (!xn) return;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = half_wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += (a * a + bb * bb) + (c * c + d * d);
  }
  const float rstd = rsqrtf(half_wave_sum(q) / D + eps);
  if (mean_out && hl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (hl + 32 * j) * 4;
    const float4 sh = *reinterpret_cast<const float4*>(shift + (size_t)b * mod_stride + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + (size_t)b * mod_stride + c);
    float y0 = (v[j].x - mean) * rstd * (1.f + sc.x) + sh.x;
    float y1 = (v[j].y - mean) * rstd * (1.f + sc.y) + sh.y;
    float y2 = (v[j].z - mean) * rstd * (1.f + sc.z) + sh.z;
    float y3 = (v[j].w - mean) * rstd * (1.f + sc.w) + sh.w;
    st_u2(xn + base + c, pack_bf16x4(y0, y1, y2, y3));
  }
}

constexpr int BWD_ROWS = 16;  // rows per half-wave in the reducing backward kernels
#ifndef BWD_ROW_MAP
#define BWD_ROW_MAP 1          // 1 = the block's 8 half-waves walk its 128 rows interleaved (row = first + h + 8 i: the block reads 8 consecutive rows =
#endif                         //     37 KB at a time); 0 = 16 consecutive rows per half-wave (4096 concurrent streams 72 KB = 9 x 2^13 bytes apart)
__device__ __forceinline__ int bwd_row(int blk_first, int h, int i) { return BWD_ROW_MAP ? blk_first + h + 8 * i : blk_first + h * BWD_ROWS + i; }

// Block-level combine of two per-lane column-sum sets (a0 -> columns [0, D), a1 -> columns [D, 2 D)) held by the 8 half-waves of a 256-thread
// block, WITHOUT LDS atomics (they retire at about one lane per clock per CU: 8 half-waves x 72 words x 32 lanes = 18 k cycles per block, a
// quarter of the block's streaming time).  The two halves of a wave cover the same columns: one v_permlane32_swap per (a0, a1) pair adds them so that
// the lower half owns a0 and the upper half a1 (9 ds_write_b128 per lane).  Waves 0/1 store into slots 0/1, waves 2/3 add into them:
// red = [2 slots][2 D] floats, two barriers, plain LDS reads and writes only.
__device__ __forceinline__ float half_swap_sum(float lo_owner, float hi_owner) {
  // v_permlane32_swap: dst.upper <-> src.lower.  Afterwards dst = {lo_owner.lower, hi_owner.lower}, src = {lo_owner.upper, hi_owner.upper}: their
  // sum is the wave total of lo_owner in the lower 32 lanes and of hi_owner in the upper 32 - one swap and one add per pair of values
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_owner), __float_as_uint(hi_owner), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int NV>
__device__ __forceinline__ void block_colsum_combine(float* red, float4 (&a0)[NV], float4 (&a1)[NV], int D) {
  const int hl = threadIdx.x & 31, upper = (threadIdx.x >> 5) & 1, wave = threadIdx.x >> 6;
  float* slot = red + (wave & 1) * 2 * D + upper * D;
  float4 mine[NV];
#pragma unroll
  for (int j = 0; j < NV; j++)
    mine[j] = make_float4(half_swap_sum(a0[j].x, a1[j].x), half_swap_sum(a0[j].y, a1[j].y), half_swap_sum(a0[j].z, a1[j].z), half_swap_sum(a0[j].w, a1[j].w));
  if (wave < 2) {
#pragma unroll
    for (int j = 0; j < NV; j++) *reinterpret_cast<float4*>(slot + (hl + 32 * j) * 4) = mine[j];
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      float4* p = reinterpret_cast<float4*>(slot + (hl + 32 * j) * 4);
      const float4 o = *p;
      *p = make_float4(o.x + mine[j].x, o.y + mine[j].y, o.z + mine[j].z, o.w + mine[j].w);
    }
  }
  __syncthreads();
}
#ifndef LNB_VARIANT
#define LNB_VARIANT 1          // 1 = the row's dx_in loads are issued with its other loads (dx_out may alias dx_in, so a load placed behind the
#endif                         //     previous chunk's store cannot be hoisted: 0 = that order, 312 us; 1: 250 us; 3 = all dx_in loads after the reduction: 407 us)

template <int NV, bool DB>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(
    const bf16_t* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ scale, int mod_stride, const float* dx_in, float* dx_out, bf16_t* __restrict__ dx_bf16,
    float* __restrict__ dshift, float* __restrict__ dscale, int dmod_stride, float* __restrict__ dbias, long dbias_stride, int R, int D, int rows_per_batch) {
  // Per-sample column sums (dshift / dscale): every 16-row chunk used to fire 72 global atomics per lane at the SAME 2 x D words
  // of its sample - ~1000 serialised read-modify-writes per cache line and call.  When the block's 128 rows belong to one sample
  // (always, unless a sample's token count is not a multiple of 128) the 8 half-waves first combine in LDS, then the block adds
  // once: 8x fewer, fully coalesced atomics (block_colsum_combine: no LDS atomics either).
  extern __shared__ float red[];                       // [2 slots][2][D] (+ DB: [8 half-waves][D] column sums of dx_out)
  const int hl = threadIdx.x & 31;
  const int blk_first = blockIdx.x * 8 * BWD_ROWS, blk_last = min(R, blk_first + 8 * BWD_ROWS) - 1;
  const int r_beg = bwd_row(blk_first, threadIdx.x >> 5, 0);
  const bool one_sample = (blk_first / rows_per_batch) == (blk_last / rows_per_batch);   // block-uniform
  // (round 5) DB: column sums of dx_out - the bias gradient of the Linear whose output gradient dx_bf16 is (cross_attn.proj) - into slotted partials like
  // gate_bwd's: the rows are in registers anyway, the separate colsum pass over dx_bf16 (151 MB read per block) goes.  The sums live in LDS, one private
  // float4 slot per lane and column group (plain read-add-write, 9 KB per row of a CU that streams 256 rows: nothing beside the HBM time): a third register
  // accumulator set took the kernel from 284 to 334 registers and from 204 to 345 us for EVERY call (session 7 step profile); instances without DB
  // compile to the round-4 kernel.
  float4* adb = reinterpret_cast<float4*>(red + 4 * D + (threadIdx.x >> 5) * D);
  float4 ash[NV], asc[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) { ash[j] = make_float4(0, 0, 0, 0); asc[j] = make_float4(0, 0, 0, 0); }
  if constexpr (DB) {
#pragma unroll
    for (int j = 0; j < NV; j++) adb[hl + 32 * j] = make_float4(0, 0, 0, 0);
  }
  int cur_b = r_beg / rows_per_batch;
  auto flush = [&](int b) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      float* ps = dshift + (size_t)b * dmod_stride + c;
      float* pc = dscale + (size_t)b * dmod_stride + c;
      atomicAdd(ps + 0, ash[j].x); atomicAdd(ps + 1, ash[j].y); atomicAdd(ps + 2, ash[j].z); atomicAdd(ps + 3, ash[j].w);
      atomicAdd(pc + 0, asc[j].x); atomicAdd(pc + 1, asc[j].y); atomicAdd(pc + 2, asc[j].z); atomicAdd(pc + 3, asc[j].w);
      ash[j] = make_float4(0, 0, 0, 0); asc[j] = make_float4(0, 0, 0, 0);
      if constexpr (DB) {
        float* pb = dbias + (size_t)(b % PXA_COLSUM_SLOTS) * dbias_stride + c;
        const float4 a = adb[hl + 32 * j];
        atomicAdd(pb + 0, a.x); atomicAdd(pb + 1, a.y); atomicAdd(pb + 2, a.z); atomicAdd(pb + 3, a.w);
        adb[hl + 32 * j] = make_float4(0, 0, 0, 0);
      }
    }
  };
  for (int i = 0; i < BWD_ROWS; i++) {
    const int row = bwd_row(blk_first, threadIdx.x >> 5, i);
    if (row >= R) break;
    const int b = row / rows_per_batch;
    if (b != cur_b) { flush(cur_b); cur_b = b; }
    const size_t base = (size_t)row * D;
    const float mu = mean[row], rs = rstd[row];
    float4 g[NV], xh[NV];
#if LNB_VARIANT == 1 || LNB_VARIANT == 3
    float4 di[NV];
#endif
#if LNB_VARIANT == 1
    if (dx_in) {
#pragma unroll
      for (int j = 0; j < NV; j++) di[j] = ld_f4(dx_in + base + (hl + 32 * j) * 4);
    }
#endif
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      const uint2 dd = ld_u2(dy + base + c);
      const float4 xv = ld_f4(x + base + c);
      const float4 sc = *reinterpret_cast<const float4*>(scale + (size_t)b * mod_stride + c);
      float d0, d1, d2, d3;
      unpack_bf16x2(dd.x, d0, d1); unpack_bf16x2(dd.y, d2, d3);
      xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      ash[j].x += d0; ash[j].y += d1; ash[j].z += d2; ash[j].w += d3;
      asc[j].x += d0 * xh[j].x; asc[j].y += d1 * xh[j].y; asc[j].z += d2 * xh[j].z; asc[j].w += d3 * xh[j].w;
      g[j] = make_float4(d0 * (1.f + sc.x), d1 * (1.f + sc.y), d2 * (1.f + sc.z), d3 * (1.f + sc.w));
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    const float c1 = half_wave_sum(s1) / D, c2 = half_wave_sum(s2) / D;
#if LNB_VARIANT == 3
    if (dx_in) {
#pragma unroll
      for (int j = 0; j < NV; j++) di[j] = *reinterpret_cast<const float4*>(dx_in + base + (hl + 32 * j) * 4);
    }
#endif
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      float4 o = make_float4(rs * (g[j].x - c1 - xh[j].x * c2), rs * (g[j].y - c1 - xh[j].y * c2),
                             rs * (g[j].z - c1 - xh[j].z * c2), rs * (g[j].w - c1 - xh[j].w * c2));
      if (dx_in) {
#if LNB_VARIANT == 1 || LNB_VARIANT == 3
        o.x += di[j].x; o.y += di[j].y; o.z += di[j].z; o.w += di[j].w;
#else
        const float4 di = *reinterpret_cast<const float4*>(dx_in + base + c);
        o.x += di.x; o.y += di.y; o.z += di.z; o.w += di.w;
#endif
      }
      st_f4(dx_out + base + c, o);
      if (dx_bf16) st_u2(dx_bf16 + base + c, pack_bf16x4(o.x, o.y, o.z, o.w));
      if constexpr (DB) { float4 a = adb[hl + 32 * j]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; adb[hl + 32 * j] = a; }
    }
  }
  if (one_sample) {
    block_colsum_combine<NV>(red, ash, asc, D);
    const int b = blk_first / rows_per_batch;
    for (int i = threadIdx.x; i < D; i += 256) {
      atomicAdd(dshift + (size_t)b * dmod_stride + i, red[i] + red[2 * D + i]);
      atomicAdd(dscale + (size_t)b * dmod_stride + i, red[D + i] + red[3 * D + i]);
    }
    if constexpr (DB) {                                // the 8 half-waves' rows of sums (every lane's last write is in front of the combine's barriers)
      const float* hw = red + 4 * D;
      for (int i = threadIdx.x; i < D; i += 256)
        atomicAdd(dbias + (size_t)(b % PXA_COLSUM_SLOTS) * dbias_stride + i,
                  ((hw[i] + hw[D + i]) + (hw[2 * D + i] + hw[3 * D + i])) + ((hw[4 * D + i] + hw[5 * D + i]) + (hw[6 * D + i] + hw[7 * D + i])));
    }
  } else if (r_beg < R) {
    flush(cur_b);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void gate_bwd_kernel(
    const float* dx, const bf16_t* __restrict__ add, const bf16_t* __restrict__ u, const float* __restrict__ gate,
    int mod_stride, float* dx_out, bf16_t* __restrict__ du, float* __restrict__ dgate, int dmod_stride, float* __restrict__ dbias,
    long dbias_stride, int R, int D, int rows_per_batch) {
  extern __shared__ float red[];                       // [2 slots][2][D]: block-level combine of the column sums (block_colsum_combine)
  const int hl = threadIdx.x & 31;
  const int blk_first = blockIdx.x * 8 * BWD_ROWS, blk_last = min(R, blk_first + 8 * BWD_ROWS) - 1;
  const int r_beg = bwd_row(blk_first, threadIdx.x >> 5, 0);
  const bool one_sample = (blk_first / rows_per_batch) == (blk_last / rows_per_batch);   // block-uniform
  float4 ag[NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) { ag[j] = make_float4(0, 0, 0, 0); ab[j] = make_float4(0, 0, 0, 0); }
  int cur_b = r_beg / rows_per_batch;
  auto flush = [&](int b) {
    if (dgate) {
#pragma unroll
      for (int j = 0; j < NV; j++) {
        float* pg = dgate + (size_t)b * dmod_stride + (hl + 32 * j) * 4;
        atomicAdd(pg + 0, ag[j].x); atomicAdd(pg + 1, ag[j].y); atomicAdd(pg + 2, ag[j].z); atomicAdd(pg + 3, ag[j].w);
        ag[j] = make_float4(0, 0, 0, 0);
      }
    }
    if (dbias) {   // partial slot b % PXA_COLSUM_SLOTS: bounds same-address atomic contention exactly like the per-sample dgate
#pragma unroll
      for (int j = 0; j < NV; j++) {
        float* pb = dbias + (size_t)(b % PXA_COLSUM_SLOTS) * dbias_stride + (hl + 32 * j) * 4;
        atomicAdd(pb + 0, ab[j].x); atomicAdd(pb + 1, ab[j].y); atomicAdd(pb + 2, ab[j].z); atomicAdd(pb + 3, ab[j].w);
        ab[j] = make_float4(0, 0, 0, 0);
      }
    }
  };
  for (int i = 0; i < BWD_ROWS; i++) {
    const int row = bwd_row(blk_first, threadIdx.x >> 5, i);
    if (row >= R) break;
    const int b = row / rows_per_batch;
    if (b != cur_b) { flush(cur_b); cur_b = b; }
    const size_t base = (size_t)row * D;
    // every load of the row is issued before its first store: dx_out may alias dx (in place), so a load behind a store could not be
    // hoisted over it and the row became nine serialised load -> store round trips (the same fix took ln_mod_bwd from 3.4 to 4.2 TB/s)
    float4 gv[NV];
    uint2 av[NV], uv[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      gv[j] = ld_f4(dx + base + c);
      if (add) av[j] = ld_u2(add + base + c);
      if (gate) uv[j] = ld_u2(u + base + c);
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      float4 g = gv[j];
      if (add) {
        float a0, a1, a2, a3;
        unpack_bf16x2(av[j].x, a0, a1); unpack_bf16x2(av[j].y, a2, a3);
        g.x += a0; g.y += a1; g.z += a2; g.w += a3;
      }
      if (dx_out) st_f4(dx_out + base + c, g);
      float4 o = g;
      if (gate) {
        const float4 gt = *reinterpret_cast<const float4*>(gate + (size_t)b * mod_stride + c);
        float u0, u1, u2, u3;
        unpack_bf16x2(uv[j].x, u0, u1); unpack_bf16x2(uv[j].y, u2, u3);
        ag[j].x += g.x * u0; ag[j].y += g.y * u1; ag[j].z += g.z * u2; ag[j].w += g.w * u3;
        o = make_float4(g.x * gt.x, g.y * gt.y, g.z * gt.z, g.w * gt.w);
      }
      if (du) st_u2(du + base + c, pack_bf16x4(o.x, o.y, o.z, o.w));
      ab[j].x += o.x; ab[j].y += o.y; ab[j].z += o.z; ab[j].w += o.w;
    }
  }
  if (one_sample) {
    block_colsum_combine<NV>(red, ag, ab, D);
    const int b = blk_first / rows_per_batch;
    for (int i = threadIdx.x; i < D; i += 256) {
      if (dgate) atomicAdd(dgate + (size_t)b * dmod_stride + i, red[i] + red[2 * D + i]);
      if (dbias) atomicAdd(dbias + (size_t)(b % PXA_COLSUM_SLOTS) * dbias_stride + i, red[D + i] + red[3 * D + i]);
    }
  } else if (r_beg < R) {
    flush(cur_b);
  }
}

// db[n] += sum_r dY[r][n].  Block = 48 column-threads (8 columns = one 16-byte load each) x 5 row-lanes over rows_per_block rows;
// 8 independent row loads in flight per thread; the row-lanes are combined through LDS, one atomic per column and block.
constexpr int CS_CT = 48, CS_RL = 5;                  // 48 column-threads x 8 columns = 384 (divides 1152 / 3456 / 4608), 5 row-lanes
__global__ __launch_bounds__(CS_CT * CS_RL) void colsum_kernel(const bf16_t* __restrict__ dy, int ld, float* __restrict__ out, int R, int N, int rows_per_block) {
  __shared__ float red[CS_RL][CS_CT][8];
  const int ct = threadIdx.x % CS_CT, rl = threadIdx.x / CS_CT;
  const int c = (blockIdx.x * CS_CT + ct) * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // (1, 0) and (0, 1) in the operand type, kept in registers the compiler cannot see through: as a literal, 0x00003f80 is emitted as the INLINE constant
  // "1.0" of v_dot2c_f32_bf16, which the hardware expands to fp32 1.0 = 0x3f800000 - i.e. (0, 1), the wrong half (first GPU run of this loop: every
  // even column received its odd neighbour's sum).
  uint32_t one_lo = PXA_OPERAND_ONE_BITS, one_hi = PXA_OPERAND_ONE_BITS << 16;
  asm volatile("" : "+s"(one_lo), "+s"(one_hi));
  if (c < N) {
    for (int r = r0 + rl; r < r1; r += 8 * CS_RL) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int rr = r + CS_RL * u;
        v[u] = rr < r1 ? *reinterpret_cast<const uint4*>(dy + (size_t)rr * ld + c) : make_uint4(0, 0, 0, 0);
      }
      // one v_dot2 per column against (1, 0) / (0, 1) instead of unpack + add: half the VALU instructions, and no f16 -> f32 conversions in the fp16
      // build (where this kernel ran 2.2x longer than in the bf16 build: profiles/r03i_step_kernel_stats_*.csv)
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[2 * j] = dot2_acc(w[j], one_lo, acc[2 * j]);
          acc[2 * j + 1] = dot2_acc(w[j], one_hi, acc[2 * j + 1]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[rl][ct][e] = acc[e];
  __syncthreads();
  if (rl == 0 && c < N) {
#pragma unroll
    for (int e = 0; e < 8; e++) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < CS_RL; k++) t += red[k][ct][e];
      atomicAdd(out + c + e, t);
    }
  }
}


// ------------------------------------------------------------------------------------------------ q / k LayerNorm (qk_norm=True)
// nn.LayerNorm(dim) with affine weight/bias and eps 1e-5 over the full channel dimension, applied to the q and k column blocks
// of the qkv GEMM output before compression / attention (reference PixArt_blocks.py:90-92,133-134).  bf16 rows in, bf16 rows out
// (in place in the qkv buffer), statistics in fp32; the un-normalised rows are copied to `xsave` for the backward pass.
template <int NV>
__global__ __launch_bounds__(256) void ln_affine_fwd_kernel(const bf16_t* x, long x_stride, const float* __restrict__ w, const float* __restrict__ b,
                                                            bf16_t* y, long y_stride, bf16_t* __restrict__ xsave, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int R, int D, float eps) {
  const int hl = threadIdx.x & 31, row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= R) return;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (hl + 32 * j) * 4;
    const uint2 u = *reinterpret_cast<const uint2*>(x + (size_t)row * x_stride + c);
    if (xsave) *reinterpret_cast<uint2*>(xsave + (size_t)row * D + c) = u;
    unpack_bf16x2(u.x, v[j].x, v[j].y); unpack_bf16x2(u.y, v[j].z, v[j].w);
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = half_wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += (a * a + bb * bb) + (c * c + d * d);
  }
  const float rstd = rsqrtf(half_wave_sum(q) / D + eps);
  if (hl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (hl + 32 * j) * 4;
    const float4 ww = *reinterpret_cast<const float4*>(w + c), bv = *reinterpret_cast<const float4*>(b + c);
    *reinterpret_cast<uint2*>(y + (size_t)row * y_stride + c) =
        pack_bf16x4((v[j].x - mean) * rstd * ww.x + bv.x, (v[j].y - mean) * rstd * ww.y + bv.y, (v[j].z - mean) * rstd * ww.z + bv.z,
                    (v[j].w - mean) * rstd * ww.w + bv.w);
  }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy w;   dw += sum_rows dy xhat,  db += sum_rows dy   (block-combined in LDS)
template <int NV>
__global__ __launch_bounds__(256) void ln_affine_bwd_kernel(const bf16_t* dy, long dy_stride, const bf16_t* __restrict__ xsave, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ w, bf16_t* dx, long dx_stride,
                                                            float* __restrict__ dw, float* __restrict__ db, int R, int D) {
  extern __shared__ float red[];                       // [2][D]
  const int hl = threadIdx.x & 31;
  const int chunk = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int r_beg = chunk * BWD_ROWS, r_end = min(R, r_beg + BWD_ROWS);
  for (int i = threadIdx.x; i < 2 * D; i += 256) red[i] = 0.f;
  __syncthreads();
  float4 aw[NV], ab[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) { aw[j] = make_float4(0, 0, 0, 0); ab[j] = make_float4(0, 0, 0, 0); }
  for (int row = r_beg; row < r_end; row++) {
    const float mu = mean[row], rs = rstd[row];
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      const uint2 dd = *reinterpret_cast<const uint2*>(dy + (size_t)row * dy_stride + c);
      const uint2 xx = *reinterpret_cast<const uint2*>(xsave + (size_t)row * D + c);
      const float4 ww = *reinterpret_cast<const float4*>(w + c);
      float d0, d1, d2, d3, x0, x1, x2, x3;
      unpack_bf16x2(dd.x, d0, d1); unpack_bf16x2(dd.y, d2, d3);
      unpack_bf16x2(xx.x, x0, x1); unpack_bf16x2(xx.y, x2, x3);
      xh[j] = make_float4((x0 - mu) * rs, (x1 - mu) * rs, (x2 - mu) * rs, (x3 - mu) * rs);
      ab[j].x += d0; ab[j].y += d1; ab[j].z += d2; ab[j].w += d3;
      aw[j].x += d0 * xh[j].x; aw[j].y += d1 * xh[j].y; aw[j].z += d2 * xh[j].z; aw[j].w += d3 * xh[j].w;
      g[j] = make_float4(d0 * ww.x, d1 * ww.y, d2 * ww.z, d3 * ww.w);
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    const float c1 = half_wave_sum(s1) / D, c2 = half_wave_sum(s2) / D;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      const int c = (hl + 32 * j) * 4;
      *reinterpret_cast<uint2*>(dx + (size_t)row * dx_stride + c) =
          pack_bf16x4(rs * (g[j].x - c1 - xh[j].x * c2), rs * (g[j].y - c1 - xh[j].y * c2), rs * (g[j].z - c1 - xh[j].z * c2),
                      rs * (g[j].w - c1 - xh[j].w * c2));
    }
  }
#pragma unroll
  for (int j = 0; j < NV; j++) {
    float* pw = red + (hl + 32 * j) * 4;
    float* pb = red + D + (hl + 32 * j) * 4;
    atomicAdd(pw + 0, aw[j].x); atomicAdd(pw + 1, aw[j].y); atomicAdd(pw + 2, aw[j].z); atomicAdd(pw + 3, aw[j].w);
    atomicAdd(pb + 0, ab[j].x); atomicAdd(pb + 1, ab[j].y); atomicAdd(pb + 2, ab[j].z); atomicAdd(pb + 3, ab[j].w);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) { atomicAdd(dw + i, red[i]); atomicAdd(db + i, red[D + i]); }
}

#define DISPATCH_NV(D, CALL)                                            \
  switch ((D) / 128) {                                                  \
    case 9: { constexpr int NV = 9; CALL; } break;                      \
    case 8: { constexpr int NV = 8; CALL; } break;                      \
    case 4: { constexpr int NV = 4; CALL; } break;                      \
    case 2: { constexpr int NV = 2; CALL; } break;                      \
    case 1: { constexpr int NV = 1; CALL; } break;                      \
    default: pxa_set_error("unsupported hidden size %d (supported: 128,256,512,1024,1152)", (D)); return -1; \
  }
}  // namespace

extern "C" int pxa_ln_mod_fwd(const float* x, const void* u_bf16, const float* gate, int gate_stride, const float* shift, const float* scale,
                              int mod_stride, float* x_out, void* xn_bf16, void* xb_bf16, float* mean, float* rstd,
                              int R, int D, int rows_per_batch, float eps, hipStream_t stream) {
  PXA_CHECK(x && R > 0 && D % 128 == 0 && rows_per_batch > 0, "pxa_ln_mod_fwd: bad args");
  PXA_CHECK(!xn_bf16 || (shift && scale), "pxa_ln_mod_fwd: LN output needs shift/scale");
  PXA_CHECK(!mean == !rstd, "pxa_ln_mod_fwd: mean/rstd must both be given or both null");
  DISPATCH_NV(D, hipLaunchKernelGGL(ln_mod_fwd_kernel<NV>, dim3((R + 7) / 8), dim3(256), 0, stream, x, (const bf16_t*)u_bf16, gate, shift, scale,
                                     mod_stride, gate_stride, x_out, (bf16_t*)xn_bf16, (bf16_t*)xb_bf16, mean, rstd, R, D, rows_per_batch, eps));
  PXA_LAUNCH_CHECK();
  return 0;
}


extern "C" int pxa_ln_affine_fwd(const void* x_bf16, long x_stride, const float* w, const float* b, void* y_bf16, long y_stride, void* xsave_bf16,
                                 float* mean, float* rstd, int R, int D, float eps, hipStream_t stream) {
  PXA_CHECK(x_bf16 && w && b && y_bf16 && mean && rstd, "pxa_ln_affine_fwd: null pointer");
  PXA_CHECK(R > 0 && D % 128 == 0 && x_stride % 4 == 0 && y_stride % 4 == 0, "pxa_ln_affine_fwd: bad shape / strides");
  DISPATCH_NV(D, hipLaunchKernelGGL(ln_affine_fwd_kernel<NV>, dim3((R + 7) / 8), dim3(256), 0, stream, (const bf16_t*)x_bf16, x_stride, w, b,
                                     (bf16_t*)y_bf16, y_stride, (bf16_t*)xsave_bf16, mean, rstd, R, D, eps));
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_ln_affine_bwd(const void* dy_bf16, long dy_stride, const void* xsave_bf16, const float* mean, const float* rstd, const float* w,
                                 void* dx_bf16, long dx_stride, float* dw, float* db, int R, int D, hipStream_t stream) {
  PXA_CHECK(dy_bf16 && xsave_bf16 && mean && rstd && w && dx_bf16 && dw && db, "pxa_ln_affine_bwd: null pointer");
  PXA_CHECK(R > 0 && D % 128 == 0 && dy_stride % 4 == 0 && dx_stride % 4 == 0, "pxa_ln_affine_bwd: bad shape / strides");
  const int chunks = (R + BWD_ROWS - 1) / BWD_ROWS;
  DISPATCH_NV(D, hipLaunchKernelGGL(ln_affine_bwd_kernel<NV>, dim3((chunks + 7) / 8), dim3(256), 2 * D * sizeof(float), stream, (const bf16_t*)dy_bf16, dy_stride,
                                     (const bf16_t*)xsave_bf16, mean, rstd, w, (bf16_t*)dx_bf16, dx_stride, dw, db, R, D));
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_ln_mod_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* scale,
                              int mod_stride, const float* dx_in, float* dx_out, void* dx_bf16, float* dshift, float* dscale, int dmod_stride,
                              float* dbias, long dbias_stride, int R, int D, int rows_per_batch, hipStream_t stream) {
  PXA_CHECK(dy_bf16 && x && mean && rstd && scale && dx_out && dshift && dscale, "pxa_ln_mod_bwd: null pointer");
  PXA_CHECK(R > 0 && D % 128 == 0 && rows_per_batch > 0, "pxa_ln_mod_bwd: bad shape");
  const int chunks = (R + BWD_ROWS - 1) / BWD_ROWS;
  if (dbias) {
    DISPATCH_NV(D, hipLaunchKernelGGL((ln_mod_bwd_kernel<NV, true>), dim3((chunks + 7) / 8), dim3(256), 12 * D * sizeof(float), stream, (const bf16_t*)dy_bf16, x, mean, rstd,
                                       scale, mod_stride, dx_in, dx_out, (bf16_t*)dx_bf16, dshift, dscale, dmod_stride, dbias, dbias_stride, R, D, rows_per_batch));
  } else {
    DISPATCH_NV(D, hipLaunchKernelGGL((ln_mod_bwd_kernel<NV, false>), dim3((chunks + 7) / 8), dim3(256), 4 * D * sizeof(float), stream, (const bf16_t*)dy_bf16, x, mean, rstd,
                                       scale, mod_stride, dx_in, dx_out, (bf16_t*)dx_bf16, dshift, dscale, dmod_stride, dbias, dbias_stride, R, D, rows_per_batch));
  }
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_gate_bwd(const float* dx, const void* add_bf16, const void* u_bf16, const float* gate, int mod_stride,
                            float* dx_out, void* du_bf16, float* dgate, int dmod_stride, float* dbias, long dbias_stride, int R, int D,
                            int rows_per_batch, hipStream_t stream) {
  PXA_CHECK(dx && R > 0 && D % 128 == 0 && rows_per_batch > 0, "pxa_gate_bwd: bad args");
  PXA_CHECK(!gate || (u_bf16 && dgate), "pxa_gate_bwd: gate needs u and dgate");
  const int chunks = (R + BWD_ROWS - 1) / BWD_ROWS;
  DISPATCH_NV(D, hipLaunchKernelGGL(gate_bwd_kernel<NV>, dim3((chunks + 7) / 8), dim3(256), 4 * D * sizeof(float), stream, dx, (const bf16_t*)add_bf16,
                                     (const bf16_t*)u_bf16, gate, mod_stride, dx_out, (bf16_t*)du_bf16, dgate, dmod_stride, dbias, dbias_stride, R, D, rows_per_batch));
  PXA_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, long stride, float* __restrict__ out, long n) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  float s = out[i];
#pragma unroll
  for (int p = 0; p < PXA_COLSUM_SLOTS; p++) s += part[p * stride + i];
  out[i] = s;
}
}  // namespace
extern "C" int pxa_colsum_reduce(const float* part, long stride, float* out, long n, hipStream_t stream) {
  PXA_CHECK(part && out && n > 0 && stride >= n, "pxa_colsum_reduce: bad args");
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, part, stride, out, n);
  PXA_LAUNCH_CHECK();
  return 0;
}

extern "C" int pxa_colsum_bf16(const void* dy_bf16, int ld, float* out, int R, int N, hipStream_t stream) {
  PXA_CHECK(dy_bf16 && out && R > 0 && N % 8 == 0 && ld % 8 == 0, "pxa_colsum_bf16: bad args");
  // ~80 row blocks per column group: every output address then receives ~80 same-address atomics.  That contention, not the streaming, set
  // the time: at R = 65,536 a 1152-wide sum took 78 us with 1024 workgroups (328 atomics per address) and 34-36 us with 192-256, a 3456-wide
  // one 77-79 us anywhere between 384 and 1024 (profiles/r02_elementwise.txt).
  const int bx = (N / 8 + CS_CT - 1) / CS_CT;
  static const int forced = getenv("PXA_COLSUM_BLOCKS") ? atoi(getenv("PXA_COLSUM_BLOCKS")) : 0;         // A/B: workgroups per launch
  const int target = forced > 0 ? forced : (80 * bx < 192 ? 192 : (80 * bx > 1024 ? 1024 : 80 * bx));
  int rpb = (int)(((long)R * bx + target - 1) / target);
  rpb = (rpb + 8 * CS_RL - 1) / (8 * CS_RL) * (8 * CS_RL);
  dim3 grid(bx, (R + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(CS_CT * CS_RL), 0, stream, (const bf16_t*)dy_bf16, ld, out, R, N, rpb);
  PXA_LAUNCH_CHECK();
  return 0;
}
