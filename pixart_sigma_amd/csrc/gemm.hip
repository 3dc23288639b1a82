// bf16 MFMA GEMM family for the PixArt-Sigma token linears (qkv/proj/q/kv/fc1/fc2/final/caption) and their
// backward (dX, dW).  One kernel, three operand layouts, runtime epilogue:
//   layout 0 (NT): C[m][n] = sum_k A[m][k] * B[n][k]   forward  y = x W^T        (nn.Linear, PixArt_blocks.py:47-48,130,155)
//   layout 1 (NN): C[m][n] = sum_k A[m][k] * B[k][n]   dX = dY W
//   layout 2 (TN): C[m][n] = sum_k A[k][m] * B[k][n]   dW = dY^T X   (split-K, fp32 atomic accumulate)
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles.
// The MFMA "A" operand is fed from the n-side tile and the "B" operand from the m-side tile, so that a lane owns
// one output row m and 4 consecutive n per accumulator quad -> 8-byte bf16 / 16-byte fp32 stores.
// k-contiguous operands sit in LDS as [128][64] with a 16-byte-chunk XOR swizzle (conflict-free ds_read_b128);
// k-strided operands sit as [64][128+32] and are read with ds_read_b64_tr_b16 (hardware transpose).
#include <atomic>
#include "common.h"
#include "gemm_params.h"
#include "../../include/pixart_hip.h"
#include <cstdlib>
#include <cstring>

namespace {
using namespace pxa;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KC_BYTES = 128 * BK * 2;       // 16384
constexpr int RC_STRIDE = (128 + 32) * 2;    // 320 B per k-row (pad: 4 tr-read rows hit disjoint banks)
constexpr int RC_BYTES = BK * RC_STRIDE;     // 20480


// ---- global -> registers (4 x 16 B per thread per operand tile), zero-filled out of bounds
template <bool KC>
__device__ __forceinline__ void g2r(const bf16_t* __restrict__ X, int ld, int r0, int R, int k0, int kend, uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bf16_t* src;
    bool ok;
    if (KC) {
      int c = tid & 7, row = (tid >> 3) + 32 * i;
      int gr = r0 + row, gk = k0 + c * 8;
      ok = (gr < R) && (gk < kend);
      src = X + (size_t)gr * ld + gk;
    } else {
      int c = tid & 15, kr = (tid >> 4) + 16 * i;
      int gk = k0 + kr, gr = r0 + c * 8;
      ok = (gk < kend) && (gr < R);
      src = X + (size_t)gk * ld + gr;
    }
    reg[i] = ok ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
  }
}
template <bool KC>
__device__ __forceinline__ void r2s(char* lds, const uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int off;
    if (KC) {
      int c = tid & 7, row = (tid >> 3) + 32 * i;
      off = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    } else {
      int c = tid & 15, kr = (tid >> 4) + 16 * i;
      off = kr * RC_STRIDE + c * 16;
    }
    *reinterpret_cast<uint4*>(lds + off) = reg[i];
  }
}
// ---- LDS -> MFMA fragment: lane l gets X[rbase + (l&31)][ks*16 + 8*(l>>5) + 0..7]
template <bool KC>
__device__ __forceinline__ bf16x8 frag(const char* lds, int rbase, int ks, int lane) {
  if (KC) {
    int row = rbase + (lane & 31), c = ks * 2 + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    int gg = lane >> 4, tt = lane & 15, hi = gg >> 1;
    int krow = ks * 16 + 8 * hi + (tt >> 2), col = rbase + 16 * (gg & 1) + (tt & 3) * 4;
    const char* p = lds + krow * RC_STRIDE + col * 2;
    return concat_tr(lds_tr_read(p), lds_tr_read(p + 4 * RC_STRIDE));
  }
}

// blockIdx -> output tile.  Workgroup b runs on XCD b % 8 (each XCD has a private 4 MiB L2), so every XCD is given a
// CONTIGUOUS range of the logical tile order, and the logical order walks the tile grid in groups of GROUP_M m-tiles
// (column-major inside a group): the ~64 workgroups resident on one XCD then cover an 8 x 8 super-tile whose 8 A panels
// and 8 B panels (K = 1152: 2 x 2.4 MB) are fetched once into that L2 and reused 8x, instead of 8 XCDs each streaming
// every panel.  Bijective for any tile count.
constexpr int NXCD = 8, GROUP_M = 8;
__device__ __forceinline__ void tile_coords(int bid, int mt, int nt, int split, int& tm, int& tn, int& z) {
  const int tiles = mt * nt, T = tiles * split, q = T / NXCD, r = T % NXCD;
  const int x = bid % NXCD, idx = bid / NXCD;
  const int L = x * q + min(x, r) + idx;               // logical id: XCD x owns [x*q + min(x,r), +q + (x<r))
  z = L / tiles;                                       // split-K slice major: an XCD's range shares its k-range
  const int t = L - z * tiles;
  const int per_group = GROUP_M * nt;
  const int g = t / per_group, first_m = g * GROUP_M;
  const int gsz = min(mt - first_m, GROUP_M);
  const int in_g = t - g * per_group;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}

// ---- epilogue: lane owns row m, 4 consecutive n per accumulator quad (bias / GELU / GELU' / bf16 + fp32 / atomic stores)
__device__ __forceinline__ void epilogue(const GemmParams& p, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int lane, int hi, int z) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * hi;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][q * 4 + e];
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.act == 1) {
          if (p.out2) *reinterpret_cast<uint2*>(p.out2 + (size_t)m * p.ldo + n) = pack_bf16x4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = gelu_tanh(v[e]);
        } else if (p.act == 3) {
          float g[4];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = gelu_tanh_both(v[e], g[e]);
          *reinterpret_cast<uint2*>(p.out2 + (size_t)m * p.ldo + n) = pack_bf16x4(g[0], g[1], g[2], g[3]);
        } else if (p.act == 2 || p.act == 4 || p.act == 5) {
          const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (size_t)m * p.ldaux + n);
          float a0, a1, a2, a3;
          unpack_bf16x2(a.x, a0, a1); unpack_bf16x2(a.y, a2, a3);
          if (p.act == 2) { a0 = gelu_tanh_grad(a0); a1 = gelu_tanh_grad(a1); a2 = gelu_tanh_grad(a2); a3 = gelu_tanh_grad(a3); }
          if (p.act == 5) { v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3; }
          else { v[0] *= a0; v[1] *= a1; v[2] *= a2; v[3] *= a3; }
        }
        if (p.out) *reinterpret_cast<uint2*>(p.out + (size_t)m * p.ldo + n) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        if (p.outf) {
          float* dst = p.outf + (size_t)m * p.ldf + n;
          if (p.accumulate == 3) {            // split-K partial slab z (dense [M][N]), summed into outf by splitk_reduce_kernel
            *reinterpret_cast<float4*>(p.slab + ((size_t)z * p.M + m) * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else if (p.accumulate == 2) {     // single k-slice: this thread owns the element -> plain read-modify-write
            float4 o = *reinterpret_cast<const float4*>(dst);
            *reinterpret_cast<float4*>(dst) = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
          } else if (p.accumulate) {          // fallback: fp32 atomics (scattered 4-byte atomics: ~40 G/s, avoid)
#pragma unroll
            for (int e = 0; e < 4; e++) atomicAdd(dst + e, v[e]);
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int A_BYTES = A_KC ? KC_BYTES : RC_BYTES, B_BYTES = B_KC ? KC_BYTES : RC_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = A_BYTES + B_BYTES;  // stage s: [A tile][B tile] at smem + s*STAGE

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, hi = lane >> 5;
  int tm_, tn_, z_;
  tile_coords(blockIdx.x, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.split, tm_, tn_, z_);
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int kbeg = z_ * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;

  uint4 ra[4], rb[4];
  if (nk > 0) {
    g2r<A_KC>(p.A, p.lda, m0, p.M, kbeg, kend, ra, tid);
    g2r<B_KC>(p.B, p.ldb, n0, p.N, kbeg, kend, rb, tid);
    r2s<A_KC>(smem, ra, tid);
    r2s<B_KC>(smem + A_BYTES, rb, tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
    const bool more = (kt + 1 < nk);
    if (more) {
      g2r<A_KC>(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, ra, tid);
      g2r<B_KC>(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, rb, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      bf16x8 af[2], bf[2];
      af[0] = frag<A_KC>(sA, wm * 64, ks, lane);
      af[1] = frag<A_KC>(sA, wm * 64 + 32, ks, lane);
      bf[0] = frag<B_KC>(sB, wn * 64, ks, lane);
      bf[1] = frag<B_KC>(sB, wn * 64 + 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma32(bf[j], af[i], acc[i][j]);
    }
    if (more) {
      r2s<A_KC>(smem + (cur ^ 1) * STAGE, ra, tid);
      r2s<B_KC>(smem + (cur ^ 1) * STAGE + A_BYTES, rb, tid);
    }
    __syncthreads();
  }

  epilogue(p, acc, m0, n0, wm, wn, lane, hi, z_);
}

// =====================================================================================================================
// Fast path (every K range a multiple of 64): operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction, no VGPR round trip and no ds_write issue slots).  The DMA writes lane-linear (wave-uniform base + lane*16),
// so the bank swizzles move to the per-lane SOURCE address and the same XOR is applied on the fragment reads (guide rule 21):
//   k-contiguous tile  [R r][64 k]: chunk' = chunk ^ ((r>>1)&7)                         (ds_read_b128 conflict-free)
//   k-strided   tile  [64 k][R r]: 64-byte block' = block ^ (k&3), rows unpadded         (4 tr-read rows -> 4 bank quarters)
// Rows beyond M / N are clamped to the last valid row / column 0 (their products are never stored).
// Tile shapes: the 128x128 tile moves 1 byte per 64 FLOP from L2 — at the MI355X's ~35 TB/s aggregate L2 that caps a GEMM
// near 2.2 PF and in practice (L2 misses served by MALL/HBM) well under 1 PF — so the large token GEMMs use a 256x256 tile
// (8 waves, 128x64 per wave, 128 FLOP/B) or 256x128 (8 waves, 64x64 per wave) when N is not a multiple of 256.
// Pipeline: 2 LDS stages, ONE barrier per k-tile: barrier (vmcnt(0): tile t landed, stage t^1 free) -> issue DMA of tile
// t+1 -> MFMA on tile t.
template <bool KC, int ROWS, int NW>
__device__ __forceinline__ void dma_tile(char* lds, const bf16_t* __restrict__ X, int ld, int r0, int R, int k0, int wave, int lane) {
  constexpr int NINST = ROWS * 128 / 1024 / NW;        // 1 KiB DMA instructions per wave
  constexpr int CPR = ROWS / 8;                         // 16-byte chunks per k-row of the k-strided image
#pragma unroll
  for (int i = 0; i < NINST; i++) {
    const int p = (i * NW + wave) * 64 + lane;          // linear 16-byte chunk position inside the tile
    const bf16_t* src;
    if (KC) {
      const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
      const int gr = min(r0 + row, R - 1);
      src = X + (size_t)gr * ld + k0 + c * 8;
    } else {
      const int kr = p / CPR, cl = p % CPR, c = ((((cl >> 2) ^ (kr & 3)) << 2) | (cl & 3));
      int gc = r0 + c * 8;
      gc = gc < R ? gc : 0;
      src = X + (size_t)(k0 + kr) * ld + gc;
    }
    lds_dma16(src, lds + (i * NW + wave) * 1024);
  }
}
template <bool KC, int ROWS>
__device__ __forceinline__ bf16x8 frag_g(const char* lds, int rbase, int ks, int lane) {
  if (KC) {
    const int row = rbase + (lane & 31), c = ks * 2 + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  } else {
    const int gg = lane >> 4, tt = lane & 15, hi = gg >> 1;
    const int kr = ks * 16 + 8 * hi + (tt >> 2), col = rbase + 16 * (gg & 1) + (tt & 3) * 4;
    const int blk = col >> 5, inblk = (col & 31) * 2;
    const char* p0 = lds + kr * (ROWS * 2) + ((blk ^ (kr & 3)) << 6) + inblk;
    const char* p1 = lds + (kr + 4) * (ROWS * 2) + ((blk ^ ((kr + 4) & 3)) << 6) + inblk;
    return concat_tr(lds_tr_read(p0), lds_tr_read(p1));
  }
}

// bf16 epilogue staged through LDS: the accumulator layout gives every lane 4 consecutive columns of ONE row, i.e. a store
// instruction touches 32 rows x 16 bytes (32 partial cache lines; measured 6-17 us per 256x256 tile, 15-30 % of a K=1152 GEMM).
// Here each wave first parks its (TM*32) x 64 tile in its own LDS region (128-byte rows, 16-byte chunk XOR (row&7)), then reads it back row-wise so one
// instruction stores 8 complete 128-byte row segments, 16 bytes per lane.  Bias / GELU are applied on the fp32 accumulators
// before the tile is parked; with the dual-output GELU epilogue the tile makes two trips (pre-activation, activation).
constexpr int EPI_STRIDE = 128;
template <int TM>
__device__ __forceinline__ void stage_store(const char* wl, bf16_t* __restrict__ out, int ldo, int mw, int nw, int M, int N, int lane) {
#pragma unroll
  for (int t = 0; t < TM * 4; t++) {                   // 8 rows x 128 B per instruction
    const int row = t * 8 + (lane >> 3), ch = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(wl + row * EPI_STRIDE + ((ch ^ (row & 7)) << 4));
    const int m = mw + row, n = nw + ch * 8;
    if (m < M && n < N) *reinterpret_cast<uint4*>(out + (size_t)m * ldo + n) = v;
  }
}
template <int TM, int TN>
__device__ __forceinline__ void epilogue_staged(const GemmParams& p, f32x16 (&acc)[TM][TN], char* smem, int wave, int mw, int nw, int lane, int hi) {
  static_assert(TN == 2, "staged epilogue assumes 64-column wave tiles");
  char* wl = smem + wave * (TM * 32 * EPI_STRIDE);
  __syncthreads();                                     // every wave is done reading the operand stages
  const bool dual = (p.act == 1 && p.out2 != nullptr);
  float cs[TN][16];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int g = 0; g < 16; g++) cs[j][g] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int n = nw + j * 32 + 8 * q + 4 * hi;
        if (n < p.N) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
          for (int i = 0; i < TM; i++) { acc[i][j][q * 4] += b.x; acc[i][j][q * 4 + 1] += b.y; acc[i][j][q * 4 + 2] += b.z; acc[i][j][q * 4 + 3] += b.w; }
        }
      }
  }
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {               // pass 0: pre-activation copy (dual output only); pass 1: final values
    if (pass == 0 && !dual) continue;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int col = j * 32 + 8 * q + 4 * hi;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = acc[i][j][q * 4 + e];
          if (pass == 1) {
            if (p.act == 1) {
#pragma unroll
              for (int e = 0; e < 4; e++) v[e] = gelu_tanh(v[e]);
            } else if (p.act == 2 || p.act == 4 || p.act == 5) {
              const int m = mw + i * 32 + (lane & 31), n = nw + col;
              if (m < p.M && n < p.N) {
                const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (size_t)m * p.ldaux + n);
                float a0, a1, a2, a3;
                unpack_bf16x2(a.x, a0, a1); unpack_bf16x2(a.y, a2, a3);
                if (p.act == 2) { a0 = gelu_tanh_grad(a0); a1 = gelu_tanh_grad(a1); a2 = gelu_tanh_grad(a2); a3 = gelu_tanh_grad(a3); }
                if (p.act == 5) { v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3; }
                else { v[0] *= a0; v[1] *= a1; v[2] *= a2; v[3] *= a3; }
              }
            }
          }
          *reinterpret_cast<uint2*>(wl + (i * 32 + (lane & 31)) * EPI_STRIDE + (((col >> 3) ^ (lane & 7)) << 4) + (col & 4) * 2) = pack_bf16x4(v[0], v[1], v[2], v[3]);
          if (pass == 1 && p.colsum && mw + i * 32 + (lane & 31) < p.M) {
#pragma unroll
            for (int e = 0; e < 4; e++) cs[j][q * 4 + e] += v[e];
          }
        }
    // the region is private to this wave: only its own LDS writes must have landed before the row-wise reads
    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
    stage_store<TM>(wl, pass == 0 ? p.out2 : p.out, p.ldo, mw, nw, p.M, p.N, lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);                // reads returned before the next trip overwrites the region
  }
  if (p.colsum) {                                      // column sums over this wave's rows: lane tree, one atomic per column
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int g = 0; g < 16; g++) {
        float v = cs[j][g];
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        const int n = nw + j * 32 + 8 * (g >> 2) + 4 * hi + (g & 3);
        if ((lane & 31) == 0 && n < p.N) atomicAdd(p.colsum + (size_t)((mw >> 7) % PXA_COLSUM_SLOTS) * p.colsum_stride + n, v);
      }
  }
}

template <int TM, int TN>
__device__ __forceinline__ void epilogue_t(const GemmParams& p, const f32x16 (&acc)[TM][TN], int mw, int nw, int lane, int hi, int z) {
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const int m = mw + i * 32 + (lane & 31);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < TN; j++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int n = nw + j * 32 + 8 * q + 4 * hi;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][q * 4 + e];
        if (p.bias) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.act == 1) {
          if (p.out2) *reinterpret_cast<uint2*>(p.out2 + (size_t)m * p.ldo + n) = pack_bf16x4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = gelu_tanh(v[e]);
        } else if (p.act == 3) {
          float g[4];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = gelu_tanh_both(v[e], g[e]);
          *reinterpret_cast<uint2*>(p.out2 + (size_t)m * p.ldo + n) = pack_bf16x4(g[0], g[1], g[2], g[3]);
        } else if (p.act == 2 || p.act == 4 || p.act == 5) {
          const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (size_t)m * p.ldaux + n);
          float a0, a1, a2, a3;
          unpack_bf16x2(a.x, a0, a1); unpack_bf16x2(a.y, a2, a3);
          if (p.act == 2) { a0 = gelu_tanh_grad(a0); a1 = gelu_tanh_grad(a1); a2 = gelu_tanh_grad(a2); a3 = gelu_tanh_grad(a3); }
          if (p.act == 5) { v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3; }
          else { v[0] *= a0; v[1] *= a1; v[2] *= a2; v[3] *= a3; }
        }
        if (p.out) *reinterpret_cast<uint2*>(p.out + (size_t)m * p.ldo + n) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        if (p.outf) {
          float* dst = p.outf + (size_t)m * p.ldf + n;
          if (p.accumulate == 3) {            // split-K partial slab z (dense [M][N]), summed into outf by splitk_reduce_kernel
            *reinterpret_cast<float4*>(p.slab + ((size_t)z * p.M + m) * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
          } else if (p.accumulate == 2) {     // single k-slice: this thread owns the element -> plain read-modify-write
            float4 o = *reinterpret_cast<const float4*>(dst);
            *reinterpret_cast<float4*>(dst) = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
          } else if (p.accumulate) {          // fallback: fp32 atomics (scattered 4-byte atomics: ~40 G/s, avoid)
#pragma unroll
            for (int e = 0; e < 4; e++) atomicAdd(dst + e, v[e]);
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  }
}

// SEG: the k range of A is cut into segments of p.k_seg elements, segment s starting p.seg_jump elements further on than a plain
// row would have it (rows may overlap: lda < K).  A 3x3 convolution over a zero-padded NHWC image is exactly that GEMM: the three
// taps of one kernel row are 3*C contiguous elements of the padded image, the next kernel row is one image row further on.
template <int LAYOUT, int TBM, int TBN, int WM, int WN, int EPI, bool SEG = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_kernel(GemmParams p) {
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int NW = WM * WN, TM = TBM / WM / 32, TN = TBN / WN / 32;
  constexpr int A_BYTES = TBM * 128, B_BYTES = TBN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN, hi = lane >> 5;
  int tm_, tn_, z_;
  tile_coords(blockIdx.x, (p.M + TBM - 1) / TBM, (p.N + TBN - 1) / TBN, p.split, tm_, tn_, z_);
  const int m0 = tm_ * TBM, n0 = tn_ * TBN;
  const int kbeg = z_ * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;

  const bf16_t* Ab = p.A;                              // SEG: A base of the segment the next DMA reads from
  int seg_left = SEG ? p.k_seg : 0;                    // ... and the k elements left in it (split-K is not combined with SEG)
  int kx = 0, ky = 0;                                  // tap-interleaved order: the tap the next DMA reads
  auto seg_advance = [&]() {
    if (SEG) {
      if (p.k_tap) {                                   // Ab + 64 t = A + offset of k-tile t: correct the linear +64 at every tap change
        long adj;
        if (kx < 2) { adj = p.k_tap - BK; kx++; }                                      // next tap of the kernel row: + C
        else if (ky < 2) { adj = p.tap_s - 2L * p.k_tap - BK; kx = 0; ky++; }          // next kernel row, tap 0
        else { adj = -2L * p.tap_s - 2L * p.k_tap; kx = 0; ky = 0; }                    // next 64-channel chunk, first tap
        Ab += adj;
      } else {
        seg_left -= BK;
        if (seg_left == 0) { Ab += p.seg_jump; seg_left = p.k_seg; }
      }
    }
  };
  if (nk > 0) {
    dma_tile<A_KC, TBM, NW>(smem, Ab, p.lda, m0, p.M, kbeg, wave, lane);
    dma_tile<B_KC, TBN, NW>(smem + A_BYTES, p.B, p.ldb, n0, p.N, kbeg, wave, lane);
    seg_advance();
  }
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    lds_dma_wait<0>();                                 // this wave's DMA pieces of tile kt have landed (written out: the compiler does not see the asm DMA)
    __syncthreads();                                   // stage hand-over
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
    // fragments are double-buffered in registers: the ds_reads of k-step ks+1 are in flight under the MFMAs of ks, and
    // the DMA of the next k-tile is issued behind the first MFMA group so the matrix pipe restarts right after the barrier
    bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; i++) af[0][i] = frag_g<A_KC, TBM>(sA, wm * (TM * 32) + i * 32, 0, lane);
#pragma unroll
    for (int j = 0; j < TN; j++) bf[0][j] = frag_g<B_KC, TBN>(sB, wn * (TN * 32) + j * 32, 0, lane);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      if (ks + 1 < BK / 16) {
#pragma unroll
        for (int i = 0; i < TM; i++) af[(ks + 1) & 1][i] = frag_g<A_KC, TBM>(sA, wm * (TM * 32) + i * 32, ks + 1, lane);
#pragma unroll
        for (int j = 0; j < TN; j++) bf[(ks + 1) & 1][j] = frag_g<B_KC, TBN>(sB, wn * (TN * 32) + j * 32, ks + 1, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = mfma32(bf[ks & 1][j], af[ks & 1][i], acc[i][j]);
      if (ks == 0 && kt + 1 < nk) {
        char* nxt = smem + (cur ^ 1) * STAGE;
        dma_tile<A_KC, TBM, NW>(nxt, Ab, p.lda, m0, p.M, kbeg + (kt + 1) * BK, wave, lane);
        dma_tile<B_KC, TBN, NW>(nxt + A_BYTES, p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, wave, lane);
        seg_advance();
      }
    }
  }
  if constexpr (EPI == 1) epilogue_staged<TM, TN>(p, acc, smem, wave, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, hi);
  else epilogue_t<TM, TN>(p, acc, m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, hi, z_);
}

// =====================================================================================================================
// Deep-pipelined variant: NST LDS stages of depth BKT, NST-1 k-tiles of LDS-DMA in flight.  ~20 % of the operand requests miss
// the XCD's L2 (inherent to a 32-workgroup super-tile) and come back from MALL/HBM after microseconds; with a single tile of
// prefetch the slowest request of every k-tile gates the barrier (SQ_WAIT_ANY ~47 % of wave cycles on the 2-stage kernel).
// Here a wave only waits for the OLDEST tile (counted s_waitcnt vmcnt(N), never 0 in the steady state) and the barrier is a
// raw s_barrier, so younger DMAs stay in flight across it (guide T3/T4).
template <bool KC, int ROWS, int NW, int BKT>
__device__ __forceinline__ void dma_tile_p(char* lds, const bf16_t* __restrict__ X, int ld, int r0, int R, int k0, int wave, int lane) {
  constexpr int CPRK = BKT / 8;                         // chunks per row, k-contiguous image
  constexpr int CPR = ROWS / 8;                         // chunks per k-row, k-strided image
  constexpr int NINST = ROWS * BKT * 2 / 1024 / NW;
#pragma unroll
  for (int i = 0; i < NINST; i++) {
    const int p = (i * NW + wave) * 64 + lane;
    const bf16_t* src;
    if (KC) {
      const int row = p / CPRK, cl = p % CPRK;
      const int c = CPRK == 8 ? (cl ^ ((row >> 1) & 7)) : (cl ^ ((row >> 2) & 3));
      const int gr = min(r0 + row, R - 1);
      src = X + (size_t)gr * ld + k0 + c * 8;
    } else {
      const int kr = p / CPR, cl = p % CPR, c = ((((cl >> 2) ^ (kr & 3)) << 2) | (cl & 3));
      int gc = r0 + c * 8;
      gc = gc < R ? gc : 0;
      src = X + (size_t)(k0 + kr) * ld + gc;
    }
    lds_dma16(src, lds + (i * NW + wave) * 1024);
  }
}
// Same stream with running per-lane source pointers (the persistent kernel): dma_ptrs() gives the addresses of k-unit 0, every
// issue afterwards costs one 64-bit add per instruction instead of re-deriving row * ld + k.
template <bool KC, int ROWS, int NW, int BKT>
__device__ __forceinline__ void dma_ptrs(const bf16_t* __restrict__ X, int ld, int r0, int R, int wave, int lane, const bf16_t* (&ptr)[ROWS * BKT * 2 / 1024 / NW]) {
  constexpr int CPRK = BKT / 8, CPR = ROWS / 8, NINST = ROWS * BKT * 2 / 1024 / NW;
#pragma unroll
  for (int i = 0; i < NINST; i++) {
    const int p = (i * NW + wave) * 64 + lane;
    if (KC) {
      const int row = p / CPRK, cl = p % CPRK;
      const int c = CPRK == 8 ? (cl ^ ((row >> 1) & 7)) : (cl ^ ((row >> 2) & 3));
      ptr[i] = X + (size_t)min(r0 + row, R - 1) * ld + c * 8;
    } else {
      const int kr = p / CPR, cl = p % CPR, c = ((((cl >> 2) ^ (kr & 3)) << 2) | (cl & 3));
      int gc = r0 + c * 8;
      gc = gc < R ? gc : 0;
      ptr[i] = X + (size_t)kr * ld + gc;
    }
  }
}
template <int NW, int NINST>
__device__ __forceinline__ void dma_issue(char* lds, const bf16_t* (&ptr)[NINST], long step, int wave) {
#pragma unroll
  for (int i = 0; i < NINST; i++) {
    lds_dma16(ptr[i], lds + (i * NW + wave) * 1024);
    ptr[i] += step;
  }
}
template <bool KC, int ROWS, int BKT>
__device__ __forceinline__ bf16x8 frag_p(const char* lds, int rbase, int ks, int lane) {
  if (KC) {
    constexpr int CPRK = BKT / 8;
    const int row = rbase + (lane & 31), c = ks * 2 + (lane >> 5);
    const int cs = CPRK == 8 ? (c ^ ((row >> 1) & 7)) : (c ^ ((row >> 2) & 3));
    return *reinterpret_cast<const bf16x8*>(lds + row * (BKT * 2) + (cs << 4));
  } else {
    return frag_g<false, ROWS>(lds, rbase, ks, lane);
  }
}
template <int V> struct IntC { static constexpr int value = V; };
#ifndef GEMM_WAIT_ALL
#define GEMM_WAIT_ALL 0     // debug build: every COUNTED vmcnt wait of this file becomes vmcnt(0).  The counts below mirror issue orders by hand; a count that
#endif                      // over-estimates lets registers / LDS be read before they land.  tests/test_kernels_gpu.py::test_gemm_counted_waits_match_full_waits builds
                            // this variant and demands bit-identical outputs from every epilogue flavour.
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (GEMM_WAIT_ALL != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// A value the compiler must treat as new at this point: per-lane constants derived from it are RECOMPUTED where they are used (a few VALU per item)
// instead of being hoisted above the item loop - where, at the 256-register ceiling of the main loop, they were spilled and came back through
// `scratch_load; s_waitcnt vmcnt(0)` pairs that also drained the epilogue's own stores and the next item's DMA (round 3: 15 registers / 64 B of
// scratch in the NN instance, one reload-and-drain per 8 stored rows).
// Used by the instances that spilled (NN; the aux / statistics / run-time epilogue flavours): same-box A/B NN +1.5 ... +4 %, but -0.5 ... -1 % on plain NT, which never spilled and only pays
// for the recomputation (profiles/r03g_kbench_gemm_ab.txt).
#ifndef GEMM_OPAQUE
#define GEMM_OPAQUE 1       // 0 = let the compiler hoist everywhere (A/B builds)
#endif
template <bool ON>
__device__ __forceinline__ int opaque(int v) {
  if constexpr ((ON && GEMM_OPAQUE) || GEMM_OPAQUE == 2) asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ void wait_vm_upto(int n) {      // n (wave-uniform) in {0, 2, 4, 6}: LDS-DMA instructions that may stay in flight
  if (n >= 6) wait_vmcnt<6>();
  else if (n == 4) wait_vmcnt<4>();
  else if (n == 2) wait_vmcnt<2>();
  else wait_vmcnt<0>();
}
#define PXA_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef GEMM_ABL
#define GEMM_ABL 0          // ablation bits (tools/build_variant.py; wrong results on purpose): 1 = every k-unit re-fetches the item's FIRST unit (cache-hot DMA), 2 = no bf16 epilogue,
#endif                      // 4 = the main loop's FLOPs issued as v_mfma_f32_16x16x32 on the quarters of each accumulator tile (what the other MFMA shape would buy)
                            // round 5, the two heavy epilogues taken apart: 8 = no bias-gradient column sums, 16 = no aux loads (x 1), 32 = the second output is
                            // computed and parked but never stored, 64 = no GELU arithmetic (both outputs carry the pre-activation)
#if GEMM_ABL & 4
__device__ __forceinline__ void mfma_abl16(f32x4 (&q)[4], bf16x8 b0, bf16x8 a0, bf16x8 b1, bf16x8 a1) {
  q[0] = mfma16(b0, a0, q[0]); q[1] = mfma16(b1, a1, q[1]); q[2] = mfma16(b0, a1, q[2]); q[3] = mfma16(b1, a0, q[3]);
}
#endif
#ifndef GEMM_STORE_OVERLAP
#define GEMM_STORE_OVERLAP 1   // next item's main loop starts over the draining epilogue stores (counted vmcnt); 0 = wait for them (A/B)
#endif
#ifndef GEMM_M16
#define GEMM_M16 3          // bit per layout (1 NT, 2 NN, 4 TN): main loop on v_mfma_f32_16x16x32; a cleared bit keeps that layout on 32x32x16.  TN measured
                            // 2-4 % SLOWER in the 16-row shape (its read phase - 24 transpose reads + ~39 address VALU per k-unit - no longer fits into the issue
                            // slots 32 back-to-back 16-cycle MFMAs of the partner wave leave: 3 per MFMA instead of 7), so it stays on 32x32x16
#endif
#ifndef GEMM_NT_STORE
#define GEMM_NT_STORE 0     // bf16 output rows of the persistent kernel as non-temporal stores (A/B builds)
#endif
#ifndef GEMM_M16_PRIO
#define GEMM_M16_PRIO 1     // s_setprio 1 around the 16-row matrix phases of the NT instances (0: none; A/B builds).  NN measured +1.2 % WITHOUT it (profiles/r03h_kbench_gemm_variants.txt)
#endif
#ifndef GEMM_STAGGER
#define GEMM_STAGGER 0      // > 0 (A/B builds): workgroups on odd CU slots of their XCD start GEMM_STAGGER x 10 ns late (s_memrealtime, 100 MHz), so that the two halves
#endif                      // of the chip reach their epilogues - 64 to 128 MB of stores chip-wide per round of items - half an item apart instead of together
#ifndef GEMM_EPI_EARLY
#define GEMM_EPI_EARLY 1    // bias add in front of the prefetch + asm aux loads with counted waits (see the hand-over of gemm_pers_kernel); 0 = the round-4 order (A/B builds)
#endif
#ifndef GEMM_PHASE16
#define GEMM_PHASE16 -1     // matrix phases per k-unit: -1 = per layout (below), 0 = two 8-MFMA phases everywhere, 1 = one 16-MFMA phase everywhere
#endif

// =====================================================================================================================
// Persistent ping-pong kernel for the big token GEMMs with bf16 outputs (NT forward, NN dX): one 512-thread workgroup per CU
// walks the XCD-aware tile order; 256x256 tile, 8 waves (2 x 4, 128x64 each), k-units of 32 in a 4-deep LDS ring (4 x 32 KiB)
// plus a 32 KiB epilogue staging area (8 x 4 KiB wave-private) = the CU's whole 160 KiB.
// Main loop.  A 512-thread workgroup puts two waves on every SIMD; left alone they leave each barrier in lockstep, so their
// LDS-read phases coincide and the matrix pipe idles while fragments arrive.  Every k-unit is two phases
//   [R: ds_reads of this phase's fragments + 2 LDS-DMA instructions] barrier [M: 8 MFMAs under s_setprio 1] barrier
// and the second wave of each SIMD (waves 4-7, the wm = 1 row) runs ONE barrier late: in every barrier interval one wave of a
// SIMD is in M while its partner is in R (guide "8-phase" / T3-T5 schedule).
// DMA bookkeeping (per wave, in issue order): unit u = A part (2 instructions, issued in R of phase b of unit u-3) then B part
// (2, issued in R of phase a of unit u-2).  The only in-loop wait is a COUNTED vmcnt in phase b of unit t: everything up to unit
// t+1 has landed while unit t+2 and half of t+3 (6 instructions) stay in flight across the barriers; two barriers separate it
// from the first read of unit t+1, for both wave groups.  A ring slot is refilled at the earliest three barriers after the late
// group's last read of it.
// Tile hand-over.  With K = 1152 a tile's main loop is only ~30 us, and an epilogue that drains 128 KiB of stores before the
// workgroup retires (then a fresh workgroup waits ~2 us for its first operands) costs 13-40 % of the GEMM.  Here, after the
// last k-unit, a wave first issues the NEXT tile's first 2.5 k-units of DMA (the ring is idle), then converts its accumulators
// 32 rows at a time through its private staging slice into full-row 16-byte stores - which are fire-and-forget - and waits once
// (vmcnt(0): its stores and the prefetched units) before the next main loop.  Store drain and operand latency of consecutive
// tiles overlap; bias / GELU / GELU' / second (pre-activation) output / bias-gradient column sums all ride in that epilogue.
// Geometry: 256 x 256 block tile, 8 waves as 2 (m) x 4 (n), wave tile 128 x 64.  (A 192 x 384 tile - same 128 FLOP per operand
// byte, divides every DiT width - was measured 5-20 % SLOWER: 144 accumulators leave the epilogue spilling and its phases must
// split k, not rows.)  A k-unit is 32-40 one-KiB DMA pieces; wave w owns pieces w, w + 8, ...: the first two are issued in phase b
// of unit u - 3, the rest in phase a of unit u - 2.
// Half-width remainder columns (N = 1152 = 4.5 x 256: every projection back to the model width) are not padded to a fifth tile
// column that wastes half its MFMAs: the remainder columns of TWO consecutive m-tiles form one "paired" item - 512 rows x 128
// columns, the same 65,536 outputs, the same 128 x 64 per wave (waves wn = 0,1 take the first m-tile, wn = 2,3 the second) - whose
// k-unit is 32 A pieces + 8 B pieces (40 KiB, hence the 40 KiB ring slots).
// Bank swizzle of the k-contiguous images (64-byte rows, 4 chunks): chunk' = chunk ^ swz(row), applied to the DMA's per-lane SOURCE address
// and mirrored by the fragment reads.  ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32):
//   32-row fragments (lane = row & 31, chunk 2 ks + hi): swz = (row >> 2) & 3;
//   16-row fragments (lane = row & 15, chunk lane >> 4): a group holds rows r, r + 12 with one chunk and r + 4, r + 8 with the next, so the
//   four rows of a residue class mod 4 need the XOR pattern (0, 0, 3, 3): swz = 3 ((row >> 3) & 1).  Both conflict-free (checked per group).
template <bool M16>
__device__ __forceinline__ int kc_swz(int row) { return M16 ? ((row >> 3) & 1) * 3 : (row >> 2) & 3; }
template <bool KC, bool M16 = false>
__device__ __forceinline__ const bf16_t* piece_ptr_rt(const bf16_t* __restrict__ X, int ld, int r0a, int r0b, int R, int q, int rows_log2) {
  // q: 16-byte chunk index in the LDS image of one operand of a k-unit; image rows/columns [0, 256) come from r0a, [256, 512) from r0b
  if constexpr (KC) {
    const int row = q >> 2, c = (q & 3) ^ kc_swz<M16>(row);
    const int gr = (row < 256 ? r0a : r0b - 256) + row;
    return X + (size_t)min(gr, R - 1) * ld + c * 8;
  } else {
    // k-strided image [32 k][rows]: 64-byte block' = block ^ (k & 3) puts the 4 k-rows of a transpose read on the 4 bank quarters.  16-row
    // fragments read the same 32 bytes of rows k and k + 8 from the two halves of a 32-lane LDS group: their images also swap the 32-byte
    // halves of a block where k has bit 3 set.
    const int sh = rows_log2 - 3, kr = q >> sh, cl = q & ((1 << sh) - 1);
    const int c = ((((cl >> 2) ^ (kr & 3)) << 2) | ((cl & 3) ^ (M16 ? ((kr >> 3) & 1) << 1 : 0)));
    const int col = c * 8;
    int gc = (col < 256 ? r0a : r0b - 256) + col;
    gc = gc < R ? gc : 0;
    return X + (size_t)kr * ld + gc;
  }
}
template <bool KC>
__device__ __forceinline__ bf16x8 frag_rt(const char* lds, int rbase, int ks, int lane, int rows_log2) {
  if constexpr (KC) {
    return frag_p<true, 256, 32>(lds, rbase, ks, lane);           // 64-byte rows whatever the row count
  } else {
    const int gg = lane >> 4, tt = lane & 15, hi2 = gg >> 1;
    const int kr = ks * 16 + 8 * hi2 + (tt >> 2), col = rbase + 16 * (gg & 1) + (tt & 3) * 4;
    const int blk = col >> 5, inblk = (col & 31) * 2;
    const char* p0 = lds + (kr << (rows_log2 + 1)) + ((blk ^ (kr & 3)) << 6) + inblk;
    const char* p1 = lds + ((kr + 4) << (rows_log2 + 1)) + ((blk ^ ((kr + 4) & 3)) << 6) + inblk;
    return concat_tr(lds_tr_read(p0), lds_tr_read(p1));
  }
}

// Operand of v_mfma_f32_16x16x32 from a k-contiguous image: rows rbase .. rbase + 15, the k-unit's whole depth of 32 (lane l: row l & 15,
// k = 8 (l >> 4) .. + 7).
__device__ __forceinline__ bf16x8 frag16_kc(const char* lds, int rbase, int lane) {
  const int row = rbase + (lane & 15);
  return *reinterpret_cast<const bf16x8*>(lds + row * 64 + (((lane >> 4) ^ kc_swz<true>(row)) << 4));
}
// The same operand from a k-strided image [32 k][2^rows_log2]: rows (of the operand) rbase .. rbase + 15 are 16 columns of the image; lane group
// G = l >> 4 takes k = 8 G .. 8 G + 7 - the order the k-contiguous partner's chunk G has - as two transpose reads of [4 k][16] blocks.
template <bool KC>
__device__ __forceinline__ bf16x8 frag16(const char* lds, int rbase, int lane, int rows_log2) {
  if constexpr (KC) {
    return frag16_kc(lds, rbase, lane);
  } else {
    const int gg = lane >> 4, tt = lane & 15;
    const int kr = 8 * gg + (tt >> 2), col = rbase + (tt & 3) * 4;
    const int blk = col >> 5, inblk = ((col & 31) * 2) ^ ((gg & 1) << 5);       // (kr >> 3) & 1 == gg & 1, also for kr + 4
    const char* p0 = lds + (kr << (rows_log2 + 1)) + ((blk ^ (kr & 3)) << 6) + inblk;
    const char* p1 = lds + ((kr + 4) << (rows_log2 + 1)) + ((blk ^ ((kr + 4) & 3)) << 6) + inblk;
    return concat_tr(lds_tr_read(p0), lds_tr_read(p1));
  }
}

// EPI (bf16 epilogue flavour, compiled separately so that none carries the others' registers - the epilogue runs with all
// 128 accumulators live and spills at the slightest extra state): 0 = (+bias), 1 = act 3 (bias + GELU, GELU' as the second
// output), 2 = act 4 (x aux) + bias-gradient column sums, 3 = everything decided at run time (acts 1 / 2 and odd mixes), 7 = act 1 (bias + GELU,
// one output: the fc1 of a forward whose backward never runs - inference, the discarded forward of a checkpointed step).
// Dynamic item scheduler of the bf16 (NT / NN) instances.  A static "workgroup b takes items b, b + 256, ..." split makes the kernel
// twice as long whenever another kernel (an RCCL all-reduce overlapping the backward) holds a few CUs: the workgroups that could not
// start run their whole list after the others have finished.  Instead every XCD's contiguous item range has an atomic cursor; a
// workgroup takes the next item of its own XCD's range (same L2 locality as the static order) and, when that is exhausted, of the
// other XCDs' ranges.  The cursor fetch for item i+2 is issued at the hand-over i -> i+1 and resolved after the next vmcnt(0), so its
// latency is never waited for.  Cursors live in a 64-slot global table (one slot per launch, round robin); the last workgroup to
// retire zeroes its slot.
#ifndef GEMM_TRACE
#define GEMM_TRACE 0        // diagnostics build (tools/build_variant.py): workgroup 8 / wave 0 records s_memtime at 6 points of its first 12 items
#endif
#if GEMM_TRACE
__device__ unsigned long long g_trace[12][8];
#define PXA_TR(k) do { if (blockIdx.x == 8 && tid == 0 && tr_i < 12) g_trace[tr_i][k] = __builtin_readcyclecounter(); } while (0)
#else
#define PXA_TR(k) do {} while (0)
#endif
__device__ unsigned g_sched[64][16];                   // [slot][0..7] per-XCD cursors, [8] retired workgroups
__device__ unsigned g_sched_word[64][512];             // paired-tile instances (no spare LDS): per-workgroup broadcast word

// RM: how a half-width remainder column (0 < N % 256 <= 128) is handled; compiled in only where it is used, because the run-time
// geometry costs the plain path 3 %.
//   RM = 1 (fp32 weight gradients): PAIRED items, as described above - with split-K the item count is free to fill the CUs, so the
//           saved half tile is saved time (dW of qkv / fc1: -12 %).
//   RM = 2 (bf16 NT / NN): HALF items - one m-tile's remainder columns alone, 256 x 128 outputs on all 8 waves as 4 x 2 waves of
//           64 x 64 (one row tile per phase: 8 MFMAs per k-unit and wave instead of 16).  Pairing would not help here: 256 CUs run
//           whole rounds of equal items, 4.5 rounds of work still take 5; a round of half-duration items under the dynamic
//           cursors makes it ~4.6.
// SEG: segmented-K A operand (see gemm_glds_kernel): the running A pointers take an extra p.seg_jump every p.k_seg elements.
template <int LAYOUT, int EPI, int RM, bool SEG = false>
__global__ __launch_bounds__(512) void gemm_pers_kernel(GemmParams p) {
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int NW = 8, TM = 4, TN = 2, BKT = 32, H1 = 2;
  // One 16-MFMA matrix phase per k-unit (2 barriers) instead of two 8-MFMA phases (4 barriers) wherever an operand is read through the
  // LDS transpose (NN: B, TN: both): those read phases are twice as long in instructions and did not fit under the partner's 8 MFMAs.
  // Measured at M = 65,536 (profiles/r02_gemm_phase16.txt): TN +18-20 %, NN +3-6 %, NT +-0 (keeps the finer interleave).
  constexpr bool PH16 = GEMM_PHASE16 < 0 ? (LAYOUT != 0) : (GEMM_PHASE16 != 0);
  // MFMA shape.  Under the chip's power limit an MFMA stream of 16x16x32 instructions on N(0,1) operands runs 12 % faster than the same FLOPs as
  // 32x32x16 (half the accumulator words through the register file per FLOP; probe/mfma_power.hip), and this kernel with its main-loop FLOPs
  // re-issued in that shape (-DGEMM_ABL=4) measured NT +8-14 %, NN +2-8 %, TN +2-5 % (profiles/r02c_gemm_mfma16_ablation.txt); run for real:
  // NT +8-11 %, NN +2-7 %, the VAE's implicit convolutions -10 % decode time, TN -2-4 % (profiles/r02d_gemm_nt16_ab.txt, r02e_gemm_m16_ab.txt).  Wave tile 128 x 64 = 8 x 4 tiles of 16 x 16, the n side still the MFMA's A operand, so a lane
  // owns 4 consecutive columns of one output row (lane (R, c): row 16 im + c, columns 16 jn + 4 R + g).  Fragment counts per k-unit are those of
  // the 32-row form: 12 ds_read_b128 (k-contiguous operands) or transpose-read pairs (k-strided), each now the unit's whole depth of 32.
  constexpr bool M16 = ((GEMM_M16 >> LAYOUT) & 1) && !(GEMM_ABL & 4);
  constexpr int UNIT = 40960;                          // ring slot: A image at 0 (16 KiB; 32 KiB paired), B image behind it (16 KiB; 8 KiB paired)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3, hi = lane >> 5;
  const bool late = wave >= NW / 2;
  // epilogue staging: 4 KiB per wave inside ring slot 3, which the next item's prefetch (slots 0, 1 and part of 2) leaves alone
  // until every wave has passed the barrier that opens the next main loop
  char* stg = smem + 3 * UNIT + wave * 4096;
  // work items: output tile (or paired half-width tile) x k-slice (split-K only for the fp32 weight-gradient layout), XCD-aware order
  const int mt = (p.M + 255) / 256, ntf = p.N / 256, remn = p.N - ntf * 256;
  constexpr bool PAIR = (RM == 1), HALF = (RM == 2);
  const bool remcol = RM != 0 && remn > 0 && remn <= 128;   // PAIR with an odd m-tile count: the last pair's second half lies beyond M (loads clamp, stores are guarded)
  const int ntp = remcol ? ntf : (p.N + 255) / 256;    // n-tile columns handled as (possibly padded) full tiles
  const int rem_pg = remcol ? (PAIR ? 4 : 8) : 0;      // remainder items per group of 8 m-tiles
  const int per_group = 8 * ntp + rem_pg, tiles = mt * ntp + (remcol ? (PAIR ? (mt + 1) / 2 : mt) : 0), T = tiles * p.split;
  auto units_of = [&](int z) { return (min(p.K, (z + 1) * p.k_per_split) - z * p.k_per_split) / BKT; };   // >= 2: k ranges are multiples of 64

  // current / prefetched item: row origins of the (two) A row blocks, column origin, paired flag, k-slice
  int m0a = 0, m0b = 0, n0 = 0, z_ = 0;
  bool vt = false;
  // workgroup b runs on XCD b % 8; every XCD owns a contiguous range [xs(x), xs(x) + xc(x)) of the grouped (8 m-tiles), k-slice
  // major item order
  const int xq = T / 8, xr = T % 8;
  auto xs = [&](int x) { return x * xq + min(x, xr); };
  auto xc = [&](int x) { return xq + (x < xr ? 1 : 0); };
  auto locate_g = [&](int Lg) {                        // position in that order -> coordinates
    z_ = Lg / tiles;
    const int t = Lg - z_ * tiles;
    const int g = t / per_group, in_g = t - g * per_group, first_m = g * 8, gsz = min(mt - first_m, 8), nfull = gsz * ntp;
    if (in_g < nfull) { vt = false; m0a = (first_m + in_g % gsz) * 256; m0b = m0a; n0 = (in_g / gsz) * 256; }
    else if (PAIR) { vt = true; m0a = (first_m + 2 * (in_g - nfull)) * 256; m0b = m0a + 256; n0 = ntf * 256; }
    else if (HALF) { vt = true; m0a = (first_m + (in_g - nfull)) * 256; m0b = m0a; n0 = ntf * 256; }
  };
  // static order: workgroup-strided index.  p.desc (round 5): every XCD walks its contiguous range from the END - the grouped order puts the token rows' tiles in
  // ascending order, so a launch behind a producer that swept the rows upwards starts on the rows still in the Infinity Cache (-1.7 ms per training step,
  // profiles/r5_16_step_ab_reverse.txt); the same for the dynamic cursors
  const bool rev = LAYOUT != 2 && p.desc != 0;
  auto pos_of = [&](int x, int i) { return xs(x) + (rev ? xc(x) - 1 - i : i); };
  auto locate = [&](int L) { locate_g(pos_of(L % 8, L / 8)); };
  const bool DYN = p.sched_slot >= 0;                  // slot < 0: static workgroup-strided split (A/B experiments)
  const int myx = blockIdx.x % 8;
  unsigned* sched = g_sched[p.sched_slot & 63];
  // the fetched position travels from wave 0 to the others through a word the DMA never touches: bytes 32..40 KiB of ring slot 0
  // are unused without pairing; the paired instances have no spare LDS and use a per-workgroup word in global memory instead
  unsigned* sched_lds = reinterpret_cast<unsigned*>(smem + 32768);
  unsigned* sched_glb = &g_sched_word[p.sched_slot & 63][blockIdx.x & 511];
  auto publish = [&](int v) {                          // wave 0 / lane 0 writes, then the caller's barrier
    if (PAIR) { if (lane == 0) __hip_atomic_store(sched_glb, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wait_vmcnt<0>(); }
    else { if (lane == 0) sched_lds[0] = (unsigned)v; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  };
  auto receive = [&]() -> int {
    const unsigned v = PAIR ? __hip_atomic_load(sched_glb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : sched_lds[0];
    return __builtin_amdgcn_readfirstlane((int)v);
  };
  unsigned pend = 0u;                                  // wave 0 / lane 0: cursor value of the fetch in flight
  int nxt = -1;                                        // the item after the current one (position in the order), -1: none
  // (asm: the compiler put `s_waitcnt vmcnt(0)` right behind its own atomic - wave 0 then waited for the whole prefetch it had just issued; the value is
  // only read in fetch_resolve, behind the top-of-loop wait that covers it: the fetch is older than every epilogue store that wait leaves in flight)
  auto fetch_issue = [&]() {
    if (wave == 0 && lane == 0) {
      unsigned* cur = &sched[myx];
      const unsigned one = 1u;
      asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(pend) : "v"(cur), "v"(one) : "memory");
    }
  };
  auto fetch_resolve = [&]() -> int {                  // wave 0 only; the own-XCD fetch has returned (a vmcnt wait that covers it was executed)
    int res = -1;
    asm volatile("" : "+v"(pend));
    if (lane == 0) {
      if ((int)pend < xc(myx)) res = pos_of(myx, (int)pend);
      else
        for (int k = 1; k < 8 && res < 0; k++) {       // own range exhausted: take from the others (blocking; only at a kernel's tail)
          const int x2 = (myx + k) & 7;
          const unsigned i2 = atomicAdd(&sched[x2], 1u);
          if ((int)i2 < xc(x2)) res = pos_of(x2, (int)i2);
        }
    }
    return __builtin_amdgcn_readfirstlane(res);
  };
  int L = blockIdx.x;
#if GEMM_STAGGER
  if (LAYOUT != 2 && ((blockIdx.x >> 3) & 1)) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)GEMM_STAGGER) __builtin_amdgcn_s_sleep(32);
  }
#endif
  if (DYN) {
    fetch_issue();
    wait_vmcnt<0>();
    if (wave == 0) publish(fetch_resolve());
    __syncthreads();
    const int first = receive();
    __syncthreads();                                   // everybody has read the word before it is reused
    if (first < 0) {                                   // nothing left for this workgroup (it started late): retire
      if (wave == 0 && lane == 0 && atomicAdd(&sched[8], 1u) == gridDim.x - 1)
        for (int k = 0; k < 9; k++) sched[k] = 0u;
      return;
    }
    locate_g(first);
    fetch_issue();                                     // the item after it
  } else {
    locate(L);
  }
  int nk = units_of(z_), nk_pf = nk;                   // nk_pf: units of the item being prefetched
  // running source pointers, per-unit strides (elements) and LDS offsets of this wave's (up to) 5 DMA pieces per k-unit
  const bf16_t* pp[5];
  long st[5];
  int dof[5];
  const long stepA = A_KC ? BKT : (long)BKT * p.lda, stepB = B_KC ? BKT : (long)BKT * p.ldb;
  int s_lo = 0, s_hi = 0;                              // ring slots of the next first-half / second-half issue
  bool vt_pf = false;                                  // paired flag of the item whose DMA is being issued
  int dma_count = 0;                                   // LDS-DMA instructions this wave has issued since the last prefetch() began (wave-uniform; what the epilogue's
                                                       // counted waits must allow in flight - counted where they are issued, not re-derived: ADVICE r05)
  auto piece = [&](int i, int slot) {
    dma_count++;
    lds_dma16(pp[i], smem + slot * UNIT + dof[i]);
#if !(GEMM_ABL & 1)
    pp[i] += st[i];
#endif
  };
  // SEG: the A pieces' walk over the segments / taps.  State 0 belongs to pieces 0, 1 (issued by issue_lo), state 1 to pieces 2, 3 of a PAIRED item (its
  // other two A pieces, issued by issue_hi half a unit later): the same walk, one issue apart.  (Scalars and references, not arrays indexed by a lambda
  // parameter: with those the compiler kept pp[] / dof[] in SCRATCH - 768 bytes of private segment, the LDS offset in a VGPR, the kernel 16x slower.)
  int seg_left0 = 0, seg_left1 = 0;                    // k elements left in the A segment the next issue reads
  int tap_kx0 = 0, tap_ky0 = 0, tap_half0 = 0, tap_kx1 = 0, tap_ky1 = 0, tap_half1 = 0;   // tap-interleaved order: position of the next issue
  // (the walk is written out in both issue lambdas, and both are always_inline: as one more lambda called from the two, the inliner gave up on the PAIR && SEG
  // instances and the closure - kernel arguments, item state, pp[] / dof[] - went to scratch: 816 bytes of private segment, the kernel 16x slower)
#define PXA_SEG_ADVANCE(seg_left, tap_kx, tap_ky, tap_half, pa, pb)                                                          \
  do {                                                                                                                       \
    if (p.k_tap) {                                     /* tap-interleaved order: 64 channels = two k-units per tap */        \
      if (tap_half) {                                                                                                        \
        long adj;                                                                                                            \
        if (tap_kx < 2) { adj = p.k_tap - 64; tap_kx++; }                                                                    \
        else if (tap_ky < 2) { adj = p.tap_s - 2L * p.k_tap - 64; tap_kx = 0; tap_ky++; }                                    \
        else { adj = -2L * p.tap_s - 2L * p.k_tap; tap_kx = 0; tap_ky = 0; }                                                 \
        pa += adj; pb += adj;                                                                                                \
      }                                                                                                                      \
      tap_half ^= 1;                                                                                                         \
    } else {                                                                                                                 \
      seg_left -= BKT;                                                                                                       \
      if (seg_left == 0) { pa += p.seg_jump; pb += p.seg_jump; seg_left = p.k_seg; }                                         \
    }                                                                                                                        \
  } while (0)
  auto issue_lo = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < H1; i++) piece(i, s_lo);
    if (SEG) PXA_SEG_ADVANCE(seg_left0, tap_kx0, tap_ky0, tap_half0, pp[0], pp[1]);      // pieces 0 and 1 are A pieces of every SEG item
    s_lo = (s_lo + 1) & 3;
  };
  auto issue_hi = [&]() __attribute__((always_inline)) {
    piece(2, s_hi);
    if (!(HALF && vt_pf)) piece(3, s_hi);
    if (PAIR && vt_pf) piece(4, s_hi);
    if (SEG && PAIR) { if (vt_pf) PXA_SEG_ADVANCE(seg_left1, tap_kx1, tap_ky1, tap_half1, pp[2], pp[3]); }   // paired item: pieces 2 and 3 are its other two A pieces
    s_hi = (s_hi + 1) & 3;
  };
#undef PXA_SEG_ADVANCE
  auto prefetch = [&]() __attribute__((always_inline)) {   // units 0, 1 and the first half of unit 2 of the item at (m0a, m0b, n0, z_)
    vt_pf = (RM != 0) && vt;
    const long k0 = (long)z_ * p.k_per_split;
    const bool vq = PAIR && vt, hq = HALF && vt;
    const int rla = vq ? 9 : 8, rlb = (vq || hq) ? 7 : 8;   // log2 of the A / B image row (column) counts
    const int ln = opaque<LAYOUT == 1 || (EPI >= 2 && EPI != 7) || (GEMM_EPI_EARLY && RM == 2)>(lane);
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int id = wave + NW * i;                    // full tile: 16 A + 16 B pieces; paired: 32 A + 8 B
      const bool is_a = vq ? (i < 4) : (i < 2);                      // full: A A B B -, paired: A A A A B, half: A A B - -
      const int pid = is_a ? id : (vq ? id - 32 : id - 16);
      if (vq || (hq ? i < 3 : i < 4)) {
        if (is_a) pp[i] = piece_ptr_rt<A_KC, M16>(p.A, p.lda, m0a, m0b, p.M, pid * 64 + ln, rla) + (A_KC ? k0 : k0 * p.lda);
        else pp[i] = piece_ptr_rt<B_KC, M16>(p.B, p.ldb, n0, n0, p.N, pid * 64 + ln, rlb) + (B_KC ? k0 : k0 * p.ldb);
      }
      st[i] = is_a ? stepA : stepB;
      dof[i] = (is_a ? 0 : (vq ? 32768 : 16384)) + pid * 1024;
    }
    s_lo = s_hi = 0;
    dma_count = 0;
    if (SEG) { seg_left0 = seg_left1 = p.k_seg; tap_kx0 = tap_kx1 = tap_ky0 = tap_ky1 = tap_half0 = tap_half1 = 0; }
    issue_lo(); issue_hi(); issue_lo(); issue_hi();
    if (nk_pf > 2) { issue_lo(); if (PH16) issue_hi(); }      // PH16: three whole units ahead; else 2.5
  };
  prefetch();
  int tr_i = 0; (void)tr_i;
  int prev_stores = 0;                                 // epilogue store instructions of the previous item still allowed in flight (0: none)

  while (true) {
    f32x16 acc[TM][TN];                                // 32 x 32 tiles; M16 instances use acc4 instead (the unused set costs nothing)
    f32x4 acc4[2 * TM][2 * TN];                        // 16 x 16 tiles: [m tile][n tile] of the wave's 128 x 64
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * TM; i++)
#pragma unroll
      for (int j = 0; j < 2 * TN; j++) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if GEMM_ABL & 4
    f32x4 a4[TM][TN][4];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) a4[i][j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    // this item's fragment geometry (the prefetch of the next item will overwrite vt / m0a / ...)
    const bool vtc = PAIR && vt, hfc = HALF && vt;
    const int a_rb = hfc ? (wave >> 1) * 64 : (vtc ? (wn >> 1) * 256 : 0) + wm * 128, b_rb = (vtc || hfc) ? (wn & 1) * 64 : wn * 64;
    const int rla = vtc ? 9 : 8, rlb = (vtc || hfc) ? 7 : 8, boff = vtc ? 32768 : 16384;
    const int mw = hfc ? m0a + a_rb : ((vtc && (wn >> 1)) ? m0b : m0a) + wm * 128, nw = n0 + b_rb;   // this wave's output origin
    const int tm_eff = hfc ? 2 : TM;                    // row tiles of this wave's output
    const int zw = z_;                                  // this item's k-slice (fp32 slab index)
    PXA_TR(0);
    // This item's first units (and the cursor fetch) have landed once only the previous item's epilogue stores - issued after them, and
    // vmcnt retires in issue order - are still in flight: interior tiles of the plain / dual-output bf16 epilogues issue exactly
    // `prev_stores` of them per wave (4 per 32-row slice and output), so the main loop starts while 128 KiB of stores drain instead of
    // behind them (K = 1152 tiles: the store drain was most of a 16 % epilogue cost).  Anything else waits for everything.
    if (GEMM_STORE_OVERLAP && LAYOUT != 2 && EPI != 3) {
      if (prev_stores == 40) wait_vmcnt<40>();
      else if (prev_stores == 36) wait_vmcnt<36>();
      else if (prev_stores == 32) wait_vmcnt<32>();
      else if (prev_stores == 28) wait_vmcnt<28>();
      else if (prev_stores == 24) wait_vmcnt<24>();
      else if (prev_stores == 20) wait_vmcnt<20>();
      else if (prev_stores == 16) wait_vmcnt<16>();
      else if (prev_stores == 12) wait_vmcnt<12>();
      else if (prev_stores == 8) wait_vmcnt<8>();
      else wait_vmcnt<0>();
    } else {
      wait_vmcnt<0>();                                 // this item's first units have landed; last item's stores are out
    }
    if (DYN && wave == 0) {                            // ... and so has the cursor fetch issued at the last hand-over
      publish(fetch_resolve());
    }
    __builtin_amdgcn_s_barrier();
    if (DYN) nxt = receive();
    if (late) __builtin_amdgcn_s_barrier();            // stagger the second wave of each SIMD by one barrier interval
    PXA_TR(1);
    // ---- epilogue operands loaded through asm statements with counted waits (round 5; the reasons are at the hand-over below.  Requesting them two k-units
    // before the end of the main loop was tried and dropped: 36 more live registers there spill 46-82 registers in every aux flavour)
    constexpr bool EARLY = GEMM_EPI_EARLY && M16 && LAYOUT != 2;      // (the 32-row accumulator layout exists for A/B builds only and keeps the round-4 order)
    constexpr bool AUXA = EARLY && (EPI == 2 || EPI == 4 || EPI == 6) && !(GEMM_ABL & 16);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 axa[2][2][4];                                // AUXA: aux of two 32-row slices, [slice parity][m tile of the slice][n tile]
    f32x4 bq[2 * TN];                                  // EARLY: the wave's 64 bias values in accumulator layout
    const bool has_bias = EARLY && p.bias != nullptr;  // wave-uniform
    auto bias_issue = [&]() {                          // asm: no compiler-made wait; bias_apply runs behind a counted one
      const int R4b = lane >> 4;
#pragma unroll
      for (int jn = 0; jn < 2 * TN; jn++) {
        const float* src = p.bias + min(nw + jn * 16 + 4 * R4b, p.N - 4);      // columns beyond N: clamped, never stored
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[jn]) : "v"(src) : "memory");
      }
    };
    auto aux_issue = [&](int i, u32x2 (&dst)[2][4]) {  // rows / columns beyond M / N are clamped: their products are never stored or summed
      const int R4a = lane >> 4, c16a = lane & 15;
#pragma unroll
      for (int mh = 0; mh < 2; mh++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
          const int m = min(mw + i * 32 + 16 * mh + c16a, p.M - 1), n = min(nw + jn * 16 + 4 * R4a, p.N - 4);
          const bf16_t* src = p.aux + (size_t)m * p.ldaux + n;
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst[mh][jn]) : "v"(src) : "memory");
        }
    };
    // one k-unit = ONE read phase + ONE matrix phase of 16 MFMAs (two barriers per unit instead of four: half as many hand-overs of the
    // matrix pipe between the two waves of a SIMD).  R: all 12 fragment reads of unit t, the 4 (5 / 3) LDS-DMA pieces of unit t+3, the
    // counted wait (unit t+1 landed; t+2, t+3 in flight) and lgkmcnt(0) - reads are COMPLETE at the barrier, so one barrier separates
    // the late group's last read of a ring slot from its refill by the early group.  REM = units that follow this one.
    auto unit16 = [&](int t, auto rem_c, auto th_c) {
      constexpr int REM = decltype(rem_c)::value, TH = decltype(th_c)::value;   // 2 * TH row tiles (4; 2 for half items)
      const char* sA = smem + (t & 3) * UNIT;
      const char* sB = sA + boff;
      if constexpr (M16) {
        bf16x8 bq[2 * TN], aq[4 * TH];
#pragma unroll
        for (int j = 0; j < 2 * TN; j++) bq[j] = frag16<B_KC>(sB, b_rb + j * 16, lane, rlb);
#pragma unroll
        for (int i = 0; i < 4 * TH; i++) aq[i] = frag16<A_KC>(sA, a_rb + i * 16, lane, rla);
        if (REM >= 3) { issue_lo(); issue_hi(); }
        if (REM >= 3) { if (vtc) wait_vmcnt<10>(); else if (TH == 1) wait_vmcnt<6>(); else wait_vmcnt<8>(); }
        else if (REM == 2) { if (vtc) wait_vmcnt<5>(); else if (TH == 1) wait_vmcnt<3>(); else wait_vmcnt<4>(); }
        else if (REM == 1) wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
        if (GEMM_M16_PRIO && LAYOUT != 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4 * TH; i++)
#pragma unroll
          for (int j = 0; j < 2 * TN; j++) acc4[i][j] = mfma16(bq[j], aq[i], acc4[i][j]);
        if (GEMM_M16_PRIO && LAYOUT != 1) __builtin_amdgcn_s_setprio(0);
        PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
        return;
      }
      bf16x8 af[2][2 * TH], bf[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < TN; j++) bf[ks][j] = frag_rt<B_KC>(sB, b_rb + j * 32, ks, lane, rlb);
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 2 * TH; i++) af[ks][i] = frag_rt<A_KC>(sA, a_rb + i * 32, ks, lane, rla);
      if (REM >= 3) { issue_lo(); issue_hi(); }
      if (REM >= 3) { if (vtc) wait_vmcnt<10>(); else if (TH == 1) wait_vmcnt<6>(); else wait_vmcnt<8>(); }
      else if (REM == 2) { if (vtc) wait_vmcnt<5>(); else if (TH == 1) wait_vmcnt<3>(); else wait_vmcnt<4>(); }
      else if (REM == 1) wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
      __builtin_amdgcn_s_setprio(1);
#if GEMM_ABL & 4
#pragma unroll
      for (int i = 0; i < 2 * TH; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) mfma_abl16(a4[i][j], bf[0][j], af[0][i], bf[1][j], af[1][i]);
#else
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 2 * TH; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j] = mfma32(bf[ks][j], af[ks][i], acc[i][j]);
#endif
      __builtin_amdgcn_s_setprio(0);
      PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
    };
    // one k-unit = two phases; REM = units that follow it (3 = steady state: both DMA halves issued, 6-7 instructions left in flight)
    auto unit8 = [&](int t, auto rem_c, auto th_c) {
      constexpr int REM = decltype(rem_c)::value, TH = decltype(th_c)::value;   // TH row tiles per phase (2; 1 for half items)
      const char* sA = smem + (t & 3) * UNIT;
      const char* sB = sA + boff;
      auto rest_a = [&]() { if (REM >= 2) issue_hi(); };           // phase a: rest of unit t+2
      auto rest_b = [&]() {                                          // phase b: first pieces of unit t+3, then the counted wait
        if (REM >= 3) issue_lo();
        if (REM >= 3) { if (vtc) wait_vmcnt<7>(); else if (TH == 1) wait_vmcnt<5>(); else wait_vmcnt<6>(); }   // unit t+1 landed; t+2 and the start of t+3 in flight
        else if (REM == 2) { if (vtc) wait_vmcnt<5>(); else if (TH == 1) wait_vmcnt<3>(); else wait_vmcnt<4>(); }
        else if (REM == 1) wait_vmcnt<0>();
      };
      if constexpr (M16) {
        // the same two phases on 16 x 16 tiles: 4 n fragments (kept for phase b) + 2 TH m fragments per phase, each the unit's whole depth
        bf16x8 bq[2 * TN];
#pragma unroll
        for (int j = 0; j < 2 * TN; j++) bq[j] = frag16<B_KC>(sB, b_rb + j * 16, lane, rlb);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          bf16x8 aq[2 * TH];
#pragma unroll
          for (int i = 0; i < 2 * TH; i++) aq[i] = frag16<A_KC>(sA, a_rb + (h * 2 * TH + i) * 16, lane, rla);
          if (h == 0) rest_a(); else rest_b();
          PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
          if (GEMM_M16_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 2 * TH; i++)
#pragma unroll
            for (int j = 0; j < 2 * TN; j++) acc4[h * 2 * TH + i][j] = mfma16(bq[j], aq[i], acc4[h * 2 * TH + i][j]);
          if (GEMM_M16_PRIO) __builtin_amdgcn_s_setprio(0);
          PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
        }
        return;
      }
      // phases split the wave's rows: a = upper half x all columns (B fragments stay in registers for b = lower half)
      bf16x8 af[2][TH], bf[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < TN; j++) bf[ks][j] = frag_rt<B_KC>(sB, b_rb + j * 32, ks, lane, rlb);
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < TH; i++) af[ks][i] = frag_rt<A_KC>(sA, a_rb + (h * TH + i) * 32, ks, lane, rla);
        if (h == 0) rest_a(); else rest_b();
        PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
        __builtin_amdgcn_s_setprio(1);
#if GEMM_ABL & 4
#pragma unroll
        for (int i = 0; i < TH; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) mfma_abl16(a4[h * TH + i][j], bf[0][j], af[0][i], bf[1][j], af[1][i]);
#else
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int i = 0; i < TH; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[h * TH + i][j] = mfma32(bf[ks][j], af[ks][i], acc[h * TH + i][j]);
#endif
        __builtin_amdgcn_s_setprio(0);
        PXA_SB(); __builtin_amdgcn_s_barrier(); PXA_SB();
      }
    };
    auto unit = [&](int t, auto rem_c, auto th_c) {
      if constexpr (PH16) unit16(t, rem_c, th_c); else unit8(t, rem_c, th_c);
    };
    auto run_units = [&](auto th_c) {
      int t = 0;
      for (; t < nk - 3; t++) unit(t, IntC<3>{}, th_c);
      if (nk >= 3) unit(t++, IntC<2>{}, th_c);
      unit(t++, IntC<1>{}, th_c);
      unit(t++, IntC<0>{}, th_c);
    };
    if (HALF && hfc) run_units(IntC<1>{}); else run_units(IntC<2>{});
#if GEMM_ABL & 4
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) acc[i][j][g] = a4[i][j][g >> 2][g & 3];
#endif
    PXA_TR(2);
    if constexpr (LAYOUT != 2) {                       // the epilogue's own loads go out as soon as this wave's last matrix phase is issued (see the hand-over)
      if (has_bias) bias_issue();
      if constexpr (AUXA) { aux_issue(0, axa[0]); aux_issue(1, axa[1]); }
    }
    if (!late) __builtin_amdgcn_s_barrier();           // every wave has executed the same number of barriers; the ring is idle
    PXA_TR(3);

    // ---- hand-over: prefetch the next item's first units, then this item's epilogue
    bool more = false;
    int pf_ops = 0;                                    // LDS-DMA instructions this wave issues in the prefetch below (wave-uniform)
    auto hand_over = [&]() {
      if (DYN) {
        more = nxt >= 0;
        if (more) {
          locate_g(nxt);
          nk_pf = units_of(z_);
          prefetch();
          fetch_issue();                               // cursor fetch for the item after the next; resolved after the next vmcnt(0)
        }
      } else {
        L += gridDim.x;
        more = L < T;
        if (more) {
          locate(L);
          nk_pf = units_of(z_);
          prefetch();
        }
      }
      if (more) pf_ops = dma_count;                    // = 2 (H1 + hi_ops) + (nk_pf > 2 ? H1 + (PH16 ? hi_ops : 0) : 0), hi_ops = 2 (3 paired, 1 half): counted in piece().
                                                       // NOT in the count: wave 0 / lane 0's cursor atomic (fetch_issue, issued BEHIND the prefetch): one more operation
                                                       // in flight for that wave only, so its waits are conservative by one - never the other way.
    };
    const int le = opaque<LAYOUT == 1 || (EPI >= 2 && EPI != 7) || (GEMM_EPI_EARLY && RM == 2)>(lane);          // NN: epilogue-only lane constants are rebuilt per item (see opaque())
    const int srow = le & 31;
    if constexpr (LAYOUT == 2) {
      hand_over();
      PXA_TR(4);
      // fp32 weight-gradient tile: each 32 x 32 accumulator tile is parked in the wave's staging slice (128-byte rows, 16-byte
      // chunk XOR (row & 7)) and leaves as full 128-byte row segments: split-K slab store, read-modify-write into the gradient
      // (single k-slice: this workgroup owns the element) or plain store
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
          if constexpr (M16) {                           // the 32 x 32 block as 2 x 2 tiles: lane (R, c) -> row 16 mh + c, floats 16 nh + 4 R ..
#pragma unroll
            for (int mh = 0; mh < 2; mh++)
#pragma unroll
              for (int nh = 0; nh < 2; nh++) {
                const f32x4 a = acc4[2 * i + mh][2 * j + nh];
                const int row = 16 * mh + (lane & 15);
                *reinterpret_cast<float4*>(stg + row * 128 + (((4 * nh + (lane >> 4)) ^ (row & 7)) << 4)) = make_float4(a[0], a[1], a[2], a[3]);
              }
          } else {
#pragma unroll
          for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4*>(stg + srow * 128 + (((q * 2 + hi) ^ (srow & 7)) << 4)) =
                make_float4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
          for (int t4 = 0; t4 < 4; t4++) {
            const int row = t4 * 8 + (lane >> 3), ch = lane & 7;
            float4 v = *reinterpret_cast<const float4*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
            const int mm = mw + i * 32 + row, nn = nw + j * 32 + ch * 4;
            if (mm < p.M && nn < p.N) {
              if (p.accumulate == 3) {
                *reinterpret_cast<float4*>(p.slab + ((size_t)zw * p.M + mm) * p.N + nn) = v;
              } else {
                float* dst = p.outf + (size_t)mm * p.ldf + nn;
                if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(dst) = v;
              }
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);
        }
      nk = nk_pf;
      if (!more) break;
      continue;
    }
    const int act = (EPI == 0 || EPI == 5) ? 0 : EPI == 1 ? 3 : EPI == 2 ? 4 : (EPI == 4 || EPI == 6) ? 5 : EPI == 7 ? 1 : p.act;   // EPI 4 / 6: + aux (residual connection)
    constexpr bool want_st = (EPI == 5 || EPI == 6);   // + GroupNorm statistics of the output (implicit convolutions of the VAE)
    const bool dual = EPI == 1 || (EPI == 3 && ((p.act == 1 && p.out2 != nullptr) || p.act == 3));
    const bool want_cs = !(GEMM_ABL & 8) && (EPI == 2 || (EPI == 3 && p.colsum != nullptr));
    // Round 5: what the epilogue LOADS must not queue behind the next item's prefetch.  vmcnt retires in issue order and the compiler does not see
    // the asm LDS-DMA: a compiler-visible load issued behind the prefetch is waited for with `s_waitcnt vmcnt(0)` - i.e. every item's epilogue began
    // by waiting for the NEXT item's first k-units to land (the bias loads of every flavour; profiles/r5_01: the x aux flavour, whose four row slices
    // each waited like that, cost fc2's dX 240 us per launch over its plain twin, 215 of them gone with the aux loads ablated).  So (GEMM_EPI_EARLY):
    //   * the bias is added BEFORE the prefetch is issued (its wait then covers only itself), for every 16-bit flavour;
    //   * the aux flavours with a compiled-in activation (x aux: EPI 2; + aux: EPI 4 / 6) load aux through asm statements with hand-counted waits:
    //     issue order  L0 L1 | PREFETCH (pf_ops) | wait(L0) apply(0) | L2 | S0 | wait(L1) apply(1) | L3 | S1 | wait(L2) apply(2) | S2 | wait(L3) apply(3) | S3
    //     (L = the 8 loads of a 32-row slice, S = its 4 stores): a slice's aux is two slices ahead of its use, every wait names how many YOUNGER requests
    //     may stay in flight, and none of them makes the prefetch land.  Half items (2 slices): L0 L1 | PREFETCH | wait(L0) apply(0) | S0 | wait(L1) apply(1) | S1.
    auto bias_apply = [&]() {                          // behind a counted wait that covers the four loads
      asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
      for (int jn = 0; jn < 2 * TN; jn++)
#pragma unroll
        for (int mt = 0; mt < 2 * TM; mt++) acc4[mt][jn] += bq[jn];
    };
    auto bias_add = [&]() {                            // round-4 form: compiler-visible loads, waited for with vmcnt(0) wherever they are issued
      const int R4b = le >> 4;
      if constexpr (M16) {
#pragma unroll
        for (int jn = 0; jn < 2 * TN; jn++) {
          const int n = nw + jn * 16 + 4 * R4b;
          const float4 b4 = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int mt = 0; mt < 2 * TM; mt++) { acc4[mt][jn][0] += b4.x; acc4[mt][jn][1] += b4.y; acc4[mt][jn][2] += b4.z; acc4[mt][jn][3] += b4.w; }
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int n = nw + j * 32 + 8 * q + 4 * hi;
            const float4 b4 = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < TM; i++) { acc[i][j][q * 4] += b4.x; acc[i][j][q * 4 + 1] += b4.y; acc[i][j][q * 4 + 2] += b4.z; acc[i][j][q * 4 + 3] += b4.w; }
          }
      }
    };
    auto wait_allow = [&](int allowed) {               // at most `allowed` (wave-uniform) younger vector-memory operations stay in flight
      if (allowed >= 28) wait_vmcnt<28>();
      else if (allowed >= 26) wait_vmcnt<26>();
      else if (allowed >= 24) wait_vmcnt<24>();
      else if (allowed >= 22) wait_vmcnt<22>();
      else if (allowed >= 20) wait_vmcnt<20>();
      else if (allowed >= 16) wait_vmcnt<16>();
      else if (allowed >= 12) wait_vmcnt<12>();
      else if (allowed >= 10) wait_vmcnt<10>();
      else if (allowed >= 9) wait_vmcnt<9>();
      else if (allowed >= 8) wait_vmcnt<8>();
      else if (allowed >= 4) wait_vmcnt<4>();
      else wait_vmcnt<0>();
    };
    auto aux_ready = [&](int allowed, u32x2 (&a)[2][4]) {
      wait_allow(allowed);
      // (the registers pass through an asm statement behind the wait: no use of them can be scheduled in front of it)
      asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
    };
    auto aux_apply = [&](int i, const u32x2 (&a)[2][4]) {
#pragma unroll
      for (int mh = 0; mh < 2; mh++)
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
          float a0, a1, a2f, a3;
          unpack_bf16x2(a[mh][jn][0], a0, a1); unpack_bf16x2(a[mh][jn][1], a2f, a3);
          f32x4& v = acc4[2 * i + mh][jn];
          if (act == 5) { v[0] += a0; v[1] += a1; v[2] += a2f; v[3] += a3; }
          else { v[0] *= a0; v[1] *= a1; v[2] *= a2f; v[3] *= a3; }
        }
    };
    hand_over();                                       // (the prefetch's address arithmetic and its 10-12 issues run under the latency of the bias / slice 0, requested above)
    if constexpr (AUXA) {
      aux_ready(8 + pf_ops, axa[0]);                   // the bias (older) and slice 0 have landed; slice 1 and the prefetch may still be in flight
      if (has_bias) bias_apply();
      aux_apply(0, axa[0]);
      if (tm_eff > 2) aux_issue(2, axa[0]);
    } else {
      if (has_bias) { wait_allow(pf_ops); bias_apply(); }      // the prefetch, issued behind the bias loads, stays in flight
    }
    PXA_TR(4);
    // one column group of JW 32-wide tiles (JW = 2: 128-byte staging rows, 8 rows per store; JW = 1: 64-byte rows, 16 per store)
    auto emit = [&](int j0, auto jw_c) {
      constexpr int JW = decltype(jw_c)::value, RB = JW * 64;
      constexpr int LPR = JW * 4, RPI = 64 / LPR, NST = 4 / (3 - JW);   // read-back: JW = 2: 4 x (8 rows x 128 B); JW = 1: 2 x (16 rows x 64 B)
      static_assert(!M16 || JW == 2, "16 x 16 accumulator tiles: the wave's 64 columns are one column group");
      // M16: lane (R4, c16) holds row 16 mt + c16, columns 16 jn + 4 R4 + g of the wave tile; a 32-row slice is the two m tiles 2 i, 2 i + 1 and
      // a lane's 4 values are 8 bytes of staging chunk 2 jn + (R4 >> 1).  Everything behind the staging slice is shared with the 32 x 32 form.
      const int R4 = le >> 4, c16 = le & 15;
      // bias goes into the accumulators first, unconditionally (zero when absent), so nothing extra stays live across the row loop
      // (GEMM_EPI_EARLY: already done in front of the prefetch)
      if (!EARLY) bias_add();
      // bias-gradient column sums are taken from the read-back (row-wise) copy of the bf16 output: a lane keeps 8 running sums
      // for its 8-column chunk over all row slices; one 3-step lane tree at the end
      float cs[8], sq[2] = {0.f, 0.f};                 // statistics flavours: cs[0..1] / sq[0..1] = the chunk's two channel quads
#pragma unroll
      for (int e = 0; e < 8; e++) cs[e] = 0.f;
      // statistics: this wave's rows belong to ONE image (gn_img_rows is a multiple of 256); a row counts when its padded-grid
      // position is an interior pixel.  floor((pix + 0.5) / rp) in fp32 is exact for pix < 2^22.
      const int st_img = want_st ? mw / p.gn_img_rows : 0, st_base = st_img * p.gn_img_rows;
      // aux (saved pre-activation / saved GELU') of the NEXT row slice is requested before this slice is processed
      const bool want_aux = !AUXA && (act == 2 || act == 4 || act == 5);
      const bool interior_a = mw + tm_eff * 32 <= p.M && nw + 64 <= p.N;      // AUXA: every store of this wave is issued -> the counted waits hold
      uint2 ax[2][JW][4];                              // M16: [slice parity][m tile of the slice][n tile]
      auto load_aux = [&](int i, uint2 (&dst)[JW][4]) {
        if constexpr ((GEMM_ABL & 16) != 0) {
#pragma unroll
          for (int a_ = 0; a_ < JW; a_++)
#pragma unroll
            for (int b_ = 0; b_ < 4; b_++) dst[a_][b_] = make_uint2(OPERAND_ONE_X2, OPERAND_ONE_X2);
        } else if constexpr (M16) {
#pragma unroll
          for (int mh = 0; mh < 2; mh++)
#pragma unroll
            for (int jn = 0; jn < 4; jn++) {
              const int m = mw + i * 32 + 16 * mh + c16, n = nw + jn * 16 + 4 * R4;
              dst[mh][jn] = (m < p.M && n < p.N) ? *reinterpret_cast<const uint2*>(p.aux + (size_t)m * p.ldaux + n) : make_uint2(0u, 0u);
            }
        } else {
        const int m = mw + i * 32 + srow;
#pragma unroll
        for (int jj = 0; jj < JW; jj++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int n = nw + (j0 + jj) * 32 + 8 * q + 4 * hi;
            dst[jj][q] = (m < p.M && n < p.N) ? *reinterpret_cast<const uint2*>(p.aux + (size_t)m * p.ldaux + n) : make_uint2(0u, 0u);
          }
        }
      };
      // one group of 4 values: activation / second output / aux by flavour; returns what is parked (and leaves GELU in the accumulator on pass 0)
      auto finish4 = [&](int pass, float (&v)[4], auto&& put_back, const uint2& a2) {
        if (pass == 0 && act == 3) {                   // second output = GELU'(pre-activation); the activation itself replaces
          float g[4];                                  // the accumulator so that pass 1 only has to park it
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (GEMM_ABL & 64) { g[e] = v[e]; put_back(e, v[e]); }
            else put_back(e, gelu_tanh_both(v[e], g[e]));
            v[e] = g[e];
          }
        }
        if (pass == 1) {
          if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = gelu_tanh(v[e]);
          } else if (!AUXA && (act == 2 || act == 4 || act == 5)) {
            float a0, a1, a2f, a3;
            unpack_bf16x2(a2.x, a0, a1); unpack_bf16x2(a2.y, a2f, a3);
            if (act == 2) { a0 = gelu_tanh_grad(a0); a1 = gelu_tanh_grad(a1); a2f = gelu_tanh_grad(a2f); a3 = gelu_tanh_grad(a3); }
            if (act == 5) { v[0] += a0; v[1] += a1; v[2] += a2f; v[3] += a3; }
            else { v[0] *= a0; v[1] *= a1; v[2] *= a2f; v[3] *= a3; }
          }
        }
      };
      if (want_aux) load_aux(0, ax[0]);
#pragma unroll
      for (int i = 0; i < TM; i++) {
        if (i >= tm_eff) break;                        // half items: two row tiles per wave
        if (want_aux && i + 1 < tm_eff) load_aux(i + 1, ax[(i + 1) & 1]);
        if constexpr (AUXA) {                          // issue order and counts: see the hand-over
          if (i == 1) {
            aux_ready(interior_a ? (tm_eff > 2 ? 8 : 0) + pf_ops + 4 : 0, axa[1]);
            aux_apply(1, axa[1]);
            if (tm_eff > 3) aux_issue(3, axa[1]);
          } else if (i == 2) {
            aux_ready(interior_a ? 16 : 0, axa[0]);
            aux_apply(2, axa[0]);
          } else if (i == 3) {
            aux_ready(interior_a ? 8 : 0, axa[1]);
            aux_apply(3, axa[1]);
          }
        }
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {         // pass 0: second output (dual-output flavours only); pass 1: final values
          if (pass == 0 && !dual) continue;
          if constexpr (M16) {
#pragma unroll
            for (int mh = 0; mh < 2; mh++)
#pragma unroll
              for (int jn = 0; jn < 4; jn++) {
                f32x4& a = acc4[2 * i + mh][jn];
                float v[4] = {a[0], a[1], a[2], a[3]};
                finish4(pass, v, [&](int e, float x) { a[e] = x; }, ax[i & 1][mh][jn]);
                const int row = 16 * mh + c16, ch = 2 * jn + (R4 >> 1);
                *reinterpret_cast<uint2*>(stg + row * RB + ((ch ^ (row & 7)) << 4) + (R4 & 1) * 8) = pack_bf16x4(v[0], v[1], v[2], v[3]);
              }
          } else {
#pragma unroll
          for (int jj = 0; jj < JW; jj++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int j = j0 + jj;
              float v[4] = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
              finish4(pass, v, [&](int e, float x) { acc[i][j][q * 4 + e] = x; }, ax[i & 1][jj][q]);
              const int ch = jj * 4 + q;               // 16-byte chunk of the staging row
              const int sw = JW == 2 ? (srow & 7) : ((srow >> 1) & 3);
              *reinterpret_cast<uint2*>(stg + srow * RB + ((ch ^ sw) << 4) + hi * 8) = pack_bf16x4(v[0], v[1], v[2], v[3]);
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the slice is private to this wave
          bf16_t* dst = pass == 0 ? p.out2 : p.out;
#pragma unroll
          for (int t4 = 0; t4 < NST; t4++) {
            const int row = t4 * RPI + le / LPR, ch = le % LPR;
            const int sw = JW == 2 ? (row & 7) : ((row >> 1) & 3);
            const uint4 v4 = *reinterpret_cast<const uint4*>(stg + row * RB + ((ch ^ sw) << 4));
            const int mm = mw + i * 32 + row, nn = nw + j0 * 32 + ch * 8;
            size_t orow = (size_t)mm;
            bool st_ok = mm < p.M && nn < p.N && !((GEMM_ABL & 32) && pass == 0);
            if constexpr (want_st) {
              if (p.up_rp) {                           // phase of an upsampled convolution: interior low-res pixels only, scattered into the high-res padded grid
                const int pixu = mm - st_base, pyu = (int)(((float)pixu + 0.5f) * p.gn_inv_rp), pxu = pixu - pyu * p.gn_rp;
                st_ok = st_ok && pyu >= 1 && pyu <= p.gn_h && pxu >= 1 && pxu <= p.gn_w;
                orow = (size_t)st_img * p.up_ip + (size_t)((2 * pyu - 1) * p.up_rp + 2 * pxu - 1 + p.up_off);
              }
            }
            if (st_ok) {
              if (GEMM_NT_STORE) __builtin_nontemporal_store(nt_u4{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<nt_u4*>(dst + orow * p.ldo + nn));
              else *reinterpret_cast<uint4*>(dst + orow * p.ldo + nn) = v4;
            }
            if (pass == 1 && want_cs && mm < p.M) {
              float f[8];
              unpack_bf16x8(v4, f);
#pragma unroll
              for (int e = 0; e < 8; e++) cs[e] += f[e];
            }
            if (pass == 1 && want_st && mm < p.M) {
              const int pix = mm - st_base, py = (int)(((float)pix + 0.5f) * p.gn_inv_rp), px = pix - py * p.gn_rp;
              const unsigned msk = (py >= 1 && py <= p.gn_h && px >= 1 && px <= p.gn_w) ? 0xffffffffu : 0u;
              const unsigned w0 = v4.x & msk, w1 = v4.y & msk, w2 = v4.z & msk, w3 = v4.w & msk;   // the stored (rounded) values, packed
              cs[0] = dot2_acc(w1, OPERAND_ONE_X2, dot2_acc(w0, OPERAND_ONE_X2, cs[0]));
              cs[1] = dot2_acc(w3, OPERAND_ONE_X2, dot2_acc(w2, OPERAND_ONE_X2, cs[1]));
              sq[0] = dot2_acc(w1, w1, dot2_acc(w0, w0, sq[0]));
              sq[1] = dot2_acc(w3, w3, dot2_acc(w2, w2, sq[1]));
            }
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);          // reads returned before the slice is overwritten
        }
      }
      if (want_cs) {                                   // lanes with the same chunk (lane % LPR) hold partial sums of the same 8 columns
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float v = cs[e];
          if (LPR <= 4) v += __shfl_xor(v, 4);
          v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
          const int n = nw + j0 * 32 + (lane % LPR) * 8 + e;
          if (lane < LPR && n < p.N) atomicAdd(p.colsum + (size_t)((mw >> 7) % PXA_COLSUM_SLOTS) * p.colsum_stride + n, v);
        }
      }
      if (want_st) {
        float* dst = p.gn_part + ((size_t)((mw >> 7) % PXA_COLSUM_SLOTS) * p.gn_B + st_img) * (p.N / 4) * 2;
#pragma unroll
        for (int h = 0; h < 2; h++) {                  // lanes with the same chunk hold partial sums of the same two quads
          float v = cs[h], u = sq[h];
          if (LPR <= 4) { v += __shfl_xor(v, 4); u += __shfl_xor(u, 4); }
          v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
          u += __shfl_xor(u, 8); u += __shfl_xor(u, 16); u += __shfl_xor(u, 32);
          const int n = nw + j0 * 32 + (lane % LPR) * 8 + 4 * h;
          if (lane < LPR && n < p.N) { atomicAdd(dst + (n >> 2) * 2, v); atomicAdd(dst + (n >> 2) * 2 + 1, u); }
        }
      }
    };
    if (GEMM_ABL & 2) {                                  // ablation: no epilogue (one store keeps the accumulators alive)
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) sacc += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
      if (sacc == 123.456f) p.out[lane] = (bf16_t)sacc;
    } else {
#pragma unroll
      for (int j0 = 0; j0 + 1 < TN; j0 += 2) emit(j0, IntC<2>{});
      if constexpr (TN % 2 == 1) emit(TN - 1, IntC<1>{});
    }
    PXA_TR(5);
#if GEMM_TRACE
    tr_i++;
#endif
    {   // stores this wave has just issued, if every one of them was a full (unpredicated) 16-byte row segment
      const bool interior = mw + tm_eff * 32 <= p.M && nw + 64 <= p.N;
      // (the statistics / column-sum flavours issue 4 / 8 more vector-memory instructions behind the stores: their atomics)
      // (upsampled-convolution phases predicate their stores per row: nothing may be assumed in flight)
      prev_stores = (interior && p.out && !(want_st && p.up_rp)) ? tm_eff * 4 * ((dual && !(GEMM_ABL & 32)) ? 2 : 1) + (want_st ? 4 : 0) + (want_cs ? 8 : 0) + ((AUXA && tm_eff > 2) ? 16 : 0) : 0;
    }
    nk = nk_pf;
    if (!more) break;
  }
  if (DYN && wave == 0 && lane == 0 && atomicAdd(&sched[8], 1u) == gridDim.x - 1)
    for (int k = 0; k < 9; k++) sched[k] = 0u;         // last workgroup out: the slot is clean for a later launch
}

// item hand-out of the persistent NT / NN kernels: pxa_gemm_set_dynamic_items (include/pixart_hip.h); -1 = not set yet -> the environment, else static
static std::atomic<int> g_dynamic_items{-1};
static inline bool dynamic_items() {
  int v = g_dynamic_items.load(std::memory_order_relaxed);
  if (v < 0) {
    v = getenv("PXA_GEMM_DYNAMIC") != nullptr && getenv("PXA_GEMM_STATIC") == nullptr ? 1 : 0;
    int expect = -1;
    g_dynamic_items.compare_exchange_strong(expect, v);
    v = g_dynamic_items.load(std::memory_order_relaxed);
  }
  return v != 0;
}
static std::atomic<unsigned> g_launch_seq{0};          // cursor-slot round robin, shared by every instantiation of the persistent kernel
// work items of the persistent kernel per k-slice (must match the kernel's own count)
static inline bool pers_pairing(int M, int N) { (void)M; return N % 256 > 0 && N % 256 <= 128; }
static inline bool pers_halfcol(int N) { return N % 256 > 0 && N % 256 <= 128; }
static inline int pers_tiles(int M, int N, int rm) {
  const int mt = (M + 255) / 256, ntf = N / 256;
  if (rm == 2 && pers_halfcol(N)) return mt * ntf + mt;
  const bool pairing = rm == 1 && pers_pairing(M, N);
  return mt * (pairing ? ntf : (N + 255) / 256) + (pairing ? (mt + 1) / 2 : 0);
}
template <int LAYOUT, int EPI, int RM = 0, bool SEG = false>
int launch_pers(GemmParams p, int split, hipStream_t s) {
  static_assert(!SEG || LAYOUT == 0, "segmented A: layout NT");
  p.split = split;
  constexpr int LDSP = 4 * 40960;                      // the ring (4 x 40 KiB slots) = the CU's whole 160 KiB: one workgroup per CU
  static bool attr_set_pp = false;
  static int n_cu = 0;
  if (!attr_set_pp) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pers_kernel<LAYOUT, EPI, RM, SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
    if (e != hipSuccess) { pxa_set_error("hipFuncSetAttribute(gemm_pers<%d,%d>): %s", LAYOUT, EPI, hipGetErrorString(e)); return -3; }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { pxa_set_error("gemm_pers: device query failed"); return -3; }
    n_cu = prop.multiProcessorCount;
    attr_set_pp = true;
  }
  const int tiles = pers_tiles(p.M, p.N, RM) * split;
  p.sched_slot = dynamic_items() ? (int)(g_launch_seq.fetch_add(1u) & 63u) : -1;
  hipLaunchKernelGGL((gemm_pers_kernel<LAYOUT, EPI, RM, SEG>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), LDSP, s, p);
  PXA_LAUNCH_CHECK();
  return 0;
}

template <int LAYOUT, int TBM, int TBN, int WM, int WN, int EPI, bool SEG = false>
int launch_glds_e(GemmParams p, int split, hipStream_t s) {
  p.split = split;
  constexpr int LDSG = 2 * (TBM + TBN) * 128;
  static_assert(WM * WN * (TBM / WM) * EPI_STRIDE <= LDSG, "staged epilogue must fit the operand stages");
  static bool attr_set_g = false;
  if (!attr_set_g) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<LAYOUT, TBM, TBN, WM, WN, EPI, SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSG);
    if (e != hipSuccess) { pxa_set_error("hipFuncSetAttribute(gemm_glds<%d,%d,%d>): %s", LAYOUT, TBM, TBN, hipGetErrorString(e)); return -3; }
    attr_set_g = true;
  }
  dim3 grid(((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN) * split, 1, 1);
  hipLaunchKernelGGL((gemm_glds_kernel<LAYOUT, TBM, TBN, WM, WN, EPI, SEG>), grid, dim3(WM * WN * 64), LDSG, s, p);
  PXA_LAUNCH_CHECK();
  return 0;
}
template <int LAYOUT, int TBM, int TBN, int WM, int WN>
int launch_glds(GemmParams p, int split, hipStream_t s) {
  // bf16-only outputs of the forward / dX GEMMs take the LDS-staged, fully coalesced epilogue (separate kernel instances, so
  // neither epilogue's registers burden the other)
  static const bool no_stage = getenv("PXA_GEMM_NO_STAGED_EPILOGUE") != nullptr;
  const bool dual = (p.act == 1 && p.out2 != nullptr) || p.act == 3;   // two outputs: the direct epilogue
  static const bool no_pers = getenv("PXA_GEMM_NO_PERSISTENT") != nullptr;
  if (LAYOUT != 2 && TBM == 256 && TBN == 256 && split == 1 && p.out && !p.outf && !no_pers) {
    constexpr int LY = LAYOUT == 2 ? 0 : LAYOUT;
    static const bool no_half = getenv("PXA_GEMM_NO_HALF_ITEMS") != nullptr;   // A/B: pad the remainder column to a full tile
    const bool hc = pers_halfcol(p.N) && !no_half;
    if constexpr (LAYOUT == 0) {                          // the one-wave-per-SIMD NT kernel (gemm_nt4.hip) where it applies
      if (p.act == 0 && !p.colsum) { GemmParams q = p; q.split = 1; const int rc = pxa_gemm_nt4_launch(q, s); if (rc <= 0) return rc; }
    }
    if (p.act == 0 && !p.colsum) return hc ? launch_pers<LY, 0, 2>(p, 1, s) : launch_pers<LY, 0, 0>(p, 1, s);
    if (p.act == 3 && !p.colsum) return launch_pers<LY, 1, 0>(p, 1, s);      // fc1 forward: N = 4608, no remainder column
    if (p.act == 4 && p.colsum) return hc ? launch_pers<LY, 2, 2>(p, 1, s) : launch_pers<LY, 2, 0>(p, 1, s);
    if constexpr (LY == 0) { if (p.act == 1 && !p.out2 && !p.colsum && !hc) return launch_pers<0, 7, 0>(p, 1, s); }
    return launch_pers<LY, 3, 0>(p, 1, s);
  }
  // fp32 weight gradients (TN, split-K slabs / single-slice read-modify-write / plain store): the same persistent kernel
  if (LAYOUT == 2 && TBM == 256 && TBN == 256 && p.outf && !p.out && !p.bias && p.act == 0 && p.accumulate != 1 && !p.colsum && !no_pers)
    return pers_pairing(p.M, p.N) ? launch_pers<2, 0, 1>(p, split, s) : launch_pers<2, 0, 0>(p, split, s);
  if (LAYOUT != 2 && p.out && !p.outf && !dual && !no_stage) return launch_glds_e<LAYOUT, TBM, TBN, WM, WN, 1>(p, split, s);
  if (p.colsum) {                                         // not fused on this path: separate column-sum pass over the output
    float* cs = p.colsum;
    p.colsum = nullptr;
    int rc = launch_glds_e<LAYOUT, TBM, TBN, WM, WN, 0>(p, split, s);
    return rc ? rc : pxa_colsum_bf16(p.out, p.ldo, cs, p.M, p.N, s);
  }
  return launch_glds_e<LAYOUT, TBM, TBN, WM, WN, 0>(p, split, s);
}

template <int LAYOUT>
int launch(GemmParams p, int split, hipStream_t s) {
  p.split = split;
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int LDS = 2 * ((A_KC ? KC_BYTES : RC_BYTES) + (B_KC ? KC_BYTES : RC_BYTES));
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { pxa_set_error("hipFuncSetAttribute(gemm<%d>, %d): %s", LAYOUT, LDS, hipGetErrorString(e)); return -3; }
    attr_set = true;
  }
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * split, 1, 1);
  const bool fast = (p.K % BK == 0) && (p.k_per_split % BK == 0) && !getenv("PXA_GEMM_NO_GLDS");
  if (p.k_seg) {                                          // implicit 3x3 convolution (checked by pxa_gemm: NT, K and k_seg multiples of 64)
    if constexpr (LAYOUT == 0) {
      static const bool no_pers_seg = getenv("PXA_GEMM_NO_PERSISTENT") != nullptr;
      if (p.out && !p.outf && (p.act == 0 || p.act == 5) && p.M >= 1024 && p.N >= 128 && p.N % 128 == 0 && p.k_seg % 32 == 0 && !no_pers_seg) {
        // Remainder columns of 128 (the VAE's 128-channel layers: N = 128 is ONLY a remainder column).  Round 6: PAIRED items - the remainder columns of two
        // consecutive m-tiles as one 512 x 128 item on full 128 x 64 wave tiles - instead of HALF items (256 x 128 on 64 x 64 wave tiles: half the MFMAs per
        // barrier pair, 500-640 TFLOP/s on the 512 x 512 x 128-channel layers of the decoder, profiles/r6_01_vae_layer_table.txt).  The token GEMMs keep HALF
        // items: there the remainder column is a ninth of the work and equal rounds matter more (see RM above).  PXA_GEMM_SEG_HALF=1: the round-5 choice (A/B).
        static const bool seg_half = getenv("PXA_GEMM_SEG_HALF") != nullptr;
        const bool hc = pers_halfcol(p.N), pair = hc && !seg_half && p.M >= 512;
        if (p.gn_part) {                                 // + GroupNorm statistics of the output
          if (p.act == 5) return pair ? launch_pers<0, 6, 1, true>(p, 1, s) : hc ? launch_pers<0, 6, 2, true>(p, 1, s) : launch_pers<0, 6, 0, true>(p, 1, s);
          return pair ? launch_pers<0, 5, 1, true>(p, 1, s) : hc ? launch_pers<0, 5, 2, true>(p, 1, s) : launch_pers<0, 5, 0, true>(p, 1, s);
        }
        if (p.act == 5) return pair ? launch_pers<0, 4, 1, true>(p, 1, s) : hc ? launch_pers<0, 4, 2, true>(p, 1, s) : launch_pers<0, 4, 0, true>(p, 1, s);   // conv + residual
        return pair ? launch_pers<0, 0, 1, true>(p, 1, s) : hc ? launch_pers<0, 0, 2, true>(p, 1, s) : launch_pers<0, 0, 0, true>(p, 1, s);
      }
      if (p.gn_part) { pxa_set_error("pxa_gemm: gn_part needs the persistent implicit-convolution path (bf16 output, M >= 1024, N a multiple of 128)"); return -1; }
      if (p.out && !p.outf && p.act == 0) return launch_glds_e<0, 128, 128, 2, 2, 1, true>(p, 1, s);
      return launch_glds_e<0, 128, 128, 2, 2, 0, true>(p, 1, s);
    }
    return -2;
  }
  if (fast) {
    static const char* force = getenv("PXA_GEMM_TILE");   // "128" | "256x128" | "256" : A/B experiments
    int tile = force ? atoi(force) * (strstr(force, "x128") ? -1 : 1) : 0;
    if (!tile) tile = p.tile_hint;
    if (!tile) tile = (p.M >= 1024 && p.N >= 1024) ? 256 : 128;
    if (tile == 128) return launch_glds<LAYOUT, 128, 128, 2, 2>(p, split, s);
    if (tile == -256) return launch_glds<LAYOUT, 256, 128, 4, 2>(p, split, s);
    return launch_glds<LAYOUT, 256, 256, 2, 4>(p, split, s);
  }
  float* cs_fallback = p.colsum;
  p.colsum = nullptr;
  hipLaunchKernelGGL(gemm_kernel<LAYOUT>, grid, dim3(256), LDS, s, p);
  PXA_LAUNCH_CHECK();
  if (cs_fallback) return pxa_colsum_bf16(p.out, p.ldo, cs_fallback, p.M, p.N, s);
  return 0;
}
}  // namespace

namespace {
// out[m][n] += sum_z slab[z][m][n]   (split-K combine at the launch boundary: plain 16-byte loads/stores, no atomics)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int ld, int M, int N, int split) {
  const long i4 = blockIdx.x * 256L + threadIdx.x, n4 = N / 4;
  if (i4 >= (long)M * n4) return;
  const int m = i4 / n4, n = (i4 - (long)m * n4) * 4;
  float4 s = *reinterpret_cast<const float4*>(out + (size_t)m * ld + n);
  for (int z = 0; z < split; z++) {
    const float4 v = *reinterpret_cast<const float4*>(slab + ((size_t)z * M + m) * N + n);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)m * ld + n) = s;
}
}  // namespace

extern "C" int pxa_gemm(const pxa_gemm_args* a, hipStream_t stream) {
  PXA_CHECK(a && a->A && a->B, "pxa_gemm: null operand");
  PXA_CHECK(a->layout >= 0 && a->layout <= 2, "pxa_gemm: bad layout %d", a->layout);
  PXA_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "pxa_gemm: bad shape %d %d %d", a->M, a->N, a->K);
  PXA_CHECK(a->N % 8 == 0, "pxa_gemm: N=%d must be a multiple of 8", a->N);
  PXA_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0, "pxa_gemm: lda/ldb must be multiples of 8 (16-byte rows)");
  if (a->layout != 2) PXA_CHECK(a->K % 8 == 0, "pxa_gemm: K=%d must be a multiple of 8 for k-contiguous A", a->K);
  if (a->layout == 2) PXA_CHECK(a->M % 8 == 0, "pxa_gemm: M=%d must be a multiple of 8 for layout TN", a->M);
  PXA_CHECK(a->out_bf16 || a->out_f32, "pxa_gemm: no output");
  if (a->out_bf16 || a->out2_bf16) PXA_CHECK(a->ld_out % 4 == 0, "pxa_gemm: ld_out must be a multiple of 4");
  if (a->out_f32) PXA_CHECK(a->ld_f32 % 4 == 0, "pxa_gemm: ld_f32 must be a multiple of 4");
  PXA_CHECK(a->act >= 0 && a->act <= 5, "pxa_gemm: bad act %d", a->act);
  if (a->act == 2 || a->act == 4 || a->act == 5) PXA_CHECK(a->aux && a->ldaux % 4 == 0, "pxa_gemm: act=%d needs aux", a->act);
  if (a->act == 3) PXA_CHECK(a->out_bf16 && a->out2_bf16, "pxa_gemm: act=3 needs both bf16 outputs");
  if (a->colsum) PXA_CHECK(a->out_bf16 && !a->out_f32, "pxa_gemm: colsum needs a bf16 output");
  int split = a->split_k < 1 ? 1 : a->split_k;   // 0 = choose here (only for fp32 atomic-accumulate outputs)
  if (split > 1) PXA_CHECK(a->out_f32 && a->accumulate && !a->out_bf16 && a->act == 0 && !a->bias, "pxa_gemm: split_k>1 needs fp32 atomic accumulate output only");
  GemmParams p;
  p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.lda = a->lda; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.bias = a->bias; p.aux = (const bf16_t*)a->aux; p.ldaux = a->ldaux;
  p.out = (bf16_t*)a->out_bf16; p.out2 = (bf16_t*)a->out2_bf16; p.ldo = a->ld_out;
  p.outf = a->out_f32; p.ldf = a->ld_f32;
  p.act = a->act; p.accumulate = a->accumulate;
  p.tile_hint = 0;
  static const bool force_asc = getenv("PXA_GEMM_ASCENDING") != nullptr;      // A/B: ignore items_descending
  p.desc = (a->items_descending && !force_asc) ? 1 : 0;
  p.k_seg = a->k_seg; p.seg_jump = a->k_seg ? a->a_seg_stride - a->k_seg : 0;
  p.k_tap = a->k_seg ? a->k_tap : 0; p.tap_s = a->a_seg_stride;
  if (a->k_seg && a->k_tap)
    PXA_CHECK(a->k_tap % BK == 0 && a->k_seg == 3 * a->k_tap && a->K == 3 * a->k_seg, "pxa_gemm: k_tap=%d needs k_seg = 3 k_tap, K = 9 k_tap, k_tap a multiple of %d", a->k_tap, BK);
  if (a->k_seg) {
    PXA_CHECK(a->layout == 0 && split == 1 && !a->colsum, "pxa_gemm: k_seg needs layout NT, no split-K, no colsum");
    PXA_CHECK(a->k_seg > 0 && a->k_seg % BK == 0 && a->K % a->k_seg == 0 && a->a_seg_stride % 8 == 0, "pxa_gemm: k_seg=%d must be a multiple of %d dividing K=%d (segment stride a multiple of 8)", a->k_seg, BK, a->K);
  }
  if (a->split_k == 0 && a->out_f32 && a->accumulate && !a->out_bf16 && a->act == 0 && !a->bias && a->K % BK == 0) {
    // Split-K weight-gradient GEMMs (K = tokens, 65536): a handful of long-running workgroups, so wave quantisation against the
    // 256 CUs decides the time.  Pick (tile, split) minimising  rounds x k-tiles x tile cost  + atomic epilogue traffic.
    // Candidates: 128 x 128 tiles, two workgroups per CU (512 slots, ~760 TF/s when full) and the persistent 256 x 256 kernel
    // (256 slots, ~1050 TF/s when full since the 16-MFMA phases; with the round-1 constants 800 / 1000 the 1152 x 1152 gradients still went
    // to the 128 kernel: 232-245 us against 212 for 25 padded 256 tiles x 10 k-slices).
    // time = rounds x k-range x tile FLOPs / per-slot rate  +  slab write + read + reduce.
    struct Cfg { int tile, bm, bn, slots; double rate; };
    const Cfg cfgs[2] = {{128, 128, 128, 512, 760e12}, {256, 256, 256, 256, 1050e12}};
    double best = 1e30;
    for (const Cfg& c : cfgs) {
      if (c.tile == 256 && (a->M < 256 || a->N < 256)) continue;
      const long tiles = c.tile == 256 ? pers_tiles(a->M, a->N, 1) : (long)((a->M + c.bm - 1) / c.bm) * ((a->N + c.bn - 1) / c.bn);
      for (int sp = 1; sp <= 16; sp++) {
        const int kp = ((a->K + sp - 1) / sp + BK - 1) / BK * BK;
        if ((long)kp * (sp - 1) >= a->K) continue;               // would leave an empty split
        const long rounds = (tiles * sp + c.slots - 1) / c.slots;
        const double t = rounds * (double)kp * c.bm * c.bn * 2.0 / (c.rate / c.slots) + (sp > 1 ? 2.0 * sp * a->M * a->N * 4.0 / 3.5e12 + 4e-6 : 0.0);
        if (t < best) { best = t; split = sp; p.tile_hint = c.tile; }
      }
    }
  }
  int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
  p.k_per_split = kps;
  split = (a->K + kps - 1) / kps;
  p.slab = nullptr;
  p.colsum = a->colsum; p.colsum_stride = a->colsum_stride;
  p.gn_part = a->gn_part; p.gn_img_rows = a->gn_img_rows; p.gn_rp = a->gn_row_pitch; p.gn_h = a->gn_h; p.gn_w = a->gn_w;
  p.gn_B = 0; p.gn_inv_rp = 0.f;
  p.up_rp = 0; p.up_ip = 0; p.up_off = 0;
  if (a->up_row_pitch) {
    PXA_CHECK(a->gn_part && a->k_seg && a->act == 0 && !a->k_tap, "pxa_gemm: up_row_pitch needs an implicit convolution with gn_part, act 0 and the plain segment order");
    PXA_CHECK(a->K == 2 * a->k_seg && a->up_row_pitch == 2 * a->gn_w + 2 && a->up_img_rows >= (2 * a->gn_h + 2) * a->up_row_pitch && (a->up_dy | 1) == 1 && (a->up_dx | 1) == 1,
              "pxa_gemm: bad upsampled-convolution phase (K=%d k_seg=%d up_row_pitch=%d up_img_rows=%d dy=%d dx=%d)", a->K, a->k_seg, a->up_row_pitch, a->up_img_rows, a->up_dy, a->up_dx);
    PXA_CHECK(a->M >= 1024 && a->N % 128 == 0 && (long)(a->M / a->gn_img_rows) * a->up_img_rows < (1L << 31), "pxa_gemm: upsampled-convolution phase needs the persistent path (M >= 1024, N a multiple of 128)");
    p.up_rp = a->up_row_pitch; p.up_ip = a->up_img_rows; p.up_off = a->up_dy * a->up_row_pitch + a->up_dx;
  }
  if (a->gn_part) {
    PXA_CHECK(a->k_seg && a->out_bf16 && !a->out_f32 && (a->act == 0 || a->act == 5), "pxa_gemm: gn_part needs an implicit convolution (k_seg) with a bf16 output and act 0 or 5");
    PXA_CHECK(a->gn_img_rows > 0 && a->gn_img_rows % 256 == 0 && a->M % a->gn_img_rows == 0, "pxa_gemm: gn_img_rows=%d must be a multiple of 256 dividing M=%d", a->gn_img_rows, a->M);
    PXA_CHECK(a->gn_row_pitch == a->gn_w + 2 && a->gn_h > 0 && a->gn_w > 0 && (long)(a->gn_h + 2) * a->gn_row_pitch <= a->gn_img_rows && a->gn_img_rows < (1 << 22),
              "pxa_gemm: bad padded-grid geometry for gn_part (h=%d w=%d row_pitch=%d img_rows=%d)", a->gn_h, a->gn_w, a->gn_row_pitch, a->gn_img_rows);
    p.gn_B = a->M / a->gn_img_rows; p.gn_inv_rp = 1.0f / (float)a->gn_row_pitch;
  }
  if (p.accumulate && p.outf) {
    if (split == 1) p.accumulate = 2;
    else if (a->splitk_ws && a->splitk_ws_elems >= (long)split * a->M * a->N) { p.accumulate = 3; p.slab = a->splitk_ws; }
  }
  int rc;
  switch (a->layout) {
    case 0: rc = launch<0>(p, split, stream); break;
    case 1: rc = launch<1>(p, split, stream); break;
    default: rc = launch<2>(p, split, stream); break;
  }
  if (rc == 0 && p.accumulate == 3) {
    const long n4 = (long)a->M * (a->N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, stream, p.slab, a->out_f32, a->ld_f32, a->M, a->N, split);
    PXA_LAUNCH_CHECK();
  }
  return rc;
}

extern "C" long pxa_gemm_splitk_ws_elems(int M, int N) { return 16L * M * N; }

extern "C" int pxa_gemm_set_dynamic_items(int on) {
  const bool prev = dynamic_items();
  if (getenv("PXA_GEMM_STATIC") == nullptr && getenv("PXA_GEMM_DYNAMIC") == nullptr) g_dynamic_items.store(on ? 1 : 0);     // (an A/B environment override wins)
  return prev ? 1 : 0;
}

#if GEMM_TRACE
extern "C" int pxa_gemm_trace(unsigned long long* host_out) {       // 12 x 8 counters of the last traced launch
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : -1;
}
#endif
