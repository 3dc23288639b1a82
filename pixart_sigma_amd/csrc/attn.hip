// Flash-style softmax attention, head_dim 72, bf16 MFMA, for the three attention shapes of the PixArt block:
//   * self-attention over image tokens               (xformers.ops.memory_efficient_attention, PixArt_blocks.py:153)
//   * self-attention against KV-compressed tokens    (N_q != N_kv, PixArt_blocks.py:137-139)
//   * varlen cross-attention to packed text tokens   (BlockDiagonalMask.from_seqlens, PixArt_blocks.py:50-53)
// All three are one kernel family driven by strides + optional per-sample (kv_start, kv_len).
//
// Structure ("swapped QK^T"): a wave owns 32 query columns. S^T = K Q^T is computed with
// v_mfma_f32_32x32x16_bf16 (A = K rows from LDS, B = Q^T held in registers), so every lane holds scores of ONE
// query -> softmax max/sum are in-lane plus a single lane^32 exchange, and P^T in accumulator layout is
// the B operand of the second product O^T = V^T P^T: directly in the 32-row form (the reduction index kv is permuted
// identically on both operands; dK/dV kernel), after 4 v_permlane16_swap per 32 x 32 block in the 16-row form
// (v_mfma_f32_16x16x32: forward and dQ kernels, see "second products on 16x16x32 MFMAs" below).
// V^T fragments come from a row-major [kv][d] LDS tile through ds_read_b64_tr_b16.
// head_dim 72 is zero-padded to 80 for the QK^T reduction (5 k-steps of 16) and, as OUTPUT rows, to 80 (5 tiles of 16;
// forward, dQ) or 96 (3 tiles of 32; dK/dV).
// Backward = delta pre-pass + dQ kernel (q-stationary) + dK/dV kernel (kv-stationary), both recomputing P
// from the saved log2-sum-exp.  Grids are flat and keep the blocks of a head on one XCD (block_coords).
#include "common.h"
#include "../../include/pixart_hip.h"
#include <map>
#include <mutex>
#include <utility>

namespace {
using namespace pxa;

constexpr int DH = 72;
constexpr int NCH = DH / 8;          // 9 16-byte chunks per head row
constexpr int KSTEPS = 5;            // ceil(72/16)
// LDS tile image: row-major [64 rows][12 chunks of 16 B] = 192-byte rows (9 data chunks + 3 pad), chunk index XOR ((row>>2)&3).
//   * ds_read_b64_tr_b16 (transpose reads) take 4 consecutive rows x 64 B per 32-lane group: 192 B = 48 banks puts the 4 rows on
//     the 4 disjoint 16-bank quarters; the XOR is constant over an aligned group of 4 rows and closed on aligned groups of 4
//     chunks, so it only permutes inside each row's 64-byte window -> conflict-free.
//   * ds_read_b128 serves 16-lane groups whose rows cover all residues mod 16; a bare 192-byte stride would pile rows r, r+4,
//     r+8, r+12 onto one 16-byte slot (4-way conflict), the XOR separates them -> conflict-free.
// Tiles are filled by LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, per-lane source address), so the image
// is lane-linear and the swizzle lives in the SOURCE address (guide rule 21).  Lanes that map to pad chunks are exec-masked
// (no DMA, 25 % less traffic); the pads are written once per kernel: zero, except that the forward's V tiles carry 1.0 in
// column 72 so the PV MFMA also produces the softmax row-sum (row 72 of O^T).  Rows beyond the valid range receive a duplicate
// of the last valid row (finite, masked by index).
constexpr int ROWB = 192;
constexpr int TILE_B = 64 * ROWB;        // 12288 B per tile
constexpr int BKV = 64;
constexpr float RESCALE_LOG2 = 6.0f;     // online-softmax rescale threshold, log2 domain (P <= 64)
constexpr int NDMA = TILE_B / 1024 / 4;  // 3 DMA instructions per wave per tile

struct AttnParams {
  const bf16_t *Q, *K, *V, *dO;
  bf16_t *O, *dQ, *dK, *dV;
  float* LSE;          // [B][H][Nq], log2 domain: m*c + log2(l)
  const float* Delta;  // [B][H][Nq]
  float *dq_colsum, *dk_colsum, *dv_colsum;   // optional [PXA_COLSUM_SLOTS][colsum_stride] fp32 partials: += column sums of dQ / dK / dV
  long colsum_stride;
  long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;  // element strides (batch, token)
  int q_hs, k_hs, v_hs, o_hs;                            // head strides
  long dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
  int dq_hs, dk_hs, dv_hs;
  int B, H, Nq, Nk;
  const int* kv_start; const int* kv_len;  // optional per-batch varlen (rows into the packed K/V)
  float scale, scale_log2;      // softmax scale (the dq stores); the factor in front of q k^T inside exp2: scale * log2 e, or 1 when q arrives prescaled
  float dk_scale;               // what the dk stores multiply by: scale, or ln 2 = scale / (scale * log2 e) when the q operand of dS^T q already carries the rest
  int nx;              // blocks per (batch, head) of the launch: the grid is the flat nx * H * B, see block_coords()
  const bf16_t* stats; // dK/dV kernel, round 3: [2][B][H][Nq64] rows of 8 operands {hi, lo, 0 x 6}: lse / scale_log2 and delta, see "stats rows"
  int Nq64;            // Nq rounded up to the 64-query tile
};

// Block -> (row block, head, batch).  Every block of one (batch, head) streams the same rows (K / V in the forward and dQ kernels,
// Q / dO / lse / delta in the dK/dV kernel).  Workgroup i of a launch runs on XCD i % 8 (observed dispatch rule, used for speed
// only) and the eight L2s do not share: in plain x-fastest order the nx blocks of a head are dealt round-robin to all eight XCDs and
// each of them fetches the head's rows from HBM for itself - measured on the dK/dV kernel at B16 H16 N4096: 2.72 GB fetched per launch
// for 0.60 GB of operands (profiles/r02_pmc_attention.txt).  Here XCD x takes the contiguous range [xs(x), xs(x) + xc(x)) of that order,
// i.e. whole heads: the 32 CUs of an XCD work on the same one or two heads at a time and the rows come from HBM once.
#ifndef ATTN_XCD_HEADS
#define ATTN_XCD_HEADS 1   // 0 = plain order (A/B builds, tools/build_variant.py)
#endif
__device__ __forceinline__ void block_coords(const AttnParams& p, int& bx, int& h, int& b) {
  int v = blockIdx.x;
#if ATTN_XCD_HEADS
  const int T = gridDim.x, xq = T >> 3, xr = T & 7, xcd = v & 7;
  v = xcd * xq + min(xcd, xr) + (v >> 3);
#endif
  bx = v % p.nx;
  const int hb = v / p.nx;
  h = hb % p.H;
  b = hb / p.H;
}

__device__ __forceinline__ void kv_range(const AttnParams& p, int b, long& kbase, long& vbase, long& dkbase, long& dvbase, int& len) {
  if (p.kv_start) {
    const long s = p.kv_start[b];
    kbase = s * p.k_ts; vbase = s * p.v_ts; dkbase = s * p.dk_ts; dvbase = s * p.dv_ts; len = p.kv_len[b];
  } else {
    kbase = (long)b * p.k_bs; vbase = (long)b * p.v_bs; dkbase = (long)b * p.dk_bs; dvbase = (long)b * p.dv_bs; len = p.Nk;
  }
}


__device__ __forceinline__ int soff(int r, int c) { return r * ROWB + ((c ^ ((r >> 2) & 3)) << 4); }

// The wait for this wave's own LDS-DMA pieces is written out (common.h lds_dma16: the compiler does not see the DMA), then the workgroup barrier.
__device__ __forceinline__ void tile_sync() {
  lds_dma_wait<0>();
  __syncthreads();
}
// Row fragments loaded from global memory at kernel start: make the compiler wait for them HERE.  Left alone it waits at their first use, inside the
// tile loop, with a vmcnt that counts only the loads it knows - and with the LDS-DMA invisible to it (lds_dma16) that wait would drain the next tile's
// DMA in every iteration.
__device__ __forceinline__ void settle(bf16x8 (&f)[5]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4])); }
struct DmaPlan { int row[NDMA], coff[NDMA]; };   // per-lane constants: tile row and source element offset of each DMA chunk
__device__ __forceinline__ void dma_plan(DmaPlan& pl, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NDMA; i++) {
    const int p = (i * 4 + wave) * 64 + lane, r = p / 12, cl = p - r * 12;
    const int c = cl ^ ((r >> 2) & 3);
    pl.row[i] = r;
    pl.coff[i] = c < NCH ? c * 8 : -1;       // pad chunk: this lane issues no DMA (the pads are written once at kernel start)
  }
}
// Source addresses are wave-uniform base (SGPR pair) + 32-bit per-lane element offset, so the DMA uses the saddr form and
// the per-tile address update is one integer add per chunk.  `FULL` tiles need no row clamp.
template <bool FULL>
__device__ __forceinline__ void dma_tile(char* lds, const bf16_t* __restrict__ base, int ts, int row0, int nrows, const DmaPlan& pl, int wave) {
#pragma unroll
  for (int i = 0; i < NDMA; i++) {
    const int gr = FULL ? row0 + pl.row[i] : min(row0 + pl.row[i], nrows - 1);
    const unsigned off = (unsigned)(gr * ts + pl.coff[i]);
    if (pl.coff[i] >= 0)                       // exec-masked: inactive lanes write nothing (LDS address = M0 + lane*16)
      lds_dma16(base + off, lds + (i * 4 + wave) * 1024);
  }
}
// delta folded into dP (dQ kernel).  dS = P (dP - delta) with dP = dO V^T reduced over the head dimension in 5 steps of 16 = 80 slots, of which 72..79 are
// zero padding: the lane's dO row (registers) carries delta in slots 72 .. 74 (split3: three terms of the operand type, fp32 precision) and the V
// tiles carry -1.0 there (written once with the pads), so the MFMA returns dP - delta and the 32 subtractions per tile leave the VALU; nothing is added to
// the matrix work: dQ kernel 1.990 -> 1.916 ms (profiles/r02n_attn_fold_ab.txt).  The dK/dV kernel would have to write delta into the dO tile's pad chunk
// every tile (its dO rows come by DMA): measured +5 % there; round 3's dK/dV kernel gets them as DMA'd stats rows instead.
// Both operand builds since round 3.  In the fp16 build a loss-scaled delta = rowsum(dO o O) can in principle leave the fp16 range while dO itself is
// still inside it (the scaler keeps dO finite, not a sum of 72 products): its leading term then is inf, dS and the gradients are non-finite, and the
// device-side scaler treats the step exactly like any other overflow - skipped, scale halved (dp.LossScaler, GradScaler's protocol) - i.e. the scale
// settles at most one notch lower than with an fp32 subtraction.  bench.py's fp16 runs report steps_skipped (0 at the initial 65536).
#ifndef ATTN_FOLD_DELTA
#define ATTN_FOLD_DELTA 1   // 0 = subtract delta on the VALU (A/B builds)
#endif
#define PXA_OPERAND_MINUS_ONE_X2 (((PXA_OPERAND_ONE_BITS | 0x8000u) << 16) | PXA_OPERAND_ONE_BITS | 0x8000u)
#define PXA_OPERAND_MINUS_ONE_X1 (PXA_OPERAND_ONE_BITS | 0x8000u)     // {-1.0, 0}
// x as THREE operand-type terms {hi, mid}, {lo, 0} with hi + mid + lo = x to 24 (bf16) / 33 (fp16) mantissa bits - fp32 precision.  Round 3: the two-term
// form (16 bits in bf16) is not enough where the softmax saturates: there dP - delta vanishes for the dominant key while delta itself is large, and a
// 2^-17 relative error on delta becomes a spurious dS that dwarfs the true (vanishing) gradient - measured on the depth-28 1024px training golden
// (train_xl2_1024_b1: gradients of blocks 22 and below off by 1e5 .. 1e6 in the bf16 build; the fp16 build, 22 bits, passed).
__device__ __forceinline__ uint2 split3(float x) {
  const bf16_t h = (bf16_t)x;
  const float r1 = x - (float)h;
  const bf16_t m = (bf16_t)r1;
  const bf16_t l = (bf16_t)(r1 - (float)m);
  bf16x2 a, b; a[0] = h; a[1] = m; b[0] = l; b[1] = (bf16_t)0.f;
  return make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
}
// one-time pad initialisation of a [64][12-chunk] tile: chunks 9..11 <- 0; pad 1: element (row, 72) <- 1.0; pad 2: (row, 72 .. 74) <- -1.0
__device__ __forceinline__ void init_pads(char* tile, int pad, int tid) {
  for (int i = tid; i < BKV * 3; i += 256) {
    const int r = i / 3, c = NCH + (i - r * 3);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pad == 1 && c == NCH) v.x = PXA_OPERAND_ONE_BITS;     // 1.0 in the low half = column 72
    if (pad == 2 && c == NCH) { v.x = PXA_OPERAND_MINUS_ONE_X2; v.y = PXA_OPERAND_MINUS_ONE_X1; }   // -1.0 in columns 72, 73, 74
    *reinterpret_cast<uint4*>(tile + soff(r, c)) = v;
  }
}
// B-operand fragments of a row held in registers: X[row][ks*16 + 8*hi .. +8], zero for d >= 72 or invalid row
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[KSTEPS], const bf16_t* __restrict__ rowptr, bool valid, int hi) {
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ks++) {
    const int d0 = ks * 16 + 8 * hi;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid && d0 < DH) v = *reinterpret_cast<const uint4*>(rowptr + d0);
    f[ks] = __builtin_bit_cast(bf16x8, v);
  }
}
// Fragment addressing with per-lane constants + compile-time immediates.  With s = ((lane&31)>>2)&3 the XOR swizzle only touches
// the low two chunk bits, and sub*32 / 16u / dt*4 are multiples of 4 (rows) resp. 4 (chunks), so:
//   rowfrag(sub, ks)  = lds + rb[ks&1] + sub*32*ROWB + (ks>>1)*64       rb[e] = r*ROWB + (((2e + hi) ^ s) << 4),  r = lane&31
//   trfrag(dt, u)     = lds + tb[h]    + u*16*ROWB  + dt*64            tb[h] = row_h*ROWB + ((cc ^ s_h) << 4) + sub8
struct FragAddr { int rb[2], tb[2]; };
__device__ __forceinline__ void frag_addr(FragAddr& fa, int lane) {
  const int r = lane & 31, hi = lane >> 5, s = (r >> 2) & 3;
  fa.rb[0] = r * ROWB + (((0 + hi) ^ s) << 4);
  fa.rb[1] = r * ROWB + (((2 + hi) ^ s) << 4);
  const int gg = lane >> 4, tt = lane & 15, h2 = gg >> 1;
  const int row0 = 4 * h2 + (tt >> 2), cc = 2 * (gg & 1) + ((tt & 3) >> 1), sub8 = (tt & 1) * 8;
  fa.tb[0] = row0 * ROWB + ((cc ^ ((row0 >> 2) & 3)) << 4) + sub8;
  fa.tb[1] = (row0 + 8) * ROWB + ((cc ^ (((row0 + 8) >> 2) & 3)) << 4) + sub8;
}
// A-operand from a row-major tile: rows sub*32 + (lane&31), k = d
__device__ __forceinline__ bf16x8 rowfrag(const char* lds, const FragAddr& fa, int sub, int ks) {
  return *reinterpret_cast<const bf16x8*>(lds + fa.rb[ks & 1] + sub * 32 * ROWB + (ks >> 1) * 64);
}
// A-operand X^T[d = dt*32 + (lane&31)][k-slots of step u] from a row-major [row][d] tile via transpose reads.
// slot j <-> row 16u + (j&3) + 8*(j>>2) + 4*hi : the same permutation the accumulator layout gives the B operand.
__device__ __forceinline__ bf16x8 trfrag(const char* lds, const FragAddr& fa, int dt, int u) {
  return concat_tr(lds_tr_read(lds + fa.tb[0] + u * 16 * ROWB + dt * 64), lds_tr_read(lds + fa.tb[1] + u * 16 * ROWB + dt * 64));
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int off) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = (bf16_t)v[off + j];
  return r;
}
// store a transposed accumulator set X^T[d][q] (3 tiles) as bf16 rows X[q][0..71]
__device__ __forceinline__ void store_rows(bf16_t* __restrict__ rowptr, const f32x16 (&acc)[3], float mul, int hi) {
#pragma unroll
  for (int dt = 0; dt < 3; dt++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const int d0 = dt * 32 + 8 * qd + 4 * hi;
      if (d0 < DH)
        *reinterpret_cast<uint2*>(rowptr + d0) = pack_bf16x4(acc[dt][qd * 4] * mul, acc[dt][qd * 4 + 1] * mul, acc[dt][qd * 4 + 2] * mul, acc[dt][qd * 4 + 3] * mul);
    }
}
// bias gradient of the Linear that produced this operand: colsum[h*72 + d] += sum over the wave's 32 rows of X[row][d] (invalid rows
// contribute 0).  One cross-lane tree per stored value, once per workgroup: negligible next to the tile loop, and it removes a
// full extra HBM pass over the gradient tensor (the separate column-sum kernel).
__device__ __forceinline__ void colsum_rows(float* __restrict__ dst, const f32x16 (&acc)[3], float mul, bool valid, int hi, int lane) {
#pragma unroll
  for (int dt = 0; dt < 3; dt++)
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const int d0 = dt * 32 + 8 * qd + 4 * hi;
      if (dt * 32 + 8 * qd < DH) {                       // compile-time prune of the pad tiles (hi-dependent part checked below)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float v = valid ? acc[dt][qd * 4 + e] * mul : 0.f;
          v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
          if ((lane & 31) == 0 && d0 < DH) atomicAdd(dst + d0 + e, v);
        }
      }
    }
}
__device__ __forceinline__ void zero3(f32x16 (&a)[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int g = 0; g < 16; g++) a[i][g] = 0.f;
}
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int I> struct IntC { static constexpr int value = I; };

// ------------------------------------------------------------------------------------------------ second products on 16x16x32 MFMAs
// O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS: the products whose OUTPUT rows are the head dimension.  With 32-row tiles
// 72 rows pad to 96 (3 tiles); with the 16-row shape to 80 (5 tiles): 10 v_mfma_f32_16x16x32 per 32 x 32 block of P instead of 6
// v_mfma_f32_32x32x16 = 160 instead of 192 matrix-pipe cycles, and the 16-row shape is the one the power limit favours (common.h).
// The first products stay on 32x32x16 (their REDUCTION runs over the head dimension: 80 = 5 x 16, where K = 32 steps would need 96).
// Operand B.  A 32 x 32 block of P^T sits in the first product's accumulators: lane (col n = l & 31, hi = l >> 5) holds rows
// (g & 3) + 8 (g >> 2) + 4 hi, g = 0..15.  The 16-row shape wants, per 16-lane row R = l >> 4 of the wave, 8 k-values of column l & 15.
// With a = rows {0-3, 16-19} + 4 hi (g 0-3, 8-11) and b = rows {8-11, 24-27} + 4 hi (g 4-7, 12-15), both packed to 4 dwords, one
// v_permlane16_swap per dword (odd 16-lane rows of the first operand <-> even rows of the second) leaves
//   X = {a(R0), b(R0), a(R2), b(R2)}: columns 0-15,  lane row R holds rows k16(R) + {0-3} and k16(R) + 16 + {0-3}
//   Y = {a(R1), b(R1), a(R3), b(R3)}: columns 16-31, same rows,                     k16(R) = 8 (R & 1) + 4 (R >> 1)
// in the original lane rows R0..R3 - the two B operands of the block, 4 extra VALU instructions.
// Operand A.  X^T[d = 16 t + (l & 15)][those 8 rows] of a row-major [row][d] LDS tile: two transpose reads (ds_read_b64_tr_b16: inside a
// 16-lane group, lanes 4 i .. 4 i + 3 address row i of a [4][16] block and lane c receives column c), rows k16(R) + i and k16(R) + 16 + i.
// The two 16-lane groups of a 32-lane LDS group read rows 8 apart: same bank base (8 x 192 B = 6 x 256 B), XOR swizzle differing in bit 1
// -> the two halves of one 64-byte window, as in the 32-row fragment: conflict-free.
// Result.  Lane (R, c) holds X^T[16 t + 4 R + g][column], g = 0..3, for columns c (from X) and 16 + c (from Y): 8 contiguous bytes of
// two output rows per tile.
struct Tr16Addr { int tb[2][2]; };   // [read e][tile parity]: the swizzle XOR acts on chunk bits 0-1 and a 16-wide tile starts at chunk 2 t
__device__ __forceinline__ void tr16_addr(Tr16Addr& ta, int lane) {
  const int gg = lane >> 4, tt = lane & 15, x = (tt & 3) >> 1;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const int row = 8 * (gg & 1) + 4 * (gg >> 1) + (tt >> 2) + 16 * e, sw = (row >> 2) & 3;
#pragma unroll
    for (int par = 0; par < 2; par++) ta.tb[e][par] = row * ROWB + (((2 * par + x) ^ sw) << 4) + (tt & 1) * 8;
  }
}
__device__ __forceinline__ bf16x8 trfrag16(const char* lds, const Tr16Addr& ta, int t, int sub) {
  const int off = sub * 32 * ROWB + (t >> 1) * 64;
  return concat_tr(lds_tr_read(lds + ta.tb[0][t & 1] + off), lds_tr_read(lds + ta.tb[1][t & 1] + off));
}
__device__ __forceinline__ void pack_xy(const f32x16& v, bf16x8& x, bf16x8& y) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 4; j++) { a[j] = (bf16_t)v[j]; a[4 + j] = (bf16_t)v[8 + j]; b[j] = (bf16_t)v[4 + j]; b[4 + j] = (bf16_t)v[12 + j]; }
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
  u32x4 ux, uy;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const auto r = __builtin_amdgcn_permlane16_swap(ua[w], ub[w], false, false);
    ux[w] = r[0]; uy[w] = r[1];
  }
  x = __builtin_bit_cast(bf16x8, ux);
  y = __builtin_bit_cast(bf16x8, uy);
}
// Used by the forward and dQ kernels (dQ -5.7 %, forward -1.5 % against the 32-row form, profiles/r02c_attn_pv16_ab.txt).  The dK/dV kernel - two
// such products and two re-layouts per block, and the heaviest softmax beside them - measured 3-7 % SLOWER in every variant of it (both products, one
// of them, MFMAs interleaved by hand: profiles/r02d_attn_dkv_modes.txt; the 16-cycle MFMAs leave the SIMD's other wave fewer issue slots) and keeps
// its 32-row tiles.
constexpr int NT16 = 5;              // ceil(72 / 16) output tiles
struct Acc16 { f32x4 v[NT16][2]; };  // [d tile][column half]: lane (R, c) <-> d = 16 t + 4 R + g, output row c / 16 + c of the wave's 32
__device__ __forceinline__ void zero16(Acc16& a) {
#pragma unroll
  for (int t = 0; t < NT16; t++)
#pragma unroll
    for (int h = 0; h < 2; h++) a.v[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void mma16(Acc16& acc, const char* lds, const Tr16Addr& ta, int sub, const bf16x8& x, const bf16x8& y) {
#pragma unroll
  for (int t = 0; t < NT16; t++) {
    const bf16x8 af = trfrag16(lds, ta, t, sub);
    acc.v[t][0] = mfma16(af, x, acc.v[t][0]);
    acc.v[t][1] = mfma16(af, y, acc.v[t][1]);
  }
}
// rows X[row0 + c][0..71] and X[row0 + 16 + c][0..71] (token stride ts); mul0 / mul1 scale the two rows, ok0 / ok1 guard them
__device__ __forceinline__ void store_rows16(bf16_t* __restrict__ base, long ts, const Acc16& acc, float mul0, float mul1, bool ok0, bool ok1, int lane) {
  const int R = lane >> 4, c = lane & 15;
  bf16_t* r0 = base + (long)c * ts;
  bf16_t* r1 = base + (long)(16 + c) * ts;
#pragma unroll
  for (int t = 0; t < NT16; t++) {
    const int d0 = 16 * t + 4 * R;
    if (16 * t + 12 < DH || d0 < DH) {                   // compile-time for t < 4, lane rows 0 / 1 of the last tile
      if (ok0) *reinterpret_cast<uint2*>(r0 + d0) = pack_bf16x4(acc.v[t][0][0] * mul0, acc.v[t][0][1] * mul0, acc.v[t][0][2] * mul0, acc.v[t][0][3] * mul0);
      if (ok1) *reinterpret_cast<uint2*>(r1 + d0) = pack_bf16x4(acc.v[t][1][0] * mul1, acc.v[t][1][1] * mul1, acc.v[t][1][2] * mul1, acc.v[t][1][3] * mul1);
    }
  }
}
// bias gradient of the Linear that produced this operand (see colsum_rows): colsum[d] += sum over the wave's 32 rows
__device__ __forceinline__ void colsum_rows16(float* __restrict__ dst, const Acc16& acc, float mul, bool ok0, bool ok1, int lane) {
  const int R = lane >> 4, c = lane & 15;
#pragma unroll
  for (int t = 0; t < NT16; t++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      float v = (ok0 ? acc.v[t][0][g] : 0.f) + (ok1 ? acc.v[t][1][g] : 0.f);
      v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
      const int d = 16 * t + 4 * R + g;
      if (c == 0 && d < DH) atomicAdd(dst + d, v * mul);
    }
}


// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // 2 stages x {K, V}
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const int q = bx * 128 + wave * 32 + (lane & 31);
  const bool qvalid = q < p.Nq;
  long kbase, vbase, d0_, d1_; int kvlen;
  kv_range(p, b, kbase, vbase, d0_, d1_, kvlen);
  const bf16_t* Kp = p.K + kbase + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + vbase + (long)h * p.v_hs;
  const int kts = (int)p.k_ts, vts = (int)p.v_ts;

  bf16x8 qf[KSTEPS];
  load_row_frags(qf, p.Q + (long)b * p.q_bs + (long)q * p.q_ts + (long)h * p.q_hs, qvalid, hi);
  settle(qf);
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  FragAddr fa;
  frag_addr(fa, lane);

  for (int st = 0; st < 2; st++) {
    init_pads(smem + st * 2 * TILE_B, 0, tid);                // K
    init_pads(smem + st * 2 * TILE_B + TILE_B, 1, tid);       // V: column 72 = 1 -> O^T row 72 accumulates sum_kv P = l
  }
  Acc16 o;
  zero16(o);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  float m = -INFINITY;
  const float c = p.scale_log2;

  auto tile = [&](auto tailc, const char* sK, const char* sV, int kv0) {
    constexpr bool TAIL = decltype(tailc)::value;
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
#pragma unroll
      for (int g = 0; g < 16; g++) s[sub][g] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) s[sub] = mfma32(rowfrag(sK, fa, sub, ks), qf[ks], s[sub]);
    }
    if (TAIL) {
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int g = 0; g < 16; g++)
          if (kv0 + sub * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi >= kvlen) s[sub][g] = -INFINITY;
    }
    float mt = s[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int g = 0; g < 16; g++) mt = fmaxf(mt, s[sub][g]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    // Deferred rescale (guide T13): keep the stale running max while no query of this wave grew by more than 2^RESCALE_LOG2;
    // P is then bounded by 2^RESCALE_LOG2 instead of 1 (bf16 keeps its relative precision, l and O are fp32).  When the branch
    // fires, O and l — everything still expressed against the old max — are scaled exactly once, before any P of this tile.
    if (__builtin_amdgcn_readfirstlane(__any((mt - m) * c > RESCALE_LOG2))) {
      asm volatile("" ::: "memory");       // keep this a real (rare) branch: if-conversion would run the 48 multiplies every tile
      const float mn = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
      m = mn;
      const float ao = __shfl_xor(alpha, 16);           // the accumulators of lane (R, c) belong to queries c and 16 + c, alpha to query l & 31
      const float a0 = (lane & 16) ? ao : alpha, a1 = (lane & 16) ? alpha : ao;
#pragma unroll
      for (int t = 0; t < NT16; t++) { o.v[t][0] *= a0; o.v[t][1] *= a1; }   // includes the row-sum row (d = 72)
    }
    const float mc = m * c;
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int g = 0; g < 16; g++) s[sub][g] = __builtin_amdgcn_exp2f(s[sub][g] * c - mc);
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      bf16x8 px, py;
      pack_xy(s[sub], px, py);
      mma16(o, sV, ta, sub, px, py);
    }
  };

  const int Tfull = kvlen / BKV, rem = kvlen - Tfull * BKV, T = Tfull + (rem ? 1 : 0);
  auto issue = [&](int t) {                // DMA of tile t into stage t&1
    char* nx = smem + (t & 1) * 2 * TILE_B;
    if (t < Tfull) {
      dma_tile<true>(nx, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<true>(nx + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    } else {
      dma_tile<false>(nx, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<false>(nx + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    }
  };
  if (T > 0) issue(0);
  for (int t = 0; t < Tfull; t++) {
    tile_sync();                           // own DMA drained (vmcnt(0)) + stage hand-over; ONE barrier per tile
    if (t + 1 < T) issue(t + 1);
    const char* st = smem + (t & 1) * 2 * TILE_B;
    tile(BoolC<false>{}, st, st + TILE_B, t * BKV);
  }
  if (rem) {                               // ragged last tile: the only place that pays for masking
    tile_sync();
    const char* st = smem + (Tfull & 1) * 2 * TILE_B;
    tile(BoolC<true>{}, st, st + TILE_B, Tfull * BKV);
  }
  // row 72 of O^T (tile 4, lane row 2, g = 0: lanes 32 + c) = sum over keys of the bf16 P actually multiplied into O
  const float la = __shfl(o.v[4][0][0], 32 + (lane & 15)), lb = __shfl(o.v[4][1][0], 32 + (lane & 15));
  const float l = (lane & 16) ? lb : la;               // of query l & 31
  const float inv = l > 0.f ? 1.f / l : 0.f, invo = __shfl_xor(inv, 16);
  const int q0w = bx * 128 + wave * 32;
  store_rows16(p.O + (long)b * p.o_bs + (long)q0w * p.o_ts + (long)h * p.o_hs, p.o_ts, o, (lane & 16) ? invo : inv, (lane & 16) ? inv : invo,
               q0w + (lane & 15) < p.Nq, q0w + 16 + (lane & 15) < p.Nq, lane);
  if (qvalid && hi == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Nq + q] = m * c + log2f(l);
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
// "stats rows" (round 3).  The dK/dV kernel needs lse and delta of every QUERY of a tile, i.e. along its accumulator rows - 16 LDS reads, 32 subtractions
// and a staging write per tile when they are applied on the VALU.  Instead the pre-pass also writes them as 16-byte rows of the operand type,
//   Lrow[b][h][q] = {hi, mid, lo, 0 x 5} summing to lse / scale_log2,        Drow[b][h][q] = {hi, mid, lo, 0 x 5} summing to delta     (split3),
// the dK/dV kernel brings 64 of each per tile into LDS with ONE 1 KiB LDS-DMA instruction, and its lanes read them as the k-slots 72..79 of the Q / dO
// operand (the zero padding of head_dim 72 -> 80) against -1.0 in the K / V registers: the first products return S - lse / c and dP - delta.
// Three terms carry 24 (bf16) / 33 (fp16) mantissa bits.  Rows [Nq, Nq64) hold a huge finite lse (P = 0 for queries that do not exist) and delta 0.
#ifdef PXA_OPERAND_F16
#define PXA_STAT_SENTINEL 60000.0f   // fits fp16; exp2(c (S - 6e4)) underflows to 0 for every scale this model uses (c = 0.17)
#else
#define PXA_STAT_SENTINEL 1.0e30f
#endif
__device__ __forceinline__ void write_stat_rows(bf16_t* __restrict__ stats, long rows_total, long row, float l_over_c, float dl) {
  if (!stats) return;
  const uint2 l3 = split3(l_over_c), d3 = split3(dl);
  *reinterpret_cast<uint4*>(stats + row * 8) = make_uint4(l3.x, l3.y, 0, 0);
  *reinterpret_cast<uint4*>(stats + (rows_total + row) * 8) = make_uint4(d3.x, d3.y, 0, 0);
}
__global__ __launch_bounds__(256) void attn_stats_pad_kernel(bf16_t* __restrict__ stats, int BH, int Nq, int Nq64) {
  const int pad = Nq64 - Nq, idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= BH * pad) return;
  const int bh = idx / pad, q = Nq + idx - bh * pad;
  write_stat_rows(stats, (long)BH * Nq64, (long)bh * Nq64 + q, PXA_STAT_SENTINEL, 0.f);
}
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, float* __restrict__ delta,
                                                         long o_bs, long o_ts, int o_hs, long do_bs, long do_ts, int do_hs, int B, int H, int Nq,
                                                         const float* __restrict__ lse, bf16_t* __restrict__ stats, int Nq64, float inv_c) {
  const long idx = blockIdx.x * 256L + threadIdx.x;  // (b, q, h) with h fastest -> adjacent threads read adjacent 144-byte rows
  if (idx >= (long)B * Nq * H) return;
  const int h = idx % H;
  const long t = idx / H;
  const int q = t % Nq, b = t / Nq;
  const bf16_t* po = O + b * o_bs + q * o_ts + (long)h * o_hs;
  const bf16_t* pd = dO + b * do_bs + q * do_ts + (long)h * do_hs;
  float acc = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) {
    float a[8], d[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(po + ch * 8), a);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(pd + ch * 8), d);
#pragma unroll
    for (int e = 0; e < 8; e++) acc += a[e] * d[e];
  }
  delta[((long)b * H + h) * Nq + q] = acc;
  if (stats) write_stat_rows(stats, (long)B * H * Nq64, ((long)b * H + h) * Nq64 + q, lse[((long)b * H + h) * Nq + q] * inv_c, acc);
}

// Same for token-contiguous O / dO ([B][Nq][H][72], what the engine passes): the kernel above reads adjacent 144-byte rows from adjacent
// lanes (every 16-byte load of a wave touches 64 different cache lines) and scatters its 4-byte results Nq floats apart.  Here a block
// takes 16 tokens = 2,304 16-byte chunks, reads them fully coalesced (9 per thread), parks the per-chunk partial dot products in LDS and
// lets thread (h, q) add the 9 partials of its head: 64-byte runs of delta per head.
constexpr int DELTA_TOK = 16;
__global__ __launch_bounds__(256) void attn_delta_rows_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO, float* __restrict__ delta,
                                                              int H, int Nq, long tokens, const float* __restrict__ lse, bf16_t* __restrict__ stats,
                                                              int Nq64, float inv_c) {
  __shared__ float part[DELTA_TOK * 16 * NCH + 16];
  const long tok0 = (long)blockIdx.x * DELTA_TOK;
  const int cpt = H * NCH;                                 // chunks per token (144)
  const long nchunks = min((long)DELTA_TOK, tokens - tok0) * cpt;
  const uint4* po = reinterpret_cast<const uint4*>(O) + tok0 * cpt;
  const uint4* pd = reinterpret_cast<const uint4*>(dO) + tok0 * cpt;
  uint4 a[NCH], d[NCH];
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = threadIdx.x + 256 * i;
    if (c < nchunks) { a[i] = ld_u4(po + c); d[i] = ld_u4(pd + c); }
  }
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = threadIdx.x + 256 * i;
    if (c < nchunks) {
      float fa[8], fd[8], acc = 0.f;
      unpack_bf16x8(a[i], fa); unpack_bf16x8(d[i], fd);
#pragma unroll
      for (int e = 0; e < 8; e++) acc += fa[e] * fd[e];
      part[c] = acc;
    }
  }
  __syncthreads();
  const int ql = threadIdx.x % DELTA_TOK, h = threadIdx.x / DELTA_TOK;   // H <= 16
  const long tok = tok0 + ql;
  if (h < H && tok < tokens) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; i++) acc += part[ql * cpt + h * NCH + i];
    const long b = tok / Nq, q = tok - b * Nq;
    delta[(b * H + h) * Nq + q] = acc;
    if (stats) write_stat_rows(stats, (tokens / Nq) * H * Nq64, (b * H + h) * Nq64 + q, lse[(b * H + h) * Nq + q] * inv_c, acc);
  }
}

// Forward with TWO query sub-tiles per wave (256 queries per workgroup).  A K or V^T fragment read from LDS now feeds two MFMAs,
// and a K/V tile pair fetched by LDS-DMA serves twice as many queries: the single-sub-tile kernel above moves 12.9 GB from L2 into
// LDS per self-attention launch (8.6 TB/s - the same ballpark as the GEMM's DMA-only ceiling) and issues 1.9 LDS instructions per
// MFMA.  Costs: 96 + 64 accumulator registers (two waves per SIMD instead of three).
#ifndef FWD_ABL
#define FWD_ABL 0       // ablation study of the forward loop (tools/build_variant.py): 1 = no exp2, 2 = no cvt / permlane, 3 = no softmax VALU at all, 4 = no MFMAs
#endif
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(AttnParams p) {
  constexpr int QS = 2;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // 2 stages x {K, V}
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  long kbase, vbase, d0_, d1_; int kvlen;
  kv_range(p, b, kbase, vbase, d0_, d1_, kvlen);
  const bf16_t* Kp = p.K + kbase + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + vbase + (long)h * p.v_hs;
  const int kts = (int)p.k_ts, vts = (int)p.v_ts;

  int q[QS];
  bool qvalid[QS];
  bf16x8 qf[QS][KSTEPS];
#pragma unroll
  for (int s = 0; s < QS; s++) {
    q[s] = bx * 256 + wave * 64 + s * 32 + (lane & 31);
    qvalid[s] = q[s] < p.Nq;
    load_row_frags(qf[s], p.Q + (long)b * p.q_bs + (long)q[s] * p.q_ts + (long)h * p.q_hs, qvalid[s], hi);
    settle(qf[s]);
  }
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  FragAddr fa;
  frag_addr(fa, lane);
  for (int st = 0; st < 2; st++) {
    init_pads(smem + st * 2 * TILE_B, 0, tid);                // K
    init_pads(smem + st * 2 * TILE_B + TILE_B, 1, tid);       // V: column 72 = 1 -> O^T row 72 accumulates sum_kv P = l
  }
  Acc16 o[QS];
  Tr16Addr ta;
  tr16_addr(ta, lane);
  float m[QS];
#pragma unroll
  for (int s = 0; s < QS; s++) {
    zero16(o[s]);
    m[s] = -INFINITY;
  }
  const float c = p.scale_log2;

  auto tile = [&](auto tailc, const char* sK, const char* sV, int kv0) {
    constexpr bool TAIL = decltype(tailc)::value;
    f32x16 sc[QS][2];
#pragma unroll
    for (int s = 0; s < QS; s++)
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int g = 0; g < 16; g++) sc[s][sub][g] = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const bf16x8 kf = rowfrag(sK, fa, sub, ks);          // one LDS read, two MFMAs
#if FWD_ABL == 4
        asm volatile("" :: "v"(kf));
#else
#pragma unroll
        for (int s = 0; s < QS; s++) sc[s][sub] = mfma32(kf, qf[s][ks], sc[s][sub]);
#endif
      }
#if FWD_ABL == 3
    // (ablation: no softmax at all - the score registers feed the pack as they are)
#else
#pragma unroll
    for (int s = 0; s < QS; s++) {
      if (TAIL) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
          for (int g = 0; g < 16; g++)
            if (kv0 + sub * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi >= kvlen) sc[s][sub][g] = -INFINITY;
      }
      float mt = sc[s][0][0];
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int g = 0; g < 16; g++) mt = fmaxf(mt, sc[s][sub][g]);
      mt = fmaxf(mt, __shfl_xor(mt, 32));
      if (__builtin_amdgcn_readfirstlane(__any((mt - m[s]) * c > RESCALE_LOG2))) {   // deferred rescale, see attn_fwd_kernel
        asm volatile("" ::: "memory");
        const float mn = fmaxf(m[s], mt);
        const float alpha = __builtin_amdgcn_exp2f((m[s] - mn) * c);
        m[s] = mn;
        const float ao = __shfl_xor(alpha, 16);
        const float a0 = (lane & 16) ? ao : alpha, a1 = (lane & 16) ? alpha : ao;
#pragma unroll
        for (int t = 0; t < NT16; t++) { o[s].v[t][0] *= a0; o[s].v[t][1] *= a1; }
      }
      const float mc = m[s] * c;
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
#if FWD_ABL == 1
          sc[s][sub][g] = sc[s][sub][g] * c - mc;             // (ablation: no exp2)
#else
          sc[s][sub][g] = __builtin_amdgcn_exp2f(sc[s][sub][g] * c - mc);
#endif
        }
    }
#endif
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      bf16x8 px[QS], py[QS];
#pragma unroll
      for (int s = 0; s < QS; s++) {
#if FWD_ABL == 2 || FWD_ABL == 3
        const f32x4 lo4 = {sc[s][sub][0], sc[s][sub][1], sc[s][sub][2], sc[s][sub][3]}, hi4 = {sc[s][sub][4], sc[s][sub][5], sc[s][sub][6], sc[s][sub][7]};
        px[s] = __builtin_bit_cast(bf16x8, lo4);                // (ablation: no cvt / permlane)
        py[s] = __builtin_bit_cast(bf16x8, hi4);
        asm volatile("" : "+v"(px[s]), "+v"(py[s]));
#else
        pack_xy(sc[s][sub], px[s], py[s]);
#endif
      }
#pragma unroll
      for (int t = 0; t < NT16; t++) {
        const bf16x8 vf = trfrag16(sV, ta, t, sub);          // one transposed fragment, four MFMAs
#if FWD_ABL == 4
        asm volatile("" :: "v"(vf), "v"(px[0]), "v"(py[0]), "v"(px[1]), "v"(py[1]));
#else
#pragma unroll
        for (int s = 0; s < QS; s++) {
          o[s].v[t][0] = mfma16(vf, px[s], o[s].v[t][0]);
          o[s].v[t][1] = mfma16(vf, py[s], o[s].v[t][1]);
        }
#endif
      }
    }
  };

  const int Tfull = kvlen / BKV, rem = kvlen - Tfull * BKV, T = Tfull + (rem ? 1 : 0);
  auto issue = [&](int t) {                // DMA of tile t into stage t&1
    char* nx = smem + (t & 1) * 2 * TILE_B;
    if (t < Tfull) {
      dma_tile<true>(nx, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<true>(nx + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    } else {
      dma_tile<false>(nx, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<false>(nx + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    }
  };
  if (T > 0) issue(0);
  for (int t = 0; t < Tfull; t++) {
    tile_sync();                           // own DMA drained (vmcnt(0)) + stage hand-over; ONE barrier per tile
    if (t + 1 < T) issue(t + 1);
    const char* st = smem + (t & 1) * 2 * TILE_B;
    tile(BoolC<false>{}, st, st + TILE_B, t * BKV);
  }
  if (rem) {
    tile_sync();
    const char* st = smem + (Tfull & 1) * 2 * TILE_B;
    tile(BoolC<true>{}, st, st + TILE_B, Tfull * BKV);
  }
#pragma unroll
  for (int s = 0; s < QS; s++) {
    const float la = __shfl(o[s].v[4][0][0], 32 + (lane & 15)), lb = __shfl(o[s].v[4][1][0], 32 + (lane & 15));
    const float l = (lane & 16) ? lb : la;             // row 72 of O^T = sum over keys of the bf16 P actually multiplied into O, of query l & 31
    const float inv = l > 0.f ? 1.f / l : 0.f, invo = __shfl_xor(inv, 16);
    const int q0w = bx * 256 + wave * 64 + s * 32;
    store_rows16(p.O + (long)b * p.o_bs + (long)q0w * p.o_ts + (long)h * p.o_hs, p.o_ts, o[s], (lane & 16) ? invo : inv, (lane & 16) ? inv : invo,
                 q0w + (lane & 15) < p.Nq, q0w + 16 + (lane & 15) < p.Nq, lane);
    if (qvalid[s] && hi == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Nq + q[s]] = m[s] * c + log2f(l);
  }
}

// ------------------------------------------------------------------------------------------------ forward, all keys resident (cross-attention)
// Text keys are few (L <= 300): with one 256-query workgroup per launch unit, as above, a cross-attention forward is 4,096 workgroups that each
// pay a query load, the pads, a first LDS-DMA and a barrier per 64-key tile for five tiles of work - 100 of its 165 us do not depend on L
// (profiles/r03zd_cross_vs_len.txt).  Here ONE 512-thread workgroup loads ALL of a sample's K and V rows for its head once (<= KVRES_TILES tiles, both
// operands: <= 120 KiB, one workgroup per CU, two waves per SIMD) and then walks `qpb` queries, 64 per wave and trip, with no DMA and no barrier in
// the loop: the tile body is attn_fwd2_kernel's, reading the resident tiles.
constexpr int KVRES_TILES = 5;                       // 320 keys
__global__ __launch_bounds__(512, 1) void attn_fwd_kvres_kernel(AttnParams p, int qpb, int tiles_alloc) {
  constexpr int QS = 2;
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char* smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  long kbase, vbase, d0_, d1_; int kvlen;
  kv_range(p, b, kbase, vbase, d0_, d1_, kvlen);
  const bf16_t* Kp = p.K + kbase + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + vbase + (long)h * p.v_hs;
  kvlen = min(kvlen, tiles_alloc * BKV);                       // (the host sized the LDS for max_kv_len: a longer sample would be a caller error)
  const int Tfull = kvlen / BKV, rem = kvlen - Tfull * BKV, T = Tfull + (rem ? 1 : 0);
  {   // waves 0-3 fetch the K tiles, waves 4-7 the V tiles (each group is the four-wave DMA team of the kernels above); pads once per tile
    DmaPlan pl;
    dma_plan(pl, wave & 3, lane);
    const bool vside = wave >= 4;
    for (int t = 0; t < T; t++) {
      char* dst = smem + t * 2 * TILE_B + (vside ? TILE_B : 0);
      init_pads(dst, vside ? 1 : 0, tid & 255);               // V: column 72 = 1 -> O^T row 72 accumulates sum_kv P = l
      if (t < Tfull) dma_tile<true>(dst, vside ? Vp : Kp, vside ? (int)p.v_ts : (int)p.k_ts, t * BKV, kvlen, pl, wave & 3);
      else dma_tile<false>(dst, vside ? Vp : Kp, vside ? (int)p.v_ts : (int)p.k_ts, t * BKV, kvlen, pl, wave & 3);
    }
  }
  FragAddr fa;
  frag_addr(fa, lane);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  const float c = p.scale_log2;
  tile_sync();
  for (int q0b = bx * qpb; q0b < min(p.Nq, (bx + 1) * qpb); q0b += 512) {
    const int q0w = q0b + wave * 64;
    if (q0w >= p.Nq) break;                                    // wave-uniform; nothing below synchronises the workgroup
    int q[QS];
    bool qvalid[QS];
    bf16x8 qf[QS][KSTEPS];
#pragma unroll
    for (int s = 0; s < QS; s++) {
      q[s] = q0w + s * 32 + (lane & 31);
      qvalid[s] = q[s] < p.Nq;
      load_row_frags(qf[s], p.Q + (long)b * p.q_bs + (long)q[s] * p.q_ts + (long)h * p.q_hs, qvalid[s], hi);
    }
    Acc16 o[QS];
    float m[QS];
#pragma unroll
    for (int s = 0; s < QS; s++) {
      zero16(o[s]);
      m[s] = -INFINITY;
    }
    auto tile = [&](auto tailc, const char* sK, const char* sV, int kv0) {
      constexpr bool TAIL = decltype(tailc)::value;
      f32x16 sc[QS][2];
#pragma unroll
      for (int s = 0; s < QS; s++)
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
          for (int g = 0; g < 16; g++) sc[s][sub][g] = 0.f;
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
          const bf16x8 kf = rowfrag(sK, fa, sub, ks);
#pragma unroll
          for (int s = 0; s < QS; s++) sc[s][sub] = mfma32(kf, qf[s][ks], sc[s][sub]);
        }
#pragma unroll
      for (int s = 0; s < QS; s++) {
        if (TAIL) {
#pragma unroll
          for (int sub = 0; sub < 2; sub++)
#pragma unroll
            for (int g = 0; g < 16; g++)
              if (kv0 + sub * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi >= kvlen) sc[s][sub][g] = -INFINITY;
        }
        float mt = sc[s][0][0];
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
          for (int g = 0; g < 16; g++) mt = fmaxf(mt, sc[s][sub][g]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        if (__builtin_amdgcn_readfirstlane(__any((mt - m[s]) * c > RESCALE_LOG2))) {   // deferred rescale, see attn_fwd_kernel
          asm volatile("" ::: "memory");
          const float mn = fmaxf(m[s], mt);
          const float alpha = __builtin_amdgcn_exp2f((m[s] - mn) * c);
          m[s] = mn;
          const float ao = __shfl_xor(alpha, 16);
          const float a0 = (lane & 16) ? ao : alpha, a1 = (lane & 16) ? alpha : ao;
#pragma unroll
          for (int t = 0; t < NT16; t++) { o[s].v[t][0] *= a0; o[s].v[t][1] *= a1; }
        }
        const float mc = m[s] * c;
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
          for (int g = 0; g < 16; g++) sc[s][sub][g] = __builtin_amdgcn_exp2f(sc[s][sub][g] * c - mc);
      }
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        bf16x8 px[QS], py[QS];
#pragma unroll
        for (int s = 0; s < QS; s++) pack_xy(sc[s][sub], px[s], py[s]);
#pragma unroll
        for (int t = 0; t < NT16; t++) {
          const bf16x8 vf = trfrag16(sV, ta, t, sub);
#pragma unroll
          for (int s = 0; s < QS; s++) {
            o[s].v[t][0] = mfma16(vf, px[s], o[s].v[t][0]);
            o[s].v[t][1] = mfma16(vf, py[s], o[s].v[t][1]);
          }
        }
      }
    };
    for (int t = 0; t < Tfull; t++) tile(BoolC<false>{}, smem + t * 2 * TILE_B, smem + t * 2 * TILE_B + TILE_B, t * BKV);
    if (rem) tile(BoolC<true>{}, smem + Tfull * 2 * TILE_B, smem + Tfull * 2 * TILE_B + TILE_B, Tfull * BKV);
#pragma unroll
    for (int s = 0; s < QS; s++) {
      const float la = __shfl(o[s].v[4][0][0], 32 + (lane & 15)), lb = __shfl(o[s].v[4][1][0], 32 + (lane & 15));
      const float l = (lane & 16) ? lb : la;
      const float inv = l > 0.f ? 1.f / l : 0.f, invo = __shfl_xor(inv, 16);
      const int q0s = q0w + s * 32;
      store_rows16(p.O + (long)b * p.o_bs + (long)q0s * p.o_ts + (long)h * p.o_hs, p.o_ts, o[s], (lane & 16) ? invo : inv, (lane & 16) ? inv : invo,
                   q0s + (lane & 15) < p.Nq, q0s + 16 + (lane & 15) < p.Nq, lane);
      if (qvalid[s] && hi == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Nq + q[s]] = m[s] * c + log2f(l);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
#ifndef ATTN_BWD_WAVES
#define ATTN_BWD_WAVES 2
#endif
#ifndef PXA_ATTN_DKV_DEFAULT
#define PXA_ATTN_DKV_DEFAULT 2
#endif
#ifndef PXA_ATTN_DQ_DEFAULT
#define PXA_ATTN_DQ_DEFAULT 1
#endif
#ifndef ATTN_ABL
#define ATTN_ABL 0      // ablation study of the dK/dV kernel (tools/build_variant.py; results in profiles/r02_attention_bwd_experiments.md):
#endif                  // 1 no softmax VALU, 2 no dV/dK MFMAs, 4 no S/dP MFMAs, 8 no LDS fragment reads, 16 no LDS-DMA.  0 = the product kernel.
__global__ __launch_bounds__(256, ATTN_BWD_WAVES) void attn_bwd_dq_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // 2 stages x {K, V}
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const int q = bx * 128 + wave * 32 + (lane & 31);
  const bool qvalid = q < p.Nq;
  long kbase, vbase, d0_, d1_; int kvlen;
  kv_range(p, b, kbase, vbase, d0_, d1_, kvlen);
  const bf16_t* Kp = p.K + kbase + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + vbase + (long)h * p.v_hs;
  const int kts = (int)p.k_ts, vts = (int)p.v_ts;

  bf16x8 qf[KSTEPS], dof[KSTEPS];
  load_row_frags(qf, p.Q + (long)b * p.q_bs + (long)q * p.q_ts + (long)h * p.q_hs, qvalid, hi);
  load_row_frags(dof, p.dO + (long)b * p.o_bs + (long)q * p.o_ts + (long)h * p.o_hs, qvalid, hi);
  settle(qf);
  settle(dof);
  const long sidx = ((long)b * p.H + h) * p.Nq + q;
  const float lse = qvalid ? p.LSE[sidx] : 0.f;
  const float delta = qvalid ? p.Delta[sidx] : 0.f;
  if (ATTN_FOLD_DELTA && hi == 1) {                             // slots 72 .. 74 of this lane's dO row (k-step 4, upper half: d = 72 .. 79)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, dof[KSTEPS - 1]);
    const uint2 d3 = split3(delta);
    w[0] = d3.x; w[1] = d3.y;
    dof[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
  }
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  FragAddr fa;
  frag_addr(fa, lane);

  for (int st = 0; st < 4; st++) init_pads(smem + st * TILE_B, (ATTN_FOLD_DELTA && (st & 1)) ? 2 : 0, tid);   // odd tiles = V: -1.0 in slots 72 .. 74
  Acc16 dq;
  zero16(dq);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  const float c = p.scale_log2;
  auto tile = [&](auto tailc, const char* sK, const char* sV, int kv0) {
    constexpr bool TAIL = decltype(tailc)::value;
    f32x16 s[2], dp[2];
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
#pragma unroll
      for (int g = 0; g < 16; g++) { s[sub][g] = 0.f; dp[sub][g] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        s[sub] = mfma32(rowfrag(sK, fa, sub, ks), qf[ks], s[sub]);
        dp[sub] = mfma32(rowfrag(sV, fa, sub, ks), dof[ks], dp[sub]);
      }
    }
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int g = 0; g < 16; g++) {
        float pr = __builtin_amdgcn_exp2f(s[sub][g] * c - lse);
        if (TAIL && kv0 + sub * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi >= kvlen) pr = 0.f;
        s[sub][g] = ATTN_FOLD_DELTA ? pr * dp[sub][g] : pr * (dp[sub][g] - delta);  // dS^T (without the softmax scale, applied at the end)
      }
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      bf16x8 dx, dy;
      pack_xy(s[sub], dx, dy);
      mma16(dq, sK, ta, sub, dx, dy);
    }
  };
  const int Tfull = kvlen / BKV, rem = kvlen - Tfull * BKV, T = Tfull + (rem ? 1 : 0);
  auto issue = [&](int t) {                // DMA of tile t into stage t&1
    char* nx = smem + (t & 1) * 2 * TILE_B;
    if (t < Tfull) {
      dma_tile<true>(nx, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<true>(nx + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    } else {
      dma_tile<false>(nx, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<false>(nx + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    }
  };
  if (T > 0) issue(0);
  for (int t = 0; t < Tfull; t++) {
    tile_sync();                           // own DMA drained (vmcnt(0)) + stage hand-over; ONE barrier per tile
    if (t + 1 < T) issue(t + 1);
    const char* st = smem + (t & 1) * 2 * TILE_B;
    tile(BoolC<false>{}, st, st + TILE_B, t * BKV);
  }
  if (rem) {                               // ragged last tile: the only place that pays for masking
    tile_sync();
    const char* st = smem + (Tfull & 1) * 2 * TILE_B;
    tile(BoolC<true>{}, st, st + TILE_B, Tfull * BKV);
  }
  const int q0w = bx * 128 + wave * 32;
  const bool ok0 = q0w + (lane & 15) < p.Nq, ok1 = q0w + 16 + (lane & 15) < p.Nq;
  store_rows16(p.dQ + (long)b * p.dq_bs + (long)q0w * p.dq_ts + (long)h * p.dq_hs, p.dq_ts, dq, p.scale, p.scale, ok0, ok1, lane);
  if (p.dq_colsum) colsum_rows16(p.dq_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dq, p.scale, ok0, ok1, lane);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(256, ATTN_BWD_WAVES) void attn_bwd_dkv_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B + 4 * BKV * 4];   // 2 stages x {Q, dO} + 2 stages x {lse, delta}
  float* ldsL = reinterpret_cast<float*>(smem + 4 * TILE_B);                      // [2][2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  long kbase, vbase, dkbase, dvbase; int kvlen;
  kv_range(p, b, kbase, vbase, dkbase, dvbase, kvlen);
  if (bx * 128 >= kvlen) return;  // whole block beyond this sample's keys (uniform across the block)
  const int kv = bx * 128 + wave * 32 + (lane & 31);
  const bool kvvalid = kv < kvlen;
  // A wave whose 32 keys all lie beyond the sample's length (text keys: 300 = 128 + 128 + 32 + 12) still serves the block's LDS-DMA and
  // barriers but issues no MFMA / softmax work: the matrix pipe of its SIMD is left to the co-resident workgroup's wave.
  const bool wave_active = bx * 128 + wave * 32 < kvlen;

  bf16x8 kf[KSTEPS], vf[KSTEPS];
  load_row_frags(kf, p.K + kbase + (long)kv * p.k_ts + (long)h * p.k_hs, kvvalid, hi);
  load_row_frags(vf, p.V + vbase + (long)kv * p.v_ts + (long)h * p.v_hs, kvvalid, hi);
  settle(kf);
  settle(vf);
  const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs;
  const bf16_t* Dp = p.dO + (long)b * p.o_bs + (long)h * p.o_hs;
  const float* Lp = p.LSE + ((long)b * p.H + h) * p.Nq;
  const float* Dl = p.Delta + ((long)b * p.H + h) * p.Nq;
  const int qts = (int)p.q_ts, ots = (int)p.o_ts;
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  FragAddr fa;
  frag_addr(fa, lane);

  for (int st = 0; st < 4; st++) init_pads(smem + st * TILE_B, 0, tid);
  f32x16 dk[3], dv[3];
  zero3(dk);
  zero3(dv);
  const float c = p.scale_log2;
  const int T = (p.Nq + BKV - 1) / BKV;
  float rl = INFINITY, rdl = 0.f;
  auto fetch_stats = [&](int q0) {
    if (tid < BKV) {
      const bool ok = q0 + tid < p.Nq;
      rl = ok ? Lp[q0 + tid] : INFINITY;    // +inf -> P = exp2(-inf) = 0 for rows beyond Nq
      rdl = ok ? Dl[q0 + tid] : 0.f;
    }
  };
  const int Tfull = p.Nq / BKV;
  auto issue = [&](int t) {
    char* nx = smem + (t & 1) * 2 * TILE_B;
    if (t < Tfull) {
      dma_tile<true>(nx, Qp, qts, t * BKV, p.Nq, pl, wave);
      dma_tile<true>(nx + TILE_B, Dp, ots, t * BKV, p.Nq, pl, wave);
    } else {
      dma_tile<false>(nx, Qp, qts, t * BKV, p.Nq, pl, wave);
      dma_tile<false>(nx + TILE_B, Dp, ots, t * BKV, p.Nq, pl, wave);
    }
    fetch_stats(t * BKV);
  };
  issue(0);
  for (int t = 0; t < T; t++) {
    const char* sQ = smem + (t & 1) * 2 * TILE_B;
    const char* sD = sQ + TILE_B;
    float* sL = ldsL + (t & 1) * 2 * BKV;
    if (tid < BKV) { sL[tid] = rl; sL[BKV + tid] = rdl; }   // buffer (t&1) was last read two iterations ago
    tile_sync();
    if (!(ATTN_ABL & 16) && t + 1 < T) issue(t + 1);
    if (!wave_active) continue;
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      f32x16 s, dp;
#pragma unroll
      for (int g = 0; g < 16; g++) { s[g] = 0.f; dp[g] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const bf16x8 qa = (ATTN_ABL & 8) ? vf[(ks + sub) % KSTEPS] : rowfrag(sQ, fa, sub, ks);
        const bf16x8 da = (ATTN_ABL & 8) ? kf[(ks + sub) % KSTEPS] : rowfrag(sD, fa, sub, ks);
        if (ATTN_ABL & 4) {
          s[ks] += (float)qa[0] + (float)kf[ks][1];
          dp[ks] += (float)da[0] + (float)vf[ks][1];
        } else {
          s = mfma32(qa, kf[ks], s);    // S[q][kv], col = kv (lane), rows = q
          dp = mfma32(da, vf[ks], dp);  // dP[q][kv]
        }
      }
#pragma unroll
      for (int qd = 0; qd < 4; qd++) {
        const int ql = sub * 32 + 8 * qd + 4 * hi;
        const float4 L4 = *reinterpret_cast<const float4*>(&sL[ql]);
        const float4 D4 = *reinterpret_cast<const float4*>(&sL[BKV + ql]);
        const float Lv[4] = {L4.x, L4.y, L4.z, L4.w}, Dv[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (ATTN_ABL & 1) continue;        // ablation: no softmax arithmetic (packs S, dP as they are)
          const float pr = __builtin_amdgcn_exp2f(s[qd * 4 + e] * c - Lv[e]);
          s[qd * 4 + e] = pr;
          dp[qd * 4 + e] = pr * (dp[qd * 4 + e] - Dv[e]);
        }
      }
#pragma unroll
      for (int uu = 0; uu < 2; uu++) {
        const bf16x8 pb = pack8(s, 8 * uu), db = pack8(dp, 8 * uu);
        const int u = sub * 2 + uu;
#pragma unroll
        for (int dt = 0; dt < 3; dt++) {
          const bf16x8 dot = (ATTN_ABL & 8) ? kf[dt + uu] : trfrag(sD, fa, dt, u);
          const bf16x8 qt = (ATTN_ABL & 8) ? vf[dt + uu] : trfrag(sQ, fa, dt, u);
          if (ATTN_ABL & 2) {
            dv[dt][u] += (float)dot[0] * (float)pb[dt];
            dk[dt][u] += (float)qt[0] * (float)db[dt];
          } else {
            dv[dt] = mfma32(dot, pb, dv[dt]);
            dk[dt] = mfma32(qt, db, dk[dt]);
          }
        }
      }
    }
  }
  if (kvvalid) {
    store_rows(p.dK + dkbase + (long)kv * p.dk_ts + (long)h * p.dk_hs, dk, p.dk_scale, hi);
    store_rows(p.dV + dvbase + (long)kv * p.dv_ts + (long)h * p.dv_hs, dv, 1.f, hi);
  }
  if (p.dk_colsum) colsum_rows(p.dk_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dk, p.dk_scale, kvvalid, hi, lane);
  if (p.dv_colsum) colsum_rows(p.dv_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dv, 1.f, kvvalid, hi, lane);
}


// ------------------------------------------------------------------------------------------------ backward: dK, dV (round 3)
// Same products, layouts and fragments as attn_bwd_dkv_kernel above; what changes is the instruction stream (the old loop ran
// 61 cycles per MFMA against a 32-cycle floor with 6.7 other issues per MFMA, the two waves of a SIMD each alternating long MFMA-only and VALU-only stretches):
//   * lse and delta ride in the first products ("stats rows" above): per element mul, exp2, mul and half a cvt_pk are left (4 instead of 5 VALU), the 16
//     broadcast reads and the staging write per tile are gone, and so is every branch inside the tile loop.
//   * MODE 1, software pipeline over 32-query sub-tiles j: the dV / dK MFMAs of sub-tile j-1 are issued BETWEEN the softmax instructions of sub-tile j
//     (hand-placed slots of one MFMA, the two transpose reads of the MFMA two slots ahead and 4-6 VALU, fenced with sched_barrier(0):
//     sched_group_barrier pipelines were not honoured by the scheduler at this register pressure - it fell back to read / wait / MFMA triplets), then the
//     S / dP MFMAs of j+1 with their row reads.  A wave's stream is MFMA-dense from the first to the last tile; the exp2 / mul / cvt work sits in the
//     issue slots one 32-cycle MFMA leaves (probe/overlap_probe2.hip).  The packed P / dS of one sub-tile are kept across (16 registers); S / dP reuse
//     one register set because softmax j is complete before A(j+1) issues.
//   * three-stage LDS ring {Q tile, L rows, dO tile, D rows} (78 KiB per workgroup, two workgroups per CU): C(j-1) of a tile's last sub-tile runs in the next
//     iteration, so a tile's stage is released one barrier later; the DMA of tile t+2 is issued behind that barrier.  One barrier per tile as before.
//   MODE 0 keeps the old order (A, softmax, C per sub-tile) on the same ring and stats rows: the A/B partner that isolates the pipeline.
constexpr int STAT_B = BKV * 16;                   // 64 rows x 16 B
constexpr int STAGE_B = 2 * (TILE_B + STAT_B);     // [Q tile][L rows][dO tile][D rows]
constexpr int NSTAGE = 3;
template <typename F, int... K> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, K...>) { (f(IntC<K>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// LDS reads the compiler does not see (no s_waitcnt of its own: every use is preceded by lds_wait on the destination)
template <int OFF> __device__ __forceinline__ void lds_row_asm(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_tr_asm(bf16x8& d, unsigned a0, unsigned a1) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFF) : "memory");
  d = concat_tr(lo, hi);
}
template <int N> __device__ __forceinline__ void lds_wait(bf16x8& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
#ifndef DKV2_WAVES
#define DKV2_WAVES 2        // A/B: 1 = one workgroup per CU (one wave per SIMD)
#endif
#ifndef DKV2_PRIO_X
#define DKV2_PRIO_X 0       // A/B: s_setprio of the MFMA-dense row-operand regions (0 = never raised)
#endif
#ifndef DKV2_MFMA_FIRST
#define DKV2_MFMA_FIRST 0   // A/B: fence between a slot's MFMA and its fillers (measured 1.4 % slower: profiles/r03f_dkv2_variants.txt)
#endif
template <int MODE>
__global__ __launch_bounds__(256, DKV2_WAVES) void attn_bwd_dkv2_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  long kbase, vbase, dkbase, dvbase; int kvlen;
  kv_range(p, b, kbase, vbase, dkbase, dvbase, kvlen);
  if (bx * 128 >= kvlen) return;  // whole block beyond this sample's keys (uniform across the block)
  const int kv = bx * 128 + wave * 32 + (lane & 31);
  const bool kvvalid = kv < kvlen;
  const bool wave_active = bx * 128 + wave * 32 < kvlen;   // see attn_bwd_dkv_kernel

  bf16x8 kf[KSTEPS], vf[KSTEPS];
  load_row_frags(kf, p.K + kbase + (long)kv * p.k_ts + (long)h * p.k_hs, kvvalid, hi);
  load_row_frags(vf, p.V + vbase + (long)kv * p.v_ts + (long)h * p.v_hs, kvvalid, hi);
  settle(kf);
  settle(vf);
  if (hi == 1) {                                            // k-slots 72 .. 74 (k-step 4, upper half): -1.0 against the stats rows' {hi, mid, lo}
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, kf[KSTEPS - 1]);
    w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
    kf[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
    w = __builtin_bit_cast(u32x4, vf[KSTEPS - 1]);
    w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
    vf[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
  }
  const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs;
  const bf16_t* Dp = p.dO + (long)b * p.o_bs + (long)h * p.o_hs;
  const bf16_t* Ls = p.stats + ((long)b * p.H + h) * p.Nq64 * 8;
  const bf16_t* Ds = Ls + (long)p.B * p.H * p.Nq64 * 8;
  const int qts = (int)p.q_ts, ots = (int)p.o_ts;
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  FragAddr fa;
  frag_addr(fa, lane);
  // k-step 4 of the row operand: lanes of the lower half read chunk 8 of the tile (d = 64..71), lanes of the upper half the stats row of their query
  // (16 B at tile base + TILE_B + 16 row): one ds_read_b128 whose 16-lane groups each lie entirely in one of the two regions.
  int r4[2];
#pragma unroll
  for (int sub = 0; sub < 2; sub++) r4[sub] = hi ? TILE_B + (sub * 32 + (lane & 31)) * 16 : fa.rb[0] + 2 * 64 + sub * 32 * ROWB;

  for (int st = 0; st < NSTAGE; st++) {
    init_pads(smem + st * STAGE_B, 0, tid);
    init_pads(smem + st * STAGE_B + TILE_B + STAT_B, 0, tid);
  }
  f32x16 dk[3], dv[3];
  zero3(dk);
  zero3(dv);
  const float c = p.scale_log2;
  const int T = (p.Nq + BKV - 1) / BKV, Tfull = p.Nq / BKV;
  // DMA of tile t -> stage st: 6 tile pieces per wave + the two stats rows blocks (waves 0 / 1).  Full tiles (every tile but a ragged last one) run the
  // lean form: per-lane source offsets fixed at kernel start, the Q and dO pieces of one mask under ONE exec region (3 regions per tile, not 6), no clamp.
  unsigned offQ[NDMA], offD[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; i++) { offQ[i] = (unsigned)(pl.row[i] * qts + pl.coff[i]); offD[i] = (unsigned)(pl.row[i] * ots + pl.coff[i]); }
  auto issue = [&](int t, char* st) {
    if (t < Tfull) {
      const bf16_t* qb = Qp + (long)t * BKV * qts;
      const bf16_t* db = Dp + (long)t * BKV * ots;
#pragma unroll
      for (int i = 0; i < NDMA; i++)
        if (pl.coff[i] >= 0) {
          lds_dma16(qb + offQ[i], st + (i * 4 + wave) * 1024);
          lds_dma16(db + offD[i], st + TILE_B + STAT_B + (i * 4 + wave) * 1024);
        }
    } else {
      dma_tile<false>(st, Qp, qts, t * BKV, p.Nq, pl, wave);
      dma_tile<false>(st + TILE_B + STAT_B, Dp, ots, t * BKV, p.Nq, pl, wave);
    }
    if (wave < 2) {
      const bf16_t* src = (wave == 0 ? Ls : Ds) + ((long)t * BKV + lane) * 8;
      char* dst = st + (wave == 0 ? TILE_B : 2 * TILE_B + STAT_B);
      lds_dma16(src, dst);
    }
  };
  auto rowA = [&](const char* tile, int sub, int ks) -> bf16x8 {
    return ks < KSTEPS - 1 ? rowfrag(tile, fa, sub, ks) : *reinterpret_cast<const bf16x8*>(tile + r4[sub]);
  };
  // A(j): S' = Q K^T - lse / c and dP' = dO V^T - delta of one 32-query sub-tile
  auto phaseA = [&](const char* sQ, const char* sD, int sub, f32x16& s, f32x16& dp) {
#pragma unroll
    for (int g = 0; g < 16; g++) { s[g] = 0.f; dp[g] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
      s = mfma32(rowA(sQ, sub, ks), kf[ks], s);
      dp = mfma32(rowA(sD, sub, ks), vf[ks], dp);
    }
  };
  // B(j): P = exp2(c S'), dS = P dP' (softmax scale applied at the store), packed to the B operands of the second products
  auto phaseB = [&](f32x16& s, f32x16& dp, bf16x8 (&pb)[2], bf16x8 (&db)[2]) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const float pr = __builtin_amdgcn_exp2f(s[g] * c);
      s[g] = pr;
      dp[g] = pr * dp[g];
    }
#pragma unroll
    for (int uu = 0; uu < 2; uu++) { pb[uu] = pack8(s, 8 * uu); db[uu] = pack8(dp, 8 * uu); }
  };
  // C(j): dV^T += dO^T P, dK^T += Q^T dS
  auto phaseC = [&](const char* sQ, const char* sD, int sub, const bf16x8 (&pb)[2], const bf16x8 (&db)[2]) {
#pragma unroll
    for (int uu = 0; uu < 2; uu++)
#pragma unroll
      for (int dt = 0; dt < 3; dt++) {
        dv[dt] = mfma32(trfrag(sD, fa, dt, sub * 2 + uu), pb[uu], dv[dt]);
        dk[dt] = mfma32(trfrag(sQ, fa, dt, sub * 2 + uu), db[uu], dk[dt]);
      }
  };
  auto stage = [&](int i) -> char* { return smem + i * STAGE_B; };

  // Ring protocol (both modes): iteration t = barrier (tile t landed; every wave has left the stage of tile t - 2), DMA of tile t + 1 into that
  // stage, compute.  MODE 1 still reads tile t - 1 during iteration t, hence three stages.
  // The wait for this wave's own DMA pieces is written out: the compiler only waits for an LDS-DMA in front of LDS reads IT emits (and at workgroup
  // scope a fence needs no vmcnt), so where the loop's reads are inline asm - or a wave reads nothing at all - __syncthreads() alone compiles to a
  // bare s_barrier and tiles were read before they had landed (round 3, first GPU run: run-to-run different dK in 18 of 256 heads at B16).
  auto ring_sync = [&]() { tile_sync(); };
  issue(0, stage(0));
  if (!wave_active) {                        // serves DMA and barriers only (same barrier count as the compute path)
    int in = 1;
    for (int t = 0; t < T; t++) {
      ring_sync();
      if (t + 1 < T) issue(t + 1, stage(in));
      in = in == NSTAGE - 1 ? 0 : in + 1;
    }
    return;
  }
  if (MODE == 0) {
    int ic = 0;
    for (int t = 0; t < T; t++) {
      const int in = ic == NSTAGE - 1 ? 0 : ic + 1;
      ring_sync();
      if (t + 1 < T) issue(t + 1, stage(in));
      const char* sQ = stage(ic);
      const char* sD = sQ + TILE_B + STAT_B;
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        f32x16 s, dp;
        bf16x8 pb[2], db[2];
        phaseA(sQ, sD, sub, s, dp);
        phaseB(s, dp, pb, db);
        phaseC(sQ, sD, sub, pb, db);
      }
      ic = in;
    }
  } else {
    // Hand-placed stream.  Slot = {counted wait; one MFMA; the LDS reads of the MFMA two slots ahead; a slice of the softmax}, closed by
    // sched_barrier(0) so the compiler keeps the order written here.  Fragments rotate through f[0..2].
    //   region X (10 slots): A of a sub-tile (row fragment k = 2 ks + w; w = 0: Q -> S, w = 1: dO -> dP)
    //   region Y (12 slots): C of the previous sub-tile (tr fragment k = uu 6 + dt 2 + w; w = 0: dO^T -> dV, w = 1: Q^T -> dK) || B of the current one
    // B in place, per slot k: exp2 of elements 2k, 2k+1 (k < 8), dP multiplies one slot behind (1 <= k <= 8), the four cvt_pk groups in slots 8..11:
    // 4 6 6 6 6 6 6 6 6 4 4 4 VALU.
    // The LDS reads are inline asm with hand-counted lgkmcnt waits: (1) the compiler puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 that
    // follows an LDS-DMA still in flight (the intrinsic's memory operand carries no alias scope) - in this loop that would expose the whole DMA
    // latency once per tile; (2) it re-used one register quad for every fragment (read, wait, MFMA).  lgkmcnt counts in issue order, so a wait in
    // front of MFMA k names the reads issued after fragment k's own: exactly fragment k + 1 (fragment k + 2 follows the MFMA).  tr fragment = 2 reads,
    // row fragment = 1.  Iteration = barrier; DMA of tile t + 1; X: A(t, 0) [cold start]; Y: C(t-1, 1) || B(t, 0); X: A(t, 1); Y: C(t, 0) || B(t, 1):
    // nothing is in flight at the loop's back edge.  tools/check_lds_waits.py replays the emitted ISA against these rules.
    const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
    f32x16 s, dp;
    bf16x8 pb[2], db[2], pb1[2], db1[2], f[3];
#pragma unroll
    for (int uu = 0; uu < 2; uu++)
#pragma unroll
      for (int e = 0; e < 8; e++) { pb1[uu][e] = (bf16_t)0.f; db1[uu][e] = (bf16_t)0.f; }
    struct Bases { unsigned r0, r1, r40, r41, t0, t1; };   // per-lane LDS byte addresses inside one stage (the dO tile is an immediate offset away)
    auto bases = [&](unsigned st) -> Bases { return Bases{st + (unsigned)fa.rb[0], st + (unsigned)fa.rb[1], st + (unsigned)r4[0], st + (unsigned)r4[1],
                                                          st + (unsigned)fa.tb[0], st + (unsigned)fa.tb[1]}; };
    constexpr int DOFF = TILE_B + STAT_B;                  // dO tile relative to the Q tile of its stage
    auto rd_row = [&](auto subc, auto kc, bf16x8& d, const Bases& bs) {
      constexpr int sub = decltype(subc)::value, k = decltype(kc)::value, ks = k >> 1, w = k & 1;
      if constexpr (ks < KSTEPS - 1) lds_row_asm<w * DOFF + sub * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? bs.r1 : bs.r0);
      else lds_row_asm<w * DOFF>(d, sub ? bs.r41 : bs.r40);
    };
    auto rd_tr = [&](auto subc, auto kc, bf16x8& d, unsigned t0, unsigned t1) {
      constexpr int sub = decltype(subc)::value, k = decltype(kc)::value, uu = k / 6, dt = (k % 6) >> 1, w = k & 1;
      lds_tr_asm<(w ? 0 : DOFF) + (sub * 2 + uu) * 16 * ROWB + dt * 64>(d, t0, t1);
    };
    // (the empty asm statements pin each slice to its slot: SelectionDAG otherwise emits pure arithmetic next to its first use, i.e. behind the
    // region's last sched_barrier - seen with the 16 cvt_pk)
    auto softmax_slot = [&](auto kc, bf16x8 (&npb)[2], bf16x8 (&ndb)[2]) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < 8) {
        s[2 * k] = __builtin_amdgcn_exp2f(s[2 * k] * c);
        s[2 * k + 1] = __builtin_amdgcn_exp2f(s[2 * k + 1] * c);
        asm volatile("" : "+v"(s[2 * k]), "+v"(s[2 * k + 1]));
      }
      if constexpr (k >= 1 && k <= 8) {
        dp[2 * k - 2] *= s[2 * k - 2];
        dp[2 * k - 1] *= s[2 * k - 1];
        asm volatile("" : "+v"(dp[2 * k - 2]), "+v"(dp[2 * k - 1]));
      }
      if constexpr (k == 8) { npb[0] = pack8(s, 0); asm volatile("" : "+v"(npb[0])); }
      if constexpr (k == 9) { ndb[0] = pack8(dp, 0); asm volatile("" : "+v"(ndb[0])); }
      if constexpr (k == 10) { npb[1] = pack8(s, 8); asm volatile("" : "+v"(npb[1])); }
      if constexpr (k == 11) { ndb[1] = pack8(dp, 8); asm volatile("" : "+v"(ndb[1])); }
    };
    // BASE: fragment k of the region sits in f[(k + BASE) % 3]; fragments 0 and 1 are in flight on entry.
    // region X: A(sub ASUB of the stage `ab`); PRE: its last two slots load tr fragments 0, 1 of C(sub NSUB) from the tr bases nt0 / nt1
    auto regionX = [&](auto basec, auto asubc, auto prec, auto nsubc, const Bases& ab, unsigned nt0, unsigned nt1) {
      constexpr int BASE = decltype(basec)::value;
      constexpr bool PRE = decltype(prec)::value;
      if (DKV2_PRIO_X) __builtin_amdgcn_s_setprio(DKV2_PRIO_X);
      static_for<10>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        lds_wait<(k < 9) ? 1 : (PRE ? 2 : 0)>(f[(k + BASE) % 3]);
        if constexpr (k == 0) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; s = mfma32(f[(k + BASE) % 3], kf[0], z); }
        else if constexpr (k == 1) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; dp = mfma32(f[(k + BASE) % 3], vf[0], z); }
        else if constexpr (k & 1) dp = mfma32(f[(k + BASE) % 3], vf[k >> 1], dp);
        else s = mfma32(f[(k + BASE) % 3], kf[k >> 1], s);
        if (DKV2_MFMA_FIRST) __builtin_amdgcn_sched_barrier(0);       // the MFMA leads its slot: the fillers issue in its shadow
        if constexpr (k + 2 < 10) rd_row(asubc, IntC<k + 2>{}, f[(k + 2 + BASE) % 3], ab);
        else if constexpr (PRE) rd_tr(nsubc, IntC<k + 2 - 10>{}, f[(k + 2 + BASE) % 3], nt0, nt1);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // region Y: C(sub CSUB, tr bases ct0 / ct1, operands cpb / cdb) || B(s, dp -> npb, ndb); PRE: the last two slots load row fragments 0, 1 of
    // A(sub ASUB of the stage `ab`)
    auto regionY = [&](auto basec, auto csubc, auto prec, auto asubc, unsigned ct0, unsigned ct1, const bf16x8 (&cpb)[2], const bf16x8 (&cdb)[2],
                       bf16x8 (&npb)[2], bf16x8 (&ndb)[2], const Bases& ab) {
      constexpr int BASE = decltype(basec)::value;
      constexpr bool PRE = decltype(prec)::value;
      if (DKV2_PRIO_X) __builtin_amdgcn_s_setprio(0);
      static_for<12>([&](auto kc) {
        constexpr int k = decltype(kc)::value, uu = k / 6, dt = (k % 6) >> 1;
        lds_wait<(k < 11) ? 2 : (PRE ? 1 : 0)>(f[(k + BASE) % 3]);
        if constexpr (k & 1) dk[dt] = mfma32(f[(k + BASE) % 3], cdb[uu], dk[dt]);
        else dv[dt] = mfma32(f[(k + BASE) % 3], cpb[uu], dv[dt]);
        if (DKV2_MFMA_FIRST) __builtin_amdgcn_sched_barrier(0);
        if constexpr (k + 2 < 12) rd_tr(csubc, IntC<k + 2>{}, f[(k + 2 + BASE) % 3], ct0, ct1);
        else if constexpr (PRE) rd_row(asubc, IntC<k + 2 - 12>{}, f[(k + 2 + BASE) % 3], ab);
        softmax_slot(kc, npb, ndb);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    int ic = 0;                               // stage of tile t
    unsigned cur = lds0, prv = lds0;          // LDS addresses of the stages of tile t and t - 1 (tile 0 itself at t = 0, where pb1 = db1 = 0)
    for (int t = 0; t < T; t++) {
      const int in = ic == NSTAGE - 1 ? 0 : ic + 1;
      ring_sync();                        // tile t has landed (vmcnt(0)) for every wave; every wave has left the stage of tile t - 2
      if (t + 1 < T) issue(t + 1, stage(in));
      const Bases cb = bases(cur);
      const unsigned pt0 = prv + (unsigned)fa.tb[0], pt1 = prv + (unsigned)fa.tb[1];
      __builtin_amdgcn_sched_barrier(0);
      rd_row(IntC<0>{}, IntC<0>{}, f[0], cb);                                     // cold start: row fragments 0, 1 of A(t, 0)
      rd_row(IntC<0>{}, IntC<1>{}, f[1], cb);
      __builtin_amdgcn_sched_barrier(0);
      regionX(IntC<0>{}, IntC<0>{}, BoolC<true>{}, IntC<1>{}, cb, pt0, pt1);                          // A(t, 0)               -> tr of C(t-1, 1)
      regionY(IntC<1>{}, IntC<1>{}, BoolC<true>{}, IntC<1>{}, pt0, pt1, pb1, db1, pb, db, cb);        // C(t-1, 1) || B(t, 0)  -> rows of A(t, 1)
      regionX(IntC<1>{}, IntC<1>{}, BoolC<true>{}, IntC<0>{}, cb, cb.t0, cb.t1);                      // A(t, 1)               -> tr of C(t, 0)
      regionY(IntC<2>{}, IntC<0>{}, BoolC<false>{}, IntC<0>{}, cb.t0, cb.t1, pb, db, pb1, db1, cb);   // C(t, 0) || B(t, 1)
      prv = cur;
      cur = lds0 + in * STAGE_B;
      ic = in;
    }
    phaseC(smem + (prv - lds0), smem + (prv - lds0) + TILE_B + STAT_B, 1, pb1, db1);
  }
  if (kvvalid) {
    store_rows(p.dK + dkbase + (long)kv * p.dk_ts + (long)h * p.dk_hs, dk, p.dk_scale, hi);
    store_rows(p.dV + dvbase + (long)kv * p.dv_ts + (long)h * p.dv_hs, dv, 1.f, hi);
  }
  if (p.dk_colsum) colsum_rows(p.dk_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dk, p.dk_scale, kvvalid, hi, lane);
  if (p.dv_colsum) colsum_rows(p.dv_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dv, 1.f, kvvalid, hi, lane);
}


// ------------------------------------------------------------------------------------------------ backward: dQ (round 3)
// attn_bwd_dq_kernel's products, layouts and fragments with the dK/dV kernel's treatment of the instruction stream: a software pipeline over 32-key
// sub-tiles j, placed by hand in slots {counted lgkmcnt wait; one MFMA; the LDS reads of the MFMA two slots ahead; a slice of the softmax}.
//   region R (10 slots of v_mfma_f32_32x32x16): A(j+1) - S and dP of the NEXT sub-tile into the other S / dP register set (fragment k = 2 ks + w;
//            w = 0: K rows -> S, w = 1: V rows -> dP - delta) - with B(j), the softmax of the current one, in their shadow: exp2 + the dP multiply of
//            elements 2k, 2k+1 in slots 0..7 (6 VALU), the eight cvt_pk in slot 8, the four v_permlane16_swap in slot 9
//   region C (10 slots of v_mfma_f32_16x16x32): dQ^T += K^T dS^T of sub-tile j (5 transpose-read fragments, each feeding the x and the y MFMA)
// Iteration (tile t of 64 keys) = barrier; DMA of tile t+2; R: A(t, 1) || B(t, 0); C(t, 0); R: A(t+1, 0) || B(t, 1); C(t, 1): tile t+1 is read while
// tile t is still in use, hence a three-stage ring {K, V} (72 KiB, two workgroups per CU), and nothing is in flight at the back edge.  Full tiles
// only; a ragged last tile (text keys) runs the compiler-scheduled masked path after the loop.  delta rides in the dP product (ATTN_FOLD_DELTA).
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * 2 * TILE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const int q = bx * 128 + wave * 32 + (lane & 31);
  const bool qvalid = q < p.Nq;
  long kbase, vbase, d0_, d1_; int kvlen;
  kv_range(p, b, kbase, vbase, d0_, d1_, kvlen);
  const bf16_t* Kp = p.K + kbase + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + vbase + (long)h * p.v_hs;
  const int kts = (int)p.k_ts, vts = (int)p.v_ts;

  bf16x8 qf[KSTEPS], dof[KSTEPS];
  load_row_frags(qf, p.Q + (long)b * p.q_bs + (long)q * p.q_ts + (long)h * p.q_hs, qvalid, hi);
  load_row_frags(dof, p.dO + (long)b * p.o_bs + (long)q * p.o_ts + (long)h * p.o_hs, qvalid, hi);
  settle(qf);
  settle(dof);
  const long sidx = ((long)b * p.H + h) * p.Nq + q;
  const float lse = qvalid ? p.LSE[sidx] : 0.f;
  const float delta = qvalid ? p.Delta[sidx] : 0.f;
  if (hi == 1) {                                            // slots 72 .. 74 of this lane's dO row: delta as three operand-type terms (split3)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, dof[KSTEPS - 1]);
    const uint2 d3 = split3(delta);
    w[0] = d3.x; w[1] = d3.y;
    dof[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
  }
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  FragAddr fa;
  frag_addr(fa, lane);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  for (int st = 0; st < 2 * NSTAGE; st++) init_pads(smem + st * TILE_B, (st & 1) ? 2 : 0, tid);   // odd tiles = V: -1.0 in slots 72 .. 74
  Acc16 dq;
  zero16(dq);
  const float c = p.scale_log2;
  const int Tfull = kvlen / BKV, rem = kvlen - Tfull * BKV, T = Tfull + (rem ? 1 : 0);
  auto stage = [&](int i) -> char* { return smem + i * 2 * TILE_B; };
  auto issue = [&](int t, char* st) {
    if (t < Tfull) {
      dma_tile<true>(st, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<true>(st + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    } else {
      dma_tile<false>(st, Kp, kts, t * BKV, kvlen, pl, wave);
      dma_tile<false>(st + TILE_B, Vp, vts, t * BKV, kvlen, pl, wave);
    }
  };
  // the compiler-scheduled tile of attn_bwd_dq_kernel: prologue sub-tile, pipeline drain and the ragged last tile
  auto firstA = [&](const char* sK, const char* sV, int sub, f32x16& sv, f32x16& dpv) {
#pragma unroll
    for (int g = 0; g < 16; g++) { sv[g] = 0.f; dpv[g] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
      sv = mfma32(rowfrag(sK, fa, sub, ks), qf[ks], sv);
      dpv = mfma32(rowfrag(sV, fa, sub, ks), dof[ks], dpv);
    }
  };
  auto tail_tile = [&](const char* sK, const char* sV, int kv0) {
    f32x16 sv[2], dpv[2];
#pragma unroll
    for (int sub = 0; sub < 2; sub++) firstA(sK, sV, sub, sv[sub], dpv[sub]);
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int g = 0; g < 16; g++) {
        float pr = __builtin_amdgcn_exp2f(sv[sub][g] * c - lse);
        if (kv0 + sub * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi >= kvlen) pr = 0.f;
        sv[sub][g] = pr * dpv[sub][g];
      }
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      bf16x8 dx, dy;
      pack_xy(sv[sub], dx, dy);
      mma16(dq, sK, ta, sub, dx, dy);
    }
  };

  if (T > 0) issue(0, stage(0));
  if (T > 1) issue(1, stage(1));
  if (Tfull > 0) {
    const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
    f32x16 S[2], DP[2];
    bf16x8 dx, dy, f[3];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ua, ub;                                            // pack_xy split over two slots: the cvt_pk half, then the lane exchange
    struct RowB { unsigned r0, r1; };
    struct TrB { unsigned t00, t01, t10, t11; };             // [read e][tile parity]
    auto rowb = [&](unsigned st) -> RowB { return RowB{st + (unsigned)fa.rb[0], st + (unsigned)fa.rb[1]}; };
    auto trb = [&](unsigned st) -> TrB { return TrB{st + (unsigned)ta.tb[0][0], st + (unsigned)ta.tb[0][1], st + (unsigned)ta.tb[1][0], st + (unsigned)ta.tb[1][1]}; };
    auto rd_row = [&](auto subc, auto kc, bf16x8& d, const RowB& rb) {       // row fragment k = 2 ks + w of sub-tile SUB: w = 0 K tile, w = 1 V tile
      constexpr int sub = decltype(subc)::value, k = decltype(kc)::value, ks = k >> 1, w = k & 1;
      lds_row_asm<w * TILE_B + sub * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? rb.r1 : rb.r0);
    };
    auto rd_tr = [&](auto subc, auto tc, bf16x8& d, const TrB& tb) {         // K^T fragment of output tile t (16 head dims) of sub-tile SUB
      constexpr int sub = decltype(subc)::value, t = decltype(tc)::value;
      lds_tr_asm<sub * 32 * ROWB + (t >> 1) * 64>(d, (t & 1) ? tb.t01 : tb.t00, (t & 1) ? tb.t11 : tb.t10);
    };
    auto softmax_slot = [&](auto kc, f32x16& sv, f32x16& dpv) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < 8) {
        sv[2 * k] = __builtin_amdgcn_exp2f(sv[2 * k] * c - lse) * dpv[2 * k];         // dS^T (without the softmax scale, applied at the store)
        sv[2 * k + 1] = __builtin_amdgcn_exp2f(sv[2 * k + 1] * c - lse) * dpv[2 * k + 1];
        asm volatile("" : "+v"(sv[2 * k]), "+v"(sv[2 * k + 1]));
      }
      if constexpr (k == 8) {                                 // pack_xy, first half: a = rows {0-3, 16-19} + 4 hi, b = rows {8-11, 24-27} + 4 hi
        bf16x8 a, bb;
#pragma unroll
        for (int j = 0; j < 4; j++) { a[j] = (bf16_t)sv[j]; a[4 + j] = (bf16_t)sv[8 + j]; bb[j] = (bf16_t)sv[4 + j]; bb[4 + j] = (bf16_t)sv[12 + j]; }
        ua = __builtin_bit_cast(u32x4, a);
        ub = __builtin_bit_cast(u32x4, bb);
        asm volatile("" : "+v"(ua), "+v"(ub));
      }
      if constexpr (k == 9) {
        u32x4 ux, uy;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const auto r = __builtin_amdgcn_permlane16_swap(ua[w], ub[w], false, false);
          ux[w] = r[0]; uy[w] = r[1];
        }
        dx = __builtin_bit_cast(bf16x8, ux);
        dy = __builtin_bit_cast(bf16x8, uy);
        asm volatile("" : "+v"(dx), "+v"(dy));
      }
    };
    // R: A(sub ASUB of the stage behind `ab`) -> S[NB], DP[NB] || B on S[CB], DP[CB]; fragments 0, 1 in flight on entry (f[0], f[1]); its last two
    // slots load transpose fragments 0, 1 of C(sub CSUB, bases ct)
    auto regionR = [&](auto asubc, auto nbc, auto csubc, const RowB& ab, const TrB& ct) {
      constexpr int NB = decltype(nbc)::value, CB = 1 - NB;
      static_for<10>([&](auto kc) {
        constexpr int k = decltype(kc)::value, ks = k >> 1;
        lds_wait<(k < 9) ? 1 : 2>(f[k % 3]);
        if constexpr (k & 1) {
          if constexpr (ks == 0) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; DP[NB] = mfma32(f[k % 3], dof[0], z); }
          else DP[NB] = mfma32(f[k % 3], dof[ks], DP[NB]);
        } else {
          if constexpr (ks == 0) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; S[NB] = mfma32(f[k % 3], qf[0], z); }
          else S[NB] = mfma32(f[k % 3], qf[ks], S[NB]);
        }
        if constexpr (k + 2 < 10) rd_row(asubc, IntC<k + 2>{}, f[(k + 2) % 3], ab);
        else rd_tr(csubc, IntC<k + 2 - 10>{}, f[(k + 2) % 3], ct);
        softmax_slot(kc, S[CB], DP[CB]);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // C: dQ^T += K^T dS^T of sub-tile CSUB; transpose fragment i sits in f[(i + 1) % 3] (10 row fragments went before); PRE: its last two fragment
    // slots load row fragments 0, 1 of the next A (sub ASUB of the stage behind `ab`)
    auto regionC = [&](auto csubc, auto prec, auto asubc, const TrB& ct, const RowB& ab) {
      constexpr bool PRE = decltype(prec)::value;
      static_for<10>([&](auto kc) {
        constexpr int k = decltype(kc)::value, fi = k >> 1;
        if constexpr ((k & 1) == 0) {
          lds_wait<(fi < 4) ? 2 : (PRE ? 1 : 0)>(f[(fi + 1) % 3]);
          dq.v[fi][0] = mfma16(f[(fi + 1) % 3], dx, dq.v[fi][0]);
          if constexpr (fi + 2 < NT16) rd_tr(csubc, IntC<fi + 2>{}, f[(fi + 3) % 3], ct);
          else if constexpr (PRE) rd_row(asubc, IntC<fi + 2 - NT16>{}, f[(fi + 3) % 3], ab);
        } else {
          dq.v[fi][1] = mfma16(f[(fi + 1) % 3], dy, dq.v[fi][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    tile_sync();                                             // tiles 0 (and 1) have landed, pads written
    firstA(stage(0), stage(0) + TILE_B, 0, S[0], DP[0]);     // A(0, 0), cold
    int ic = 0;
    for (int t = 0; t < Tfull; t++) {
      const int in = ic == NSTAGE - 1 ? 0 : ic + 1;
      if (t > 0) tile_sync();                                // tile t+1 has landed; every wave has left the stage of tile t-1
      if (t + 2 < T) issue(t + 2, stage(in == NSTAGE - 1 ? 0 : in + 1));
      const unsigned cur = lds0 + ic * 2 * TILE_B, nxt = lds0 + ((t + 1 < Tfull) ? in : ic) * 2 * TILE_B;   // last full tile: A(t+1, 0) re-reads tile t (unused)
      const RowB crb = rowb(cur), nrb = rowb(nxt);
      const TrB ctb = trb(cur);
      __builtin_amdgcn_sched_barrier(0);
      rd_row(IntC<1>{}, IntC<0>{}, f[0], crb);               // cold start: row fragments 0, 1 of A(t, 1)
      rd_row(IntC<1>{}, IntC<1>{}, f[1], crb);
      __builtin_amdgcn_sched_barrier(0);
      regionR(IntC<1>{}, IntC<1>{}, IntC<0>{}, crb, ctb);                   // A(t, 1) -> S[1] || B(t, 0) on S[0]      -> tr of C(t, 0)
      regionC(IntC<0>{}, BoolC<true>{}, IntC<0>{}, ctb, nrb);                // C(t, 0)                                 -> rows of A(t+1, 0)
      regionR(IntC<0>{}, IntC<0>{}, IntC<1>{}, nrb, ctb);                   // A(t+1, 0) -> S[0] || B(t, 1) on S[1]    -> tr of C(t, 1)
      regionC(IntC<1>{}, BoolC<false>{}, IntC<0>{}, ctb, nrb);               // C(t, 1)
      ic = in;
    }
    if (rem) tail_tile(stage(ic), stage(ic) + TILE_B, Tfull * BKV);         // landed and visible since the last iteration's barrier
  } else if (rem) {
    tile_sync();
    tail_tile(stage(0), stage(0) + TILE_B, 0);
  }
  const int q0w = bx * 128 + wave * 32;
  const bool ok0 = q0w + (lane & 15) < p.Nq, ok1 = q0w + 16 + (lane & 15) < p.Nq;
  store_rows16(p.dQ + (long)b * p.dq_bs + (long)q0w * p.dq_ts + (long)h * p.dq_hs, p.dq_ts, dq, p.scale, p.scale, ok0, ok1, lane);
  if (p.dq_colsum) colsum_rows16(p.dq_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dq, p.scale, ok0, ok1, lane);
}


// ------------------------------------------------------------------------------------------------ backward: dQ, all keys resident (cross-attention)
// attn_fwd_kvres_kernel's organisation for the dQ product of a cross-attention backward (max_kv_len <= 320): one 512-thread workgroup keeps the K and V
// rows of its (sample, head) in LDS and walks its queries, 32 per wave and trip - and because a wave holds the whole dO row of its query anyway, the
// delta pre-pass (a separate kernel reading O and dO: 52 us per call) folds in: the wave loads the O row beside it, forms delta = rowsum(dO o O), writes
// it (and, for the dK/dV kernel that runs afterwards, the lse / delta statistics rows) and carries it into the dP product through dO's pad slots
// (ATTN_FOLD_DELTA).  The tile body is attn_bwd_dq2_kernel's compiler-scheduled tail form on the resident tiles: no DMA, no barrier in the loop.
__global__ __launch_bounds__(512, 1) void attn_bwd_dq_kvres_kernel(AttnParams p, int qpb, int tiles_alloc, float* delta_out, bf16_t* stats_out, float inv_c) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  char* smem = smem_dyn;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  long kbase, vbase, d0_, d1_; int kvlen;
  kv_range(p, b, kbase, vbase, d0_, d1_, kvlen);
  const bf16_t* Kp = p.K + kbase + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + vbase + (long)h * p.v_hs;
  kvlen = min(kvlen, tiles_alloc * BKV);
  const int Tfull = kvlen / BKV, rem = kvlen - Tfull * BKV, T = Tfull + (rem ? 1 : 0);
  {
    DmaPlan pl;
    dma_plan(pl, wave & 3, lane);
    const bool vside = wave >= 4;
    for (int t = 0; t < T; t++) {
      char* dst = smem + t * 2 * TILE_B + (vside ? TILE_B : 0);
      init_pads(dst, vside ? 2 : 0, tid & 255);               // V: -1.0 in slots 72 .. 74 (delta rides in the dP product)
      if (t < Tfull) dma_tile<true>(dst, vside ? Vp : Kp, vside ? (int)p.v_ts : (int)p.k_ts, t * BKV, kvlen, pl, wave & 3);
      else dma_tile<false>(dst, vside ? Vp : Kp, vside ? (int)p.v_ts : (int)p.k_ts, t * BKV, kvlen, pl, wave & 3);
    }
  }
  FragAddr fa;
  frag_addr(fa, lane);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  const float c = p.scale_log2;
  const long rows_total = (long)p.B * p.H * p.Nq64;
  tile_sync();
  // the rows of the NEXT trip are requested before this trip's tiles are worked on (two waves per SIMD do not hide a 2-3 us row load by themselves)
  const int q_end = min(p.Nq, (bx + 1) * qpb);
  bf16x8 nqf[KSTEPS], ndof[KSTEPS], nof[KSTEPS];
  float nlse = 0.f;
  auto load_trip = [&](int q0w) {
    const int qq = q0w + (lane & 31);
    const bool ok = q0w < q_end && qq < p.Nq;
    load_row_frags(nqf, p.Q + (long)b * p.q_bs + (long)qq * p.q_ts + (long)h * p.q_hs, ok, hi);
    load_row_frags(ndof, p.dO + (long)b * p.o_bs + (long)qq * p.o_ts + (long)h * p.o_hs, ok, hi);
    load_row_frags(nof, p.O + (long)b * p.o_bs + (long)qq * p.o_ts + (long)h * p.o_hs, ok, hi);
    nlse = ok ? p.LSE[((long)b * p.H + h) * p.Nq + qq] : 0.f;
  };
  load_trip(bx * qpb + wave * 32);
  for (int q0b = bx * qpb; q0b < q_end; q0b += 256) {
    const int q0w = q0b + wave * 32;
    if (q0w >= p.Nq) break;                                    // wave-uniform
    const int q = q0w + (lane & 31);
    const bool qvalid = q < p.Nq;
    bf16x8 qf[KSTEPS], dof[KSTEPS], of[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) { qf[ks] = nqf[ks]; dof[ks] = ndof[ks]; of[ks] = nof[ks]; }
    const float lse = nlse;
    load_trip(q0w + 256);
    const long sidx = ((long)b * p.H + h) * p.Nq + q;
    float part = 0.f;                                          // this lane's 40 of the row's 80 slots (slots 72 .. 79 are zero in both rows)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++) part += (float)dof[ks][j] * (float)of[ks][j];
    const float delta = part + __shfl_xor(part, 32);
    if (qvalid && hi == 0) {
      delta_out[sidx] = delta;
      write_stat_rows(stats_out, rows_total, ((long)b * p.H + h) * p.Nq64 + q, lse * inv_c, delta);
    }
    if (hi == 1) {                                            // slots 72 .. 74 of this lane's dO row: delta as three operand-type terms (split3)
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 w = __builtin_bit_cast(u32x4, dof[KSTEPS - 1]);
      const uint2 d3 = split3(delta);
      w[0] = d3.x; w[1] = d3.y;
      dof[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
    }
    Acc16 dq;
    zero16(dq);
    auto tile = [&](auto tailc, const char* sK, const char* sV, int kv0) {
      constexpr bool TAIL = decltype(tailc)::value;
      f32x16 sv[2], dpv[2];
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
#pragma unroll
        for (int g = 0; g < 16; g++) { sv[sub][g] = 0.f; dpv[sub][g] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
          sv[sub] = mfma32(rowfrag(sK, fa, sub, ks), qf[ks], sv[sub]);
          dpv[sub] = mfma32(rowfrag(sV, fa, sub, ks), dof[ks], dpv[sub]);
        }
      }
#pragma unroll
      for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
          float pr = __builtin_amdgcn_exp2f(sv[sub][g] * c - lse);
          if (TAIL && kv0 + sub * 32 + (g & 3) + 8 * (g >> 2) + 4 * hi >= kvlen) pr = 0.f;
          sv[sub][g] = pr * dpv[sub][g];
        }
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        bf16x8 dx, dy;
        pack_xy(sv[sub], dx, dy);
        mma16(dq, sK, ta, sub, dx, dy);
      }
    };
    for (int t = 0; t < Tfull; t++) tile(BoolC<false>{}, smem + t * 2 * TILE_B, smem + t * 2 * TILE_B + TILE_B, t * BKV);
    if (rem) tile(BoolC<true>{}, smem + Tfull * 2 * TILE_B, smem + Tfull * 2 * TILE_B + TILE_B, Tfull * BKV);
    const bool ok0 = q0w + (lane & 15) < p.Nq, ok1 = q0w + 16 + (lane & 15) < p.Nq;
    store_rows16(p.dQ + (long)b * p.dq_bs + (long)q0w * p.dq_ts + (long)h * p.dq_hs, p.dq_ts, dq, p.scale, p.scale, ok0, ok1, lane);
    if (p.dq_colsum) colsum_rows16(p.dq_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dq, p.scale, ok0, ok1, lane);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV as a phase ping-pong (round 3, second form)
// What the hand-placed kernel above could not fix: its two waves per SIMD come from different workgroups, are in-order and uncoordinated - each blocks
// on the matrix pipe while the partner's MFMA runs and cannot issue its softmax meanwhile (47 cycles per MFMA against a 32-cycle floor).  Here ONE
// 512-thread workgroup (256 keys) owns the CU: waves w and w + 4 share a SIMD and alternate, in lock-step through s_barrier, between
//     phase M: C(j) + A(j+1) - 22 MFMAs with their LDS reads, no VALU        and        phase V: B(j+1) - the softmax, no MFMA
// with waves 4-7 one phase behind waves 0-3, so on every SIMD one wave is in M while the other is in V: the MFMAs of the two never collide and the
// softmax sits entirely in the partner's matrix phase (the persistent GEMM's two-phase scheme, csrc/gemm.hip).  S / dP are updated in place (B(j) is
// complete before A(j+1) issues), one packed P / dS set.  The Q / dO tiles are shared by 8 waves (half the LDS-DMA traffic per key).
//   per wave and tile t:  V: B(2t) | M: C(2t), A(2t+1) | V: B(2t+1) | M: C(2t+1), A(2t+2)      four barriers; group 1 (waves 4-7) runs one phase later
// MEASURED (profiles/r03o_attn_dkv_modes.txt, r03p_dkv3_depth.txt): parity green on the first run, bit-reproducible on all 256 heads at B = 16 - and
// 2.56 ms against 2.51 ms for the kernel above: no faster.  Nor does the prefetch distance of its matrix phase matter (2 / 3 / 4 / 6 fragments:
// 2.666 / 2.666 / 2.682 / 2.691 ms on one box).  Two uncoordinated waves, a hand-placed pipeline and a lock-step ping-pong all land within 3 % of each
// other: the kernel is not waiting for issue slots or LDS latency, it sits at the chip's POWER limit for this instruction mix (effective clock 1.75 GHz,
// DESIGN.md section 4 fact 2) - what moved the time this round was removing work (the stats rows: -6 %), not re-ordering it.  Kept as PXA_ATTN_DKV=3 for
// the record; mode 2 stays the default.
//   ring: tile t+2 -> stage (t+2) % 3 is issued at global phase 4t+1 (group 0: start of its first M, group 1: start of its first V - every wave has left
//   tile t-1 by then) and waited for (vmcnt(0), each wave its own pieces) in front of the barrier that ends global phase 4t+6, one phase before group 0
//   first reads it.
template <int DK>     // DK = prefetch distance of the matrix phase in fragments (DK + 1 fragment register quads)
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv3_kernel(AttnParams p) {
  constexpr int NF = DK + 1;
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  const int grp = wave >> 2;
  int bx, h, b;
  block_coords(p, bx, h, b);
  long kbase, vbase, dkbase, dvbase; int kvlen;
  kv_range(p, b, kbase, vbase, dkbase, dvbase, kvlen);
  if (bx * 256 >= kvlen) return;  // whole block beyond this sample's keys (uniform across the block)
  const int kv = bx * 256 + wave * 32 + (lane & 31);
  const bool kvvalid = kv < kvlen;
  const bool wave_active = bx * 256 + wave * 32 < kvlen;

  bf16x8 kf[KSTEPS], vf[KSTEPS];
  load_row_frags(kf, p.K + kbase + (long)kv * p.k_ts + (long)h * p.k_hs, kvvalid, hi);
  load_row_frags(vf, p.V + vbase + (long)kv * p.v_ts + (long)h * p.v_hs, kvvalid, hi);
  settle(kf);
  settle(vf);
  if (hi == 1) {                                            // k-slots 72 .. 74: -1.0 against the stats rows' {hi, mid, lo}
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, kf[KSTEPS - 1]);
    w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
    kf[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
    w = __builtin_bit_cast(u32x4, vf[KSTEPS - 1]);
    w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
    vf[KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
  }
  const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs;
  const bf16_t* Dp = p.dO + (long)b * p.o_bs + (long)h * p.o_hs;
  const bf16_t* Ls = p.stats + ((long)b * p.H + h) * p.Nq64 * 8;
  const bf16_t* Ds = Ls + (long)p.B * p.H * p.Nq64 * 8;
  const int qts = (int)p.q_ts, ots = (int)p.o_ts;
  FragAddr fa;
  frag_addr(fa, lane);
  int r4[2];
#pragma unroll
  for (int sub = 0; sub < 2; sub++) r4[sub] = hi ? TILE_B + (sub * 32 + (lane & 31)) * 16 : fa.rb[0] + 2 * 64 + sub * 32 * ROWB;
  // DMA: the 24 one-KiB pieces of a {Q, dO} tile pair over 8 waves: wave w takes pieces w, w + 8, w + 16 (0..11 = Q tile, 12..23 = dO tile)
  constexpr int DOFF = TILE_B + STAT_B;                    // dO tile relative to the Q tile of its stage
  int prow[3], pcoff[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int g = i * 8 + wave, piece = g % 12, pp = piece * 64 + lane, r = pp / 12, cl = pp - r * 12, cc = cl ^ ((r >> 2) & 3);
    prow[i] = r;
    pcoff[i] = cc < NCH ? cc * 8 : -1;
  }
  const int T = (p.Nq + BKV - 1) / BKV;
  auto issue = [&](int t, char* st) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int g = i * 8 + wave;                           // wave-uniform: which tile, which piece
      const bool isd = g >= 12;
      const int gr = min(t * BKV + prow[i], p.Nq - 1);
      if (pcoff[i] >= 0)
        lds_dma16((isd ? Dp : Qp) + (long)gr * (isd ? ots : qts) + pcoff[i], st + (isd ? DOFF : 0) + (g % 12) * 1024);
    }
    if (wave < 2) lds_dma16((wave == 0 ? Ls : Ds) + ((long)t * BKV + lane) * 8, st + (wave == 0 ? TILE_B : 2 * TILE_B + STAT_B));
  };
  for (int st = 0; st < NSTAGE; st++) {
    init_pads(smem + st * STAGE_B, 0, tid);
    init_pads(smem + st * STAGE_B + DOFF, 0, tid);
  }
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
  auto stage = [&](int i) -> char* { return smem + i * STAGE_B; };
  issue(0, stage(0));
  if (T > 1) issue(1, stage(1));
  lds_dma_wait<0>();
  __syncthreads();                                          // tiles 0 and 1 have landed, pads written
  if (!wave_active) {                                       // DMA, waits and barriers only, in the group's rhythm
    if (grp == 1) bar();
    bar();
    int in2 = 2;
    for (int t = 0; t < T; t++) {
      if (grp == 1 && t + 2 < T) issue(t + 2, stage(in2));  // group 1: start of its first V
      bar();
      if (grp == 0 && t + 2 < T) issue(t + 2, stage(in2));  // group 0: start of its first M
      if (grp == 1) lds_dma_wait<0>();
      bar();
      if (grp == 0) lds_dma_wait<0>();
      bar();
      bar();
      in2 = in2 == NSTAGE - 1 ? 0 : in2 + 1;
    }
    if (grp == 0) bar();
    return;
  }
  f32x16 dk[3], dv[3];
  zero3(dk);
  zero3(dv);
  const float c = p.scale_log2;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  f32x16 s, dp;
  bf16x8 pb[2], db[2], f[NF];
  struct Bases { unsigned r0, r1, r40, r41, t0, t1; };
  auto bases = [&](unsigned st) -> Bases { return Bases{st + (unsigned)fa.rb[0], st + (unsigned)fa.rb[1], st + (unsigned)r4[0], st + (unsigned)r4[1],
                                                        st + (unsigned)fa.tb[0], st + (unsigned)fa.tb[1]}; };
  auto rd_row = [&](auto subc, auto kc, bf16x8& d, const Bases& bs) {
    constexpr int sub = decltype(subc)::value, k = decltype(kc)::value, ks = k >> 1, w = k & 1;
    if constexpr (ks < KSTEPS - 1) lds_row_asm<w * DOFF + sub * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? bs.r1 : bs.r0);
    else lds_row_asm<w * DOFF>(d, sub ? bs.r41 : bs.r40);
  };
  auto rd_tr = [&](auto subc, auto kc, bf16x8& d, const Bases& bs) {
    constexpr int sub = decltype(subc)::value, k = decltype(kc)::value, uu = k / 6, dt = (k % 6) >> 1, w = k & 1;
    lds_tr_asm<(w ? 0 : DOFF) + (sub * 2 + uu) * 16 * ROWB + dt * 64>(d, bs.t0, bs.t1);
  };
  // phase M as ONE chain of 22 fragments: 12 transpose fragments of C(sub CSUB of the stage behind cb; 2 reads each), then 10 row fragments of A(sub ASUB
  // of the stage behind ab; 1 read each).  Fragment i sits in f[i % NF]; the first D = NF - 1 are in flight on entry (issued by the V phase before);
  // slot k = {wait for fragment k: the reads issued after it are those of fragments k+1 .. k+D-1; MFMA k; issue fragment k+D}.  Nothing in flight after.
  auto frag_issue = [&](auto ic_, auto csubc, auto asubc, const Bases& cb, const Bases& ab) {
    constexpr int i = decltype(ic_)::value;
    if constexpr (i < 12) rd_tr(csubc, IntC<i>{}, f[i % NF], cb);
    else if constexpr (i < 22) rd_row(asubc, IntC<i - 12>{}, f[i % NF], ab);
  };
  auto mChain = [&](auto csubc, auto asubc, const Bases& cb, const Bases& ab) {
    static_for<22>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int nafter = [] { int n = 0; for (int i = k + 1; i < k + DK && i < 22; i++) n += i < 12 ? 2 : 1; return n; }();
      lds_wait<nafter>(f[k % NF]);
      if constexpr (k < 12) {
        constexpr int uu = k / 6, dt = (k % 6) >> 1;
        if constexpr (k & 1) dk[dt] = mfma32(f[k % NF], db[uu], dk[dt]);
        else dv[dt] = mfma32(f[k % NF], pb[uu], dv[dt]);
      } else {
        constexpr int j = k - 12;
        if constexpr (j == 0) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; s = mfma32(f[k % NF], kf[0], z); }
        else if constexpr (j == 1) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; dp = mfma32(f[k % NF], vf[0], z); }
        else if constexpr (j & 1) dp = mfma32(f[k % NF], vf[j >> 1], dp);
        else s = mfma32(f[k % NF], kf[j >> 1], s);
      }
      frag_issue(IntC<k + DK>{}, csubc, asubc, cb, ab);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // A alone (prologue): row fragments only, cold
  auto mA0 = [&](const Bases& ab) {
    static_for<DK>([&](auto ic_) { rd_row(IntC<0>{}, ic_, f[decltype(ic_)::value % NF], ab); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<10>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int nafter = (k + DK - 1 < 10 ? DK - 1 : 9 - k);
      lds_wait<nafter>(f[k % NF]);
      if constexpr (k == 0) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; s = mfma32(f[k % NF], kf[0], z); }
      else if constexpr (k == 1) { f32x16 z; for (int g = 0; g < 16; g++) z[g] = 0.f; dp = mfma32(f[k % NF], vf[0], z); }
      else if constexpr (k & 1) dp = mfma32(f[k % NF], vf[k >> 1], dp);
      else s = mfma32(f[k % NF], kf[k >> 1], s);
      if constexpr (k + DK < 10) rd_row(IntC<0>{}, IntC<k + DK>{}, f[(k + DK) % NF], ab);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // phase V: B in place, then the first D fragments of the M phase that follows (C of sub CSUB of the stage behind cb)
  auto vB = [&](auto csubc, const Bases& cb) {
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const float pr = __builtin_amdgcn_exp2f(s[g] * c);
      s[g] = pr;
      dp[g] = pr * dp[g];
    }
#pragma unroll
    for (int uu = 0; uu < 2; uu++) { pb[uu] = pack8(s, 8 * uu); db[uu] = pack8(dp, 8 * uu); }
    asm volatile("" : "+v"(pb[0]), "+v"(pb[1]), "+v"(db[0]), "+v"(db[1]));
    __builtin_amdgcn_sched_barrier(0);
    static_for<DK>([&](auto ic_) { rd_tr(csubc, ic_, f[decltype(ic_)::value % NF], cb); });
    __builtin_amdgcn_sched_barrier(0);
  };

  if (grp == 1) bar();                                      // group 1 sits out the phase in which group 0 computes its A(0, 0)
  mA0(bases(lds0));
  bar();
  int ic = 0;
  for (int t = 0; t < T; t++) {
    const int in = ic == NSTAGE - 1 ? 0 : ic + 1, in2 = in == NSTAGE - 1 ? 0 : in + 1;
    unsigned cst = lds0 + ic * STAGE_B;
    asm volatile("" : "+s"(cst));                           // per-stage lane addresses are rebuilt where they are used, not carried through the loop
    // ---- V: B(2t)
    if (grp == 1 && t + 2 < T) issue(t + 2, stage(in2));
    vB(IntC<0>{}, bases(cst));
    bar();
    // ---- M: C(2t), A(2t+1)
    if (grp == 0 && t + 2 < T) issue(t + 2, stage(in2));
    {
      const Bases cb = bases(cst);
      mChain(IntC<0>{}, IntC<1>{}, cb, cb);
    }
    if (grp == 1) lds_dma_wait<0>();
    bar();
    // ---- V: B(2t+1)
    asm volatile("" : "+s"(cst));
    vB(IntC<1>{}, bases(cst));
    if (grp == 0) lds_dma_wait<0>();
    bar();
    // ---- M: C(2t+1), A(2t+2) (the last tile re-reads its own first sub-tile: unused)
    {
      unsigned nst = lds0 + ((t + 1 < T) ? in : ic) * STAGE_B;
      asm volatile("" : "+s"(cst), "+s"(nst));
      const Bases cb = bases(cst), nb = bases(nst);
      mChain(IntC<1>{}, IntC<0>{}, cb, nb);
    }
    bar();
    ic = in;
  }
  if (grp == 0) bar();                                      // group 1's last phase
  if (kvvalid) {
    store_rows(p.dK + dkbase + (long)kv * p.dk_ts + (long)h * p.dk_hs, dk, p.dk_scale, hi);
    store_rows(p.dV + dvbase + (long)kv * p.dv_ts + (long)h * p.dv_hs, dv, 1.f, hi);
  }
  if (p.dk_colsum) colsum_rows(p.dk_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dk, p.dk_scale, kvvalid, hi, lane);
  if (p.dv_colsum) colsum_rows(p.dv_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dv, 1.f, kvvalid, hi, lane);
}

// ------------------------------------------------------------------------------------------------ forward, ONE wave per SIMD (round 4)
// The organisation of cdna_hip_programming.md "4-wave, one-wave-per-SIMD, persistent structure", for head_dim 72: a workgroup = 4 waves = 256 queries,
// each wave 64 queries (two 32-query blocks s) and the WHOLE 512-register file of its SIMD (__launch_bounds__(256, 1)); nothing is hidden by a
// second wave, everything by software pipelining inside the wave.  Per 64-key tile j two matrix phases, every other instruction placed in their gaps:
//   phase A(j): S'(j+1) = K(j+1) Q^T   (20 v_mfma_f32_32x32x16; K fragments resident in accumulator registers)  || the tile's LDS-DMA (K(j+5), V(j+3));
//               finish-softmax(j): P = exp2(S'(j)), cvt_pk, v_permlane16_swap -> the B operands of PV(j); the first V(j) transpose reads
//   phase B(j): O^T += V(j)^T P(j)^T    (40 v_mfma_f32_16x16x32; V fragments stream through FWD4_NVQ quads)       || start-softmax(j+1): running row maxima of
//               S'(j+1) (v_max3), the K(j+2) row reads into the K registers
// S is double-buffered (2 x 64 registers); ONE s_barrier per tile hands the rings over.  K and V tiles live in two 4-slot rings (96 KiB) and are fetched
// THREE tiles ahead of their first LDS read, behind a counted vmcnt(12): with one workgroup per CU nothing else covers the L2 / HBM latency of a tile (first
// version, 2-slot rings, fetch one tile ahead: 1.42 ms, 1.10 ms with the DMA ablated - profiles/r4_01_fwd4_time.txt).
// Vector work per score: v_max3 (half), exp2, cvt_pk (half) + in the bf16 build the fma that applies scale and running maximum.  In the fp16 build
// (FWD4_FOLD) that fma rides in the first product: Q~ = c Q (c = scale log2 e) is formed once per workgroup, and the zero padding of head_dim 72 -> 80
// holds -m c in slot 72 of the query operand against 1.0 in column 72 of every K tile (the pad column that gives the V tiles their row sums), so the MFMA
// returns S' = c S - m c.  m c is rounded to the operand type when it enters the slot; the SAME rounded value is used for alpha and for
// lse = m c + log2 l, so that rounding cancels exactly (a uniform factor per query row, and l is summed from the very P that multiply V).  What does not
// cancel is the second rounding of the query operand (Q~ = op(c op(q))): +24 % on the forward error of the fp16 build (2.9e-4 -> 3.6e-4 rel-L2 at the
// headline shape), but +30 % with bf16 operands ON TOP of an 8 x coarser grid (2.3e-3 -> 2.9e-3, lse 1e-3 -> 4e-3: the backward's recomputed P follows
// lse) - so the bf16 build keeps the fma.  (Folding c into the q rows of the qkv projection's shadow weights would remove that rounding for both: next.)
// m is the *deferred* maximum: it moves only when some query of the wave sees S' > 6 (P <= 64; checked per tile, wave-uniform branch into a slow path
// that rescales O, shifts the pending S' and rewrites the slot) - on the first tile always.
// Full key tiles only (Nk % 64 == 0, dense keys): self-attention at every bucket resolution and the KV-compressed layers; everything else runs
// attn_fwd2_kernel / the keys-resident kernel.  PXA_ATTN_FWD4=0 switches it off (A/B).
#ifndef FWD4_FOLD
#define FWD4_FOLD PXA_OPERAND_DTYPE_ID      // scale + running maximum inside the first product (fp16 build only, see above)
#endif
#ifndef PXA_ATTN_FWD4_DEFAULT
#define PXA_ATTN_FWD4_DEFAULT FWD4_FOLD      // on where it wins: with the fold (fp16 build) 1.19-1.20 ms against 1.28-1.31; without it the two organisations tie (below)
#endif
// The deferred maximum.  The slow path is EXPENSIVE here (O lives in the accumulator half: 160 v_accvgpr moves, two MFMA drains, 64 S' updates - about
// one tile's time), so the kernel's speed follows the score range through the number of events: with q, k ~ 2 N(0, 1) and the usual P <= 2^6 rule 1.48 ms
// against 1.27 at N(0, 1) - the 12-17 % "in-step gap" of VERDICT r03 item 5 (the step's activations are not N(0, 1); the dQ / dK/dV kernels, which have
// no such path, show no gap; a preceding GEMM, freshly written operands or an HBM write stream in front of the launch change nothing:
// profiles/r4_09_instep_gap.txt, r4_10_fwd_vs_score_range.txt).  Two changes: the window is as wide as the operand type allows - S' <= 11, P <= 2048
// (fp16 holds 65504; l and O accumulate in fp32) - and the first tile's maximum is entered with a MARGIN of 4 (P of the first tile's own maximum = 1/16:
// fp16 keeps full precision down to 2^-14), so the maximum may grow by 15 in the log2 domain (3.3e4 x) over the keys before anything is rescaled.
#ifndef FWD4_THRESH_LOG2
#define FWD4_THRESH_LOG2 11.0f
#endif
#ifndef FWD4_MARGIN_LOG2
#define FWD4_MARGIN_LOG2 4.0f
#endif
constexpr float FWD4_THRESH = FWD4_THRESH_LOG2, FWD4_MARGIN = FWD4_MARGIN_LOG2;
#ifndef FWD4_NVQ
#define FWD4_NVQ 5          // register quads the V^T fragments rotate through: a fragment is read FWD4_NVQ - 1 groups (of 4 MFMAs) ahead of its use
#endif
#ifndef FWD4_KSPREAD
#define FWD4_KSPREAD 0      // K row reads of the next tile: 0 = two per PV group in groups 0..4, 1 = one per group in all ten (A/B)
#endif
#ifndef FWD4_DMA_A
#define FWD4_DMA_A 0        // the tile's three LDS-DMA regions: 0 = behind PV groups 5..7, 1 = in phase A gaps 14 / 16 / 18 (A/B)
#endif
#ifndef FWD4_ABL
#define FWD4_ABL 0          // ablation builds (wrong results, timing only; tools/build_variant.py): 1 no LDS-DMA, 2 no running maxima, 4 no exp2, 8 no first-product MFMAs, 16 no second-product MFMAs, 32 no cvt / lane swaps
#endif
#ifndef FWD4_TRACE
#define FWD4_TRACE 0        // diagnostics build: wave 0 of workgroup 0 sums s_memtime ticks (100 MHz) per tile section: [barrier wait, phase A, phase B, tiles]
#endif
#if FWD4_TRACE
__device__ unsigned long long fwd4_trace_buf[8];
__device__ __forceinline__ unsigned long long fwd4_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
#endif
// Register ownership.  With 512 registers per wave the compiler's own split between the two halves of the file is hopeless (first build of this kernel:
// 177 spills, ~440 v_accvgpr moves per tile), so every value that only the matrix pipe touches is pinned to the accumulator half through asm
// constraints: O ("+a", 80 registers), Q ("a", 40), the K fragments (ds_read_b128 straight into "=a", 40); S / P / the V fragments stay in the arch
// half where the VALU can reach them.  The MFMAs are therefore inline asm - and the compiler neither knows their latency nor pads their hazards: a VALU
// or v_accvgpr read of an MFMA result must sit >= 12 issue states behind it.  In the loop that holds by construction (S'(j+1) is first read half a
// tile after its last MFMA; O is read only in the slow path and the epilogue, both behind mfma_drain()).
#ifdef PXA_OPERAND_F16
#define PXA_MFMA32_ASM "v_mfma_f32_32x32x16_f16"
#define PXA_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
#define PXA_MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#define PXA_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
__device__ __forceinline__ void mfma32_aa_first(f32x16& d, const bf16x8& a, const bf16x8& b) {   // d = A B        (A, B in AGPRs, d in VGPRs)
  asm volatile(PXA_MFMA32_ASM " %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma32_aa(f32x16& d, const bf16x8& a, const bf16x8& b) {         // d += A B
  asm volatile(PXA_MFMA32_ASM " %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
}
__device__ __forceinline__ void mfma16_acc(f32x4& d, const bf16x8& a, const bf16x8& b) {         // d += A B       (A, B in VGPRs, d in AGPRs)
  asm volatile(PXA_MFMA16_ASM " %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
template <class T> __device__ __forceinline__ void to_agpr(T& v) { asm volatile("" : "+a"(v)); }
template <int OFF> __device__ __forceinline__ void lds_row_asm_a(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait_all(bf16x8 (&d)[10]) {
  asm volatile("s_waitcnt lgkmcnt(%10)" : "+a"(d[0]), "+a"(d[1]), "+a"(d[2]), "+a"(d[3]), "+a"(d[4]), "+a"(d[5]), "+a"(d[6]), "+a"(d[7]), "+a"(d[8]), "+a"(d[9]) : "n"(N));
}
// both halves of a wave exchange: x <- {x.lo, x.lo}, y <- {x.hi, x.hi} (asm: with two copies of one value as operands the builtin's second result is
// folded away by the compiler - the first version of this kernel lost the upper half's keys from its running maximum that way)
__device__ __forceinline__ void bcast_halves(float x_in, float& lo, float& hi) {
  float t;
  asm volatile("v_mov_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %1, %0" : "=&v"(t), "+v"(x_in));
  lo = x_in; hi = t;
}
// number of LDS reads that may stay in flight in front of the first MFMA of PV group g = everything issued behind V fragment g.  Issue order: V fragments
// 0 .. NVQ-2 at the end of phase A; then per group gg: [wait], K row read 2 gg, V fragment gg + NVQ - 1, K row read 2 gg + 1 (K reads in groups 0..4 only)
constexpr int fwd4_nwait(int g, int nvq) {
  int n = 0;
  bool seen = false;
  for (int f = 0; f <= nvq - 2; f++) { if (seen) n += 2; if (f == g) seen = true; }
  for (int gg = 0; gg < g; gg++) {
    if ((FWD4_KSPREAD || gg < 5) && seen) n += 1;
    if (gg + nvq - 1 < 10) { if (seen) n += 2; if (gg + nvq - 1 == g) seen = true; }
    if (!FWD4_KSPREAD && gg < 5 && seen) n += 1;
  }
  return n;
}
// one exec region, two LDS-DMA pieces (K and V piece i share their lane mask): saddr form - wave-uniform 64-bit base + per-lane 32-bit byte offset
template <int OFFK, int OFFV>
__device__ __forceinline__ void dma_pair(unsigned long long mask, unsigned wbase, unsigned voffk, const void* kb, unsigned voffv, const void* vb) {
  // (exec is all ones in this kernel - full waves, no divergence outside these statements - so it is set and reset, not saved; the exec write doubles as the
  // wait state between the first M0 write and its LDS-DMA)
  asm volatile("s_add_u32 m0, %1, %2\n\ts_mov_b64 exec, %0\n\tglobal_load_lds_dwordx4 %3, %4\n\ts_add_u32 m0, %1, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %6, %7\n\ts_mov_b64 exec, -1"
               :: "s"(mask), "s"(wbase), "n"(OFFK), "v"(voffk), "s"(kb), "n"(OFFV), "v"(voffv), "s"(vb) : "memory", "scc");
}
template <int OFF>
__device__ __forceinline__ void dma_one(unsigned long long mask, unsigned wbase, unsigned voff, const void* base) {
  unsigned long long sv;
  asm volatile("s_and_saveexec_b64 %0, %1\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b64 exec, %0"
               : "=&s"(sv) : "s"(mask), "s"(wbase), "n"(OFF), "v"(voff), "s"(base) : "memory", "scc");
}
__global__ __launch_bounds__(256, 1) void attn_fwd4_kernel(AttnParams p) {
  constexpr int QS = 2, NVQ = FWD4_NVQ;
  constexpr bool FOLD = FWD4_FOLD;
  constexpr int RING = 4, KOFF = 0, VOFF = RING * TILE_B;         // LDS: K ring [4 tiles] | V ring [4 tiles]
  constexpr int LEAD = 3;                                          // tiles between a tile's DMA and its first LDS read (vmcnt(6 (LEAD - 1)) at the barrier)
  __shared__ __attribute__((aligned(16))) char smem[2 * RING * TILE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const bf16_t* Kp = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + (long)b * p.v_bs + (long)h * p.v_hs;
  const int kts = (int)p.k_ts, vts = (int)p.v_ts;
  const int T = p.Nk / BKV;                                       // >= 1, full tiles (checked by the launcher)
  const float c = p.scale_log2;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  int q[QS];
  bool qvalid[QS];
  bf16x8 qf[QS][KSTEPS];                                           // (loaded in the prologue, behind the first tiles' DMA)
  for (int st = 0; st < 2 * RING; st++) init_pads(smem + st * TILE_B, 1, tid);   // column 72 = 1.0 in K tiles (x slot 72 of Q~) and V tiles (row sums)

  // LDS-DMA plan: per-lane byte offsets inside a tile's rows (saddr form), one lane mask per piece index
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  unsigned offK[NDMA], offV[NDMA];
  unsigned long long dmask[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; i++) {
    offK[i] = (unsigned)(pl.row[i] * kts + pl.coff[i]) * 2u;
    offV[i] = (unsigned)(pl.row[i] * vts + pl.coff[i]) * 2u;
    dmask[i] = __builtin_amdgcn_ballot_w64(pl.coff[i] >= 0);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned wbase = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
  const long kstep = (long)BKV * kts, vstep = (long)BKV * vts;     // elements per tile
  auto ktile = [&](int t) { return Kp + (long)min(t, T - 1) * kstep; };   // past the last tile: a harmless re-fetch of it into a slot nobody reads
  auto vtile = [&](int t) { return Vp + (long)min(t, T - 1) * vstep; };
  auto dma_k = [&](auto slotc, const bf16_t* kb) {
    constexpr int S = decltype(slotc)::value;
    dma_one<KOFF + S * TILE_B>(dmask[0], wbase, offK[0], kb);
    dma_one<KOFF + S * TILE_B + 4096>(dmask[1], wbase, offK[1], kb);
    dma_one<KOFF + S * TILE_B + 8192>(dmask[2], wbase, offK[2], kb);
  };
  auto dma_kv = [&](auto kslotc, auto vslotc, const bf16_t* kb, const bf16_t* vb) {
    constexpr int SK = decltype(kslotc)::value, SV = decltype(vslotc)::value;
    dma_pair<KOFF + SK * TILE_B, VOFF + SV * TILE_B>(dmask[0], wbase, offK[0], kb, offV[0], vb);
    dma_pair<KOFF + SK * TILE_B + 4096, VOFF + SV * TILE_B + 4096>(dmask[1], wbase, offK[1], kb, offV[1], vb);
    dma_pair<KOFF + SK * TILE_B + 8192, VOFF + SV * TILE_B + 8192>(dmask[2], wbase, offK[2], kb, offV[2], vb);
  };

  // fragment addressing (LDS byte addresses; slot and fragment offsets are instruction immediates)
  FragAddr fa;
  frag_addr(fa, lane);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  const unsigned kr0 = lds0 + (unsigned)fa.rb[0], kr1 = lds0 + (unsigned)fa.rb[1];
  const unsigned vt00 = lds0 + VOFF + (unsigned)ta.tb[0][0], vt01 = lds0 + VOFF + (unsigned)ta.tb[0][1], vt10 = lds0 + VOFF + (unsigned)ta.tb[1][0],
                 vt11 = lds0 + VOFF + (unsigned)ta.tb[1][1];        // (the V ring's base sits in the address registers: instruction offsets are 16 bits)
  bf16x8 kf[10];                                                   // K(j+1) fragments: kf[sub * 5 + ks]
  auto rd_k = [&](auto slotc, auto kc) {
    constexpr int S = decltype(slotc)::value, k = decltype(kc)::value, sub = k / 5, ks = k % 5;
    lds_row_asm_a<KOFF + S * TILE_B + sub * 32 * ROWB + (ks >> 1) * 64>(kf[k], (ks & 1) ? kr1 : kr0);
  };
  bf16x8 vfr[NVQ];                                                 // V(j) transposed fragments, fragment g = sub * 5 + t in vfr[g % NVQ]
  auto rd_v = [&](auto slotc, auto gc) {
    constexpr int S = decltype(slotc)::value, g = decltype(gc)::value, sub = g / 5, t = g % 5;
    lds_tr_asm<S * TILE_B + sub * 32 * ROWB + (t >> 1) * 64>(vfr[g % NVQ], (t & 1) ? vt01 : vt00, (t & 1) ? vt11 : vt10);
  };

  Acc16 o[QS];
  f32x16 sc[2][QS][2];                                             // [buffer][query block s][key half sub]
  u32x4 pxu[QS][2], pyu[QS][2];                                    // P(j) packed: the two B operands (query columns 0-15 / 16-31) of each 32 x 32 block
  float mc[QS];                                                    // m c of this lane's query (lane & 31) of block s, log2 domain - FOLD: exactly the value in the slot
  float mcsel = 0.f;                                               // hi ? mc[1] : mc[0]
  float mrel = 0.f;                                                // after a tile's maxima: lower half lanes: max S' of block 0's query, upper half: block 1's
#pragma unroll
  for (int s = 0; s < QS; s++) {
    zero16(o[s]); mc[s] = 0.f;
#pragma unroll
    for (int t = 0; t < NT16; t++) { to_agpr(o[s].v[t][0]); to_agpr(o[s].v[t][1]); }
  }

  // the slow path: the deferred maximum of some query moves (always on the first tile).  `nxt` = the pending S' buffer (computed with the old slot value).
  auto rescale = [&](auto nxtc, auto firstc) {
    constexpr int NXT = decltype(nxtc)::value;
    constexpr bool first = decltype(firstc)::value;              // first tile: O is still zero (and alpha may overflow), the maximum may move DOWN from 0
    mfma_drain();                                                 // the last PV MFMAs' results (asm MFMAs: no compiler-inserted hazard padding)
    float mx[QS];
    bcast_halves(mrel, mx[0], mx[1]);
#pragma unroll
    for (int s = 0; s < QS; s++) {
      // a query block none of whose rows left the window keeps its maximum (wave-uniform skip: usually ONE block triggers the event, and the block's
      // share of the slow path - 32 score updates, 80 register moves + 40 multiplies on O - is most of its cost)
      if (!first && __builtin_amdgcn_ballot_w64(mx[s] > FWD4_THRESH) == 0) continue;
      const float dmax = first ? mx[s] + FWD4_MARGIN : fmaxf(mx[s], 0.f);
      float mnew = mc[s] + dmax;
      bf16_t nb = (bf16_t)0.f;
      if constexpr (FOLD) { nb = (bf16_t)(-mnew); mnew = -(float)nb; }   // what the slot will hold, rounded to the operand type
      const float delta = mnew - mc[s];
      mc[s] = mnew;
      if constexpr (FOLD) {
#pragma unroll
        for (int sub = 0; sub < 2; sub++)
#pragma unroll
          for (int g = 0; g < 16; g++) sc[NXT][s][sub][g] -= delta;
        u32x4 w = __builtin_bit_cast(u32x4, qf[s][KSTEPS - 1]);
        bf16x2 sl; sl[0] = nb; sl[1] = (bf16_t)0.f;
        w[0] = hi ? __builtin_bit_cast(unsigned, sl) : w[0];
        qf[s][KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
        to_agpr(qf[s][KSTEPS - 1]);
      }
      if constexpr (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        const float ao = __shfl_xor(alpha, 16);
        const float a0 = (lane & 16) ? ao : alpha, a1 = (lane & 16) ? alpha : ao;
#pragma unroll
        for (int t = 0; t < NT16; t++) { o[s].v[t][0] *= a0; o[s].v[t][1] *= a1; }
#pragma unroll
        for (int t = 0; t < NT16; t++) { to_agpr(o[s].v[t][0]); to_agpr(o[s].v[t][1]); }
      }
    }
    mcsel = hi ? mc[1] : mc[0];
    asm volatile("s_nop 3" ::: "memory");                          // v_accvgpr_write -> MFMA operand
    __builtin_amdgcn_sched_barrier(0);
  };
  // running maxima of a pending S' buffer -> mrel (halves exchanged: 1 swap + 1 max for both query blocks); FOLD: S' is already relative to m c
  auto finish_max = [&](float m0, float m1) {              // lower half lanes <- max over both halves of m0, upper half lanes <- of m1
    // (asm: hipcc 7.2 folds fmaxf over the two results of __builtin_amdgcn_permlane32_swap to its first result - /tmp reproducer in DESIGN.md - which is
    // how the first version of this kernel lost the upper half's keys from its running maximum)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(m0), "+v"(m1));
    mrel = FOLD ? m0 : fmaf(m0, c, -mcsel);
  };

  // ---- prologue.  DMA issue order = retirement order (vmcnt counts it): K(0) K(1) | K(2) V(0) | K(3) V(1) | - K(0) into registers, barrier - K(4) V(2)
  dma_k(IntC<0>{}, ktile(0));
  dma_k(IntC<1>{}, ktile(1));
  dma_kv(IntC<2>{}, IntC<0>{}, ktile(2), vtile(0));
  dma_kv(IntC<3>{}, IntC<1>{}, ktile(3), vtile(1));
  // (the query rows are fetched BEHIND the first tiles' DMA: their latency and the tiles' overlap instead of adding up - 2 us of a 75 us workgroup)
  // query operands (FOLD: prescaled, Q~ = c Q in the operand type; slot 72 = k-step 4, upper half, element 0 carries -m c, initially 0)
#pragma unroll
  for (int s = 0; s < QS; s++) {
    q[s] = bx * 256 + wave * 64 + s * 32 + (lane & 31);
    qvalid[s] = q[s] < p.Nq;
    load_row_frags(qf[s], p.Q + (long)b * p.q_bs + (long)q[s] * p.q_ts + (long)h * p.q_hs, qvalid[s], hi);
    settle(qf[s]);
    if constexpr (FOLD) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) qf[s][ks][e] = (bf16_t)((float)qf[s][ks][e] * c);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) to_agpr(qf[s][ks]);
  }
  lds_dma_wait<12>();                                              // K(0), K(1) landed (this wave's pieces)
  __syncthreads();                                                 // ... everybody's, and the pads are written
  static_for<10>([&](auto kc) { rd_k(IntC<0>{}, kc); });
  lds_wait_all<0>(kf);
  __syncthreads();                                                 // every wave holds K(0): its slot is free
  dma_kv(IntC<0>{}, IntC<2>{}, ktile(4), vtile(2));
  const bf16_t* kdma = ktile(5);                                   // tile body j fetches K(j+5) and V(j+3)
  const bf16_t* vdma = vtile(3);
#pragma unroll
  for (int s = 0; s < QS; s++)
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      mfma32_aa_first(sc[0][s][sub], kf[sub * 5], qf[s][0]);
#pragma unroll
      for (int ks = 1; ks < KSTEPS; ks++) mfma32_aa(sc[0][s][sub], kf[sub * 5 + ks], qf[s][ks]);
    }
  static_for<10>([&](auto kc) { rd_k(IntC<1>{}, kc); });           // K(1): waited for in front of the first tile's barrier
  mfma_drain();
  {
    float m0 = sc[0][0][0][0], m1 = sc[0][1][0][0];
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int g = 0; g < 16; g++) { m0 = fmaxf(m0, sc[0][0][sub][g]); m1 = fmaxf(m1, sc[0][1][sub][g]); }
    finish_max(m0, m1);
  }
  rescale(IntC<0>{}, BoolC<true>{});

  // ---- one tile j, J = j & 3 (ring slots), CUR = j & 1 (S buffers).  ONE form for every tile: past the last tile the DMA sources are clamped to it and
  // the last tile's phase A / maxima work on a tile that does not exist (the K registers still hold K(T-1): finite scores, never used; a rescale they may
  // trigger scales O, l and m consistently) - 0.8 % extra matrix work at 64 tiles, against specialised copies of this body whose merges cost spills and
  // register copies in the first version.
#if FWD4_TRACE
  unsigned long long tr_wait = 0, tr_a = 0, tr_b = 0, tr_n = 0;
#endif
  auto body = [&](auto jc) {
    constexpr int J = decltype(jc)::value, CUR = J & 1, NXT = CUR ^ 1;
    constexpr int KRD = (J + 2) % RING, VRD = J, KWR = (J + 2 + LEAD) % RING, VWR = (J + LEAD) % RING;
#if FWD4_TRACE
    const unsigned long long t0 = fwd4_now();
#endif
    lds_wait_all<0>(kf);                                           // this wave's K(j+1) row reads are complete: their slot is refilled behind a later barrier
    lds_dma_wait<6 * (LEAD - 1)>();                                // this wave's pieces of K(j+2), V(j) have landed; the two younger tiles stay in flight
    __syncthreads();                                               // ... every wave's; and every wave is past phase B(j-1)
    __builtin_amdgcn_sched_barrier(0);
#if FWD4_TRACE
    const unsigned long long t1 = fwd4_now();
#endif
    // Issue budget (SQ counters of the first working version, profiles/r4_04_pmc_fwd4_sq.txt: ~5.4 cycles per issued instruction, exp2 and the 32-row MFMA
    // two issue quads each; a wave is alone on its SIMD, so every instruction of the tile queues behind every other): the first product's 640 matrix
    // cycles can carry ~120 quads of other work, the second product's 640 another ~120 - and the softmax alone is 64 exp2 (128 quads) + 48 cvt / swap.
    // All of it sat in phase A at first (phase A 960 cycles for 640 of MFMA, phase B matrix-bound with idle issue slots); now the tile's four 32 x 32
    // blocks of P are split: blocks 0, 1 (key half 0, needed by PV groups 0-4) in phase A, blocks 2, 3 (key half 1, needed from group 5 on) in the
    // shadow of PV groups 0-4; the running maxima of S'(j+1) follow its first product half by half (key half 0 in the tail of phase A, key half 1 in
    // PV groups 5-9, where the LDS-DMA sits too).
    // micro-instruction k = 0..27 of softmax block blk -> (s, sub) = (blk & 1, blk >> 1): 8 exp2, 4 cvt_pk, 2 swaps - twice (pack_xy's a / b parts)
    auto smx = [&](auto blkc, auto kc) {
      constexpr int blk = decltype(blkc)::value, k = decltype(kc)::value, bs = blk & 1, bsub = blk >> 1;
      if constexpr (k < 28) {
        f32x16& v = sc[CUR][bs][bsub];
        u32x4& ux = pxu[bs][bsub]; u32x4& uy = pyu[bs][bsub];
        constexpr int hk = k % 14, hb = k / 14;                          // half hb covers scores 8 hb .. 8 hb + 7 -> words 2 hb, 2 hb + 1 of both operands
        if constexpr (hk < 8) {
          constexpr int g = 8 * hb + hk;
          if (!(FWD4_ABL & 4)) v[g] = __builtin_amdgcn_exp2f(FOLD ? v[g] : fmaf(v[g], c, -mc[bs]));
          asm volatile("" : "+v"(v[g]));
        } else if constexpr (hk < 12) {
          constexpr int w = 2 * hb + ((hk - 8) >> 1), isb = (hk - 8) & 1, g0 = (w & 1) * 2 + (w >> 1) * 8 + 4 * isb;
          unsigned d;
          if (FWD4_ABL & 32) d = __builtin_bit_cast(unsigned, v[g0]); else d = pack_bf16x2(v[g0], v[g0 + 1]);
          asm volatile("" : "+v"(d));
          if constexpr (isb) uy[w] = d; else ux[w] = d;
        } else {
          constexpr int w = 2 * hb + (hk - 12);
          if (!(FWD4_ABL & 32)) { const auto r = __builtin_amdgcn_permlane16_swap(ux[w], uy[w], false, false); ux[w] = r[0]; uy[w] = r[1]; }
          asm volatile("" : "+v"(ux[w]), "+v"(uy[w]));
        }
      }
    };
    // running maxima of S'(j+1): slice m = 0..15 of key half msub -> query block m & 1, scores 2 (m >> 1), + 1 (asm: no canonicalising v_max in front)
    float m0 = 0.f, m1 = 0.f;
    auto mxs = [&](auto msubc, auto mc_) {
      constexpr int msub = decltype(msubc)::value, m = decltype(mc_)::value, ms = m & 1, e = (m >> 1) * 2;
      if constexpr (!(FWD4_ABL & 2)) {
        float& mm = ms ? m1 : m0;
        const f32x16& v = sc[NXT][ms][msub];
        if constexpr (msub == 0 && e == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mm) : "v"(v[0]), "v"(v[1]));
        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mm) : "v"(v[e]), "v"(v[e + 1]));
      }
    };
    // -------- phase A
    static_for<20>([&](auto ac) {
      constexpr int a = decltype(ac)::value, sub = a / 10, ks = (a % 10) / 2, s = a % 2;
      if constexpr (!(FWD4_ABL & 8)) {
        if constexpr (ks == 0) mfma32_aa_first(sc[NXT][s][sub], kf[sub * 5], qf[s][0]);
        else mfma32_aa(sc[NXT][s][sub], kf[sub * 5 + ks], qf[s][ks]);
      }
      if constexpr (a < 14) static_for<4>([&](auto ic) { smx(IntC<a / 7>{}, IntC<4 * (a % 7) + decltype(ic)::value>{}); });
      else {                                                             // 16 maxima slices of key half 0 over gaps 14..19: 3 3 3 3 2 2
        constexpr int first = a < 18 ? 3 * (a - 14) : 12 + 2 * (a - 18), cnt = a < 18 ? 3 : 2;
        static_for<cnt>([&](auto ic) { mxs(IntC<0>{}, IntC<first + decltype(ic)::value>{}); });
      }
      if constexpr (a >= 20 - (NVQ - 1)) rd_v(IntC<VRD>{}, IntC<a - (20 - (NVQ - 1))>{});   // V fragments 0 .. NVQ-2
      if constexpr (FWD4_DMA_A && (a == 14 || a == 16 || a == 18) && !(FWD4_ABL & 1)) {
        constexpr int i = (a - 14) / 2;
        dma_pair<KOFF + KWR * TILE_B + i * 4096, VOFF + VWR * TILE_B + i * 4096>(dmask[i], wbase, offK[i], kdma, offV[i], vdma);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // -------- phase B
    static_for<40>([&](auto bc) {
      constexpr int bb = decltype(bc)::value, g = bb / 4, i = bb % 4, sub = g / 5, t = g % 5, s = i / 2, half = i % 2;
      if constexpr (i == 0) lds_wait<fwd4_nwait(g, NVQ)>(vfr[g % NVQ]);
      if constexpr (!(FWD4_ABL & 16)) mfma16_acc(o[s].v[t][half], vfr[g % NVQ], __builtin_bit_cast(bf16x8, half ? pyu[s][sub] : pxu[s][sub]));
      else asm volatile("" :: "v"(vfr[g % NVQ]), "v"(half ? pyu[s][sub] : pxu[s][sub]));
      if constexpr (FWD4_KSPREAD && i == 0) rd_k(IntC<KRD>{}, IntC<g>{});
      if constexpr (!FWD4_KSPREAD && g < 5 && i == 0) rd_k(IntC<KRD>{}, IntC<(g < 5 ? 2 * g : 0)>{});
      if constexpr (i == 1 && g + NVQ - 1 < 10) rd_v(IntC<VRD>{}, IntC<(g + NVQ - 1 < 10 ? g + NVQ - 1 : 0)>{});
      if constexpr (!FWD4_KSPREAD && g < 5 && i == 2) rd_k(IntC<KRD>{}, IntC<(g < 5 ? 2 * g + 1 : 0)>{});
      if constexpr (bb < 20) static_for<3>([&](auto ic) { smx(IntC<2 + bb / 10>{}, IntC<3 * (bb % 10) + decltype(ic)::value>{}); });
      else if constexpr (bb < 36) mxs(IntC<1>{}, IntC<bb - 20>{});
      if constexpr (bb == 37 && !(FWD4_ABL & 2)) { finish_max(m0, m1); asm volatile("" : "+v"(mrel)); }
      if constexpr (!FWD4_DMA_A && g >= 5 && g < 8 && i == 3 && !(FWD4_ABL & 1))
        dma_pair<KOFF + KWR * TILE_B + (g - 5) * 4096, VOFF + VWR * TILE_B + (g - 5) * 4096>(dmask[g - 5], wbase, offK[g - 5], kdma, offV[g - 5], vdma);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (__builtin_amdgcn_ballot_w64(mrel > FWD4_THRESH) != 0) rescale(IntC<NXT>{}, BoolC<false>{});
  };

  for (int j = 0; j < T; j += 4) {
    body(IntC<0>{});
    kdma += j + 6 < T ? kstep : 0;  vdma += j + 4 < T ? vstep : 0;
    if (j + 1 >= T) break;
    body(IntC<1>{});
    kdma += j + 7 < T ? kstep : 0;  vdma += j + 5 < T ? vstep : 0;
    if (j + 2 >= T) break;
    body(IntC<2>{});
    kdma += j + 8 < T ? kstep : 0;  vdma += j + 6 < T ? vstep : 0;
    if (j + 3 >= T) break;
    body(IntC<3>{});
    kdma += j + 9 < T ? kstep : 0;  vdma += j + 7 < T ? vstep : 0;
  }
  lds_wait_all<0>(kf);
  lds_dma_wait<0>();                                               // the clamped re-fetches must not land in a later workgroup's LDS
#if FWD4_TRACE
  if (blockIdx.x == 0 && tid == 0) { fwd4_trace_buf[0] = tr_wait; fwd4_trace_buf[1] = tr_a; fwd4_trace_buf[2] = tr_b; fwd4_trace_buf[3] = tr_n; }
#endif

  mfma_drain();
#pragma unroll
  for (int s = 0; s < QS; s++) {
    const float la = __shfl(o[s].v[4][0][0], 32 + (lane & 15)), lb = __shfl(o[s].v[4][1][0], 32 + (lane & 15));
    const float l = (lane & 16) ? lb : la;             // row 72 of O^T = sum over keys of the P actually multiplied into O, of query l & 31
    const float inv = l > 0.f ? 1.f / l : 0.f, invo = __shfl_xor(inv, 16);
    const int q0w = bx * 256 + wave * 64 + s * 32;
    store_rows16(p.O + (long)b * p.o_bs + (long)q0w * p.o_ts + (long)h * p.o_hs, p.o_ts, o[s], (lane & 16) ? invo : inv, (lane & 16) ? inv : invo,
                 q0w + (lane & 15) < p.Nq, q0w + 16 + (lane & 15) < p.Nq, lane);
    if (qvalid[s] && hi == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Nq + q[s]] = mc[s] + log2f(l);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK / dV, ONE wave per SIMD (round 4)
// What the forward kernel above taught (profiles/r4_06_fwd4_ab.txt): on this part the ORGANISATION of a fixed instruction mix does not move the time - one
// wave per SIMD placed by hand ties with two in-order waves - the instruction COUNT does.  attn_bwd_dkv2_kernel's count is dominated by operand
// movement: its keys are stationary (32 per wave) and every MFMA consumes a fresh Q / dO fragment from LDS - 84 LDS reads for 44 MFMAs per tile and
// wave, ~125 B/clk of LDS traffic per CU against a 128 B/clk port.  Here a wave owns 64 keys (two 32-key blocks kb) and the whole register file:
// every fragment read from LDS feeds TWO MFMAs (34 read instructions for 44 MFMAs per 32-query sub-tile: 0.77 per MFMA instead of 1.9), half the waves
// read the same Q / dO tile.  dK / dV accumulators (192 registers) and the V row operands live in the accumulator half ("a" constraints, asm MFMAs:
// see "Register ownership" above); S / dP / P / dS and the fragment quads in the arch half.
// Work of one 32-query sub-tile j (step), in issue order - 44 MFMAs, each fragment f feeding key blocks 0 and 1 back to back:
//   S(j+1)  10 MFMAs   Q rows of the NEXT sub-tile x K^T       -> S'(j+1) = S - lse / c  (statistics rows ride in k-slots 72..74, see attn_bwd_dkv2_kernel)
//   dP(j)   10 MFMAs   dO rows x V^T                           -> dP' = dP - delta
//   dV(j)   12 MFMAs   dO^T (transpose reads) x P(j)
//   dK(j)   12 MFMAs   Q^T  (transpose reads) x dS(j)
// with the vector work of the step in their gaps, 4 per gap: E(j): P = exp2(c S'(j)) + the cvt_pk of P under S(j+1) / dP(j) (S is double-buffered, nothing
// else is); M(j): dS = P dP' + its cvt_pk under dV(j); the gaps of dK(j) carry the tile hand-over.  Fragments rotate through four register quads, read
// two fragments ahead (22 fragments per step: the rotation closes over the two steps of a tile), waits are hand-counted (tools/check_lds_waits.py).
// Ring: {Q tile, L rows, dO tile, D rows} x 4 stages (104 KiB, one workgroup per CU), tile t+3 fetched behind the ONE barrier of tile t, which sits in
// front of the first look-ahead read into tile t+1 (two fragments before the end of sub-tile 0) behind a counted vmcnt(7): tile t+2 stays in flight.
// Every wave issues 7 LDS-DMA pieces per tile (waves 2 / 3 repeat the statistics pieces of waves 0 / 1: same bytes, uniform vmcnt accounting).
// Dense keys in whole 64-key blocks (a workgroup's last waves may own no keys: they serve DMA and barriers on zero operands), whole 64-query tiles;
// everything else runs attn_bwd_dkv2_kernel.
// (W >= 0: the counted wait for the fragment `a` rides in the MFMA's own statement - a separate wait statement with the fragment as its output draws a
// compiler boundary s_nop in front of every consumer: 44 per tile in the first build of this kernel)
#define PXA_WAIT_STR "s_waitcnt lgkmcnt(%3)\n\t"
template <int W = -1> __device__ __forceinline__ void mfma32_va_first(f32x16& d, const bf16x8& a, const bf16x8& b) {
  if constexpr (W >= 0) asm volatile(PXA_WAIT_STR PXA_MFMA32_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b), "n"(W));
  else asm volatile(PXA_MFMA32_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}
template <int W = -1> __device__ __forceinline__ void mfma32_va(f32x16& d, const bf16x8& a, const bf16x8& b) {
  if constexpr (W >= 0) asm volatile(PXA_WAIT_STR PXA_MFMA32_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b), "n"(W));
  else asm volatile(PXA_MFMA32_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
template <int W = -1> __device__ __forceinline__ void mfma32_vv_first(f32x16& d, const bf16x8& a, const bf16x8& b) {
  if constexpr (W >= 0) asm volatile(PXA_WAIT_STR PXA_MFMA32_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b), "n"(W));
  else asm volatile(PXA_MFMA32_ASM " %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <int W = -1> __device__ __forceinline__ void mfma32_vv(f32x16& d, const bf16x8& a, const bf16x8& b) {
  if constexpr (W >= 0) asm volatile(PXA_WAIT_STR PXA_MFMA32_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b), "n"(W));
  else asm volatile(PXA_MFMA32_ASM " %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
template <int W = -1> __device__ __forceinline__ void mfma32_acc(f32x16& d, const bf16x8& a, const bf16x8& b) {
  if constexpr (W >= 0) asm volatile(PXA_WAIT_STR PXA_MFMA32_ASM " %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b), "n"(W));
  else asm volatile(PXA_MFMA32_ASM " %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}
#ifndef PXA_ATTN_DKV4_DEFAULT
#define PXA_ATTN_DKV4_DEFAULT 1
#endif
#ifndef DKV4_ABL
#define DKV4_ABL 0          // ablation builds (wrong results, timing only): 1 no exp2, 2 no cvt_pk, 4 no multiplies, 8 no LDS fragment reads, 16 no LDS-DMA, 32 no MFMAs
#endif
constexpr int DKV4_STAGES = 4;
constexpr int dkv4_nreads(int i) { const int k = ((i % 22) + 22) % 22; return k < 10 ? 1 : 2; }   // fragment i of a step: 10 row fragments, 12 transposed ones
template <bool PRE>     // PRE: q arrives as (scale log2 e) x queries (pxa_attn_args.q_prescaled): S needs no multiply in front of exp2
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv4_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DKV4_STAGES * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const long kbase = (long)b * p.k_bs, vbase = (long)b * p.v_bs, dkbase = (long)b * p.dk_bs, dvbase = (long)b * p.dv_bs;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs;
  const bf16_t* Dp = p.dO + (long)b * p.o_bs + (long)h * p.o_hs;
  const bf16_t* Ls = p.stats + ((long)b * p.H + h) * p.Nq64 * 8;
  const bf16_t* Ds = Ls + (long)p.B * p.H * p.Nq64 * 8;
  const int qts = (int)p.q_ts, ots = (int)p.o_ts;
  const int T = p.Nq / BKV;                                        // full 64-query tiles (checked by the launcher)
  const float c = p.scale_log2;

  // LDS-DMA plan (saddr form: wave-uniform tile base + per-lane byte offset; Q and dO piece i share their lane mask)
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  unsigned offQ[NDMA], offD[NDMA];
  unsigned long long dmask[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; i++) {
    offQ[i] = (unsigned)(pl.row[i] * qts + pl.coff[i]) * 2u;
    offD[i] = (unsigned)(pl.row[i] * ots + pl.coff[i]) * 2u;
    dmask[i] = __builtin_amdgcn_ballot_w64(pl.coff[i] >= 0);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned wbase = __builtin_amdgcn_readfirstlane(wave * 1024);          // (stage addresses are added per fetch)
  const long qstep = (long)BKV * qts, ostep = (long)BKV * ots;
  const unsigned stat_off = (unsigned)lane * 16u;                  // statistics rows: 64 x 16 B per tile, one piece; waves 0 / 2 fetch L, waves 1 / 3 D
  const bf16_t* statp = (wave & 1) ? Ds : Ls;
  const unsigned stat_dst = __builtin_amdgcn_readfirstlane((wave & 1) ? 2 * TILE_B + STAT_B : TILE_B);
  // tile fetch, in four parts (three {Q, dO} piece pairs + the statistics piece) so that the loop can spread them over MFMA gaps; the source pointers
  // are running ones (qnext / dnext / snext: the next tile to fetch, clamped to the last one - past it a harmless re-fetch keeps every wave's piece
  // count, and with it the counted vmcnt, uniform)
  const bf16_t* qnext = Qp;
  const bf16_t* dnext = Dp;
  const bf16_t* snext = statp;
  auto issue_part = [&](auto pc, unsigned sb) {                    // sb = LDS byte address of the stage
    constexpr int P = decltype(pc)::value;
    const unsigned wb = wbase + sb, so = stat_off, sd = stat_dst + sb;
    const bf16_t* sn = snext;
    if constexpr (P == 0) dma_pair<0, TILE_B + STAT_B>(dmask[0], wb, offQ[0], qnext, offD[0], dnext);
    if constexpr (P == 1) dma_pair<4096, TILE_B + STAT_B + 4096>(dmask[1], wb, offQ[1], qnext, offD[1], dnext);
    if constexpr (P == 2) dma_pair<8192, TILE_B + STAT_B + 8192>(dmask[2], wb, offQ[2], qnext, offD[2], dnext);
    if constexpr (P == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(sd), "v"(so), "s"(sn) : "memory");
  };
  int tfetch = 0;                                                  // tile index behind qnext / dnext / snext
  auto advance = [&]() {
    const bool more = tfetch + 1 < T;
    qnext += more ? qstep : 0; dnext += more ? ostep : 0; snext += more ? (long)BKV * 8 : 0;
    tfetch++;
  };
  auto issue = [&](unsigned sb) { issue_part(IntC<0>{}, sb); issue_part(IntC<1>{}, sb); issue_part(IntC<2>{}, sb); issue_part(IntC<3>{}, sb); advance(); };

  // fragment addressing inside a stage: per-lane bases + instruction immediates (see attn_bwd_dkv2_kernel); `cur` = stage of tile t, `nxt` = of tile t+1
  FragAddr fa;
  frag_addr(fa, lane);
  int r4[2];
#pragma unroll
  for (int sub = 0; sub < 2; sub++) r4[sub] = hi ? TILE_B + (sub * 32 + (lane & 31)) * 16 : fa.rb[0] + 2 * 64 + sub * 32 * ROWB;
  struct Bases { unsigned r0, r1, r40, r41, t0, t1; };
  auto bases = [&](unsigned st) -> Bases { return Bases{st + (unsigned)fa.rb[0], st + (unsigned)fa.rb[1], st + (unsigned)r4[0], st + (unsigned)r4[1],
                                                        st + (unsigned)fa.tb[0], st + (unsigned)fa.tb[1]}; };
  constexpr int DOFF = TILE_B + STAT_B;

  for (int st = 0; st < DKV4_STAGES; st++) {
    init_pads(smem + st * STAGE_B, 0, tid);
    init_pads(smem + st * STAGE_B + DOFF, 0, tid);
  }
  f32x16 dk[2][3], dv[2][3];
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    zero3(dk[kb]); zero3(dv[kb]);
#pragma unroll
    for (int dt = 0; dt < 3; dt++) { to_agpr(dk[kb][dt]); to_agpr(dv[kb][dt]); }
  }
  f32x16 S[2][2], dP[2];                                           // S[buffer][kb] (S'(j) in buffer j & 1; E(j) leaves P there), dP[kb]
  bf16x8 pb[2][2], db[2][2];                                       // [kb][uu]: P / dS of 16 queries each, packed (B operands of the second products)
  bf16x8 f[4];                                                     // fragment quads

  // fragment i of step (SUB): 0..4 Q rows of the next sub-tile (k-step i), 5..9 dO rows, 10..15 dO^T (uu, dt), 16..21 Q^T (uu, dt); i >= 22: the next
  // step's fragments (look-ahead).  cb = this tile's stage, nb = the next tile's.
  auto rd_frag = [&](auto subc, auto ic, bf16x8& d, const Bases& cb, const Bases& nb) {
    constexpr int SUB = decltype(subc)::value, I = decltype(ic)::value;
    if constexpr (I >= 22) {                                       // next step: its fragments 0..3 are looked ahead (Q rows, k-steps 0..3)
      constexpr int ks = I - 22;
      static_assert(ks < KSTEPS - 1, "look-ahead reaches the statistics fragment");
      // next step = (SUB ^ 1): its S block reads the sub-tile after it: SUB == 0 -> next step is sub 1 of this tile, reads (t+1, sub 0); SUB == 1 -> next
      // step is sub 0 of tile t+1, reads (t+1, sub 1)
      lds_row_asm<(SUB ? 32 * ROWB : 0) + (ks >> 1) * 64>(d, (ks & 1) ? nb.r1 : nb.r0);
    } else if constexpr (I < 5) {
      constexpr int ks = I;
      if constexpr (SUB == 0) {                                    // (t, sub 1)
        if constexpr (ks < KSTEPS - 1) lds_row_asm<32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? cb.r1 : cb.r0);
        else lds_row_asm<0>(d, cb.r41);
      } else {                                                     // (t+1, sub 0)
        if constexpr (ks < KSTEPS - 1) lds_row_asm<(ks >> 1) * 64>(d, (ks & 1) ? nb.r1 : nb.r0);
        else lds_row_asm<0>(d, nb.r40);
      }
    } else if constexpr (I < 10) {
      constexpr int ks = I - 5;
      if constexpr (ks < KSTEPS - 1) lds_row_asm<DOFF + SUB * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? cb.r1 : cb.r0);
      else lds_row_asm<DOFF>(d, SUB ? cb.r41 : cb.r40);
    } else {
      constexpr int k = (I - 10) % 6, uu = k / 3, dt = k % 3, isq = I >= 16;
      lds_tr_asm<(isq ? 0 : DOFF) + (SUB * 2 + uu) * 16 * ROWB + dt * 64>(d, cb.t0, cb.t1);
    }
  };

  // ---- prologue: tiles 0, 1, 2 in flight; S'(0); look-ahead fragments 0, 1 of step 0
  issue(lds0);
  issue(lds0 + STAGE_B);
  issue(lds0 + 2 * STAGE_B);
  // (the stationary rows are fetched BEHIND the first tiles' DMA: the two latencies overlap)
  // stationary operands: K / V rows of this wave's 2 x 32 keys (B operands: lane = key), -1.0 in k-slots 72..74 against the statistics rows
  int kv[2];
  bf16x8 kf[2][KSTEPS], vf[2][KSTEPS];
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    kv[kb] = bx * 256 + wave * 64 + kb * 32 + (lane & 31);
    const bool kvok = kv[kb] < p.Nk;                               // Nk % 64 == 0: a key block is whole or absent (its waves then carry zeros and store nothing)
    load_row_frags(kf[kb], p.K + kbase + (long)kv[kb] * p.k_ts + (long)h * p.k_hs, kvok, hi);
    load_row_frags(vf[kb], p.V + vbase + (long)kv[kb] * p.v_ts + (long)h * p.v_hs, kvok, hi);
    settle(kf[kb]);
    settle(vf[kb]);
    if (hi == 1) {
      u32x4 w = __builtin_bit_cast(u32x4, kf[kb][KSTEPS - 1]);
      w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
      kf[kb][KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
      w = __builtin_bit_cast(u32x4, vf[kb][KSTEPS - 1]);
      w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
      vf[kb][KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) to_agpr(vf[kb][ks]);
  }
  lds_dma_wait<14>();
  __syncthreads();
  {
    const Bases cb = bases(lds0);
    static_for<5>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (ks < KSTEPS - 1) lds_row_asm<(ks >> 1) * 64>(f[ks & 3], (ks & 1) ? cb.r1 : cb.r0);
      else lds_row_asm<0>(f[0], cb.r40);
      if constexpr (ks == 3) { lds_wait<0>(f[0]); }                // (quad 0 is reused by k-step 4: settle k-step 0 first)
      if constexpr (ks == 3) { mfma32_vv_first(S[0][0], f[0], kf[0][0]); mfma32_vv_first(S[0][1], f[0], kf[1][0]); }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    mfma32_vv(S[0][0], f[1], kf[0][1]); mfma32_vv(S[0][1], f[1], kf[1][1]);
    mfma32_vv(S[0][0], f[2], kf[0][2]); mfma32_vv(S[0][1], f[2], kf[1][2]);
    mfma32_vv(S[0][0], f[3], kf[0][3]); mfma32_vv(S[0][1], f[3], kf[1][3]);
    mfma32_vv(S[0][0], f[0], kf[0][4]); mfma32_vv(S[0][1], f[0], kf[1][4]);
    mfma_drain();
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int g = 0; g < 16; g++) if (!PRE) S[0][kb][g] *= c;     // (inside the loop the next step's scores are scaled under the dK MFMAs)
    static_for<4>([&](auto ic) { rd_frag(IntC<0>{}, ic, f[decltype(ic)::value], cb, cb); });   // step 0 (tile 0, sub 0): fragments 0..3 = Q rows of (0, sub 1)
  }

  // ---- one step = one 32-query sub-tile.  Quad of fragment i = (i + 2 SUB) & 3; fragments are consumed in pairs behind ONE counted wait (the next
  // pair, already issued, may stay in flight) and re-read four fragments ahead, right behind the second MFMA of the quad's previous owner.
  // Vector work per MFMA gap gi = 2 i + kb: E(j) (P = exp2(c S'), cvt_pk) in gaps 0..19, M(j) (dS = P dP', cvt_pk) in gaps 20..31; the cvt_pk of a pair
  // lags one gap behind its second exp2 / multiply (a transcendental's consumer otherwise draws a wait state: 53 s_nop per tile in the first build).
  int t = 0;
  auto step = [&](auto subc, const Bases& cb, const Bases& nb, unsigned fst) {      // fst: LDS address of the stage tile t+3 is fetched into
    constexpr int SUB = decltype(subc)::value, CUR = SUB, NXT = SUB ^ 1;
    // Stages (a producer and its consumer are always at least one MFMA apart - an exp2's consumer, and the exp2 itself behind the multiply that feeds
    // it, otherwise draw wait states: 53 / 100 s_nop per tile in the first two builds):
    //   gaps  0..19  exp2 of S'(j) c (scaled one step earlier) -> P, in place;  cvt_pk of the P pairs completed two gaps before
    //   gaps 20..31  dS = P dP', in place in dP;                                cvt_pk of the dS pairs completed two gaps before (.. gap 33)
    //   gaps 32..43  S'(j+1) *= c  (its first product finished in gap 9)
    auto pack_p = [&](auto gc) {
      constexpr int g0 = decltype(gc)::value;
      if constexpr (g0 >= 0 && g0 < 20) {
        constexpr int p0 = (32 * g0) / 20, p1 = (32 * (g0 + 1)) / 20;
        static_for<p1 - p0>([&](auto ec) {
          constexpr int e = p0 + decltype(ec)::value, kb = e >> 4, g = e & 15;
          if constexpr (g & 1) {
            unsigned w = (DKV4_ABL & 2) ? __builtin_bit_cast(unsigned, S[CUR][kb][g]) : pack_bf16x2(S[CUR][kb][g - 1], S[CUR][kb][g]);
            asm volatile("" : "+v"(w));
            u32x4 ww = __builtin_bit_cast(u32x4, pb[kb][g >> 3]);
            ww[(g & 7) >> 1] = w;
            pb[kb][g >> 3] = __builtin_bit_cast(bf16x8, ww);
          }
        });
      }
    };
    auto pack_ds = [&](auto gc) {
      constexpr int g0 = decltype(gc)::value;
      if constexpr (g0 >= 20 && g0 < 32) {
        constexpr int p0 = (32 * (g0 - 20)) / 12, p1 = (32 * (g0 - 19)) / 12;
        static_for<p1 - p0>([&](auto ec) {
          constexpr int e = p0 + decltype(ec)::value, kb = e >> 4, g = e & 15;
          if constexpr (g & 1) {
            unsigned w = (DKV4_ABL & 2) ? __builtin_bit_cast(unsigned, dP[kb][g]) : pack_bf16x2(dP[kb][g - 1], dP[kb][g]);
            asm volatile("" : "+v"(w));
            u32x4 ww = __builtin_bit_cast(u32x4, db[kb][g >> 3]);
            ww[(g & 7) >> 1] = w;
            db[kb][g >> 3] = __builtin_bit_cast(bf16x8, ww);
          }
        });
      }
    };
    auto valu = [&](auto gic) {
      constexpr int gi = decltype(gic)::value;
      pack_p(IntC<gi - 2>{});
      pack_ds(IntC<gi - 2>{});
      if constexpr (gi < 20) {
        constexpr int e0 = (32 * gi) / 20, e1 = (32 * (gi + 1)) / 20;
        static_for<e1 - e0>([&](auto ec) {
          constexpr int e = e0 + decltype(ec)::value;
          if (!(DKV4_ABL & 1)) S[CUR][e >> 4][e & 15] = __builtin_amdgcn_exp2f(S[CUR][e >> 4][e & 15]);
          asm volatile("" : "+v"(S[CUR][e >> 4][e & 15]));
        });
      } else if constexpr (gi < 32) {
        constexpr int e0 = (32 * (gi - 20)) / 12, e1 = (32 * (gi - 19)) / 12;
        static_for<e1 - e0>([&](auto ec) {
          constexpr int e = e0 + decltype(ec)::value;
          if (!(DKV4_ABL & 4)) dP[e >> 4][e & 15] *= S[CUR][e >> 4][e & 15];
          asm volatile("" : "+v"(dP[e >> 4][e & 15]));
        });
      } else {
        constexpr int e0 = (32 * (gi - 32)) / 12, e1 = (32 * (gi - 31)) / 12;
        static_for<e1 - e0>([&](auto ec) {
          constexpr int e = e0 + decltype(ec)::value;
          if (!PRE && !(DKV4_ABL & 4)) S[NXT][e >> 4][e & 15] *= c;     // (a plain `if` on the template constant: `if constexpr` here loses the lambda's capture of S)
          asm volatile("" : "+v"(S[NXT][e >> 4][e & 15]));
        });
      }
    };
    auto mma = [&](auto ic, auto kbc, auto wc) {                   // MFMA of fragment i for key block kb; W >= 0: behind the counted wait
      constexpr int i = decltype(ic)::value, kb = decltype(kbc)::value, W = decltype(wc)::value, q = (i + 2 * SUB) & 3;
      if constexpr (DKV4_ABL & 32) { if constexpr (W >= 0) lds_wait<(W >= 0 ? W : 0)>(f[q]); return; }
      if constexpr (i < 5) { if constexpr (i == 0) mfma32_vv_first<W>(S[NXT][kb], f[q], kf[kb][0]); else mfma32_vv<W>(S[NXT][kb], f[q], kf[kb][i]); }
      else if constexpr (i < 10) { if constexpr (i == 5) mfma32_va_first<W>(dP[kb], f[q], vf[kb][0]); else mfma32_va<W>(dP[kb], f[q], vf[kb][i - 5]); }
      else if constexpr (i < 16) mfma32_acc<W>(dv[kb][(i - 10) % 3], f[q], pb[kb][(i - 10) / 3]);
      else mfma32_acc<W>(dk[kb][(i - 16) % 3], f[q], db[kb][(i - 16) / 3]);
    };
    static_for<11>([&](auto kc) {
      constexpr int i = 2 * decltype(kc)::value;
      if constexpr (SUB == 0 && i == 18) {                         // the tile's barrier, in front of the first look-ahead read into tile t+1: tile t+1 has
        lds_dma_wait<7>();                                         // landed (this wave's pieces; t+2 may stay in flight) and is visible; every wave is past
        __syncthreads();                                           // tile t-1, whose stage takes tile t+3 (fetched in four parts: here, behind the step's last
        if constexpr (!(DKV4_ABL & 16)) issue_part(IntC<0>{}, fst);   // fragment pair, and behind the first two pairs of the next step)
      }
      if constexpr (SUB == 0 && i == 20 && !(DKV4_ABL & 16)) issue_part(IntC<1>{}, fst);
      if constexpr (SUB == 1 && i == 0 && !(DKV4_ABL & 16)) issue_part(IntC<2>{}, fst);
      if constexpr (SUB == 1 && i == 2 && !(DKV4_ABL & 16)) { issue_part(IntC<3>{}, fst); advance(); }
      // (each MFMA alone in its scheduling region: the gap's vector work then sits BEHIND it, never adjacent to the previous gap's)
      mma(IntC<i>{}, IntC<0>{}, IntC<dkv4_nreads(i + 2) + dkv4_nreads(i + 3)>{});
      __builtin_amdgcn_sched_barrier(0);
      valu(IntC<2 * i>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(IntC<i>{}, IntC<1>{}, IntC<-1>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(DKV4_ABL & 8)) rd_frag(subc, IntC<i + 4>{}, f[(i + 2 * SUB) & 3], cb, nb);
      valu(IntC<2 * i + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(IntC<i + 1>{}, IntC<0>{}, IntC<-1>{});
      __builtin_amdgcn_sched_barrier(0);
      valu(IntC<2 * i + 2>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(IntC<i + 1>{}, IntC<1>{}, IntC<-1>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(DKV4_ABL & 8)) rd_frag(subc, IntC<i + 5>{}, f[(i + 1 + 2 * SUB) & 3], cb, nb);
      valu(IntC<2 * i + 3>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // stage rotation without per-tile multiplies: cur / nx / (the stage of tile t+3 = the one of tile t-1) walk the ring by additions
  unsigned cur = lds0, nx = lds0 + STAGE_B, fst = lds0 + 3 * STAGE_B;
  Bases cb = bases(cur);
  for (t = 0; t < T; t++) {
    const Bases nb = bases(nx);
    step(IntC<0>{}, cb, nb, fst);
    step(IntC<1>{}, cb, nb, fst);
    cb = nb;
    fst = cur;
    cur = nx;
    nx = nx + STAGE_B == lds0 + DKV4_STAGES * STAGE_B ? lds0 : nx + STAGE_B;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
  lds_dma_wait<0>();                                               // the clamped re-fetches must not land in a later workgroup's LDS
  mfma_drain();
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    const bool kvok = kv[kb] < p.Nk;
    if (kvok) {
      store_rows(p.dK + dkbase + (long)kv[kb] * p.dk_ts + (long)h * p.dk_hs, dk[kb], p.dk_scale, hi);
      store_rows(p.dV + dvbase + (long)kv[kb] * p.dv_ts + (long)h * p.dv_hs, dv[kb], 1.f, hi);
    }
    if (p.dk_colsum) colsum_rows(p.dk_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dk[kb], p.dk_scale, kvok, hi, lane);
    if (p.dv_colsum) colsum_rows(p.dv_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dv[kb], 1.f, kvok, hi, lane);
  }
}

// attn_bwd_dkv4_kernel with the SECOND products (dV^T = dO^T P, dK^T = Q^T dS: head_dim as output rows) on v_mfma_f32_16x16x32: 72 rows pad to 80 (5 tiles of 16)
// instead of 96 (3 of 32) - 40 MFMAs of 16 cycles per sub-tile instead of 24 of 32 (640 against 768 matrix-pipe cycles; the whole step 1280 against 1408), in
// the shape the power limit favours (common.h mfma16).  Round 2 measured this 3-7 % SLOWER in the two-wave kernel - issue-bound: the 16-cycle MFMAs left the
// partner wave's softmax too few slots; the one-wave kernel has them (ablation r4_17: its vector and LDS work fit with room).  P and dS take pack_xy's
// lane exchange (4 v_permlane16_swap per 32 x 32 block), the A operands are trfrag16 reads; dK^T / dV^T leave through store_rows16 (PXA_ATTN_DKV=5).
// Alone 2.5 % faster than attn_bwd_dkv4_kernel, inside the training step 2.4 ms per step slower (profiles/r4_34_step_ab_attention.txt): an A/B partner, not the default.
template <bool PRE>     // PRE: q arrives as (scale log2 e) x queries (pxa_attn_args.q_prescaled): S needs no multiply in front of exp2
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv5_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[DKV4_STAGES * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const long kbase = (long)b * p.k_bs, vbase = (long)b * p.v_bs, dkbase = (long)b * p.dk_bs, dvbase = (long)b * p.dv_bs;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const bf16_t* Qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs;
  const bf16_t* Dp = p.dO + (long)b * p.o_bs + (long)h * p.o_hs;
  const bf16_t* Ls = p.stats + ((long)b * p.H + h) * p.Nq64 * 8;
  const bf16_t* Ds = Ls + (long)p.B * p.H * p.Nq64 * 8;
  const int qts = (int)p.q_ts, ots = (int)p.o_ts;
  const int T = p.Nq / BKV;                                        // full 64-query tiles (checked by the launcher)
  const float c = p.scale_log2;

  // LDS-DMA plan (saddr form: wave-uniform tile base + per-lane byte offset; Q and dO piece i share their lane mask)
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  unsigned offQ[NDMA], offD[NDMA];
  unsigned long long dmask[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; i++) {
    offQ[i] = (unsigned)(pl.row[i] * qts + pl.coff[i]) * 2u;
    offD[i] = (unsigned)(pl.row[i] * ots + pl.coff[i]) * 2u;
    dmask[i] = __builtin_amdgcn_ballot_w64(pl.coff[i] >= 0);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned wbase = __builtin_amdgcn_readfirstlane(wave * 1024);          // (stage addresses are added per fetch)
  const long qstep = (long)BKV * qts, ostep = (long)BKV * ots;
  const unsigned stat_off = (unsigned)lane * 16u;                  // statistics rows: 64 x 16 B per tile, one piece; waves 0 / 2 fetch L, waves 1 / 3 D
  const bf16_t* statp = (wave & 1) ? Ds : Ls;
  const unsigned stat_dst = __builtin_amdgcn_readfirstlane((wave & 1) ? 2 * TILE_B + STAT_B : TILE_B);
  // tile fetch, in four parts (three {Q, dO} piece pairs + the statistics piece) so that the loop can spread them over MFMA gaps; the source pointers
  // are running ones (qnext / dnext / snext: the next tile to fetch, clamped to the last one - past it a harmless re-fetch keeps every wave's piece
  // count, and with it the counted vmcnt, uniform)
  const bf16_t* qnext = Qp;
  const bf16_t* dnext = Dp;
  const bf16_t* snext = statp;
  auto issue_part = [&](auto pc, unsigned sb) {                    // sb = LDS byte address of the stage
    constexpr int P = decltype(pc)::value;
    const unsigned wb = wbase + sb, so = stat_off, sd = stat_dst + sb;
    const bf16_t* sn = snext;
    if constexpr (P == 0) dma_pair<0, TILE_B + STAT_B>(dmask[0], wb, offQ[0], qnext, offD[0], dnext);
    if constexpr (P == 1) dma_pair<4096, TILE_B + STAT_B + 4096>(dmask[1], wb, offQ[1], qnext, offD[1], dnext);
    if constexpr (P == 2) dma_pair<8192, TILE_B + STAT_B + 8192>(dmask[2], wb, offQ[2], qnext, offD[2], dnext);
    if constexpr (P == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(sd), "v"(so), "s"(sn) : "memory");
  };
  int tfetch = 0;                                                  // tile index behind qnext / dnext / snext
  auto advance = [&]() {
    const bool more = tfetch + 1 < T;
    qnext += more ? qstep : 0; dnext += more ? ostep : 0; snext += more ? (long)BKV * 8 : 0;
    tfetch++;
  };
  auto issue = [&](unsigned sb) { issue_part(IntC<0>{}, sb); issue_part(IntC<1>{}, sb); issue_part(IntC<2>{}, sb); issue_part(IntC<3>{}, sb); advance(); };

  // fragment addressing inside a stage: per-lane bases + instruction immediates (see attn_bwd_dkv2_kernel); `cur` = stage of tile t, `nxt` = of tile t+1
  FragAddr fa;
  frag_addr(fa, lane);
  int r4[2];
#pragma unroll
  for (int sub = 0; sub < 2; sub++) r4[sub] = hi ? TILE_B + (sub * 32 + (lane & 31)) * 16 : fa.rb[0] + 2 * 64 + sub * 32 * ROWB;
  Tr16Addr ta;
  tr16_addr(ta, lane);
  struct Bases { unsigned r0, r1, r40, r41, t00, t01, t10, t11; };
  auto bases = [&](unsigned st) -> Bases { return Bases{st + (unsigned)fa.rb[0], st + (unsigned)fa.rb[1], st + (unsigned)r4[0], st + (unsigned)r4[1],
                                                        st + (unsigned)ta.tb[0][0], st + (unsigned)ta.tb[0][1], st + (unsigned)ta.tb[1][0], st + (unsigned)ta.tb[1][1]}; };
  constexpr int DOFF = TILE_B + STAT_B;

  for (int st = 0; st < DKV4_STAGES; st++) {
    init_pads(smem + st * STAGE_B, 0, tid);
    init_pads(smem + st * STAGE_B + DOFF, 0, tid);
  }
  Acc16 dk[2], dv[2];                                              // [kb]: dK^T / dV^T in 16-row tiles: lane (R, c): d = 16 t + 4 R + g, key c (half 0) / 16 + c (half 1)
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    zero16(dk[kb]); zero16(dv[kb]);
#pragma unroll
    for (int tt = 0; tt < NT16; tt++) { to_agpr(dk[kb].v[tt][0]); to_agpr(dk[kb].v[tt][1]); to_agpr(dv[kb].v[tt][0]); to_agpr(dv[kb].v[tt][1]); }
  }
  f32x16 S[2][2], dP[2];                                           // S[buffer][kb] (S'(j) in buffer j & 1; E(j) leaves P there), dP[kb]
  u32x4 pxu[2], pyu[2], dxu[2], dyu[2];                            // [kb]: P / dS of the sub-tile's 32 queries, packed and lane-exchanged (pack_xy): keys 0-15 / 16-31 of the block
  bf16x8 f[4];                                                     // fragment quads

  // fragment i of step (SUB): 0..4 Q rows of the next sub-tile (k-step i), 5..9 dO rows, 10..14 dO^T (16-row tile t), 15..19 Q^T (tile t); i >= 20: the
  // next step's fragments (look-ahead).  cb = this tile's stage, nb = the next tile's.
  auto rd_frag = [&](auto subc, auto ic, bf16x8& d, const Bases& cb, const Bases& nb) {
    constexpr int SUB = decltype(subc)::value, I = decltype(ic)::value;
    if constexpr (I >= 20) {                                       // next step: its fragments 0..3 are looked ahead (Q rows, k-steps 0..3)
      constexpr int ks = I - 20;
      static_assert(ks < KSTEPS - 1, "look-ahead reaches the statistics fragment");
      // next step = (SUB ^ 1): its S block reads the sub-tile after it: SUB == 0 -> next step is sub 1 of this tile, reads (t+1, sub 0); SUB == 1 -> next
      // step is sub 0 of tile t+1, reads (t+1, sub 1)
      lds_row_asm<(SUB ? 32 * ROWB : 0) + (ks >> 1) * 64>(d, (ks & 1) ? nb.r1 : nb.r0);
    } else if constexpr (I < 5) {
      constexpr int ks = I;
      if constexpr (SUB == 0) {                                    // (t, sub 1)
        if constexpr (ks < KSTEPS - 1) lds_row_asm<32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? cb.r1 : cb.r0);
        else lds_row_asm<0>(d, cb.r41);
      } else {                                                     // (t+1, sub 0)
        if constexpr (ks < KSTEPS - 1) lds_row_asm<(ks >> 1) * 64>(d, (ks & 1) ? nb.r1 : nb.r0);
        else lds_row_asm<0>(d, nb.r40);
      }
    } else if constexpr (I < 10) {
      constexpr int ks = I - 5;
      if constexpr (ks < KSTEPS - 1) lds_row_asm<DOFF + SUB * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? cb.r1 : cb.r0);
      else lds_row_asm<DOFF>(d, SUB ? cb.r41 : cb.r40);
    } else {
      constexpr int tt = (I - 10) % 5, isq = I >= 15;               // trfrag16: rows of the sub-tile's 32 queries, 16 head dims
      lds_tr_asm<(isq ? 0 : DOFF) + SUB * 32 * ROWB + (tt >> 1) * 64>(d, (tt & 1) ? cb.t01 : cb.t00, (tt & 1) ? cb.t11 : cb.t10);
    }
  };

  // ---- prologue: tiles 0, 1, 2 in flight; S'(0); look-ahead fragments 0, 1 of step 0
  issue(lds0);
  issue(lds0 + STAGE_B);
  issue(lds0 + 2 * STAGE_B);
  // (the stationary rows are fetched BEHIND the first tiles' DMA: the two latencies overlap)
  // stationary operands: K / V rows of this wave's 2 x 32 keys (B operands: lane = key), -1.0 in k-slots 72..74 against the statistics rows
  int kv[2];
  bf16x8 kf[2][KSTEPS], vf[2][KSTEPS];
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    kv[kb] = bx * 256 + wave * 64 + kb * 32 + (lane & 31);
    const bool kvok = kv[kb] < p.Nk;                               // Nk % 64 == 0: a key block is whole or absent (its waves then carry zeros and store nothing)
    load_row_frags(kf[kb], p.K + kbase + (long)kv[kb] * p.k_ts + (long)h * p.k_hs, kvok, hi);
    load_row_frags(vf[kb], p.V + vbase + (long)kv[kb] * p.v_ts + (long)h * p.v_hs, kvok, hi);
    settle(kf[kb]);
    settle(vf[kb]);
    if (hi == 1) {
      u32x4 w = __builtin_bit_cast(u32x4, kf[kb][KSTEPS - 1]);
      w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
      kf[kb][KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
      w = __builtin_bit_cast(u32x4, vf[kb][KSTEPS - 1]);
      w[0] = PXA_OPERAND_MINUS_ONE_X2; w[1] = PXA_OPERAND_MINUS_ONE_X1;
      vf[kb][KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) to_agpr(vf[kb][ks]);
  }
  lds_dma_wait<14>();
  __syncthreads();
  {
    const Bases cb = bases(lds0);
    static_for<5>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      if constexpr (ks < KSTEPS - 1) lds_row_asm<(ks >> 1) * 64>(f[ks & 3], (ks & 1) ? cb.r1 : cb.r0);
      else lds_row_asm<0>(f[0], cb.r40);
      if constexpr (ks == 3) { lds_wait<0>(f[0]); }                // (quad 0 is reused by k-step 4: settle k-step 0 first)
      if constexpr (ks == 3) { mfma32_vv_first(S[0][0], f[0], kf[0][0]); mfma32_vv_first(S[0][1], f[0], kf[1][0]); }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    mfma32_vv(S[0][0], f[1], kf[0][1]); mfma32_vv(S[0][1], f[1], kf[1][1]);
    mfma32_vv(S[0][0], f[2], kf[0][2]); mfma32_vv(S[0][1], f[2], kf[1][2]);
    mfma32_vv(S[0][0], f[3], kf[0][3]); mfma32_vv(S[0][1], f[3], kf[1][3]);
    mfma32_vv(S[0][0], f[0], kf[0][4]); mfma32_vv(S[0][1], f[0], kf[1][4]);
    mfma_drain();
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int g = 0; g < 16; g++) if (!PRE) S[0][kb][g] *= c;     // (inside the loop the next step's scores are scaled under the dK MFMAs)
    static_for<4>([&](auto ic) { rd_frag(IntC<0>{}, ic, f[decltype(ic)::value], cb, cb); });   // step 0 (tile 0, sub 0): fragments 0..3 = Q rows of (0, sub 1)
  }

  // ---- one step = one 32-query sub-tile: 20 fragments (quad of fragment i = i & 3), consumed in pairs behind ONE counted wait, re-read four ahead.
  // MFMA gaps: 0..19 S(j+1) / dP(j) (v_mfma_f32_32x32x16, two per row fragment: key blocks 0, 1), 20..39 dV(j), 40..59 dK(j) (v_mfma_f32_16x16x32, four per
  // transposed fragment: {kb 0, kb 1} x {keys 0-15, keys 16-31}).  Vector work, a producer always at least one MFMA in front of its consumer:
  //   gaps  0..15  exp2 of S'(j) c (2 scores per gap) -> P;  cvt_pk two gaps behind;  the four lane swaps of key block 0 in gaps 13 / 14, of 1 in 18 / 19
  //   gaps 20..35  dS = P dP' (2 per gap);  cvt_pk two gaps behind (.. 37);  swaps of key block 0 in gaps 30 / 31, of key block 1 in 38 / 39
  //   gaps 40..59  S'(j+1) *= c
  int t = 0;
  auto step = [&](auto subc, const Bases& cb, const Bases& nb, unsigned fst) {      // fst: LDS address of the stage tile t+3 is fetched into
    constexpr int SUB = decltype(subc)::value, CUR = SUB, NXT = SUB ^ 1;
    // pair pr = 0..15 of a 32-score block set (kb = pr >> 3, scores g = 2 (pr & 7), + 1) -> word of ua (-> x) or ub (-> y), pack_xy's layout
    auto cvt_pair = [&](auto prc, f32x16 (&src)[2], u32x4 (&xu)[2], u32x4 (&yu)[2]) {
      constexpr int pr = decltype(prc)::value;
      if constexpr (pr >= 0 && pr < 16) {
        constexpr int kb = pr >> 3, g = 2 * (pr & 7), w = (g >> 3) * 2 + ((g & 3) >> 1);
        constexpr bool isb = (g & 4) != 0;
        unsigned v = (DKV4_ABL & 2) ? __builtin_bit_cast(unsigned, src[kb][g]) : pack_bf16x2(src[kb][g], src[kb][g + 1]);
        asm volatile("" : "+v"(v));
        if constexpr (isb) yu[kb][w] = v; else xu[kb][w] = v;
      }
    };
    auto swap2 = [&](auto kbc, auto w0c, u32x4 (&xu)[2], u32x4 (&yu)[2]) {
      constexpr int kb = decltype(kbc)::value, w0 = decltype(w0c)::value;
      static_for<2>([&](auto wc) {
        constexpr int w = w0 + decltype(wc)::value;
        if (!(DKV4_ABL & 2)) { const auto r = __builtin_amdgcn_permlane16_swap(xu[kb][w], yu[kb][w], false, false); xu[kb][w] = r[0]; yu[kb][w] = r[1]; }
        asm volatile("" : "+v"(xu[kb][w]), "+v"(yu[kb][w]));
      });
    };
    auto valu = [&](auto gic) {
      constexpr int gi = decltype(gic)::value;
      if constexpr (gi < 20) {
        if constexpr (gi < 16) static_for<2>([&](auto ec) {
          constexpr int e = 2 * gi + decltype(ec)::value;
          if (!(DKV4_ABL & 1)) S[CUR][e >> 4][e & 15] = __builtin_amdgcn_exp2f(S[CUR][e >> 4][e & 15]);
          asm volatile("" : "+v"(S[CUR][e >> 4][e & 15]));
        });
        cvt_pair(IntC<gi - 2>{}, S[CUR], pxu, pyu);
        if constexpr (gi == 13) swap2(IntC<0>{}, IntC<0>{}, pxu, pyu);
        if constexpr (gi == 14) swap2(IntC<0>{}, IntC<2>{}, pxu, pyu);
        if constexpr (gi == 18) swap2(IntC<1>{}, IntC<0>{}, pxu, pyu);
        if constexpr (gi == 19) swap2(IntC<1>{}, IntC<2>{}, pxu, pyu);
      } else if constexpr (gi < 40) {
        if constexpr (gi < 36) static_for<2>([&](auto ec) {
          constexpr int e = 2 * (gi - 20) + decltype(ec)::value;
          if (!(DKV4_ABL & 4)) dP[e >> 4][e & 15] *= S[CUR][e >> 4][e & 15];
          asm volatile("" : "+v"(dP[e >> 4][e & 15]));
        });
        cvt_pair(IntC<gi - 22>{}, dP, dxu, dyu);
        if constexpr (gi == 30) swap2(IntC<0>{}, IntC<0>{}, dxu, dyu);
        if constexpr (gi == 31) swap2(IntC<0>{}, IntC<2>{}, dxu, dyu);
        if constexpr (gi == 38) swap2(IntC<1>{}, IntC<0>{}, dxu, dyu);
        if constexpr (gi == 39) swap2(IntC<1>{}, IntC<2>{}, dxu, dyu);
      } else if constexpr (gi < 56) {
        static_for<2>([&](auto ec) {
          constexpr int e = 2 * (gi - 40) + decltype(ec)::value;
          if (!PRE && !(DKV4_ABL & 4)) S[NXT][e >> 4][e & 15] *= c;     // (a plain `if` on the template constant: `if constexpr` here loses the lambda's capture of S)
          asm volatile("" : "+v"(S[NXT][e >> 4][e & 15]));
        });
      }
    };
    // MFMA m of fragment i (row fragments: m = kb; transposed ones: m = 2 kb + half); W >= 0: behind the counted wait
    auto mma = [&](auto ic, auto mc, auto wc) {
      constexpr int i = decltype(ic)::value, m = decltype(mc)::value, W = decltype(wc)::value, q = i & 3;
      if constexpr (DKV4_ABL & 32) { if constexpr (W >= 0) lds_wait<(W >= 0 ? W : 0)>(f[q]); return; }
      if constexpr (i < 5) { if constexpr (i == 0) mfma32_vv_first<W>(S[NXT][m], f[q], kf[m][0]); else mfma32_vv<W>(S[NXT][m], f[q], kf[m][i]); }
      else if constexpr (i < 10) { if constexpr (i == 5) mfma32_va_first<W>(dP[m], f[q], vf[m][0]); else mfma32_va<W>(dP[m], f[q], vf[m][i - 5]); }
      else {
        constexpr int kb = m >> 1, half = m & 1, tt = (i - 10) % 5;
        if constexpr (W >= 0) lds_wait<(W >= 0 ? W : 0)>(f[q]);
        if constexpr (i < 15) mfma16_acc(dv[kb].v[tt][half], f[q], __builtin_bit_cast(bf16x8, half ? pyu[kb] : pxu[kb]));
        else mfma16_acc(dk[kb].v[tt][half], f[q], __builtin_bit_cast(bf16x8, half ? dyu[kb] : dxu[kb]));
      }
    };
    auto nrd = [](int i) { const int k = ((i % 20) + 20) % 20; return k < 10 ? 1 : 2; };
    static_for<10>([&](auto kc) {
      constexpr int i = 2 * decltype(kc)::value;
      constexpr int G0 = i < 10 ? 2 * i : 20 + 4 * (i - 10);       // gap index of the pair's first MFMA
      if constexpr (SUB == 0 && i == 16) {                         // the tile's barrier, in front of the first look-ahead read into tile t+1
        lds_dma_wait<7>();
        __syncthreads();
        if constexpr (!(DKV4_ABL & 16)) issue_part(IntC<0>{}, fst);
      }
      if constexpr (SUB == 0 && i == 18 && !(DKV4_ABL & 16)) issue_part(IntC<1>{}, fst);
      if constexpr (SUB == 1 && i == 0 && !(DKV4_ABL & 16)) issue_part(IntC<2>{}, fst);
      if constexpr (SUB == 1 && i == 2 && !(DKV4_ABL & 16)) { issue_part(IntC<3>{}, fst); advance(); }
      constexpr int W = nrd(i + 2) + nrd(i + 3);
      if constexpr (i < 10) {                                      // row fragments: two MFMAs each
        mma(IntC<i>{}, IntC<0>{}, IntC<W>{});
        __builtin_amdgcn_sched_barrier(0);
        valu(IntC<G0>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(IntC<i>{}, IntC<1>{}, IntC<-1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DKV4_ABL & 8)) rd_frag(subc, IntC<i + 4>{}, f[i & 3], cb, nb);
        valu(IntC<G0 + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(IntC<i + 1>{}, IntC<0>{}, IntC<-1>{});
        __builtin_amdgcn_sched_barrier(0);
        valu(IntC<G0 + 2>{});
        __builtin_amdgcn_sched_barrier(0);
        mma(IntC<i + 1>{}, IntC<1>{}, IntC<-1>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DKV4_ABL & 8)) rd_frag(subc, IntC<i + 5>{}, f[(i + 1) & 3], cb, nb);
        valu(IntC<G0 + 3>{});
        __builtin_amdgcn_sched_barrier(0);
      } else {                                                     // transposed fragments: four MFMAs each
        static_for<8>([&](auto mc) {
          constexpr int mm = decltype(mc)::value, fi = i + (mm >> 2), m = mm & 3;
          mma(IntC<fi>{}, IntC<m>{}, IntC<(mm == 0 ? W : -1)>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (m == 3 && !(DKV4_ABL & 8)) rd_frag(subc, IntC<fi + 4>{}, f[fi & 3], cb, nb);
          valu(IntC<G0 + mm>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      }
    });
  };
  // stage rotation without per-tile multiplies: cur / nx / (the stage of tile t+3 = the one of tile t-1) walk the ring by additions
  unsigned cur = lds0, nx = lds0 + STAGE_B, fst = lds0 + 3 * STAGE_B;
  Bases cb = bases(cur);
  for (t = 0; t < T; t++) {
    const Bases nb = bases(nx);
    step(IntC<0>{}, cb, nb, fst);
    step(IntC<1>{}, cb, nb, fst);
    cb = nb;
    fst = cur;
    cur = nx;
    nx = nx + STAGE_B == lds0 + DKV4_STAGES * STAGE_B ? lds0 : nx + STAGE_B;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
  lds_dma_wait<0>();                                               // the clamped re-fetches must not land in a later workgroup's LDS
  mfma_drain();
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    const long k0 = (long)bx * 256 + wave * 64 + kb * 32;          // first key of the block: lane (R, c) stores rows k0 + c and k0 + 16 + c
    const bool kvok = k0 < p.Nk;                                   // whole block or none (Nk % 64 == 0)
    store_rows16(p.dK + dkbase + k0 * p.dk_ts + (long)h * p.dk_hs, p.dk_ts, dk[kb], p.dk_scale, p.dk_scale, kvok, kvok, lane);
    store_rows16(p.dV + dvbase + k0 * p.dv_ts + (long)h * p.dv_hs, p.dv_ts, dv[kb], 1.f, 1.f, kvok, kvok, lane);
    if (p.dk_colsum) colsum_rows16(p.dk_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dk[kb], p.dk_scale, kvok, kvok, lane);
    if (p.dv_colsum) colsum_rows16(p.dv_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dv[kb], 1.f, kvok, kvok, lane);
  }
}


// ------------------------------------------------------------------------------------------------ backward: dQ, ONE wave per SIMD (round 4)
// attn_bwd_dkv4_kernel's idea for the query-stationary half: a wave owns 64 queries (two 32-query blocks qb) and the whole register file, every K / V /
// K^T fragment read from LDS feeds both blocks (20 read instructions for 40 MFMAs per 32-key sub-tile; attn_bwd_dq2_kernel: 35), half the waves read the
// same tiles.  Q / dO rows (B operands, 80 registers) and dQ^T (80) live in the accumulator half; S / dP / dS in the arch half.  Products, layouts,
// the delta fold and the 16-row second product are attn_bwd_dq2_kernel's; the pipeline, per 32-key sub-tile j (step):
//   S(j+1)   10 v_mfma_f32_32x32x16   K rows of the NEXT sub-tile x Q^T                  || E(j): P = exp2(c S^T - lse): the fma one gap ahead of its exp2
//   dP(j)    10 v_mfma_f32_32x32x16   V rows x dO^T  (delta rides in slots 72..74)       ||
//   dQ(j-1)  20 v_mfma_f32_16x16x32   K^T (transpose reads) x dS^T(j-1)                   || M(j): dS^T = P dP', cvt_pk, v_permlane16_swap -> the x / y operands of dQ(j)
// S and the packed dS are double-buffered.  Fragments: two sets of five quads, a block's set is loaded in the shadow of the block before it and waited for
// ONCE (lgkmcnt(0): nothing else is in flight at a block boundary) - three waits per step.  Ring: {K tile, V tile} x 5 stages (120 KiB, one workgroup
// per CU); tile t+3 is fetched behind the ONE barrier of tile t (in front of the dQ block of its first step, the first reads of tile t+1 behind it),
// counted vmcnt(6): tile t+2 stays in flight.  Dense keys in whole 64-key tiles; everything else runs attn_bwd_dq2_kernel / the keys-resident kernel.
#ifndef PXA_ATTN_DQ4_DEFAULT
#define PXA_ATTN_DQ4_DEFAULT 1
#endif
constexpr int DQ4_STAGES = 5;
// PRE (round 5): q arrives as (scale log2 e) x queries (pxa_attn_args.q_prescaled), so S^T = K Q~^T is already the exponent's argument up to - lse - and
// that rides in the matrix product too, the way delta rides in dP: the lane's Q~ row carries lse in slots 72 .. 74 (split3) against -1.0 in the K tiles'
// pad columns.  E(j) is then the exp2 alone: one vector instruction per score less (the fma S c - lse) in a kernel whose limit is vector issue.
#ifndef DQ4_FOLD_LSE
#define DQ4_FOLD_LSE 1      // 0 (A/B builds): the prescaled instance keeps the fma in front of exp2
#endif
template <bool PRE_>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq4_kernel(AttnParams p) {
  constexpr bool PRE = PRE_ && DQ4_FOLD_LSE;
  constexpr int STG = 2 * TILE_B;
  __shared__ __attribute__((aligned(16))) char smem[DQ4_STAGES * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5;
  int bx, h, b;
  block_coords(p, bx, h, b);
  const bf16_t* Kp = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
  const bf16_t* Vp = p.V + (long)b * p.v_bs + (long)h * p.v_hs;
  const int kts = (int)p.k_ts, vts = (int)p.v_ts;
  const int T = p.Nk / BKV;                                        // full 64-key tiles (checked by the launcher), >= 1
  const float c = p.scale_log2;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  // LDS-DMA plan (saddr form; K and V piece i share their lane mask); running source pointers, clamped to the last tile
  DmaPlan pl;
  dma_plan(pl, wave, lane);
  unsigned offK[NDMA], offV[NDMA];
  unsigned long long dmask[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; i++) {
    offK[i] = (unsigned)(pl.row[i] * kts + pl.coff[i]) * 2u;
    offV[i] = (unsigned)(pl.row[i] * vts + pl.coff[i]) * 2u;
    dmask[i] = __builtin_amdgcn_ballot_w64(pl.coff[i] >= 0);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const unsigned wbase = __builtin_amdgcn_readfirstlane(wave * 1024);
  const long kstep = (long)BKV * kts, vstep = (long)BKV * vts;
  const bf16_t* knext = Kp;
  const bf16_t* vnext = Vp;
  int tfetch = 0;
  auto issue_part = [&](auto pc, unsigned sb) {
    constexpr int P = decltype(pc)::value;
    dma_pair<P * 4096, TILE_B + P * 4096>(dmask[P], wbase + sb, offK[P], knext, offV[P], vnext);
  };
  auto advance = [&]() {
    const bool more = tfetch + 1 < T;
    knext += more ? kstep : 0; vnext += more ? vstep : 0;
    tfetch++;
  };
  auto issue = [&](unsigned sb) { issue_part(IntC<0>{}, sb); issue_part(IntC<1>{}, sb); issue_part(IntC<2>{}, sb); advance(); };

  FragAddr fa;
  frag_addr(fa, lane);
  Tr16Addr ta;
  tr16_addr(ta, lane);
  struct Bases { unsigned r0, r1, t00, t01, t10, t11; };
  auto bases = [&](unsigned st) -> Bases { return Bases{st + (unsigned)fa.rb[0], st + (unsigned)fa.rb[1], st + (unsigned)ta.tb[0][0], st + (unsigned)ta.tb[0][1],
                                                        st + (unsigned)ta.tb[1][0], st + (unsigned)ta.tb[1][1]}; };
  for (int st = 0; st < 2 * DQ4_STAGES; st++) init_pads(smem + st * TILE_B, ((st & 1) || PRE) ? 2 : 0, tid);   // odd tiles = V: -1.0 in slots 72 .. 74 (PRE: the K tiles too)

  Acc16 dq[2];
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    zero16(dq[qb]);
#pragma unroll
    for (int t = 0; t < NT16; t++) { to_agpr(dq[qb].v[t][0]); to_agpr(dq[qb].v[t][1]); }
  }
  f32x16 S[2][2], DP[2];                                           // S[buffer][qb] (S^T(j) in buffer j & 1; E(j) leaves P there), DP[qb]
  u32x4 dxu[2][2], dyu[2][2];                                      // [buffer][qb]: dS^T(j) packed in buffer j & 1 - the x / y operands of dQ(j), run one step later
  bf16x8 fs[2][5];                                                 // fragment sets
#pragma unroll
  for (int qb = 0; qb < 2; qb++)
#pragma unroll
    for (int w = 0; w < 4; w++) { dxu[1][qb][w] = 0u; dyu[1][qb][w] = 0u; }     // "dS^T(-1)" = 0 for the first step's dQ block

  // fragment loads.  K rows of sub-tile SUBR of the stage behind b: row fragment ks; V rows likewise (+ TILE_B); K^T: output tile tt (16 head dims)
  auto rd_krow = [&](auto subc, auto kc, bf16x8& d, const Bases& bs) {
    constexpr int sub = decltype(subc)::value, ks = decltype(kc)::value;
    lds_row_asm<sub * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? bs.r1 : bs.r0);
  };
  auto rd_vrow = [&](auto subc, auto kc, bf16x8& d, const Bases& bs) {
    constexpr int sub = decltype(subc)::value, ks = decltype(kc)::value;
    lds_row_asm<TILE_B + sub * 32 * ROWB + (ks >> 1) * 64>(d, (ks & 1) ? bs.r1 : bs.r0);
  };
  auto rd_ktr = [&](auto subc, auto tc, bf16x8& d, const Bases& bs) {
    constexpr int sub = decltype(subc)::value, tt = decltype(tc)::value;
    lds_tr_asm<sub * 32 * ROWB + (tt >> 1) * 64>(d, (tt & 1) ? bs.t01 : bs.t00, (tt & 1) ? bs.t11 : bs.t10);
  };
  auto wait_set = [&](bf16x8 (&d)[5]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4])); };

  // ---- prologue: tiles 0, 1, 2 in flight; S^T(0) cold; the first step's K rows (0, sub 1) into set 0
  issue(lds0);
  issue(lds0 + STG);
  issue(lds0 + 2 * STG);
  // (the stationary rows are fetched BEHIND the first tiles' DMA: the two latencies overlap)
  // stationary operands: Q / dO rows of this wave's 2 x 32 queries (B operands: lane = query), delta in slots 72..74 of the dO rows (split3)
  int q[2];
  bool qvalid[2];
  float lse[2];
  bf16x8 qf[2][KSTEPS], dof[2][KSTEPS];
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    q[qb] = bx * 256 + wave * 64 + qb * 32 + (lane & 31);
    qvalid[qb] = q[qb] < p.Nq;
    load_row_frags(qf[qb], p.Q + (long)b * p.q_bs + (long)q[qb] * p.q_ts + (long)h * p.q_hs, qvalid[qb], hi);
    load_row_frags(dof[qb], p.dO + (long)b * p.o_bs + (long)q[qb] * p.o_ts + (long)h * p.o_hs, qvalid[qb], hi);
    settle(qf[qb]);
    settle(dof[qb]);
    const long sidx = ((long)b * p.H + h) * p.Nq + q[qb];
    lse[qb] = qvalid[qb] ? p.LSE[sidx] : 0.f;
    const float delta = qvalid[qb] ? p.Delta[sidx] : 0.f;
    if (hi == 1) {
      u32x4 w = __builtin_bit_cast(u32x4, dof[qb][KSTEPS - 1]);
      const uint2 d3 = split3(delta);
      w[0] = d3.x; w[1] = d3.y;
      dof[qb][KSTEPS - 1] = __builtin_bit_cast(bf16x8, w);
      if constexpr (PRE) {                                         // slots 72 .. 74 of this lane's Q~ row: lse (x -1.0 of the K tiles' pads)
        u32x4 wq = __builtin_bit_cast(u32x4, qf[qb][KSTEPS - 1]);
        const uint2 l3 = split3(lse[qb]);
        wq[0] = l3.x; wq[1] = l3.y;
        qf[qb][KSTEPS - 1] = __builtin_bit_cast(bf16x8, wq);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) { to_agpr(qf[qb][ks]); to_agpr(dof[qb][ks]); }
  }

  lds_dma_wait<12>();
  __syncthreads();
  {
    const Bases cb0 = bases(lds0);
    static_for<5>([&](auto kc) { rd_krow(IntC<0>{}, kc, fs[0][decltype(kc)::value], cb0); });
    wait_set(fs[0]);
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      mfma32_va_first(S[0][qb], fs[0][0], qf[qb][0]);
      mfma32_va(S[0][qb], fs[0][1], qf[qb][1]);
      mfma32_va(S[0][qb], fs[0][2], qf[qb][2]);
      mfma32_va(S[0][qb], fs[0][3], qf[qb][3]);
      mfma32_va(S[0][qb], fs[0][4], qf[qb][4]);
    }
    mfma_drain();
    static_for<5>([&](auto kc) { rd_krow(IntC<1>{}, kc, fs[0][decltype(kc)::value], cb0); });
  }

  // ---- one step = one 32-key sub-tile j (tile t, sub SUB).  Fragment sets: the S block uses set SUB, the dP block set SUB ^ 1, the dQ block set SUB.
  // cb: this tile's stage, nb: the next tile's, pb_: the stage of sub-tile j-1 (this tile for SUB = 1, the previous one for SUB = 0).
  int t = 0;
  auto step = [&](auto subc, const Bases& cb, const Bases& nb, const Bases& pbs, unsigned fst) {
    constexpr int SUB = decltype(subc)::value, CUR = SUB, NXT = SUB ^ 1, SA = SUB, SB = SUB ^ 1;
    // E(j): elements e = 0..31 (qb = e >> 4, g = e & 15): gap ge: fma of the elements of gap ge, exp2 of those of gap ge - 1 (gaps 0..20)
    auto e_fma = [&](auto gc) {
      constexpr int ge = decltype(gc)::value;
      if constexpr (ge >= 0 && ge < 20) {
        constexpr int e0 = (32 * ge) / 20, e1 = (32 * (ge + 1)) / 20;
        static_for<e1 - e0>([&](auto ec) {
          constexpr int e = e0 + decltype(ec)::value, qb = e >> 4, g = e & 15;
          if (!PRE) S[CUR][qb][g] = fmaf(S[CUR][qb][g], c, -lse[qb]);   // (PRE: S' = S~ - lse came out of the matrix product)
          asm volatile("" : "+v"(S[CUR][qb][g]));
        });
      }
    };
    auto e_exp = [&](auto gc) {
      constexpr int ge = decltype(gc)::value;
      if constexpr (ge >= 0 && ge < 20) {
        constexpr int e0 = (32 * ge) / 20, e1 = (32 * (ge + 1)) / 20;
        static_for<e1 - e0>([&](auto ec) {
          constexpr int e = e0 + decltype(ec)::value, qb = e >> 4, g = e & 15;
          S[CUR][qb][g] = __builtin_amdgcn_exp2f(S[CUR][qb][g]);
          asm volatile("" : "+v"(S[CUR][qb][g]));
        });
      }
    };
    // M(j) over the 20 gaps of the dQ block: dS of the element pair gm (two scores) one gap behind, its cvt_pk two gaps behind that; the lane swaps of a
    // query block when its eight words are packed.  pack_xy's layout: ua = rows {0-3, 16-19} + 4 hi, ub = rows {8-11, 24-27} + 4 hi.
    auto m_mul = [&](auto gc) {
      constexpr int gm = decltype(gc)::value;
      if constexpr (gm >= 0 && gm < 16) {
        static_for<2>([&](auto ec) {
          constexpr int e = 2 * gm + decltype(ec)::value, qb = e >> 4, g = e & 15;
          DP[qb][g] *= S[CUR][qb][g];
          asm volatile("" : "+v"(DP[qb][g]));
        });
      }
    };
    auto m_cvt = [&](auto gc) {                                    // the pair (2 gm, 2 gm + 1) -> one word of ua (-> dxu) or ub (-> dyu)
      constexpr int gm = decltype(gc)::value;
      if constexpr (gm >= 0 && gm < 16) {
        constexpr int qb = gm >> 3, pr = gm & 7, g = 2 * pr;       // pairs in score order: g = 0, 2, .., 14
        constexpr bool isb = (g & 4) != 0;                         // g 0-3 -> ua[0..1], 4-7 -> ub[0..1], 8-11 -> ua[2..3], 12-15 -> ub[2..3]
        constexpr int w = (g >> 3) * 2 + ((g & 3) >> 1);
        unsigned v = pack_bf16x2(DP[qb][g], DP[qb][g + 1]);
        asm volatile("" : "+v"(v));
        if constexpr (isb) dyu[CUR][qb][w] = v; else dxu[CUR][qb][w] = v;
      }
    };
    auto m_swap = [&](auto gc) {                                   // gaps 12, 13 (qb 0: its words are packed by gap 10), gap 19 (qb 1: by gap 18)
      constexpr int gm = decltype(gc)::value;
      if constexpr (gm == 12 || gm == 13 || gm == 19) {
        constexpr int qb = gm == 19, w0 = gm == 13 ? 2 : 0, nw = gm == 19 ? 4 : 2;
        static_for<nw>([&](auto wc) {
          constexpr int w = w0 + decltype(wc)::value;
          const auto r = __builtin_amdgcn_permlane16_swap(dxu[CUR][qb][w], dyu[CUR][qb][w], false, false);
          dxu[CUR][qb][w] = r[0]; dyu[CUR][qb][w] = r[1];
          asm volatile("" : "+v"(dxu[CUR][qb][w]), "+v"(dyu[CUR][qb][w]));
        });
      }
    };
    // ---- S block: S^T(j+1) = K rows (set SA) x Q^T; loads the V rows of sub-tile j into set SB
    wait_set(fs[SA]);
    static_for<10>([&](auto gc) {
      constexpr int g = decltype(gc)::value, ks = g >> 1, qb = g & 1;
      if constexpr (ks == 0) mfma32_va_first(S[NXT][qb], fs[SA][0], qf[qb][0]); else mfma32_va(S[NXT][qb], fs[SA][ks], qf[qb][ks]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (qb == 1) rd_vrow(subc, IntC<ks>{}, fs[SB][ks], cb);
      e_fma(gc);                                                   // (the fma first: the exp2 of the previous gap's elements then sits two instructions behind
      e_exp(IntC<g - 1>{});                                        //  its own fma - directly behind the MFMA it drew a wait state, 36 per tile)
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- dP block: dP^T(j) = V rows (set SB) x dO^T; loads the K^T fragments of sub-tile j-1 into set SA
    wait_set(fs[SB]);
    static_for<10>([&](auto gc) {
      constexpr int g = decltype(gc)::value, ks = g >> 1, qb = g & 1;
      if constexpr (ks == 0) mfma32_va_first(DP[qb], fs[SB][0], dof[qb][0]); else mfma32_va(DP[qb], fs[SB][ks], dof[qb][ks]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (qb == 1) rd_ktr(IntC<SUB ^ 1>{}, IntC<ks>{}, fs[SA][ks], pbs);
      e_fma(IntC<10 + g>{});
      e_exp(IntC<10 + g - 1>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- dQ block: dQ^T += K^T (set SA) x dS^T(j-1) (buffer NXT); loads the K rows of sub-tile j+2 into set SB; M(j) in its gaps
    if constexpr (SUB == 0) {                                      // the tile's barrier: tile t+1 has landed (this wave's pieces; t+2 may stay in flight) and is
      lds_dma_wait<6>();                                           // visible; every wave is past the dQ block of tile t-2's last sub-tile: its stage takes t+3
      __syncthreads();
      issue_part(IntC<0>{}, fst);
    }
    wait_set(fs[SA]);
    static_for<20>([&](auto gc) {
      constexpr int g = decltype(gc)::value, tt = g >> 2, qb = (g >> 1) & 1, half = g & 1;
      mfma16_acc(dq[qb].v[tt][half], fs[SA][tt], __builtin_bit_cast(bf16x8, half ? dyu[NXT][qb] : dxu[NXT][qb]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((g & 3) == 3) {                                // K rows of sub-tile j+2: SUB = 0 -> (t+1, sub 0), SUB = 1 -> (t+1, sub 1)
        rd_krow(subc, IntC<tt>{}, fs[SB][tt], nb);
      }
      if constexpr (g == 0) e_exp(IntC<19>{});
      m_swap(gc);                                                  // (dS of pair gm: multiply in gap gm + 1 - the dP MFMAs' results need their distance -,
      m_cvt(IntC<g - 3>{});                                        //  cvt_pk in gap gm + 3)
      m_mul(IntC<g - 1>{});
      if constexpr (SUB == 0 && g == 4) issue_part(IntC<1>{}, fst);
      if constexpr (SUB == 0 && g == 8) { issue_part(IntC<2>{}, fst); advance(); }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // stage rotation by additions: cur (tile t), nx (t+1), pv (t-1; tile 0 itself at t = 0, against dS^T(-1) = 0), fst (the stage of tile t+3 = of t-2)
  const unsigned ring_end = lds0 + DQ4_STAGES * STG;
  unsigned cur = lds0, nx = lds0 + STG, fst = lds0 + 3 * STG;
  Bases cb = bases(cur), pbs = cb;
  for (t = 0; t < T; t++) {
    const Bases nb = bases(nx);
    step(IntC<0>{}, cb, nb, pbs, fst);
    step(IntC<1>{}, cb, nb, cb, fst);
    pbs = cb;
    cb = nb;
    cur = nx;
    nx = nx + STG == ring_end ? lds0 : nx + STG;
    fst = fst + STG == ring_end ? lds0 : fst + STG;
  }
  // drain: dQ^T += K^T x dS^T of the last sub-tile (tile T-1, sub 1; buffer 1); the look-ahead K rows in set 0 are dropped
  {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fs[0][0]), "+v"(fs[0][1]), "+v"(fs[0][2]), "+v"(fs[0][3]), "+v"(fs[0][4]));
    static_for<5>([&](auto tc) { rd_ktr(IntC<1>{}, tc, fs[1][decltype(tc)::value], pbs); });
    wait_set(fs[1]);
    static_for<20>([&](auto gc) {
      constexpr int g = decltype(gc)::value, tt = g >> 2, qb = (g >> 1) & 1, half = g & 1;
      mfma16_acc(dq[qb].v[tt][half], fs[1][tt], __builtin_bit_cast(bf16x8, half ? dyu[1][qb] : dxu[1][qb]));
    });
  }
  lds_dma_wait<0>();                                               // the clamped re-fetches must not land in a later workgroup's LDS
  mfma_drain();
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    const int q0w = bx * 256 + wave * 64 + qb * 32;
    const bool ok0 = q0w + (lane & 15) < p.Nq, ok1 = q0w + 16 + (lane & 15) < p.Nq;
    store_rows16(p.dQ + (long)b * p.dq_bs + (long)q0w * p.dq_ts + (long)h * p.dq_hs, p.dq_ts, dq[qb], p.scale, p.scale, ok0, ok1, lane);
    if (p.dq_colsum) colsum_rows16(p.dq_colsum + (b % PXA_COLSUM_SLOTS) * p.colsum_stride + h * DH, dq[qb], p.scale, ok0, ok1, lane);
  }
}

int fill(AttnParams& p, const pxa_attn_args* a) {
  PXA_CHECK(a, "attn: null args");
  PXA_CHECK(a->head_dim == DH, "attn: head_dim %d unsupported (PixArt XL/2 uses 72)", a->head_dim);
  PXA_CHECK(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk >= 0, "attn: bad shape");
  PXA_CHECK(!a->kv_start == !a->kv_len, "attn: kv_start/kv_len must both be given");
  p.Q = (const bf16_t*)a->q; p.K = (const bf16_t*)a->k; p.V = (const bf16_t*)a->v; p.dO = (const bf16_t*)a->d_o;
  p.O = (bf16_t*)a->o; p.dQ = (bf16_t*)a->dq; p.dK = (bf16_t*)a->dk; p.dV = (bf16_t*)a->dv;
  p.LSE = a->lse; p.Delta = a->delta;
  p.dq_colsum = a->dq_colsum; p.dk_colsum = a->dk_colsum; p.dv_colsum = a->dv_colsum; p.colsum_stride = a->colsum_stride;
  p.q_bs = a->q_bs; p.q_ts = a->q_ts; p.q_hs = a->q_hs;
  p.k_bs = a->k_bs; p.k_ts = a->k_ts; p.k_hs = a->k_hs;
  p.v_bs = a->v_bs; p.v_ts = a->v_ts; p.v_hs = a->v_hs;
  p.o_bs = a->o_bs; p.o_ts = a->o_ts; p.o_hs = a->o_hs;
  p.dq_bs = a->dq_bs; p.dq_ts = a->dq_ts; p.dq_hs = a->dq_hs;
  p.dk_bs = a->dk_bs; p.dk_ts = a->dk_ts; p.dk_hs = a->dk_hs;
  p.dv_bs = a->dv_bs; p.dv_ts = a->dv_ts; p.dv_hs = a->dv_hs;
  p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.kv_start = a->kv_start; p.kv_len = a->kv_len;
  p.scale = a->scale; p.scale_log2 = a->scale * 1.4426950408889634f; p.dk_scale = a->scale;
  if (a->q_prescaled) { p.scale_log2 = 1.0f; p.dk_scale = 0.6931471805599453f; }      // q = (scale log2 e) x queries: see pxa_attn_args.q_prescaled
  p.stats = (const bf16_t*)a->bwd_stats; p.Nq64 = (a->Nq + BKV - 1) / BKV * BKV;
  const long strides[] = {p.q_ts, p.k_ts, p.v_ts, p.o_ts, p.q_hs, p.k_hs, p.v_hs, p.o_hs, p.q_bs, p.k_bs, p.v_bs, p.o_bs};
  for (long s : strides) PXA_CHECK(s % 8 == 0, "attn: strides must be multiples of 8 elements (16-byte rows)");
  return 0;
}
}  // namespace

// The keys-resident kernels need 120 KiB of dynamic LDS: the opt-in is per device (and per function), tried once per device under a lock; where it fails
// the callers fall back to the streaming kernels instead of failing every call.
static bool kvres_lds_ok(const void* fn) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, bool> state;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(mu);
  auto it = state.find({dev, fn});
  if (it != state.end()) return it->second;
  const bool ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, KVRES_TILES * 2 * TILE_B) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  state[{dev, fn}] = ok;
  return ok;
}

extern "C" int pxa_attn_fwd(const pxa_attn_args* a, hipStream_t stream) {
  AttnParams p;
  if (int rc = fill(p, a)) return rc;
  PXA_CHECK(p.Q && p.K && p.V && p.O, "pxa_attn_fwd: null tensor");
  static const bool one_sub = getenv("PXA_ATTN_FWD1") != nullptr;   // A/B: the one-sub-tile kernel
  static const bool no_kvres = getenv("PXA_ATTN_NO_KVRES") != nullptr;   // A/B: cross-attention on the streaming kernel
  const int max_k = a->max_kv_len > 0 ? a->max_kv_len : p.Nk;
  if (!no_kvres && !one_sub && max_k > 0 && max_k <= KVRES_TILES * BKV && p.Nq >= 512) {   // every key of a sample fits one workgroup's LDS: attn_fwd_kvres_kernel
    const int tiles = (max_k + BKV - 1) / BKV, lds = tiles * 2 * TILE_B;
    const int qpb = p.Nq >= 4096 ? 4096 : (p.Nq + 511) / 512 * 512;        // a whole head per workgroup up to 4,096 queries
    if (kvres_lds_ok(reinterpret_cast<const void*>(attn_fwd_kvres_kernel))) {     // else (a part with less LDS): the streaming kernels below
      p.nx = (p.Nq + qpb - 1) / qpb;
      hipLaunchKernelGGL(attn_fwd_kvres_kernel, dim3(p.nx * p.H * p.B), dim3(512), lds, stream, p, qpb, tiles);
      PXA_LAUNCH_CHECK();
      return 0;
    }
  }
  const bool two = !one_sub && p.Nq >= 256;
  p.nx = two ? (p.Nq + 255) / 256 : (p.Nq + 127) / 128;
  PXA_CHECK((long)p.nx * p.H * p.B < (1L << 31), "pxa_attn_fwd: grid too large");
  // one-wave-per-SIMD kernel: dense keys in full 64-key tiles.  PXA_ATTN_FWD4 = 1 forces it wherever it applies (tests), 0 never; by default the
  // builds that fold scale and maximum into the first product use it from 8 key tiles on (its prologue fills a 4-deep ring and its last tile
  // computes one first product too many: not worth it for the 256-token grids)
  const char* f4 = getenv("PXA_ATTN_FWD4");
  const bool f4_ok = two && !p.kv_start && p.Nk >= BKV && p.Nk % BKV == 0;
  if (f4_ok && (f4 ? atoi(f4) != 0 : (PXA_ATTN_FWD4_DEFAULT && p.Nk >= 8 * BKV))) {
    hipLaunchKernelGGL(attn_fwd4_kernel, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
    PXA_LAUNCH_CHECK();
    return 0;
  }
  if (two) hipLaunchKernelGGL(attn_fwd2_kernel, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(attn_fwd_kernel, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
  PXA_LAUNCH_CHECK();
  return 0;
}

#if FWD4_TRACE
extern "C" int pxa_attn_fwd4_trace(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(fwd4_trace_buf), sizeof(fwd4_trace_buf)) == hipSuccess ? 0 : -1; }
#endif
extern "C" long pxa_attn_bwd_stats_bytes(int B, int H, int Nq) { return 2L * B * H * ((Nq + BKV - 1) / BKV * BKV) * 16; }

extern "C" int pxa_attn_bwd(const pxa_attn_args* a, hipStream_t stream) {
  AttnParams p;
  if (int rc = fill(p, a)) return rc;
  PXA_CHECK(p.Q && p.K && p.V && p.O && p.dO && p.LSE && a->delta, "pxa_attn_bwd: null tensor");
  PXA_CHECK((p.dQ || p.dK) && (!p.dK == !p.dV), "pxa_attn_bwd: need dq and/or both of dk, dv (a NULL gradient skips the kernel that produces it)");
  for (long s : {p.dq_ts, p.dk_ts, p.dv_ts, (long)p.dq_hs, (long)p.dk_hs, (long)p.dv_hs, p.dq_bs, p.dk_bs, p.dv_bs})
    PXA_CHECK(s % 4 == 0, "pxa_attn_bwd: gradient strides must be multiples of 4 elements");
  const long total = (long)p.B * p.Nq * p.H;
  // dK/dV kernel: 0 = round-2 kernel (lse / delta on the VALU), 1 = stats rows + three-stage ring, old issue order, 2 = + software pipeline.
  // 1 and 2 need the caller's bwd_stats workspace (pxa_attn_bwd_stats_bytes); without it the round-2 kernel runs.
  const char* env = getenv("PXA_ATTN_DKV");
  int dkv_mode = env ? atoi(env) : PXA_ATTN_DKV_DEFAULT;
  if (!p.stats || !p.dK) dkv_mode = 0;
  bf16_t* stats = dkv_mode ? (bf16_t*)a->bwd_stats : nullptr;
  const float inv_c = 1.0f / p.scale_log2;
  if (stats && p.Nq64 != p.Nq) {
    const int n = p.B * p.H * (p.Nq64 - p.Nq);
    hipLaunchKernelGGL(attn_stats_pad_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, stats, p.B * p.H, p.Nq, p.Nq64);
  }
  const bool no_prepass = getenv("PXA_ATTN_BWD_NO_PREPASS") != nullptr;   // measurement only (bench.py): delta / stats rows of the SAME inputs are still in the workspace
  // cross-attention (every key of a sample fits one workgroup's LDS): the dQ kernel keeps them resident and takes over the delta / statistics pre-pass
  static const bool no_kvres = getenv("PXA_ATTN_NO_KVRES") != nullptr;
  const int max_kr = a->max_kv_len > 0 ? a->max_kv_len : p.Nk;
  const bool dq_kvres = !no_kvres && ATTN_FOLD_DELTA && p.dQ && max_kr > 0 && max_kr <= KVRES_TILES * BKV && p.Nq >= 512 &&
                        kvres_lds_ok(reinterpret_cast<const void*>(attn_bwd_dq_kvres_kernel));
  if (no_prepass || dq_kvres) {
  } else if (p.o_hs == DH && p.o_ts == (long)p.H * DH && p.o_bs == (long)p.Nq * p.o_ts && p.H <= 16 && ((uintptr_t)p.O % 16) == 0 && ((uintptr_t)p.dO % 16) == 0) {
    const long tokens = (long)p.B * p.Nq;                  // token-contiguous rows: the coalesced form
    hipLaunchKernelGGL(attn_delta_rows_kernel, dim3((tokens + DELTA_TOK - 1) / DELTA_TOK), dim3(256), 0, stream, p.O, p.dO, a->delta, p.H, p.Nq, tokens,
                       p.LSE, stats, p.Nq64, inv_c);
  } else {
    hipLaunchKernelGGL(attn_delta_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, p.O, p.dO, a->delta,
                       p.o_bs, p.o_ts, p.o_hs, p.o_bs, p.o_ts, p.o_hs, p.B, p.H, p.Nq, p.LSE, stats, p.Nq64, inv_c);
  }
  PXA_LAUNCH_CHECK();
  if (dq_kvres) {
    const int tiles = (max_kr + BKV - 1) / BKV, lds = tiles * 2 * TILE_B;
    const int qpb = p.Nq >= 4096 ? 4096 : (p.Nq + 255) / 256 * 256;
    p.nx = (p.Nq + qpb - 1) / qpb;
    hipLaunchKernelGGL(attn_bwd_dq_kvres_kernel, dim3(p.nx * p.H * p.B), dim3(512), lds, stream, p, qpb, tiles, a->delta, stats, inv_c);
    PXA_LAUNCH_CHECK();
  } else if (p.dQ) {
    p.nx = (p.Nq + 127) / 128;
    PXA_CHECK((long)p.nx * p.H * p.B < (1L << 31), "pxa_attn_bwd: grid too large");
    const char* dqe = getenv("PXA_ATTN_DQ");                // 0 = round-2 kernel (compiler-scheduled), 1 = hand-placed pipeline (needs the delta fold)
    int dq_mode = ATTN_FOLD_DELTA ? (dqe ? atoi(dqe) : PXA_ATTN_DQ_DEFAULT) : 0;
    // 4 = one wave per SIMD, 64 queries per wave (dense keys in whole 64-key tiles); PXA_ATTN_DQ=4 asks for it, the default takes it where it applies
    const bool dq4_ok = ATTN_FOLD_DELTA && !p.kv_start && p.Nk % BKV == 0 && p.Nk >= 2 * BKV && p.Nq >= 256;
    if (dq_mode == 4 && !dq4_ok) dq_mode = 1;
    if (!dqe && dq_mode == 1 && PXA_ATTN_DQ4_DEFAULT && dq4_ok) dq_mode = 4;
    if (dq_mode == 4) {
      p.nx = (p.Nq + 255) / 256;
      if (a->q_prescaled) hipLaunchKernelGGL(attn_bwd_dq4_kernel<true>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL(attn_bwd_dq4_kernel<false>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
    } else if (dq_mode) hipLaunchKernelGGL(attn_bwd_dq2_kernel, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
    PXA_LAUNCH_CHECK();
  }
  if (p.dK) {
    const int max_k = a->max_kv_len > 0 ? a->max_kv_len : p.Nk;
    // 4 / 5 = one wave per SIMD, 64 keys per wave (dense keys in whole 64-key groups, whole 64-query tiles); PXA_ATTN_DKV=4 / 5 asks for them, the default takes 4
    // where it applies and falls back to 2 elsewhere
    const bool dkv4_ok = p.stats && !p.kv_start && p.Nk % BKV == 0 && p.Nk >= 256 && p.Nq % BKV == 0 && p.Nq >= 2 * BKV;
    if (dkv_mode >= 4 && !dkv4_ok) dkv_mode = 2;
    // default: the one-wave kernel with ALL products on the 32-row shape (4).  Its variant with the second products on 16-row tiles (5) is 2.5 % faster alone
    // (profiles/r4_19_*) and 2-3 ms per step SLOWER inside the training step (profiles/r4_34_step_ab_attention.txt, one box, alternating: 411.6 / 412.2 ms
    // against 415.2 / 413.3) - the step decides; 4 is also bit-identical to the two-wave kernel.
    if (!env && dkv_mode == 2 && PXA_ATTN_DKV4_DEFAULT && dkv4_ok) dkv_mode = 4;
    p.nx = dkv_mode >= 3 ? (max_k + 255) / 256 : (max_k + 127) / 128;
    PXA_CHECK((long)p.nx * p.H * p.B < (1L << 31), "pxa_attn_bwd: grid too large");
    if (p.nx > 0) {
      if (dkv_mode == 5 && a->q_prescaled) hipLaunchKernelGGL(attn_bwd_dkv5_kernel<true>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else if (dkv_mode == 5) hipLaunchKernelGGL(attn_bwd_dkv5_kernel<false>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else if (dkv_mode == 4 && a->q_prescaled) hipLaunchKernelGGL(attn_bwd_dkv4_kernel<true>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else if (dkv_mode == 4) hipLaunchKernelGGL(attn_bwd_dkv4_kernel<false>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else if (dkv_mode == 3) hipLaunchKernelGGL(attn_bwd_dkv3_kernel<2>, dim3(p.nx * p.H * p.B), dim3(512), 0, stream, p);   // prefetch distances 3 / 4 / 6 measured the same
      else if (dkv_mode == 2) hipLaunchKernelGGL(attn_bwd_dkv2_kernel<1>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else if (dkv_mode == 1) hipLaunchKernelGGL(attn_bwd_dkv2_kernel<0>, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(p.nx * p.H * p.B), dim3(256), 0, stream, p);
    }
    PXA_LAUNCH_CHECK();
  }
  return 0;
}
