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
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KC_BYTES = 128 * BK * 2;       // 16384
constexpr int RC_STRIDE = (128 + 32) * 2;    // 320 B per k-row (pad: 4 tr-read rows hit disjoint banks)
constexpr int RC_BYTES = BK * RC_STRIDE;     // 20480

struct GemmParams {
  const bf16_t* A; const bf16_t* B; int lda, ldb;
  int M, N, K;
  const float* bias; const bf16_t* aux; int ldaux;
  bf16_t* out; bf16_t* out2; int ldo;
  float* outf; int ldf;
  int act, accumulate, k_per_split;
};

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

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int A_BYTES = A_KC ? KC_BYTES : RC_BYTES, B_BYTES = B_KC ? KC_BYTES : RC_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = A_BYTES + B_BYTES;  // stage s: [A tile][B tile] at smem + s*STAGE

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, hi = lane >> 5;
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
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

  // ---- epilogue: lane owns row m, 4 consecutive n per accumulator quad
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
        } else if (p.act == 2) {
          const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (size_t)m * p.ldaux + n);
          float a0, a1, a2, a3;
          unpack_bf16x2(a.x, a0, a1); unpack_bf16x2(a.y, a2, a3);
          v[0] *= gelu_tanh_grad(a0); v[1] *= gelu_tanh_grad(a1); v[2] *= gelu_tanh_grad(a2); v[3] *= gelu_tanh_grad(a3);
        }
        if (p.out) *reinterpret_cast<uint2*>(p.out + (size_t)m * p.ldo + n) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        if (p.outf) {
          float* dst = p.outf + (size_t)m * p.ldf + n;
          if (p.accumulate) {
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
int launch(const GemmParams& p, int split, hipStream_t s) {
  constexpr bool A_KC = (LAYOUT != 2), B_KC = (LAYOUT == 0);
  constexpr int LDS = 2 * ((A_KC ? KC_BYTES : RC_BYTES) + (B_KC ? KC_BYTES : RC_BYTES));
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) { pxa_set_error("hipFuncSetAttribute(gemm<%d>, %d): %s", LAYOUT, LDS, hipGetErrorString(e)); return -3; }
    attr_set = true;
  }
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, split);
  hipLaunchKernelGGL(gemm_kernel<LAYOUT>, grid, dim3(256), LDS, s, p);
  PXA_LAUNCH_CHECK();
  return 0;
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
  PXA_CHECK(a->act >= 0 && a->act <= 2, "pxa_gemm: bad act %d", a->act);
  if (a->act == 2) PXA_CHECK(a->aux && a->ldaux % 4 == 0, "pxa_gemm: act=2 needs aux");
  int split = a->split_k < 1 ? 1 : a->split_k;
  if (split > 1) PXA_CHECK(a->out_f32 && a->accumulate && !a->out_bf16 && a->act == 0 && !a->bias, "pxa_gemm: split_k>1 needs fp32 atomic accumulate output only");
  GemmParams p;
  p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.lda = a->lda; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.bias = a->bias; p.aux = (const bf16_t*)a->aux; p.ldaux = a->ldaux;
  p.out = (bf16_t*)a->out_bf16; p.out2 = (bf16_t*)a->out2_bf16; p.ldo = a->ld_out;
  p.outf = a->out_f32; p.ldf = a->ld_f32;
  p.act = a->act; p.accumulate = a->accumulate;
  int kps = ((a->K + split - 1) / split + BK - 1) / BK * BK;
  p.k_per_split = kps;
  split = (a->K + kps - 1) / kps;
  switch (a->layout) {
    case 0: return launch<0>(p, split, stream);
    case 1: return launch<1>(p, split, stream);
    default: return launch<2>(p, split, stream);
  }
}
