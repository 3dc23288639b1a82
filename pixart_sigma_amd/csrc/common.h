// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the PixArt-Sigma denoiser path.
// Lane layouts used here were verified on MI355X by probe/probe.hip (see DESIGN.md "Verified hardware semantics"):
//   v_mfma_f32_32x32x16_bf16:  A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31], j=0..7;
//                              C: col = l&31, row = (g&3) + 8*(g>>2) + 4*(l>>5), g = 0..15
//   ds_read_b64_tr_b16:        inside each 16-lane group, result lane t elem j = (source lane 4j + (t>>2)) elem (t&3)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit operand type of this build.  The library is compiled twice from the same sources: libpixart_hip.so with bf16
// operands (training and the default), libpixart_hip_f16.so (-DPXA_OPERAND_F16) with IEEE fp16 operands - the reference's own
// inference dtype, 8x finer mantissa: the build that meets the 1e-3 forward-parity tolerance.  The type keeps its historical name.
#ifdef PXA_OPERAND_F16
typedef _Float16 bf16_t;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
#define PXA_OPERAND_ONE_BITS 0x3c00u
#define PXA_OPERAND_DTYPE_ID 1
#else
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define PXA_OPERAND_ONE_BITS 0x3f80u
#define PXA_OPERAND_DTYPE_ID 0
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PXA_WAVE 64
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

namespace pxa {

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  bf16x2 v; v[0] = (bf16_t)a; v[1] = (bf16_t)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  return make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& a, float& b) {
#ifdef PXA_OPERAND_F16
  const bf16x2 v = __builtin_bit_cast(bf16x2, u);
  a = (float)v[0];
  b = (float)v[1];
#else
  a = __builtin_bit_cast(float, u << 16);
  b = __builtin_bit_cast(float, u & 0xffff0000u);
#endif
}
__device__ __forceinline__ void unpack_bf16x8(const uint4& u, float (&f)[8]) {
  unpack_bf16x2(u.x, f[0], f[1]); unpack_bf16x2(u.y, f[2], f[3]);
  unpack_bf16x2(u.z, f[4], f[5]); unpack_bf16x2(u.w, f[6], f[7]);
}

// c + a.lo*b.lo + a.hi*b.hi on packed operand pairs (v_dot2c_f32_bf16 / v_dot2c_f32_f16): sums and sums of squares of stored outputs
// without unpacking them.  OPERAND_ONE_X2 = (1, 1).
__device__ __forceinline__ float dot2_acc(uint32_t a, uint32_t b, float c) {
#ifdef PXA_OPERAND_F16
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, a), __builtin_bit_cast(h16x2, b), c, false);
#else
  typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b16x2, a), __builtin_bit_cast(b16x2, b), c, false);
#endif
}
#ifdef PXA_OPERAND_F16
constexpr uint32_t OPERAND_ONE_X2 = 0x3c003c00u;
#else
constexpr uint32_t OPERAND_ONE_X2 = 0x3f803f80u;
#endif

// LDS transpose read: 4 bf16 (see layout note above). `p` must be 8-byte aligned.
__device__ __forceinline__ s16x4 lds_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p));
}
__device__ __forceinline__ bf16x8 concat_tr(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef PXA_OPERAND_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);   // same operand / accumulator lane layout as the bf16 form
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x32 (probe/probe.hip): A[i = l&15][k = 8*(l>>4) + j], B[k = 8*(l>>4) + j][n = l&15], j = 0..7; C: col = l&15, row = 4*(l>>4) + g,
// g = 0..3.  Same FLOP rate on paper as the 32x32x16 form, but per FLOP it moves half the accumulator words through the register file:
// on N(0,1) operands the chip's power limit lets an MFMA-only loop of this shape run 12 % faster (1.95 vs 1.75 PFLOP/s, probe/mfma_power.hip,
// profiles/r02b_mfma_power.txt); with zero operands it is 2 % slower (17 vs 16 issue cycles per 16 cycles of work).
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef PXA_OPERAND_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// GELU(approximate="tanh") and its derivative (reference: nn.GELU(approximate="tanh"), PixArtMS.py:66), written through the
// identity 0.5 (1 + tanh u) = sigmoid(2u) = 1 / (1 + 2^w),  w = -2 log2(e) k0 x (1 + k1 x^2):
//   gelu(x)  = x s                                   7 VALU ops, 2 of them transcendental (v_exp_f32, v_rcp_f32)
//   gelu'(x) = s + x s (1 - s) 2 k0 (1 + 3 k1 x^2)   with 1 - s = 2^w s: 12 ops
// (the epilogues that apply them run with every accumulator live and are VALU-bound: the textbook tanh form with a true
// division cost 2.5x more and made the fc2-dX GEMM 65 % slower than its plain twin).  w is clamped so 2^w stays finite.
__device__ __forceinline__ float gelu_sigmoid_terms(float x, float x2, float& e) {
  const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;
  const float w = fminf(x * fmaf(x2, c1, c0), 126.0f);
  e = __builtin_amdgcn_exp2f(w);
  return __builtin_amdgcn_rcpf(1.0f + e);              // s = sigmoid(2u)
}
__device__ __forceinline__ float gelu_tanh(float x) {
  float e;
  return x * gelu_sigmoid_terms(x, x * x, e);
}
#ifndef PXA_GELU_V2
#define PXA_GELU_V2 0       // 1 = the 9 + 2 form below (A/B builds)
#endif
__device__ __forceinline__ float gelu_tanh_both(float x, float& grad) {   // returns gelu(x), grad = gelu'(x)
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
#if PXA_GELU_V2
  // 9 plain + 2 transcendental instructions instead of 11 + 2: 1 - s by subtraction (no 2^w * s product, so 2^w may overflow to +inf - s is then an
  // exact 0 - and the clamp goes), x s shared between the value and the derivative.  |error| of 1 - s is 6e-8 where the exact product is smaller
  // still: invisible behind the 16-bit store.
  const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * fmaf(x2, c1, c0));
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  const float g = x * s;
  grad = fmaf(g * (1.0f - s), fmaf(x2, 6.0f * k0 * k1, 2.0f * k0), s);
  return g;
#else
  float e;
  const float s = gelu_sigmoid_terms(x, x2, e);
  grad = fmaf(s * (e * s), x * fmaf(x2, 6.0f * k0 * k1, 2.0f * k0), s);
  return x * s;
#endif
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  float e;
  const float s = gelu_sigmoid_terms(x, x2, e);
  const float one_minus_s = e * s;
  return fmaf(s * one_minus_s, x * fmaf(x2, 6.0f * k0 * k1, 2.0f * k0), s);
}

// Streaming accesses: rows / flat buffers whose every byte is touched once per launch (the HBM-bound kernels of norm.hip, optim.hip, the
// delta pre-pass).  Non-temporal loads and stores measured ln_mod_fwd 200 -> 167 us (4.5 -> 5.4 TB/s), ln_mod_bwd 247 -> 227, gate_bwd 234 -> 225,
// the training step -2.8 ms (profiles/r02h_elem_nt.txt).  PXA_STREAM_NT (A/B builds): bit 0 = stores, bit 1 = loads.
#ifndef PXA_STREAM_NT
#define PXA_STREAM_NT 3
#endif
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_f4(float* p, const float4& v) {
  if (PXA_STREAM_NT & 1) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ void st_u2(void* p, const uint2& v) {
  if (PXA_STREAM_NT & 1) __builtin_nontemporal_store(nt_u2{v.x, v.y}, reinterpret_cast<nt_u2*>(p));
  else *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ float4 ld_f4(const float* p) {
  if (PXA_STREAM_NT & 2) { const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); return make_float4(t[0], t[1], t[2], t[3]); }
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ uint2 ld_u2(const void* p) {
  if (PXA_STREAM_NT & 2) { const nt_u2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p)); return make_uint2(t[0], t[1]); }
  return *reinterpret_cast<const uint2*>(p);
}
__device__ __forceinline__ uint4 ld_u4(const void* p) {
  if (PXA_STREAM_NT & 2) { const nt_u4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p)); return make_uint4(t[0], t[1], t[2], t[3]); }
  return *reinterpret_cast<const uint4*>(p);
}

// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane from a per-lane global address to the wave-uniform LDS address `lds` + 16 lane) issued as inline
// asm, so that the compiler's waitcnt pass does not know an LDS-DMA is in flight (round 3).  With the builtin it puts `s_waitcnt vmcnt(0)` in front of
// every ds_read_b64_tr_b16 that follows (the transpose-read intrinsic's memory operand has no alias scope; plain LDS loads have one and are left
// alone): the persistent GEMM's NN / TN main loops drained their four-deep DMA ring once per k-unit, the attention kernels waited for the NEXT tile's
// DMA at their first transpose read of the current one.  The price: the compiler no longer waits for a DMA at all - every consumer barrier must be
// preceded by an EXPLICIT `s_waitcnt vmcnt(N)` (lds_dma_wait / the GEMM's wait_vmcnt<N>).  M0 is written inside the statement (s_nop: the M0 hazard of
// LDS-DMA); no kernel that uses this helper may also use the builtin (the compiler tracks M0 only for its own).  PXA_ASM_DMA = 0: the builtin (A/B).
#ifndef PXA_ASM_DMA
#define PXA_ASM_DMA 1
#endif
__device__ __forceinline__ void lds_dma16(const void* gptr, const char* lds) {
#if PXA_ASM_DMA
  const unsigned l = (unsigned)(uintptr_t)LDS_PTR(const char, lds);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(gptr) : "memory");
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
#endif
}
template <int N> __device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float half_wave_sum(float v) {  // reduce inside each 32-lane half
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { v = half_wave_sum(v); return v + __shfl_xor(v, 32); }

}  // namespace pxa

// thread-local last-error string for the C ABI (api.hip)
void pxa_set_error(const char* fmt, ...);
#define PXA_CHECK(cond, ...) do { if (!(cond)) { pxa_set_error(__VA_ARGS__); return -1; } } while (0)
#define PXA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { pxa_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return -2; } } while (0)
