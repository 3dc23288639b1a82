// pxa_mfma_rate_probe: the part's power-limited matrix rate, measured where the benchmark runs.
// A register-resident stream of v_mfma_f32_32x32x16 (SHAPE 32) or v_mfma_f32_16x16x32 (SHAPE 16) on caller-supplied 16-bit operand data - no loads, no LDS traffic,
// no vector work inside the loop - from ONE wave per SIMD of every CU (the whole 160 KiB of LDS is claimed, so a CU takes one workgroup), every MFMA of the
// unrolled body on a different (A fragment, B fragment) pair so that the multiplier inputs toggle from instruction to instruction as they do in a kernel's
// main loop.  bench.py times it with events and reports `roofline.mfma_only_rate`: what the sheet peak (2.5 PFLOP/s) comes down to on random data under this
// box's power limit, before a single operand is moved (round 4 found 1.42 PFLOP/s by ablating the dK/dV kernel to its MFMAs: profiles/r4_17_dkv4_ablations.txt).
// Measurement infrastructure of the product library - it computes nothing the denoiser uses.
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_rate_kernel(const bf16x8* __restrict__ data, int iters, float* sink) {
  extern __shared__ char smem_unused[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16x8* src = data + (size_t)((blockIdx.x * 4 + wave) & 255) * 64 * 16 + lane;   // 16 fragments per wave, 64 lanes each
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    bf16x8 af[2][4], bf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++) { af[ks][i] = src[(ks * 4 + i) * 64]; bf[ks][i] = src[(8 + ks * 4 + i) * 64]; }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) acc[i][j][g] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = mfma32(bf[ks][j], af[ks][i], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int g = 0; g < 16; g++) s += acc[i][j][g];
  } else {
    bf16x8 af[8], bf[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { af[i] = src[i * 64]; bf[i] = src[(8 + i) * 64]; }
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = mfma16(bf[j], af[i], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) s += acc[i][j][g];
  }
  if (s == 12345.678f) sink[0] = s;      // keeps the accumulators alive; never true on the probe's data
}
}  // namespace

extern "C" long pxa_mfma_rate_probe_bytes(void) { return 256L * 16 * 64 * 16; }

extern "C" int pxa_mfma_rate_probe(const void* operands, int shape, int iters, float* sink, double* flops_per_launch, hipStream_t stream) {
  PXA_CHECK(operands && sink && iters > 0, "pxa_mfma_rate_probe: null operand buffer / sink or iters <= 0");
  PXA_CHECK(shape == 32 || shape == 16, "pxa_mfma_rate_probe: shape must be 32 (32x32x16) or 16 (16x16x32)");
  int cus = 0;
  if (pxa_device_info(&cus, nullptr) != 0 || cus <= 0) return -3;
  constexpr int LDS = 160 * 1024;
  static bool attr32 = false, attr16 = false;
  bool& attr = shape == 32 ? attr32 : attr16;
  const void* fn = shape == 32 ? reinterpret_cast<const void*>(mfma_rate_kernel<32>) : reinterpret_cast<const void*>(mfma_rate_kernel<16>);
  if (!attr) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) { pxa_set_error("pxa_mfma_rate_probe: cannot claim %d bytes of LDS", LDS); return -3; }
    attr = true;
  }
  if (shape == 32) hipLaunchKernelGGL(mfma_rate_kernel<32>, dim3(cus), dim3(256), LDS, stream, (const bf16x8*)operands, iters, sink);
  else hipLaunchKernelGGL(mfma_rate_kernel<16>, dim3(cus), dim3(256), LDS, stream, (const bf16x8*)operands, iters, sink);
  PXA_LAUNCH_CHECK();
  // 32 MFMAs of 32 x 32 x 16 (or 64 of 16 x 16 x 32) per iteration and wave = 2 * 128 * 128 * 32 FLOP
  if (flops_per_launch) *flops_per_launch = (double)cus * 4.0 * (double)iters * 2.0 * 128.0 * 128.0 * 32.0;
  return 0;
}
