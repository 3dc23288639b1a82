// Does VALU work overlap with MFMA work on one SIMD of gfx950?  One 256-thread workgroup per CU slot (1, 2 or 3 waves per SIMD), each wave
// runs ITER iterations of { NM independent v_mfma_f32_32x32x16_bf16 ; NV v_fma_f32 (independent chains) }.  Prints time for MFMA only,
// VALU only, both in one wave, and MFMA waves beside VALU waves.   hipcc --offload-arch=gfx950 -O3 probe/overlap_probe.hip -o probe/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int MODE>   // MODE 0: every wave does both; 1: even waves MFMA, odd waves VALU (needs >= 2 waves per SIMD)
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)in[(lane * 8 + i) & 1023]; b[i] = (__bf16)in[(lane * 8 + i + 512) & 1023]; }
  f32x16 acc[4];
  for (int j = 0; j < 4; j++) for (int g = 0; g < 16; g++) acc[j][g] = 0.f;
  float v[8];
  for (int j = 0; j < 8; j++) v[j] = in[(lane + j) & 1023];
  const float c0 = in[3], c1 = in[5];
  const bool do_m = MODE == 0 || ((blockIdx.x & 1) == 0), do_v = MODE == 0 || ((blockIdx.x & 1) == 1);
  for (int it = 0; it < iters; it++) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < NM; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int n = 0; n < NV; n++) v[n & 7] = __builtin_fmaf(v[n & 7], c0, c1);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++) for (int g = 0; g < 16; g++) s += acc[j][g];
  for (int j = 0; j < 8; j++) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s + wave;
}

template <int NM, int NV, int MODE>
float run(int blocks_per_cu, float* out, const float* in, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(grid), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<NM, NV, MODE>), dim3(grid), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && argv[1][0] == 'z';
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = zeros ? 0.f : (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  const int iters = 4000;
  printf("operands: %s; per iteration 8 MFMA (= 256 MFMA-pipe cycles) and/or 64 v_fma per wave; ms per launch\n", zeros ? "zeros" : "random");
  for (int bpc = 1; bpc <= 3; bpc++) {
    float m = run<8, 0, 0>(bpc, out, in, iters), v = run<0, 64, 0>(bpc, out, in, iters), both = run<8, 64, 0>(bpc, out, in, iters);
    float v32 = run<0, 32, 0>(bpc, out, in, iters), both32 = run<8, 32, 0>(bpc, out, in, iters);
    printf("%d wave/SIMD: MFMA only %.3f  VALU64 only %.3f  both-in-one-wave %.3f (sum %.3f)   VALU32 only %.3f  both %.3f (sum %.3f)\n", bpc, m, v, both, m + v, v32,
           both32, m + v32);
    if (bpc >= 2) {
      float sp = run<8, 64, 1>(bpc, out, in, iters);
      printf("            MFMA waves beside VALU waves (split by workgroup parity): %.3f\n", sp);
    }
  }
  return 0;
}
