// Issue cost (cycles per wave64 instruction, SIMD saturated with 3 waves) of the VALU ops the attention softmax uses on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  float v[8];
  const int lane = threadIdx.x;
  for (int j = 0; j < 8; j++) v[j] = in[(lane + j) & 1023];
  const float c0 = in[3], c1 = in[5];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int n = 0; n < 64; n++) {
      float& x = v[n & 7];
      if (OP == 0) x = __builtin_fmaf(x, c0, c1);
      if (OP == 1) x = __builtin_amdgcn_exp2f(x);
      if (OP == 2) { f2 a = {x, v[(n + 1) & 7]}; f2 b = {c0, c1}; a = a * b; x = a[0]; v[(n + 1) & 7] = a[1]; }
      if (OP == 3) { bf2 r; r[0] = (__bf16)x; r[1] = (__bf16)v[(n + 1) & 7]; x = __builtin_bit_cast(float, r); }
      if (OP == 4) x = x - c1;
      if (OP == 5) x = __builtin_amdgcn_rcpf(x);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 8; j++) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* out, const float* in) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * 3;
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  // per SIMD: 3 waves x iters x 64 instr
  printf("%-22s %.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz (3 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / (3.0 * iters * 64));
}
int main() {
  float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 256 * 3 * 256 * 4);
  float h[1024]; for (int i = 0; i < 1024; i++) h[i] = 0.001f * (i % 97) - 0.03f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  run<0>("v_fma_f32", out, in); run<1>("v_exp_f32", out, in); run<2>("v_pk_mul_f32 (2 el)", out, in); run<3>("v_cvt_pk_bf16_f32", out, in);
  run<4>("v_sub_f32", out, in); run<5>("v_rcp_f32", out, in);
  return 0;
}
