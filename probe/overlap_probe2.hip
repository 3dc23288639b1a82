// Round 3: does non-matrix work hide under MFMAs on gfx950?  (probe/overlap_probe.hip issued 8 MFMAs and THEN 64 v_fma in
// program order - an in-order wave cannot slip VALU under MFMAs it has already issued, so its "sum" result could not show overlap.)
// Here the stream is hand-interleaved in ONE asm block per loop iteration, so neither the compiler nor the assembler can move anything:
//     4 x { v_mfma (4 rotating accumulators) ; K fillers }        K = 0..8
// fillers: v_fma_f32 | v_exp_f32 | v_cvt_pk_bf16_f32 | ds_read_b128 | a softmax-like mix (2 fma, 1 exp, 1 cvt, 1 ds_read per 5).
// Reported per variant: shader cycles per MFMA (s_memtime over the loop, first wave of block 0), wall time, effective clock, at one and two waves
// per SIMD (one / two 256-thread workgroups per CU).  Also: 8-wave workgroups whose waves 0-3 run the bare MFMA loop while waves 4-7 (the other wave
// of each SIMD) run a bare filler loop - cross-wave overlap with no interleave at all.
//   hipcc --offload-arch=gfx950 -O3 probe/overlap_probe2.hip -o probe/overlap_probe2 && probe/overlap_probe2 [z]     (z = zero operands)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define M32(i) "v_mfma_f32_32x32x16_bf16 %[a" #i "], %[A], %[B], %[a" #i "]\n"
#define M16(i) "v_mfma_f32_16x16x32_bf16 %[b" #i "], %[A], %[B], %[b" #i "]\n"
#define F_FMA(i) "v_fma_f32 %[v" #i "], %[v" #i "], %[c0], %[c1]\n"
#define F_EXP(i) "v_exp_f32 %[v" #i "], %[v" #i "]\n"
#define F_CVT(i) "v_cvt_pk_bf16_f32 %[v" #i "], %[v" #i "], %[c0]\n"
#define LOFF0 "0"
#define LOFF1 "1024"
#define F_LDS(i) "ds_read_b128 %[d" #i "], %[la] offset:" LOFF##i "\n"
#define G0(F) ""
#define G1(F) F(0)
#define G2(F) F(0) F(1)
#define G3(F) F(0) F(1) F(2)
#define G4(F) F(0) F(1) F(2) F(3)
#define G5(F) F(0) F(1) F(2) F(3) F(4)
#define G6(F) F(0) F(1) F(2) F(3) F(4) F(5)
#define G7(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6)
#define G8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
// softmax-like mixes (issue counts per gap): 3 = fma exp cvt, 5 = fma exp fma cvt lds, 7 = fma exp fma cvt lds fma lds
#define X3(F) F_FMA(0) F_EXP(1) F_CVT(2)
#define X5(F) F_FMA(0) F_EXP(1) F_FMA(3) F_CVT(2) F_LDS(0)
#define X7(F) F_FMA(0) F_EXP(1) F_FMA(3) F_CVT(2) F_LDS(0) F_FMA(4) F_LDS(1)
#define BODY32(G, F) M32(0) G(F) M32(1) G(F) M32(2) G(F) M32(3) G(F)
#define BODY16(G, F) M16(0) G(F) M16(1) G(F) M16(2) G(F) M16(3) G(F) M16(0) G(F) M16(1) G(F) M16(2) G(F) M16(3) G(F)
#define BODYV(G, F) G(F) G(F) G(F) G(F)

struct State {
  bf16x8 A, B;
  f32x16 a0, a1, a2, a3;
  f32x4 b0, b1, b2, b3;
  float v0, v1, v2, v3, v4, v5, v6, v7, c0, c1;
  f32x4 d0, d1;
  unsigned la;
};
#define OPERANDS(s)                                                                                                                               \
  [a0] "+v"(s.a0), [a1] "+v"(s.a1), [a2] "+v"(s.a2), [a3] "+v"(s.a3), [b0] "+v"(s.b0), [b1] "+v"(s.b1), [b2] "+v"(s.b2), [b3] "+v"(s.b3),           \
      [v0] "+v"(s.v0), [v1] "+v"(s.v1), [v2] "+v"(s.v2), [v3] "+v"(s.v3), [v4] "+v"(s.v4), [v5] "+v"(s.v5), [v6] "+v"(s.v6), [v7] "+v"(s.v7),       \
      [d0] "+v"(s.d0), [d1] "+v"(s.d1)                                                                                                             \
      : [A] "v"(s.A), [B] "v"(s.B), [c0] "v"(s.c0), [c1] "v"(s.c1), [la] "v"(s.la)

__device__ __forceinline__ void init(State& s, const float* in, int lane) {
  for (int i = 0; i < 8; i++) { s.A[i] = (__bf16)in[(lane * 8 + i) & 1023]; s.B[i] = (__bf16)in[(lane * 8 + i + 512) & 1023]; }
  for (int g = 0; g < 16; g++) s.a0[g] = s.a1[g] = s.a2[g] = s.a3[g] = 0.f;
  for (int g = 0; g < 4; g++) s.b0[g] = s.b1[g] = s.b2[g] = s.b3[g] = s.d0[g] = s.d1[g] = 0.f;
  s.v0 = in[lane & 1023]; s.v1 = in[(lane + 1) & 1023]; s.v2 = in[(lane + 2) & 1023]; s.v3 = in[(lane + 3) & 1023];
  s.v4 = in[(lane + 4) & 1023]; s.v5 = in[(lane + 5) & 1023]; s.v6 = in[(lane + 6) & 1023]; s.v7 = in[(lane + 7) & 1023];
  s.c0 = 0.75f + 0.f * in[3]; s.c1 = in[5] * 0.01f;
  s.la = lane * 16;                                    // conflict-free b128 rows
}
__device__ __forceinline__ float fold(const State& s) {
  float r = 0.f;
  for (int g = 0; g < 16; g++) r += s.a0[g] + s.a1[g] + s.a2[g] + s.a3[g];
  for (int g = 0; g < 4; g++) r += s.b0[g] + s.b1[g] + s.b2[g] + s.b3[g] + s.d0[g] + s.d1[g];
  return r + s.v0 + s.v1 + s.v2 + s.v3 + s.v4 + s.v5 + s.v6 + s.v7;
}

#define KERNEL(name, BODY)                                                                                        \
  __global__ __launch_bounds__(256, 2) void name(float* out, const float* in, int iters, long long* cyc) {       \
    __shared__ float lds[2 * 1024 + 64 * 4];                                                                      \
    lds[threadIdx.x] = in[threadIdx.x]; lds[threadIdx.x + 256] = 0.f; lds[threadIdx.x + 512] = 1.f;              \
    __syncthreads();                                                                                              \
    State s; init(s, in, threadIdx.x & 63);                                                                       \
    const long long t0 = __builtin_readcyclecounter();                                                            \
    for (int it = 0; it < iters; it++) asm volatile(BODY : OPERANDS(s));                                          \
    const long long t1 = __builtin_readcyclecounter();                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\ns_nop 15\ns_nop 15" ::: "memory");                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = fold(s);                                                                \
  }
// 512-thread workgroup: waves 0-3 run BODYM, waves 4-7 BODYF (roles: 1 = matrix only, 2 = filler only, 3 = both)
#define KERNEL_SPLIT(name, BODYM, BODYF)                                                                          \
  __global__ __launch_bounds__(512, 2) void name(float* out, const float* in, int iters, long long* cyc, int roles) { \
    __shared__ float lds[2 * 1024 + 64 * 4];                                                                      \
    lds[threadIdx.x] = in[threadIdx.x]; lds[threadIdx.x + 512] = 0.f;                                             \
    __syncthreads();                                                                                              \
    State s; init(s, in, threadIdx.x & 63);                                                                       \
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                               \
    const long long t0 = __builtin_readcyclecounter();                                                            \
    if (w < 4) { if (roles & 1) for (int it = 0; it < iters; it++) asm volatile(BODYM : OPERANDS(s)); }           \
    else       { if (roles & 2) for (int it = 0; it < iters; it++) asm volatile(BODYF : OPERANDS(s)); }           \
    const long long t1 = __builtin_readcyclecounter();                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\ns_nop 15\ns_nop 15" ::: "memory");                                                              \
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;            \
    out[blockIdx.x * 512 + threadIdx.x] = fold(s);                                                                \
  }

KERNEL(k32_fma0, BODY32(G0, F_FMA)) KERNEL(k32_fma1, BODY32(G1, F_FMA)) KERNEL(k32_fma2, BODY32(G2, F_FMA)) KERNEL(k32_fma3, BODY32(G3, F_FMA))
KERNEL(k32_fma4, BODY32(G4, F_FMA)) KERNEL(k32_fma5, BODY32(G5, F_FMA)) KERNEL(k32_fma6, BODY32(G6, F_FMA)) KERNEL(k32_fma7, BODY32(G7, F_FMA))
KERNEL(k32_fma8, BODY32(G8, F_FMA))
KERNEL(k32_exp1, BODY32(G1, F_EXP)) KERNEL(k32_exp2, BODY32(G2, F_EXP)) KERNEL(k32_exp4, BODY32(G4, F_EXP))
KERNEL(k32_cvt2, BODY32(G2, F_CVT)) KERNEL(k32_cvt4, BODY32(G4, F_CVT))
KERNEL(k32_lds1, BODY32(G1, F_LDS)) KERNEL(k32_lds2, BODY32(G2, F_LDS))
KERNEL(k32_mix3, BODY32(X3, F_FMA)) KERNEL(k32_mix5, BODY32(X5, F_FMA)) KERNEL(k32_mix7, BODY32(X7, F_FMA))
KERNEL(k16_fma0, BODY16(G0, F_FMA)) KERNEL(k16_fma1, BODY16(G1, F_FMA)) KERNEL(k16_fma2, BODY16(G2, F_FMA)) KERNEL(k16_fma3, BODY16(G3, F_FMA))
KERNEL(k16_mix3, BODY16(X3, F_FMA))
KERNEL(kv_fma4, BODYV(G4, F_FMA)) KERNEL(kv_fma8, BODYV(G8, F_FMA)) KERNEL(kv_exp4, BODYV(G4, F_EXP)) KERNEL(kv_mix5, BODYV(X5, F_FMA))
KERNEL_SPLIT(ks_fma4, BODY32(G0, F_FMA), BODYV(G4, F_FMA)) KERNEL_SPLIT(ks_fma8, BODY32(G0, F_FMA), BODYV(G8, F_FMA))
KERNEL_SPLIT(ks_mix5, BODY32(G0, F_FMA), BODYV(X5, F_FMA))

static float *g_in, *g_out;
static long long* g_cyc;
template <typename K>
void run(const char* name, K kern, int bpc, int iters, int mfma_per_iter, int fill_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * bpc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, g_out, g_in, iters, g_cyc);
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, g_out, g_in, iters, g_cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  long long c; hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  const double per = mfma_per_iter ? (double)c / ((double)iters * mfma_per_iter) : (double)c / ((double)iters * fill_per_iter);
  printf("%-10s %d wave/SIMD: %8.3f ms  %7.1f cyc/%s  (%d fillers per MFMA)  eff.clock %.2f GHz\n", name, bpc, ms, per, mfma_per_iter ? "MFMA" : "filler",
         mfma_per_iter ? fill_per_iter / mfma_per_iter : 0, c / (ms * 1e6));
}
template <typename K>
void run_split(const char* name, K kern, int iters, int roles) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, g_out, g_in, iters, g_cyc, roles);
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, g_out, g_in, iters, g_cyc, roles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  long long c[2]; hipMemcpy(c, g_cyc, 16, hipMemcpyDeviceToHost);
  printf("%-10s roles %d (1 = MFMA waves 0-3, 2 = filler waves 4-7, 3 = both, one of each per SIMD): %8.3f ms   MFMA wave %lld cyc, filler wave %lld cyc\n", name, roles, ms,
         (roles & 1) ? c[0] : 0LL, (roles & 2) ? c[1] : 0LL);
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && argv[1][0] == 'z';
  hipMalloc(&g_in, 4096); hipMalloc(&g_out, 256 * 8 * 512 * 4); hipMalloc(&g_cyc, 64);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = zeros ? 0.f : (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(g_in, h.data(), 4096, hipMemcpyHostToDevice);
  const int it = 20000;
  printf("operands: %s; one asm block per iteration = 4 x {v_mfma_f32_32x32x16_bf16 ; K fillers} (8 x for 16x16x32); floor 32 (16) cycles per MFMA\n", zeros ? "zeros" : "random");
  for (int bpc = 1; bpc <= 2; bpc++) {
#define R32(k, K) run(#k, k, bpc, it, 4, 4 * K)
    R32(k32_fma0, 0); R32(k32_fma1, 1); R32(k32_fma2, 2); R32(k32_fma3, 3); R32(k32_fma4, 4); R32(k32_fma5, 5); R32(k32_fma6, 6); R32(k32_fma7, 7); R32(k32_fma8, 8);
    R32(k32_exp1, 1); R32(k32_exp2, 2); R32(k32_exp4, 4); R32(k32_cvt2, 2); R32(k32_cvt4, 4); R32(k32_lds1, 1); R32(k32_lds2, 2);
    R32(k32_mix3, 3); R32(k32_mix5, 5); R32(k32_mix7, 7);
#define R16(k, K) run(#k, k, bpc, it, 8, 8 * K)
    R16(k16_fma0, 0); R16(k16_fma1, 1); R16(k16_fma2, 2); R16(k16_fma3, 3); R16(k16_mix3, 3);
    run("kv_fma4", kv_fma4, bpc, it, 0, 16); run("kv_fma8", kv_fma8, bpc, it, 0, 32); run("kv_exp4", kv_exp4, bpc, it, 0, 16); run("kv_mix5", kv_mix5, bpc, it, 0, 20);
  }
  for (int roles = 1; roles <= 3; roles++) { run_split("ks_fma4", ks_fma4, it, roles); run_split("ks_fma8", ks_fma8, it, roles); run_split("ks_mix5", ks_mix5, it, roles); }
  return 0;
}
