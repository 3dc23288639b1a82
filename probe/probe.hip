// Hardware-semantics probe for gfx950 (MI355X). Test infrastructure, not product.
// Verifies the lane layouts every MFMA/LDS kernel in pixart_sigma_amd/csrc relies on:
//   1. v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 operand + accumulator layouts
//   2. ds_read_b64_tr_b16 (LDS transpose read) address -> lane/element mapping
//   3. v_permlane32_swap semantics
//   4. HBM copy bandwidth and MFMA issue-rate ceilings (roofline sanity)
// Build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe ; run on the GPU box, writes text to stdout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// ---- 1a. 32x32x16: D[i][n] = sum_k A[i][k] B[k][n];  A row-major [32][16], Bt row-major [32 n][16 k]
__global__ void mfma32(const unsigned short* A, const unsigned short* Bt, float* D) {
  int l = threadIdx.x, r = l & 31, hi = l >> 5;
  bf8 a, b;
  for (int j = 0; j < 8; j++) {
    unsigned short av = A[r * 16 + 8 * hi + j], bv = Bt[r * 16 + 8 * hi + j];
    a[j] = __builtin_bit_cast(__bf16, av); b[j] = __builtin_bit_cast(__bf16, bv);
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int g = 0; g < 16; g++) {
    int row = (g & 3) + 8 * (g >> 2) + 4 * hi, col = r;
    D[row * 32 + col] = c[g];
  }
}
// ---- 1b. 16x16x32: A [16][32], Bt [16 n][32 k]
__global__ void mfma16(const unsigned short* A, const unsigned short* Bt, float* D) {
  int l = threadIdx.x, r = l & 15, q = l >> 4;
  bf8 a, b;
  for (int j = 0; j < 8; j++) {
    unsigned short av = A[r * 32 + 8 * q + j], bv = Bt[r * 32 + 8 * q + j];
    a[j] = __builtin_bit_cast(__bf16, av); b[j] = __builtin_bit_cast(__bf16, bv);
  }
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int g = 0; g < 4; g++) D[(q * 4 + g) * 16 + r] = c[g];
}
// ---- 2. tr read: LDS[i] = i (ushort). pattern 0: addr = lane*8 bytes; pattern 1: per-lane table
__global__ void trread(const int* addr_tab, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int off = addr_tab[threadIdx.x];  // element offset (multiple of 4)
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + off));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
// ---- 3. permlane32_swap
__global__ void plswap(unsigned* out) {
  unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}
// ---- 4a. copy bandwidth
__global__ void copyk(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) out[i] = in[i];
}
// ---- 4b. MFMA issue rate (4 independent accumulators per wave)
template <int SHAPE>
__global__ void mfmarate(float* out, int iters) {
  bf8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
  if (SHAPE == 32) {
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f4v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
}
// ---- 5. float atomicAdd throughput (split-K dW accumulate pattern: coalesced, distinct addresses)
__global__ void atomk(float* out, size_t n, int reps) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (int r = 0; r < reps; r++)
    for (size_t k = i; k < n; k += st) atomicAdd(&out[k], 1.0f);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
  printf("device %s arch %s CUs %d clock %d kHz memclk %d kHz LDS/block %zu regs/block %d mem free %.1f GB total %.1f GB\n",
         p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.sharedMemPerBlock, p.regsPerBlock, fr / 1e9, tot / 1e9);
  srand(1);
  // 1a
  {
    std::vector<unsigned short> A(32 * 16), Bt(32 * 16); std::vector<float> Af(32 * 16), Bf(32 * 16), D(32 * 32);
    for (int i = 0; i < 32 * 16; i++) { Af[i] = (float)(rand() % 17 - 8); Bf[i] = (float)(rand() % 13 - 6); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    unsigned short *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    mfma32<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 32; i++) for (int n = 0; n < 32; n++) { float s = 0; for (int k = 0; k < 16; k++) s += Af[i * 16 + k] * Bf[n * 16 + k]; if (s != D[i * 32 + n]) bad++; }
    printf("MFMA32x32x16 layout check: %s (%d mismatches of 1024)\n", bad ? "FAIL" : "PASS", bad);
  }
  // 1b
  {
    std::vector<unsigned short> A(16 * 32), Bt(16 * 32); std::vector<float> Af(16 * 32), Bf(16 * 32), D(16 * 16);
    for (int i = 0; i < 16 * 32; i++) { Af[i] = (float)(rand() % 17 - 8); Bf[i] = (float)(rand() % 13 - 6); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
    unsigned short *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
    mfma16<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) { float s = 0; for (int k = 0; k < 32; k++) s += Af[i * 32 + k] * Bf[n * 32 + k]; if (s != D[i * 16 + n]) bad++; }
    printf("MFMA16x16x32 layout check: %s (%d mismatches of 256)\n", bad ? "FAIL" : "PASS", bad);
  }
  // 2
  {
    int tab[64]; int* dT; unsigned short* dO; unsigned short o[256];
    CK(hipMalloc(&dT, 256)); CK(hipMalloc(&dO, 512));
    for (int pat = 0; pat < 3; pat++) {
      for (int l = 0; l < 64; l++) {
        if (pat == 0) tab[l] = l * 4;                                   // lane-linear
        else if (pat == 1) tab[l] = (l >> 4) * 1024 + ((l & 15) >> 2) * 100 * 4 + (l & 3) * 4;  // rows 100*4 elems apart
        else tab[l] = ((63 - l) * 4);                                   // reversed
      }
      CK(hipMemcpy(dT, tab, 256, hipMemcpyHostToDevice));
      trread<<<1, 64>>>(dT, dO); CK(hipMemcpy(o, dO, 512, hipMemcpyDeviceToHost));
      printf("TRREAD pattern %d (lane: addr -> 4 values)\n", pat);
      for (int l = 0; l < 64; l++) printf("  l%02d a%4d -> %4d %4d %4d %4d\n", l, tab[l], o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
      // check model: out[l][j] == lds[ tab[16*(l>>4) + 4*j + ((l&15)>>2)] + (l&3) ]
      int bad = 0; for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) { int src = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2); if (o[l * 4 + j] != tab[src] + (l & 3)) bad++; }
      printf("  model 'lane l elem j <- lane(16g+4j+(t>>2)) elem (t&3)': %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
    }
  }
  // 3
  {
    unsigned* dO; unsigned o[128]; CK(hipMalloc(&dO, 512)); plswap<<<1, 64>>>(dO); CK(hipMemcpy(o, dO, 512, hipMemcpyDeviceToHost));
    printf("PERMLANE32_SWAP(a=1000+l, b=2000+l): lane0 r=(%u,%u) lane31 (%u,%u) lane32 (%u,%u) lane63 (%u,%u)\n", o[0], o[1], o[62], o[63], o[64], o[65], o[126], o[127]);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // 4a copy BW
  {
    size_t bytes = (size_t)2 << 30; float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    for (int it = 0; it < 2; it++) copyk<<<2048, 256>>>(a, b, bytes / 16);
    CK(hipEventRecord(e0)); for (int it = 0; it < 5; it++) copyk<<<2048, 256>>>(a, b, bytes / 16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("COPY 2GiB float4: %.3f ms/iter -> %.2f TB/s (read+write)\n", ms / 5, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
    CK(hipFree(a)); CK(hipFree(b));
  }
  // 4b MFMA rate
  {
    float* o; CK(hipMalloc(&o, 256 * 8 * 256 * 4)); int iters = 20000;
    for (int shape : {32, 16}) for (int wpb : {256, 512}) {
      int blocks = 256 * (wpb == 256 ? 2 : 1);
      auto launch = [&]() { if (shape == 32) mfmarate<32><<<blocks, wpb>>>(o, iters); else mfmarate<16><<<blocks, wpb>>>(o, iters); };
      launch(); CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = (double)blocks * (wpb / 64) * iters * 4 * (shape == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
      printf("MFMA rate shape %d blocks %d x %d thr: %.3f ms -> %.1f TF/s\n", shape, blocks, wpb, ms, fl / (ms * 1e-3) / 1e12);
    }
  }
  // 5 atomics
  {
    size_t n = (size_t)64 << 20; float* o; CK(hipMalloc(&o, n * 4)); CK(hipMemset(o, 0, n * 4));
    atomk<<<2048, 256>>>(o, n, 1); CK(hipEventRecord(e0)); atomk<<<2048, 256>>>(o, n, 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("ATOMIC fp32 add coalesced: %.3f ms for %zu M atomics -> %.1f G atomics/s (%.2f TB/s equiv 4B)\n", ms, 4 * n >> 20, 4.0 * n / (ms * 1e-3) / 1e9, 16.0 * n / (ms * 1e-3) / 1e12);
  }
  printf("PROBE DONE\n");
  return 0;
}
