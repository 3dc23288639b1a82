// How fast are wave-coalesced fp32 atomic adds to DISTINCT addresses (each address hit REPS times by different workgroups at different times)?
// The pattern a fused attention backward would need for dQ: 75.5 M floats (B16 H16 N4096 d72), one add per key block.
// Variants: agent-scope atomicAdd (default), workgroup-scope (executes in the issuing XCD's L2 - only valid when all writers of an address
// share an XCD), plain load+add+store (the bandwidth reference, races ignored).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, long n, int reps, int xcd_local) {
  // workgroup w handles chunk (w % chunks) in repetition (w / chunks); with xcd_local the chunk -> XCD mapping is fixed (chunk % 8 == w % 8)
  const long chunk_elems = 256 * 16;                       // 16 floats per thread
  const long chunks = n / chunk_elems;
  long w = blockIdx.x;
  long chunk = w % chunks;
  if (xcd_local) { const long per = chunks / 8; chunk = (w % 8) * per + (w / 8) % per; }
  float* p = buf + chunk * chunk_elems + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const float v = 1.0f + i;
    if (MODE == 0) atomicAdd(p + i * 256, v);
    else if (MODE == 1) __hip_atomic_fetch_add(p + i * 256, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else p[i * 256] += v;
  }
}
template <int MODE> void run(const char* name, float* buf, long n, int reps, int xcd_local) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long chunks = n / 4096;
  const long grid = chunks * reps;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, buf, n, reps, xcd_local);
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, buf, n, reps, xcd_local);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%-44s reps %2d: %8.3f ms  %7.1f G lane-adds/s  (%.2f TB/s of read-modify-write payload)\n", name, reps, ms, n * (double)reps / ms / 1e6,
         n * (double)reps * 8 / ms / 1e9);
}
int main() {
  const long n = 16L * 16 * 4096 * 72;          // 75.5 M floats
  float* buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
  for (int reps : {1, 16}) {
    run<0>("atomicAdd f32, agent scope", buf, n, reps, 0);
    run<0>("atomicAdd f32, agent scope, XCD-local chunks", buf, n, reps, 1);
    run<1>("atomic add f32, workgroup scope, XCD-local", buf, n, reps, 1);
    run<2>("plain load + add + store (no atomicity)", buf, n, reps, 0);
  }
  return 0;
}
