// v_permlane16_swap_b32 semantics on gfx950 (csrc/attn.hip pack_xy relies on it).  Test infrastructure, not product.
// Expected: odd 16-lane rows of the first operand are exchanged with even rows of the second.
// Build: hipcc --offload-arch=gfx950 -O3 perm16.hip -o perm16
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  const auto r = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  out[l] = r[0];
  out[64 + l] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  if (hipMalloc(&d, sizeof h) != hipSuccess) { printf("no device\n"); return 1; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const int row = l >> 4, c = l & 15;
    const unsigned e0 = (row & 1) ? 100u + 16 * (row - 1) + c : (unsigned)l;          // first result: a(R0) b(R0) a(R2) b(R2)
    const unsigned e1 = (row & 1) ? 100u + l : (unsigned)(16 * (row + 1) + c);        // second result: a(R1) b(R1) a(R3) b(R3)
    bad += h[l] != e0 || h[64 + l] != e1;
  }
  for (int r = 0; r < 4; r++) printf("row %d: first %3u..%3u  second %3u..%3u\n", r, h[16 * r], h[16 * r + 15], h[64 + 16 * r], h[64 + 16 * r + 15]);
  printf("permlane16_swap semantics %s\n", bad ? "DIFFER from the assumption" : "as assumed");
  return bad != 0;
}
