// Optimizer-side HBM-bound kernels for the data-parallel training step (train_scripts/train.py:182-184):
//   sumsq        : global gradient L2 norm (accelerator.clip_grad_norm_) as one reduction over the flat gradient buffer
//   clip_coef    : device-side scalar  coef = min(1, max_norm / (sqrt(sumsq)*inv_world + 1e-6)) * inv_world   (no host sync)
//   adamw_step   : torch.optim.AdamW update (configs/PixArt_xl2_internal.py:48: lr, weight_decay=3e-2, eps=1e-10) over the flat
//                  fp32 master weights, fused with the bf16 shadow-weight refresh the MFMA GEMMs read
//   cast         : fp32 -> bf16
// fp16-operand training (the reference's own mixed precision: configs/PixArt_xl2_internal.py:57 mixed_precision='fp16' ->
// accelerate's GradScaler around train_scripts/train.py:180-184): the loss is multiplied by a dynamic scale before backward; the
// `_scaled` entry points fold 1/scale into the clip coefficient, detect inf/nan through the norm itself, skip the update and keep
// torch.cuda.amp.GradScaler's growth/backoff bookkeeping in a 5-float device record - still no host synchronisation in the step.
#include "common.h"
#include "../../include/pixart_hip.h"

namespace {
using namespace pxa;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];      // plain load: the optimizer step re-reads the gradient right behind this pass
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 * 4; i < n; i++) acc += x[i] * x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float* __restrict__ out, float max_norm, float inv_world) {
  const float norm = sqrtf(*sumsq) * inv_world;
  float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
  coef = coef > 1.f ? 1.f : coef;
  out[0] = coef * inv_world;  // multiplier applied to the (summed) gradient
  out[1] = norm;              // total norm of the averaged gradient (what clip_grad_norm_ returns)
}

// scaler record: [0] loss scale, [1] growth tracker (clean steps since the last change), [2] found_inf of this step (0/1),
//                [3] number of optimizer steps actually applied, [4] number of skipped steps
__global__ void clip_coef_scaled_kernel(const float* __restrict__ sumsq, float* __restrict__ out, float max_norm, float inv_world,
                                        float* __restrict__ sc, float growth, float backoff, float interval) {
  const float scale = sc[0];
  const float norm = sqrtf(*sumsq) * inv_world / scale;           // norm of the averaged, UNSCALED gradient
  if (!(fabsf(norm) <= 3.0e38f)) {                                 // inf or nan anywhere in the gradient buffer -> skip this step
    out[0] = 0.f; out[1] = norm;
    sc[0] = scale * backoff; sc[1] = 0.f; sc[2] = 1.f; sc[4] += 1.f;
    return;
  }
  float coef = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
  coef = coef > 1.f ? 1.f : coef;
  out[0] = coef * inv_world / scale;
  out[1] = norm;
  float tr = sc[1] + 1.f;
  if (tr >= interval) { sc[0] = scale * growth; tr = 0.f; }
  sc[1] = tr; sc[2] = 0.f; sc[3] += 1.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, long n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, const float* __restrict__ gscale, const float* __restrict__ scaler) {
  if (scaler) {                                                    // loss-scaled step: skip on overflow, bias correction from the applied-step count
    if (scaler[2] != 0.f) return;
    const float t = scaler[3];
    bc1 = 1.f - powf(b1, t);
    bc2_sqrt = sqrtf(1.f - powf(b2, t));
  }
  const float gs = gscale ? gscale[0] : 1.f;
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pv = ld_f4(p + 4 * i);
    const float4 gv = ld_f4(g + 4 * i);
    float4 mv = ld_f4(m + 4 * i), vv = ld_f4(v + 4 * i);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x * gs, gv.y * gs, gv.z * gs, gv.w * gs};
    float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      pa[e] *= (1.f - lr * wd);
      ma[e] = b1 * ma[e] + (1.f - b1) * ga[e];
      va[e] = b2 * va[e] + (1.f - b2) * ga[e] * ga[e];
      const float denom = sqrtf(va[e]) / bc2_sqrt + eps;
      pa[e] -= (lr / bc1) * (ma[e] / denom);
    }
    st_f4(p + 4 * i, make_float4(pa[0], pa[1], pa[2], pa[3]));
    st_f4(m + 4 * i, make_float4(ma[0], ma[1], ma[2], ma[3]));
    st_f4(v + 4 * i, make_float4(va[0], va[1], va[2], va[3]));
    if (pb) st_u2(pb + 4 * i, pack_bf16x4(pa[0], pa[1], pa[2], pa[3]));
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = pack_bf16x4(v.x, v.y, v.z, v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 * 4; i < n; i++) y[i] = (bf16_t)x[i];
}
inline int grid_for(long n4) { long g = (n4 + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }
}  // namespace

extern "C" int pxa_sumsq_f32(const float* x, long n, float* out, hipStream_t stream) {
  PXA_CHECK(x && out && n > 0 && ((uintptr_t)x % 16) == 0, "pxa_sumsq_f32: bad args");
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, n, out);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_clip_coef(const float* sumsq, float* out2, float max_norm, float inv_world, hipStream_t stream) {
  PXA_CHECK(sumsq && out2, "pxa_clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, out2, max_norm, inv_world);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, const float* gscale, hipStream_t stream) {
  PXA_CHECK(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "pxa_adamw_step: bad args (n must be a multiple of 4)");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2), gscale, (const float*)nullptr);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_clip_coef_scaled(const float* sumsq, float* out2, float max_norm, float inv_world, float* scaler, float growth_factor,
                                    float backoff_factor, int growth_interval, hipStream_t stream) {
  PXA_CHECK(sumsq && out2 && scaler, "pxa_clip_coef_scaled: null pointer");
  PXA_CHECK(growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0, "pxa_clip_coef_scaled: bad scaler constants");
  hipLaunchKernelGGL(clip_coef_scaled_kernel, dim3(1), dim3(1), 0, stream, sumsq, out2, max_norm, inv_world, scaler, growth_factor, backoff_factor,
                     (float)growth_interval);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_adamw_step_scaled(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, const float* gscale, const float* scaler, hipStream_t stream) {
  PXA_CHECK(p && g && m && v && gscale && scaler && n > 0 && n % 4 == 0, "pxa_adamw_step_scaled: bad args (n must be a multiple of 4)");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
                     weight_decay, 1.f, 1.f, gscale, scaler);
  PXA_LAUNCH_CHECK();
  return 0;
}
namespace {
__global__ __launch_bounds__(256) void scale_copy_kernel(const float* __restrict__ x, long xs, bf16_t* __restrict__ yb, float* __restrict__ yf, long ys, long n_scaled,
                                                        long n_total, float scale) {
  const float* xb = x + (long)blockIdx.y * xs;
  const long n4 = n_total / 4;                         // (checked by the entry point: n_scaled and n_total are multiples of 4)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(xb)[i];
    if (4 * i < n_scaled) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
    if (yb) reinterpret_cast<uint2*>(yb + (long)blockIdx.y * ys)[i] = pxa::pack_bf16x4(v.x, v.y, v.z, v.w);
    if (yf) reinterpret_cast<float4*>(yf + (long)blockIdx.y * ys)[i] = v;
  }
}
}  // namespace
extern "C" int pxa_scale_copy_f32(const float* src, long src_stride, void* y_bf16, float* y_f32, long dst_stride, int nblocks, long n_scaled, long n_total,
                                  float scale, hipStream_t stream) {
  PXA_CHECK(src && (y_bf16 || y_f32) && nblocks > 0 && n_total > 0 && n_scaled >= 0 && n_scaled <= n_total, "pxa_scale_copy_f32: bad args");
  PXA_CHECK(n_total % 4 == 0 && n_scaled % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 && ((uintptr_t)src % 16) == 0 &&
            ((uintptr_t)y_bf16 % 8) == 0 && ((uintptr_t)y_f32 % 16) == 0, "pxa_scale_copy_f32: counts / strides must be multiples of 4 elements, pointers 16-byte aligned");
  const long per = (n_total / 4 + 255) / 256;
  hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)(per < 2048 ? per : 2048), nblocks), dim3(256), 0, stream, src, src_stride, (bf16_t*)y_bf16, y_f32, dst_stride,
                     n_scaled, n_total, scale);
  PXA_LAUNCH_CHECK();
  return 0;
}
extern "C" int pxa_cast_f32_bf16(const float* x, void* y_bf16, long n, hipStream_t stream) {
  PXA_CHECK(x && y_bf16 && n > 0, "pxa_cast_f32_bf16: bad args");
  PXA_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)y_bf16 % 8) == 0, "pxa_cast_f32_bf16: unaligned");
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream, x, (bf16_t*)y_bf16, n);
  PXA_LAUNCH_CHECK();
  return 0;
}
