// Kernel-side argument block of the GEMM kernels (filled by pxa_gemm in gemm.hip from pxa_gemm_args) - shared by gemm.hip and gemm_nt4.hip.
#pragma once
#include "common.h"

struct GemmParams {
  const bf16_t* A; const bf16_t* B; int lda, ldb;
  int M, N, K;
  const float* bias; const bf16_t* aux; int ldaux;
  bf16_t* out; bf16_t* out2; int ldo;
  float* outf; int ldf;
  int act, accumulate, k_per_split, tile_hint, split, sched_slot;
  int desc;        // persistent NT / NN kernels: every XCD walks its item range from the end (pxa_gemm_args.items_descending)
  float* slab;
  float* colsum;   // optional [PXA_COLSUM_SLOTS][colsum_stride] partials: += column sums of the bf16 output, staged epilogue only
  long colsum_stride;
  int k_seg;       // segmented-K A operand (implicit 3x3 convolution, layout NT): A[m][k] = A[m*lda + k + (k / k_seg) * seg_jump]
  long seg_jump;   // = a_seg_stride - k_seg
  int k_tap;       // > 0: tap-interleaved K order of the 3x3 convolution (see pxa_gemm_args): [k_tap/64 chunks][3 rows][3 taps][64]
  long tap_s;      // = a_seg_stride (elements between kernel rows)
  // GroupNorm statistics of an implicit-convolution output (persistent SEG instances, EPI 5 / 6): per-channel sum and sum of squares of
  // the bf16 output over the INTERIOR pixels of each image, per QUAD of adjacent channels (GroupNorm groups are multiples of 4 channels
  // wide), added into gn_part[slot][image][N/4][2] (slot = 128-row block % PXA_COLSUM_SLOTS)
  float* gn_part; int gn_img_rows, gn_rp, gn_h, gn_w, gn_B; float gn_inv_rp;
  // one phase of a 3x3 convolution over a 2x nearest-upsampled input (pxa_gemm_args.up_*): interior low-res pixel (py, px) of image b is stored at output row
  // b * up_ip + (2 py - 1) * up_rp + 2 px - 1 + up_off (up_off = dy * up_rp + dx); up_rp = 0: off
  int up_rp, up_ip, up_off;
};

// gemm_nt4.hip: the one-wave-per-SIMD NT kernel.  Returns 1 when the call is not one it takes (the caller goes on to the other kernels), 0 after a launch,
// < 0 on error.
int pxa_gemm_nt4_launch(const GemmParams& p, hipStream_t stream);
