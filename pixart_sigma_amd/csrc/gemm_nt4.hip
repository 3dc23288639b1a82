// NT token GEMM, one wave per SIMD (round 4): C[m][n] = sum_k A[m][k] B[n][k] (+ bias), 16-bit output - the forward linears y = x W^T
// (nn.Linear of the reference blocks: PixArt_blocks.py:47-48, 130, 155; PixArtMS.py:66-67, 77) at the token counts of the training step.
//
// Geometry (the vendor library's for this problem: MT 256 x 256, four waves, 128 x 128 per wave; profiles/r4_22_pmc_gemm_sq.txt measured its kernel 10 %
// ahead of the eight-wave ping-pong kernel of gemm.hip on the fc1 shape, with a third fewer LDS reads):
//   workgroup = 4 waves = 256 x 256 outputs (256 x 128 for the half-width remainder column of N = 1152 ...), wave (wm, wn) owns 128 x 128 (128 x 64):
//   8 x 8 (8 x 4) accumulator tiles of v_mfma_f32_16x16x32 = 256 (128) registers, ALL in the accumulator half of the 512-register file ("+a" constraints:
//   every MFMA is inline asm, as in the one-wave attention kernels of attn.hip); the arch half holds two sets of operand fragments (8 A + 8 B row
//   fragments of one k-unit each) so that the reads of unit u+1 run under the MFMAs of unit u.
//   k-units of 32 in a 4-slot LDS ring (A image 256 rows x 64 B + B image 256 rows x 64 B = 32 KiB per slot), filled by LDS-DMA as ONE continuous stream
//   across the workgroup's items (the next item's first units arrive under this item's last MFMAs and its epilogue), in LINE PAIRS (see `unit`): the two
//   64-byte halves of an operand line - units 2v and 2v+1 of the same rows - are fetched by consecutive instructions so that the texture cache sends the L2 one
//   request per line; the kernel is bound by the rate at which operand bytes reach the CU, and that rate is 20 % higher with whole lines.
//   Two barriers per two units (the ping-pong kernel: eight), 0.25 LDS reads per MFMA (0.375).
//   Measured, repeated launches of one shape (profiles/r4_30_nt4_pairs_midbarrier.txt, one box, bf16): qkv 0.437 ms (ping-pong kernel 0.47-0.51, vendor
//   library 0.474), N = K = 1152 projections 0.146 (0.167, 0.152), fc1 shape without GELU 0.557 (0.589, 0.510), fc2 shape 0.615 (0.604, 0.523).  With cold
//   operands (rotating sets, profiles/r4_32_nt4_rotating.txt): qkv 0.455-0.461 (0.469, 0.489), projections 0.181 (0.183, 0.169).  At the MLP shapes the
//   vendor's register-staged pipeline keeps whole lines AND a deeper look-ahead than 128 KiB of ring allow an LDS-DMA stream (probe/gemm_nt4_regstaged.hip:
//   the same idea rebuilt here, parity-green, slower - the experiments and what bounds each).
// Epilogue: accumulators -> (+ bias) -> 16-bit -> the wave's private 8 KiB staging slice (XOR-swizzled) -> whole 256-byte row segments, 32 rows at a time;
// the stores drain under the next item's first units (counted vmcnt: they retire in issue order behind the units already in flight).
// Items: XCD-aware order as in gemm.hip (an XCD's 32 workgroups cover 8 m-tiles x 4 n-tiles per round), static persistent split - `b, b + G, ...`.
// Takes: M % 256 == 0 (>= 2048), N % 128 == 0 (>= 256), K % 128 == 0 (>= 256), act 0, 16-bit output only.  An A/B partner, OFF by default (see
// pxa_gemm_nt4_launch: its bench advantage does not survive cold operands); PXA_GEMM_NT4 = 1 turns it on for every call it can take.
#include "common.h"
#include "gemm_params.h"
#include <cstdlib>
#include <map>
#include <mutex>

namespace {
using namespace pxa;

template <int V> struct IntC { static constexpr int value = V; };
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(IntC<I>{}); static_for<N, I + 1>(f); }
}

#ifdef PXA_OPERAND_F16
#define NT4_MFMA "v_mfma_f32_16x16x32_f16"
#else
#define NT4_MFMA "v_mfma_f32_16x16x32_bf16"
#endif
// d (accumulator half) (+)= X Y, X / Y in the arch half.  The compiler neither knows an asm MFMA's latency nor pads its hazards: the accumulators are read
// only behind mfma_drain().
__device__ __forceinline__ void mma(f32x4& d, const bf16x8& x, const bf16x8& y) { asm volatile(NT4_MFMA " %0, %1, %2, %0" : "+a"(d) : "v"(x), "v"(y)); }
__device__ __forceinline__ void mma0(f32x4& d, const bf16x8& x, const bf16x8& y) { asm volatile(NT4_MFMA " %0, %1, %2, 0" : "=a"(d) : "v"(x), "v"(y)); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
template <int OFF> __device__ __forceinline__ void lds_read16(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// LDS-DMA, saddr form: 16 bytes per lane from (wave-uniform base + per-lane byte offset) to LDS address dst + 16 lane
__device__ __forceinline__ void dma16(unsigned dst, unsigned voff, const char* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int NT4_SLOT = 32768, NT4_RING = 4, NT4_STG = 8192;
constexpr int NT4_LDS = NT4_RING * NT4_SLOT + 4 * NT4_STG;          // 163,840 B: the CU's whole LDS, one workgroup per CU
#ifndef NT4_PAIRS
#define NT4_PAIRS 1          // 1: the LDS-DMA stream in line pairs (below); 0: every unit fetches its own half lines (A/B builds)
#endif
#ifndef NT4_ABL
#define NT4_ABL 0            // ablation builds (wrong results, timing only): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop, 4 no epilogue stores
#endif

// TNB: 16-column accumulator tiles per wave (8: 256-column items, 4: the 128-column remainder items).  n_begin: first output column of this launch's items,
// nt: its number of n-tiles.
template <int TNB, bool BIAS>
__global__ __launch_bounds__(256, 1) void gemm_nt4_kernel(GemmParams p, int n_begin, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NDB = TNB / 2, NDMA = 4 + NDB;                     // LDS-DMA pieces (1 KiB = 16 rows x 64 B) per wave and k-unit: 4 of A, NDB of B
  constexpr int NST = 4 * TNB;                                      // epilogue stores per wave and item (16 bytes per lane each)
  constexpr int CH = 2 * TNB;                                       // 16-byte chunks per staged output row
  constexpr int NBL = BIAS ? TNB : 0;                               // bias loads per wave and item
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  const int nk = p.K / 32;                                          // k-units per item (a multiple of 4: the ring slot of unit u is u & 3 in every item)

  // ---- items of this workgroup: XCD x = blockIdx % 8 owns a contiguous range of the logical order (groups of 8 m-tiles, m fastest)
  const int mt = p.M / 256, T = mt * nt;
  const int G8 = gridDim.x >> 3, x = blockIdx.x & 7, sx = blockIdx.x >> 3;
  const int qx = T >> 3, rx = T & 7, cnt = qx + (x < rx ? 1 : 0), first = x * qx + min(x, rx);
  auto item_bases = [&](int idx, const char*& a, const char*& b, int& m0, int& n0) {
    const int L = first + idx, per_group = 8 * nt, g = L / per_group, first_m = g * 8, gsz = min(mt - first_m, 8), in_g = L - g * per_group;
    m0 = (first_m + in_g % gsz) * 256;
    n0 = n_begin + (in_g / gsz) * (32 * TNB);
    a = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    b = reinterpret_cast<const char*>(p.B + (size_t)n0 * p.ldb);
  };
  if (sx >= cnt) return;                                            // (whole workgroup: nothing issued yet)

  // ---- LDS-DMA plan: piece q of an image = rows 16 q .. 16 q + 15, lane l -> row 16 q + (l >> 2), image chunk l & 3 = global chunk (l & 3) ^ swz(row),
  // swz(row) = 3 ((row >> 3) & 1) (conflict-free for the 16-row fragment reads: gemm.hip kc_swz<true>); wave w takes A pieces 4 w .. 4 w + 3 and B pieces
  // NDB w .. NDB w + NDB - 1
  unsigned offA[4], offB[NDB];
  {
    const int rl = lane >> 2, c = (lane & 3) ^ (3 * ((rl >> 3) & 1));
#pragma unroll
    for (int i = 0; i < 4; i++) offA[i] = (unsigned)(((wave * 4 + i) * 16 + rl) * p.lda + c * 8) * 2u;
#pragma unroll
    for (int i = 0; i < NDB; i++) offB[i] = (unsigned)(((wave * NDB + i) * 16 + rl) * p.ldb + c * 8) * 2u;
  }
  const unsigned dstA = __builtin_amdgcn_readfirstlane(lds0 + wave * 4096), dstB = __builtin_amdgcn_readfirstlane(lds0 + 16384 + wave * NDB * 1024);
  auto issue_piece = [&](auto ic, int slot, const char* a, const char* b) {      // piece ic of this wave's NDMA pieces of one k-unit
    constexpr int i = decltype(ic)::value;
    if constexpr (NT4_ABL & 1) return;
    if constexpr (i < 4) dma16(dstA + slot * NT4_SLOT + i * 1024, offA[i], a);
    else dma16(dstB + slot * NT4_SLOT + (i - 4) * 1024, offB[i - 4], b);
  };
  // ---- fragment reads: lane l reads row (l & 15) of a 16-row block, k-group l >> 4 (image chunk (l >> 4) ^ swz(row))
  unsigned addrA, addrB;
  {
    const int r = lane & 15, ch = (lane >> 4) ^ (3 * ((r >> 3) & 1));
    addrA = lds0 + (wm * 128 + r) * 64 + ch * 16;
    addrB = lds0 + 16384 + (wn * 16 * TNB + r) * 64 + ch * 16;
  }
  bf16x8 fa[2][8], fb[2][TNB];
  auto read_frag = [&](auto rc, auto setc, int slot) {              // read rc of the 8 + TNB fragment reads of one k-unit
    constexpr int r = decltype(rc)::value, S = decltype(setc)::value;
    if constexpr (NT4_ABL & 2) return;
    if constexpr (r < 8) lds_read16<r * 1024>(fa[S][r], addrA + slot * NT4_SLOT);
    else lds_read16<(r - 8) * 1024>(fb[S][r - 8], addrB + slot * NT4_SLOT);
  };
  f32x4 acc[8][TNB];

  // ---- stream state: cA / cB = this item's operand rows, nA / nB = the next item's (or this item's again behind the last one: harmless re-fetches keep
  // every wave's piece count, and with it the counted vmcnt, uniform)
  const char *cA, *cB, *nA, *nB;
  int m0, n0, m0n, n0n;
  item_bases(sx, cA, cB, m0, n0);
  // prologue: units 0 .. 3 of the first item, then the fragments of unit 0
  static_for<4>([&](auto uc) {
    constexpr int u = decltype(uc)::value;
    static_for<NDMA>([&](auto ic) { issue_piece(ic, u, cA + u * 64, cB + u * 64); });
  });
  wait_vm<NT4_PAIRS ? 0 : 3 * NDMA>();
  __builtin_amdgcn_s_barrier();
  static_for<8 + TNB>([&](auto rc) { read_frag(rc, IntC<0>{}, 0); });

  // one k-unit.  U = u & 3 (ring slot, register set U & 1), FIRST: the item's first unit (accumulators start from zero), EXTRA: vmcnt allowance for the
  // previous item's epilogue stores still draining behind the units in flight
  auto unit = [&](auto uc, auto firstc, auto extrac, int u) {
    constexpr int U = decltype(uc)::value, CUR = U & 1, NXT = CUR ^ 1;
    constexpr bool FIRST = decltype(firstc)::value;
    constexpr int EXTRA = decltype(extrac)::value;
    constexpr int NM = 8 * TNB, NR = 8 + TNB;
    if constexpr (NT4_PAIRS) {
      // LINE PAIRS.  The two 64-byte halves of a 128-byte operand line belong to units 2v and 2v+1; fetched a unit apart (the scheme below) they cost the L2 two
      // requests per line and the kernel is bound by that request rate (profiles/r4_24_nt4_ablations.txt, r4_25*: 0.63 ms on the fc2 shape, 0.52 with the same
      // bytes as whole lines).  Fetched by CONSECUTIVE instructions of a wave they merge in the texture cache - but a pair needs two free slots at once.  So:
      // the fragment reads of unit u+1 open unit u (the register set of unit u-1 is free then); an EVEN unit follows them with lgkmcnt(0) + a barrier - now every
      // wave holds units u and u+1 in registers, both slots are free - and issues the pair (u+4, u+5) under the rest of its MFMAs, 2.6 units ahead of its first
      // read; an ODD unit opens with the counted wait for the pair (u+1, u+2) and the barrier that publishes it.  Two barriers per two units, as before.
      constexpr bool ODD = (U & 1) != 0;
      constexpr int RB = NR + 4;                                    // even units: the barrier sits behind MFMA RB - 1
      constexpr int DSTEP = TNB == 8 ? 2 : 1;
      const char *da = nullptr, *db = nullptr;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this unit's fragments
      if constexpr (ODD) {
        wait_vm<2 * NDMA + EXTRA>();                                // the pair (u + 1, u + 2) has landed (the pair issued in unit u - 1 may be in flight)
        __builtin_amdgcn_s_barrier();
      } else {
        const int v = u + 4;
        const bool cross = v >= nk;
        da = cross ? nA + (size_t)(v - nk) * 64 : cA + (size_t)v * 64;
        db = cross ? nB + (size_t)(v - nk) * 64 : cB + (size_t)v * 64;
      }
      static_for<NM>([&](auto tc) {
        constexpr int t = decltype(tc)::value, i = t / TNB, j = t % TNB;
        if constexpr (!ODD && t == RB) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if constexpr (FIRST) mma0(acc[i][j], fb[CUR][j], fa[CUR][i]); else mma(acc[i][j], fb[CUR][j], fa[CUR][i]);
        if constexpr (t < NR) read_frag(IntC<t>{}, IntC<NXT>{}, (U + 1) & 3);
        if constexpr (!ODD && t >= RB && (t - RB) % DSTEP == 0 && (t - RB) / DSTEP < 2 * NDMA) {
          constexpr int q = (t - RB) / DSTEP;                      // piece q / 2, line half q & 1 -> the slot of unit u / u + 1
          issue_piece(IntC<q / 2>{}, U + (q & 1), da + (q & 1) * 64, db + (q & 1) * 64);
        }
      });
      return;
    }
    // (NT4_PAIRS = 0, the first form: every unit fetches its own half lines three units ahead)
    // the DMA pieces this unit issues: unit u + 4 of the stream, into the slot unit u frees
    const bool cross = u + 4 >= nk;
    const char* da = cross ? nA + (size_t)(u + 4 - nk) * 64 : cA + (size_t)(u + 4) * 64;
    const char* db = cross ? nB + (size_t)(u + 4 - nk) * 64 : cB + (size_t)(u + 4) * 64;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this unit's fragments (read during the previous unit)
    wait_vm<2 * NDMA + EXTRA>();                                    // unit u + 1 has landed (u + 2, u + 3 may be in flight)
    __builtin_amdgcn_s_barrier();
    constexpr int RSTEP = TNB == 8 ? 2 : 1;                         // a fragment read behind every RSTEP-th MFMA of the first half
    constexpr int D0 = TNB == 8 ? 33 : 14, DSTEP = TNB == 8 ? 4 : 3;
    static_for<NM>([&](auto tc) {
      constexpr int t = decltype(tc)::value, i = t / TNB, j = t % TNB;
      if constexpr (FIRST) mma0(acc[i][j], fb[CUR][j], fa[CUR][i]); else mma(acc[i][j], fb[CUR][j], fa[CUR][i]);
      if constexpr (t % RSTEP == RSTEP - 1 && t / RSTEP < NR) read_frag(IntC<t / RSTEP>{}, IntC<NXT>{}, (U + 1) & 3);
      if constexpr (t >= D0 && (t - D0) % DSTEP == 0 && (t - D0) / DSTEP < NDMA) issue_piece(IntC<(t - D0) / DSTEP>{}, U, da, db);
    });
  };

  f32x4 bias4[TNB];
  for (int idx = sx; idx < cnt; idx += G8) {
    const bool has_next = idx + G8 < cnt;
    if (has_next) item_bases(idx + G8, nA, nB, m0n, n0n); else { nA = cA; nB = cB; m0n = m0; n0n = n0; }
    const int mw = m0 + wm * 128, nw = n0 + wn * 16 * TNB;
    // TNB bias loads, as asm so that they stay HERE - between the previous item's stores and this item's first DMA pieces, where the counted waits below
    // expect them (NBL); unit 3's wait, which allows only DMA pieces issued behind them, also completes them.  (Only when there is a bias: a load whose
    // result nobody reads leaves its destination registers free for the allocator while the data is still in flight.)
    if constexpr (BIAS) {
      const float* bp = p.bias + nw + 4 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < TNB; j++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4[j]) : "v"(bp + 16 * j) : "memory");
    }
    if constexpr (NT4_PAIRS) {                                      // (first item: the prologue waited for everything, any allowance is safe - one code path)
      unit(IntC<0>{}, IntC<true>{}, IntC<0>{}, 0);
      unit(IntC<1>{}, IntC<false>{}, IntC<NST + NBL>{}, 1);         // behind the pair (2, 3): the previous item's stores, this item's bias loads, the pair (4, 5)
      unit(IntC<2>{}, IntC<false>{}, IntC<0>{}, 2);
      unit(IntC<3>{}, IntC<false>{}, IntC<0>{}, 3);
    } else {
      if (idx == sx) {
        unit(IntC<0>{}, IntC<true>{}, IntC<NBL>{}, 0);
        unit(IntC<1>{}, IntC<false>{}, IntC<NBL>{}, 1);
        unit(IntC<2>{}, IntC<false>{}, IntC<NBL>{}, 2);
      } else {
        unit(IntC<0>{}, IntC<true>{}, IntC<NST + NBL>{}, 0);
        unit(IntC<1>{}, IntC<false>{}, IntC<NST + NBL>{}, 1);
        unit(IntC<2>{}, IntC<false>{}, IntC<NST + NBL>{}, 2);
      }
      unit(IntC<3>{}, IntC<false>{}, IntC<0>{}, 3);
    }
    for (int u = 4; u < nk; u += 4) {
      unit(IntC<0>{}, IntC<false>{}, IntC<0>{}, u);
      unit(IntC<1>{}, IntC<false>{}, IntC<0>{}, u + 1);
      unit(IntC<2>{}, IntC<false>{}, IntC<0>{}, u + 2);
      unit(IntC<3>{}, IntC<false>{}, IntC<0>{}, u + 3);
    }
    mfma_drain();
    // ---- epilogue: 32 rows at a time through the wave's staging slice
    char* stg = smem + NT4_RING * NT4_SLOT + wave * NT4_STG;
    int le = lane;                                                  // a value the compiler must treat as new per item: the epilogue's per-lane constants (24 LDS
    asm volatile("" : "+v"(le));                                    // addresses, row offsets) are recomputed here instead of living - spilled - across the main loop
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        __builtin_amdgcn_sched_barrier(0);                          // (else all 256 accumulator reads are hoisted to the top and spill)
        const int rl = 16 * ib + (le & 15);
#pragma unroll
        for (int j = 0; j < TNB; j++) {
          f32x4 v;                                                  // explicit accumulator reads, in program order (the compiler's own copies are all hoisted to the
          asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"     // epilogue's top: 256 live registers)
                       : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                       : "a"(acc[2 * c + ib][j][0]), "a"(acc[2 * c + ib][j][1]), "a"(acc[2 * c + ib][j][2]), "a"(acc[2 * c + ib][j][3]));
          if constexpr (BIAS) v += bias4[j];
          const int chunk = 2 * j + (le >> 5);
          *reinterpret_cast<uint2*>(stg + rl * (CH * 16) + ((chunk ^ (rl & (CH - 1))) << 4) + ((le >> 4) & 1) * 8) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int RPI = 64 / CH;                                  // rows per 64-lane read
#pragma unroll
      for (int t = 0; t < 32 / RPI; t++) {
        const int rl = t * RPI + le / CH, ch = le % CH;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + rl * (CH * 16) + ((ch ^ (rl & (CH - 1))) << 4));
        if (!(NT4_ABL & 4)) *reinterpret_cast<uint4*>(p.out + (size_t)(mw + 32 * c + rl) * p.ldo + nw + ch * 8) = v;
      }
    }
    cA = nA; cB = nB; m0 = m0n; n0 = n0n;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the look-ahead fragment reads
  wait_vm<0>();                                                     // the re-fetches behind the last item must not land in a later workgroup's LDS
}

// per device: the kernel's LDS request granted?  and the CU count (-1: this part cannot run it - the caller goes on to the other kernels)
int nt4_device_cus(const void* fn) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> state;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lk(mu);
  auto it = state.find({dev, fn});
  if (it != state.end()) return it->second;
  int cus = -1;
  hipDeviceProp_t prop;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NT4_LDS) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  else (void)hipGetLastError();
  state[{dev, fn}] = cus;
  return cus;
}

template <int TNB, bool BIAS>
int launch_nt4(const GemmParams& p, int n_begin, int nt, hipStream_t s) {
  const int n_cu = nt4_device_cus(reinterpret_cast<const void*>(gemm_nt4_kernel<TNB, BIAS>));
  if (n_cu < 8) return 1;
  const int T = (p.M / 256) * nt;
  int g = (T < n_cu ? T : n_cu) & ~7;
  if (g < 8) g = 8;
  hipLaunchKernelGGL((gemm_nt4_kernel<TNB, BIAS>), dim3(g), dim3(256), NT4_LDS, s, p, n_begin, nt);
  PXA_LAUNCH_CHECK();
  return 0;
}
}  // namespace

int pxa_gemm_nt4_launch(const GemmParams& p, hipStream_t stream) {
  const char* env = getenv("PXA_GEMM_NT4");                        // (read per call: tests and benches switch it inside one process)
#ifndef PXA_GEMM_NT4_DEFAULT
#define PXA_GEMM_NT4_DEFAULT 0
#endif
  // OFF by default.  The repeated-launch bench shows this kernel 8-12 % ahead of the ping-pong kernel at the qkv / projection shapes (and ahead of the vendor
  // library there), but that bench serves every launch its operands from L2 / MALL: with rotating operand sets (tools/kbench_nt4.py KB_NT4_ROTATE=6,
  // profiles/r4_32_nt4_rotating.txt) the gain is 2-3 % at qkv and nothing at the projections, and the training step measured the same with and without it
  // (profiles/r4_31_step_ab_nt4.txt: 425.5-426.2 / 425.2-426.0 ms).  PXA_GEMM_NT4 = 1: every call it can take (tests, benches), 0: none.
  const bool on = env ? atoi(env) != 0 : PXA_GEMM_NT4_DEFAULT != 0;
  if (!on) return 1;
  if (!p.out || p.outf || p.out2 || p.act != 0 || p.colsum || p.k_seg || p.gn_part || p.split > 1) return 1;
  if (p.M % 256 || p.N % 128 || p.K % 128 || p.K < 256 || p.N < 256 || p.M < 2048) return 1;
  if (p.lda % 8 || p.ldb % 8 || p.ldo % 8 || (reinterpret_cast<uintptr_t>(p.out) & 15) || (reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15)) return 1;
  if ((long)256 * p.lda * 2 >= (1L << 32) || (long)256 * p.ldb * 2 >= (1L << 32)) return 1;     // 32-bit per-lane DMA offsets
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return 1;
  const int nf = p.N / 256;
  if (p.N % 256) {                                                  // the remainder column's kernel must be runnable too before anything is launched
    const void* f4 = p.bias ? reinterpret_cast<const void*>(gemm_nt4_kernel<4, true>) : reinterpret_cast<const void*>(gemm_nt4_kernel<4, false>);
    if (nt4_device_cus(f4) < 8) return 1;
  }
  int rc = p.bias ? launch_nt4<8, true>(p, 0, nf, stream) : launch_nt4<8, false>(p, 0, nf, stream);
  if (rc) return rc;
  if (p.N % 256) rc = p.bias ? launch_nt4<4, true>(p, nf * 256, 1, stream) : launch_nt4<4, false>(p, nf * 256, 1, stream);
  return rc;
}
