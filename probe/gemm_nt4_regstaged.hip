// EXPERIMENT, not built into the library (round 4, sessions 27-29): gemm_nt4_kernel with REGISTER-STAGED LINE PAIRS instead of the LDS-DMA k-unit stream of
// csrc/gemm_nt4.hip.  Parity-green on both builds (bit-identical to the ping-pong kernel, profiles/r4_27_nt4_regpairs.txt, r4_28_nt4_tile192.txt), slower:
//   fc2 shape (65536 x 1152 x 4608):  ping-pong 0.617 ms | LDS-DMA k32 units, 3 units of look-ahead 0.630 | this file, 256 x 256 items, 2 units 0.672 |
//                                      this file, 256 x 192 items, second buffer in the free accumulator registers, 4 units 0.655 | vendor library 0.535
// What the ablations say (profiles/r4_24_nt4_ablations.txt, r4_25*, r4_29_nt4_ablations.txt): the kernel is bound by the rate at which operand bytes reach the CU
// (~9 TB/s chip-wide with 64-byte row segments, 10-11 TB/s with whole 128-byte lines - the vendor kernel sits at the same 10.7 TB/s); whole lines need the two
// k-halves of a line fetched together, which in 128 KiB of ring leaves two k-units of look-ahead (not enough: 0.67) or costs a narrower item (17 % more bytes: 0.65).
// Kept for the record of the register ownership it demonstrates: VMEM loads into and DS stores from the ACCUMULATOR half ("=a" / "a" operands).
// NT token GEMM, one wave per SIMD (round 4): C[m][n] = sum_k A[m][k] B[n][k] (+ bias), 16-bit output - the forward linears y = x W^T
// (nn.Linear of the reference blocks: PixArt_blocks.py:47-48, 130, 155; PixArtMS.py:66-67, 77) at the token counts of the training step.
//
// Geometry (the vendor library's for this problem: MT 256 x 256, four waves, 128 x 128 per wave; profiles/r4_22_pmc_gemm_sq.txt measured its kernel 10 %
// ahead of the eight-wave ping-pong kernel of gemm.hip on the fc1 shape, with a third fewer LDS reads):
//   workgroup = 4 waves = 256 x 256 outputs (256 x 128 for the half-width remainder column of N = 1152 ...), wave (wm, wn) owns 128 x 128 (128 x 64):
//   8 x 8 (8 x 4) accumulator tiles of v_mfma_f32_16x16x32 = 256 (128) registers, ALL in the accumulator half of the 512-register file ("+a" constraints:
//   every MFMA is inline asm, as in the one-wave attention kernels of attn.hip); the arch half holds two sets of operand fragments (8 A + 8 B row
//   fragments of one k-unit each) so that the reads of unit u+1 run under the MFMAs of unit u.
//   k-units of 32 in a 4-slot LDS ring (A image 256 rows x 64 B + B image 256 rows x 64 B = 32 KiB per slot), filled by LDS-DMA four units ahead as ONE
//   continuous stream across the workgroup's items (the next item's first units arrive under this item's last MFMAs and its epilogue);
//   one barrier per k-unit: at the top of unit u every wave has finished reading unit u (it did so during u-1) and unit u+1 has landed, so the body is
//   64 back-to-back MFMAs with the 16 fragment reads of u+1 in their first half and the 8 DMA pieces of u+4 (into the slot unit u just freed) in the second.
//   Per MFMA: 0.25 LDS reads (ping-pong kernel: 0.375), 1/64 barrier (1/16).
// Epilogue: accumulators -> (+ bias) -> 16-bit -> the wave's private 8 KiB staging slice (XOR-swizzled) -> whole 256-byte row segments, 32 rows at a time;
// the stores drain under the next item's first units (counted vmcnt: they retire in issue order behind the units already in flight).
// Items: XCD-aware order as in gemm.hip (an XCD's 32 workgroups cover 8 m-tiles x 4 n-tiles per round), static persistent split - `b, b + G, ...`.
// Takes: M % 256 == 0, N % 128 == 0, K % 128 == 0, K >= 256, act 0; everything else stays with gemm.hip.  PXA_GEMM_NT4 = 0 / 1 (A/B).
#include "common.h"
#include "gemm_params.h"
#include <cstdlib>

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
#ifndef NT4_ABL
#define NT4_ABL 0            // ablation builds (wrong results, timing only): 1 no LDS-DMA in the loop, 2 no fragment reads in the loop, 4 no epilogue stores, 32 no LDS writes of the staged pieces (their loads stay), 8 the DMA stream as whole 128-byte lines (same bytes, wrong rows), 16 as line halves from consecutive pieces
#endif

// TNB: 16-column accumulator tiles per wave (8: 256-column items, 4: the 128-column remainder items).  n_begin: first output column of this launch's items,
// nt: its number of n-tiles.
template <int TNB, bool BIAS>
__global__ __launch_bounds__(256, 1) void gemm_nt4_kernel(GemmParams p, int n_begin, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NG = 8 + TNB;                                       // line-pair pieces (8 rows x 128 B = two k-units of those rows) per wave: 8 of A, TNB of B
  constexpr int CH = 2 * TNB, CHP = TNB > 4 ? 16 : 8;               // 16-byte chunks per staged output row / the row pitch in chunks (a power of two for the XOR swizzle)
  constexpr int NST = 2 * CHP;                                      // epilogue store instructions per wave and item
  constexpr int DEPTH = TNB == 8 ? 2 : 4;                           // k-units between a piece's load and its LDS write: 4 where a second register buffer fits (the 64
                                                                    // accumulator-half registers the 8 x TNB tiles leave free), else 2
  constexpr int NBL = BIAS ? 1 : 0;                                 // bias pieces per wave and item
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

  // ---- producer: REGISTER-STAGED LINE PAIRS.  A piece = 8 operand rows x 128 bytes = the whole 128-byte lines that hold k-units 2v and 2v+1 of those rows:
  // lane l loads row 8 q + (l >> 3), 16-byte chunk c = l & 7 (global_load_dwordx4: 8 lanes = one line, so the texture cache issues ONE request per line;
  // fetching the two halves a k-unit apart - what the LDS-DMA form of this kernel did - doubles the L2 request count and was measured 20 % slower on the
  // fc2 shape, profiles/r4_25*), holds it in 4 registers for two k-units, then ds_write_b128 puts chunk c into the image of unit 2v + (c >> 2) at
  // (row, chunk (c & 3) ^ swz(row)), swz(row) = 3 ((row >> 3) & 1) (conflict-free for the 16-row fragment reads: gemm.hip kc_swz<true>).
  // Wave w takes A pieces 8 w .. 8 w + 7 and B pieces TNB w .. TNB w + TNB - 1; piece i of a wave has row-block parity i & 1.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 Gv[NG], Ga[DEPTH == 4 ? NG : 1];                           // pieces in flight: arch half / accumulator half (VMEM and DS take either)
  const unsigned strideA = (unsigned)p.lda * 16u, strideB = (unsigned)p.ldb * 16u;      // bytes between consecutive pieces of a wave (8 rows)
  unsigned voffA = (unsigned)((wave * 64 + (lane >> 3)) * p.lda + (lane & 7) * 8) * 2u;
  unsigned voffB = (unsigned)((wave * 8 * TNB + (lane >> 3)) * p.ldb + (lane & 7) * 8) * 2u;
  unsigned wrA[2][2], wrB[2][2];                                    // [row-block parity][slot pair]: LDS byte address of this lane's chunk in piece 0 of the wave
#pragma unroll
  for (int par = 0; par < 2; par++)
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
      const unsigned lanepart = (lane >> 3) * 64 + (((lane & 3) ^ (3 * par)) << 4) + ((lane >> 2) & 1) * NT4_SLOT + sp * 2 * NT4_SLOT;
      wrA[par][sp] = lds0 + wave * 4096 + lanepart;
      wrB[par][sp] = lds0 + 16384 + wave * TNB * 512 + lanepart;
    }
  // (clang: asm operands inside `if constexpr` of a generic lambda cannot name captured locals - they go through local copies)
  auto load_piece = [&](auto ic, auto bufc, const char* a, const char* b) {   // piece ic of the pair at (a, b) into buffer bufc; the running offsets walk the wave's pieces in order
    constexpr int i = decltype(ic)::value, BUF = decltype(bufc)::value;
    if constexpr (NT4_ABL & 1) return;
    u32x4 g;
    const unsigned vo = i < 8 ? voffA : voffB;
    const char* sb = i < 8 ? a : b;
    if constexpr (BUF == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g) : "v"(vo), "s"(sb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=a"(g) : "v"(vo), "s"(sb) : "memory");
    if constexpr (i < 8) { if constexpr (i < 7) voffA += strideA; else voffA -= 7 * strideA; }
    else { if constexpr (i < NG - 1) voffB += strideB; else voffB -= (TNB - 1) * strideB; }
    if constexpr (BUF == 0) Gv[i] = g; else Ga[i] = g;
  };
  auto write_piece = [&](auto ic, auto spc, auto bufc) {           // piece ic of buffer bufc -> the images of the slot pair spc
    constexpr int i = decltype(ic)::value, SP = decltype(spc)::value, BUF = decltype(bufc)::value;
    if constexpr (NT4_ABL & (1 | 32)) return;
    const unsigned wa = i < 8 ? wrA[i & 1][SP] : wrB[i & 1][SP];
    if constexpr (BUF == 0) { const u32x4 g = Gv[i]; asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wa), "v"(g), "n"((i < 8 ? i : i - 8) * 512) : "memory"); }
    else { const u32x4 g = Ga[i]; asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wa), "a"(g), "n"((i < 8 ? i : i - 8) * 512) : "memory"); }
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
  // prologue: pairs (0, 1) and (2, 3) of the first item into the ring, pair (4, 5) (and (6, 7)) into the registers, then the fragments of unit 0
  static_for<NG>([&](auto ic) { load_piece(ic, IntC<0>{}, cA, cB); });
  wait_vm<0>();
  static_for<NG>([&](auto ic) { write_piece(ic, IntC<0>{}, IntC<0>{}); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  static_for<NG>([&](auto ic) { load_piece(ic, IntC<0>{}, cA + 128, cB + 128); });
  wait_vm<0>();
  static_for<NG>([&](auto ic) { write_piece(ic, IntC<1>{}, IntC<0>{}); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  static_for<NG>([&](auto ic) { load_piece(ic, IntC<0>{}, cA + 256, cB + 256); });
  if constexpr (DEPTH == 4) static_for<NG>([&](auto ic) { load_piece(ic, IntC<1>{}, cA + 384, cB + 384); });
  wait_vm<0>();                                                     // (one code path for every item's first odd units: nothing of the prologue is in flight)
  __builtin_amdgcn_s_barrier();
  static_for<8 + TNB>([&](auto rc) { read_frag(rc, IntC<0>{}, 0); });

  // one k-unit.  U = u & 3 (ring slot, register set U & 1), FIRST: the item's first unit (accumulators start from zero), EXTRA: vmcnt allowance for what the
  // item boundary put behind the pair in flight (the previous item's epilogue stores, this item's bias loads).
  // An odd unit u opens with the only barrier of its pair of units: behind it every wave holds units u-1 and u in registers / has consumed them, the images of
  // u+1, u+2 (written two units ago) are complete, and the pair (u+3, u+4) has landed in G.  Under its first MFMAs the unit writes G into the two freed slots
  // and - piece by piece, behind a counted lgkmcnt that proves the write has read its registers - loads the pair (u+5, u+6) into the same registers: every
  // piece has two k-units (~2,000 cycles) from issue to use, issued at an even rate.  Even units carry the fragment reads only.
  auto unit = [&](auto uc, auto firstc, auto extrac, int u) {
    constexpr int U = decltype(uc)::value, CUR = U & 1, NXT = CUR ^ 1;
    constexpr bool FIRST = decltype(firstc)::value, ODD = (U & 1) != 0;
    constexpr int EXTRA = decltype(extrac)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this unit's fragments (read during the previous unit); this wave's image writes
    const char *da = nullptr, *db = nullptr;
    if constexpr (ODD) {
      const int v = u + DEPTH + 3;
      const bool cross = v >= nk;
      da = cross ? nA + (size_t)(v - nk) * 64 : cA + (size_t)v * 64;
      db = cross ? nB + (size_t)(v - nk) * 64 : cB + (size_t)v * 64;
      wait_vm<(DEPTH == 4 ? NG : 0) + EXTRA>();                     // the pair (u + 3, u + 4) is in its buffer (DEPTH 4: the pair behind it may be in flight)
      __builtin_amdgcn_s_barrier();
    }
    constexpr int NM = 8 * TNB, NR = 8 + TNB;
    constexpr int LAG = 4;                                          // a piece is re-loaded LAG MFMAs behind its write
    constexpr int RSTEP = ODD ? 1 : 2;                              // a fragment read behind every RSTEP-th MFMA ...
    constexpr int BUF = DEPTH == 4 ? (U - 1) / 2 : 0;               // odd units alternate between the two register buffers
    constexpr int R0 = ODD ? NG + LAG : 0;                          // ... from the start (even units) / behind the producer's work (odd units)
    static_for<NM>([&](auto tc) {
      constexpr int t = decltype(tc)::value, i = t / TNB, j = t % TNB;
      if constexpr (FIRST) mma0(acc[i][j], fb[CUR][j], fa[CUR][i]); else mma(acc[i][j], fb[CUR][j], fa[CUR][i]);
      if constexpr (ODD && t < NG) write_piece(IntC<t>{}, IntC<(U - 1) / 2>{}, IntC<BUF>{});
      if constexpr (ODD && t >= LAG && t < NG + LAG) {
        constexpr int behind = (t < NG ? t : NG - 1) - (t - LAG);  // writes issued behind the one whose registers are re-used
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(behind) : "memory");
        load_piece(IntC<t - LAG>{}, IntC<BUF>{}, da, db);
      }
      if constexpr (t >= R0 && (t - R0) % RSTEP == RSTEP - 1 && (t - R0) / RSTEP < NR) read_frag(IntC<(t - R0) / RSTEP>{}, IntC<NXT>{}, (U + 1) & 3);
    });
  };

  for (int idx = sx; idx < cnt; idx += G8) {
    const bool has_next = idx + G8 < cnt;
    if (has_next) item_bases(idx + G8, nA, nB, m0n, n0n); else { nA = cA; nB = cB; m0n = m0; n0n = n0; }
    const int mw = m0 + wm * 128, nw = n0 + wn * 16 * TNB;
    // This item's bias columns (this wave's 16 TNB floats) go by ONE LDS-DMA piece into the head of the wave's staging slice - idle until the epilogue - so that
    // no register carries them across the main loop (an asynchronous load into registers the allocator then spills or re-uses is a wrong result, not a
    // slow one).  It sits between the previous item's stores and this item's first loads, where the counted wait of unit 1 expects it (NBL).
    if constexpr (BIAS) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + NT4_RING * NT4_SLOT + wave * NT4_STG);
      const unsigned vo = lane * 16u;
      const char* bsrc = reinterpret_cast<const char*>(p.bias + nw);
      const unsigned long long em = (1ull << (4 * TNB)) - 1;        // 4 TNB lanes x 16 bytes (exec is all ones everywhere else in this kernel: set and reset, not saved)
      asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                   ::"s"(dst), "v"(vo), "s"(bsrc), "s"(em) : "memory");
    }
    unit(IntC<0>{}, IntC<true>{}, IntC<0>{}, 0);
    unit(IntC<1>{}, IntC<false>{}, IntC<NST + NBL>{}, 1);          // (first item: the prologue waited for everything, any allowance is safe - one code path)
    unit(IntC<2>{}, IntC<false>{}, IntC<0>{}, 2);
    unit(IntC<3>{}, IntC<false>{}, IntC<(DEPTH == 4 ? NST + NBL : 0)>{}, 3);
    for (int u = 4; u < nk; u += 4) {
      unit(IntC<0>{}, IntC<false>{}, IntC<0>{}, u);
      unit(IntC<1>{}, IntC<false>{}, IntC<0>{}, u + 1);
      unit(IntC<2>{}, IntC<false>{}, IntC<0>{}, u + 2);
      unit(IntC<3>{}, IntC<false>{}, IntC<0>{}, u + 3);
    }
    // ---- epilogue: 32 rows at a time through the wave's staging slice
    char* stg = smem + NT4_RING * NT4_SLOT + wave * NT4_STG;
    int le = lane;                                                  // a value the compiler must treat as new per item: the epilogue's per-lane constants (24 LDS
    asm volatile("" : "+v"(le));                                    // addresses, row offsets) are recomputed here instead of living - spilled - across the main loop
    f32x4 bias4[TNB];
    if constexpr (BIAS) {                                           // (landed long ago: unit 3's wait; read back before the slice is used for the rows)
#pragma unroll
      for (int j = 0; j < TNB; j++) bias4[j] = *reinterpret_cast<const f32x4*>(stg + (16 * j + 4 * (le >> 4)) * 4);
    }
    mfma_drain();
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
          *reinterpret_cast<uint2*>(stg + rl * (CHP * 16) + ((chunk ^ (rl & (CHP - 1))) << 4) + ((le >> 4) & 1) * 8) = pack_bf16x4(v[0], v[1], v[2], v[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int RPI = 64 / CHP;                                 // rows per 64-lane read
#pragma unroll
      for (int t = 0; t < 32 / RPI; t++) {
        const int rl = t * RPI + le / CHP, ch = le % CHP;
        const uint4 v = *reinterpret_cast<const uint4*>(stg + rl * (CHP * 16) + ((ch ^ (rl & (CHP - 1))) << 4));
        if (!(NT4_ABL & 4) && (CH == CHP || ch < CH)) *reinterpret_cast<uint4*>(p.out + (size_t)(mw + 32 * c + rl) * p.ldo + nw + ch * 8) = v;
      }
    }
    cA = nA; cB = nB; m0 = m0n; n0 = n0n;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the look-ahead fragment reads
  wait_vm<0>();                                                     // the re-fetches behind the last item must not land in a later workgroup's LDS
}

template <int TNB, bool BIAS>
int launch_nt4(const GemmParams& p, int n_begin, int nt, hipStream_t s) {
  static bool attr_set = false;
  static int n_cu = 0;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt4_kernel<TNB, BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, NT4_LDS);
    if (e != hipSuccess) { pxa_set_error("hipFuncSetAttribute(gemm_nt4<%d>): %s", TNB, hipGetErrorString(e)); return -3; }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { pxa_set_error("gemm_nt4: device query failed"); return -3; }
    n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
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
  const bool on = env ? atoi(env) != 0 : PXA_GEMM_NT4_DEFAULT != 0;
  if (!on) return 1;
  if (!p.out || p.outf || p.out2 || p.act != 0 || p.colsum || p.k_seg || p.gn_part || p.split > 1) return 1;
  if (p.M % 256 || p.N % 128 || p.K % 128 || p.K < 256 || p.N < 256 || p.M < 2048) return 1;
  if (p.lda % 8 || p.ldb % 8 || p.ldo % 8 || (reinterpret_cast<uintptr_t>(p.out) & 15) || (reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15)) return 1;
  if ((long)256 * p.lda * 2 >= (1L << 32) || (long)256 * p.ldb * 2 >= (1L << 32)) return 1;     // 32-bit per-lane DMA offsets
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 15)) return 1;
  if (p.N % 192 == 0) return p.bias ? launch_nt4<6, true>(p, 0, p.N / 192, stream) : launch_nt4<6, false>(p, 0, p.N / 192, stream);   // 256 x 192 items, deep pipeline
  const int nf = p.N / 256;
  int rc = p.bias ? launch_nt4<8, true>(p, 0, nf, stream) : launch_nt4<8, false>(p, 0, nf, stream);
  if (rc) return rc;
  if (p.N % 256) rc = p.bias ? launch_nt4<4, true>(p, nf * 256, 1, stream) : launch_nt4<4, false>(p, nf * 256, 1, stream);
  return rc;
}
