// 256x256 output tiles on a BK = 32 mainloop (64-byte K rows in LDS) for the MFMA-bound shapes of the large grids
// (M >= ~3000 rows: bs = 8 per GPU, the 8-GPU C4 job's per-GPU share, the 30 s C5 clip) - round 4.
//
// Why another mainloop ("BK = 32"): with 128-byte K rows a 256x256 tile cannot keep a ring in 160 KiB
// (one slice of both operands is 64 KiB), so the large grids ran 256x128 tiles whose eight 64x64 wave tiles read 1 KiB of LDS
// fragments per MFMA and pull (256 + 128) * 128 B out of the L2 per 1024 matrix-pipe cycles.  Halving the K extent of a
// slice halves every stage: a 256x256 tile then fits a 6-deep weight ring (6 x 16 KiB) + three activation chunks, its eight
// 128x64 wave tiles read 0.75 KiB per MFMA, and per MFMA it pulls half the weight bytes of a 256x128 tile out of the L2.
//
// Structure: no dedicated loader waves - with 128 accumulator registers per lane the 768-thread form of gemm_ws_impl.h (cap
// 168 registers) is out of reach, and the loads cost no VALU anyway: every wave issues its share of each slice's
// `buffer_load ... lds` pieces (SGPR resource + loop-invariant lane offset + scalar K offset, gemm_ws_impl.h).  The two
// waves of a SIMD (w and w + 4) alternate STRICTLY between the matrix pipe and the memory side - two s_barriers per 32-deep
// slice (the guide's "8-phase" idea at this loop's granularity: 16 MFMAs = 512 matrix-pipe cycles per phase):
//
//   early half : wait(own pieces of slice kt landed) ; barrier A ; issue slice kt+NSB-1, read fragments(kt) ; barrier B ; MFMA(kt)
//   late half  : wait ; barrier A ; MFMA(kt-1)                                    ; barrier B ; issue, read fragments(kt)
//
// Measured (tools/wide_bench.py, M = 4000, one box): w1/w3 249 us where the 256x128 tap-fused tile takes 298 - 326 (271 - 283
// with the same second barrier), w2 134 vs 158, linear2 59 vs 66, fc2 79 vs 92; with ONE barrier per slice (the halves drift
// into matrix || matrix) the same tile took 312 us, and when hipcc was free to move the register-only MFMAs across the
// s_barrier builtins (it sank the late half's block below barrier B) 336 us - hence the sched_barrier fences.  Variants
// measured and dropped: four waves of 128x128 (one per SIMD, software-pipelined, 256 accumulator registers: 296 us), DMA
// issue after the fragment reads (=), inside the MFMA phase (263 us), s_setprio around the MFMA phase (=).
//
// TAPS = 3 is the tap-fused channels-last conv k=3 of the single-stream blocks (mlp_layers.py:104-110: the BM + 2 activation
// rows of a 32-channel chunk are staged once for the three taps - 16 main pieces + one halo piece holding rows m0-1 and
// m0+256, whose other rows are zero-filled by the range check and double as the conv's padding row); TAPS = 1 is a plain
// linear layer.  K order inside a slice: k-step s, lane half kh takes elements [16 kh + 8 s, +8) of the slice - one
// ds_read_b128 of a 32-byte fp8 weight row then feeds both k-steps of its lane half (gemm_ws_impl.h's trick at BK = 64).
// LDS rows are unpadded; 16-byte slot p of row r holds source chunk p ^ ((r >> 2) & 3) (64-byte rows: four rows per
// 256-byte bank row), applied on the source side of the DMA and again by the fragment reads (fp8 rows: p ^ ((r >> 3) & 1)).
// The accumulator tile is 256 KiB, so the epilogue runs as two 256x128 column passes through the (dead) ring, each a
// gemm_common.h epilogue on a virtual tile whose writer waves are the pass's column half.
// reference ops: F.linear, ChannelLastConv1d (mlp_layers.py:45-49,104-110,144-149), hifi_foley.py:218-226,268-289,370.
#pragma once
#include "gemm_ws_impl.h"   // buf_lds16, cvt_fp8x16

namespace {

// loads of the younger slices kt+1 .. kt+NSB-2 that may still be in flight when slice kt must have landed (per wave)
constexpr int wide_inflight(int nsb, int taps, int tap, int ai, int bi) {
  int n = 0;
  for (int j = 1; j <= nsb - 2; ++j) n += bi + (((tap + j) % taps == 0) ? ai : 0);
  return n;
}

template <int N> __device__ __forceinline__ void wide_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SEL: which problem of a two-problem launch (the audio and the visual stream of a two-stream block: hifi_foley.py:218-226,
// 324-334) this workgroup runs - a template parameter for the reason given at gemm_ws_body (constant kernel-argument offsets).
template <typename T, int NW, int TAPS, int NSB, int NAB, int EPI, int WF, int SEL>
__device__ __forceinline__ void gemm_wide_body(const GemmPair& pr) {
  const GemmArgs& g = pr.g[SEL];
  constexpr int BM = 256, BN = 256, BK = 32, ESZ = 2, OOB = 0x7ffffff0;
  constexpr int WM = 2, WN = NW / 2, TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int WSZ = WF ? 1 : 2, BROW = BK * WSZ;       // 64-byte (bf16 / fp16) or 32-byte (fp8) weight rows
  constexpr int APC = 16 + (TAPS == 3 ? 1 : 0);           // 1 KiB pieces of an activation chunk: 16 x 16 rows (+ the halo piece)
  constexpr int ABUF = APC * 1024;
  constexpr int APW = 16 / NW;                            // main activation pieces per wave and chunk
  constexpr int BSL = BN * BROW;                          // bytes per weight slice
  constexpr int BPW = BSL / 1024 / NW;                    // weight pieces per wave and slice
  constexpr int ZROW = 258;                               // a row of the halo piece nobody loads: zeros (the conv's padding)
  constexpr int DUMMY = NAB * ABUF + NSB * BSL;           // scratch KiB: target of the dummy pieces (see `issue`)
  constexpr int AIW = APW + (TAPS == 3 ? 1 : 0);          // activation pieces every wave issues per chunk
  static_assert(NW == 8, "two waves per SIMD (a four-wave form with 128x128 wave tiles - one wave per SIMD, nothing to alternate with - was built and measured: w1/w3 at M = 4000 296 us against 249)");
  static_assert(TAPS == 1 || TAPS == 3, "plain linear or conv k=3");
  static_assert(BPW >= 1 && APW >= 1, "bad tile");
  static_assert(TAPS * NAB >= NSB - 1 + TAPS, "an activation buffer would be refilled while its chunk is still being consumed");
  static_assert(wide_inflight(NSB, TAPS, 0, AIW, BPW) < 64, "vmcnt is a 6-bit counter");
  static_assert(EPI != EPI_SILUGATE_T || (FN % 2 == 0), "gated epilogue needs fragment pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = (int)blockIdx.x - (SEL ? pr.tiles0 : 0);
  {  // bijective XCD remap: consecutive tile ids (same weight panel) share an XCD / L2 (gemm_ws_kernel)
    const int nwg = tiles_m * tiles_n * (EPI == EPI_GATE_RES ? g.ksplit : 1);
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int ks = 0;
  if constexpr (EPI == EPI_GATE_RES) {
    if (g.ks_major) {   // K-range-major order: only the one or two XCDs that run a range fetch its activation columns (gemm_ws_impl.h)
      const int tiles = tiles_m * tiles_n;
      ks = bid / tiles;
      bid -= ks * tiles;
    } else {
      ks = bid % g.ksplit;
      bid /= g.ksplit;
    }
  }
  int tm, tn;
  tile_coords(bid, tiles_m, tiles_n, g.n_groups, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  tl_stamp(g, 0);
  const int C = g.tapC;                 // channels per tap (TAPS == 1: K)
  int kc_begin = 0, nkc = C / BK;       // channel chunks; K ranges of a split are chunk ranges
  if constexpr (EPI == EPI_GATE_RES) {
    const int tot = nkc;
    kc_begin = (int)((long)tot * ks / g.ksplit);
    nkc = (int)((long)tot * (ks + 1) / g.ksplit) - kc_begin;
  }
  const int nk = TAPS * nkc;            // slices: slice kt = (chunk kt / TAPS, tap kt % TAPS)

  // ------------------------------------------------------------------ this wave's share of the loads
  const int lr = lane >> 2, lp = lane & 3;   // row inside a 16-row piece, 16-byte slot
  int vA[APW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int j = (wave * APW + i) * 16 + lr;   // tile row j <-> activation row m0 + j
    const int r = m0 + j;
    vA[i] = r < g.M ? (int)((unsigned)r * (unsigned)(g.lda * ESZ) + (unsigned)((lp ^ ((j >> 2) & 3)) * 16)) : OOB;
  }
  int vH = OOB;                               // halo piece: row 0 = activation row m0 - 1, row 1 = m0 + 256, rows 2.. = zeros
  const bool halo_wave = TAPS == 3 && wave == NW - 1;
  if constexpr (TAPS == 3) {
    const int r = lr == 0 ? m0 - 1 : (lr == 1 ? m0 + BM : -1);
    if (halo_wave && r >= 0 && r < g.M) vH = (int)((unsigned)r * (unsigned)(g.lda * ESZ) + (unsigned)(lp * 16));   // rows 256, 257: swizzle term 0
  }
  int vW[BPW];
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    if constexpr (WF == 0) {
      const int rl = (wave * BPW + i) * 16 + lr;
      const int n = n0 + rl;
      vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)(g.ldw * ESZ) + (unsigned)((lp ^ ((rl >> 2) & 3)) * 16)) : OOB;
    } else {   // fp8: a 1 KiB piece is 32 rows of 32 bytes (two slots)
      const int rl = (wave * BPW + i) * 32 + (lane >> 1);
      const int n = n0 + rl;
      vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)g.ldw + (unsigned)(((lane & 1) ^ ((rl >> 3) & 1)) * 16)) : OOB;
    }
  }
  // Branch-free issue path: the slice's tap is a compile-time value of the unrolled loop, and a slice beyond the end of the K
  // range is issued all the same with an out-of-range SCALAR offset (the range check covers voffset + soffset,
  // tools/ubench/buf_oob.hip): it zero-fills a dead stage, costs no memory traffic, and keeps every wave's vmcnt arithmetic
  // a constant.  For the same reason the waves that do not own the halo piece issue a dummy piece (all lanes out of range)
  // into a scratch KiB behind the ring.
  int is_ch = kc_begin * BK, is_ab = 0, is_bs = 0, is_n = 0;   // the next slice to issue: channel offset, A buffer, B stage, index
  auto issue = [&](int tapc) {
    const bool valid = is_n < nk;
    if (tapc == 0) {
      unsigned char* Ab = lds + is_ab * ABUF;
      const int sA = valid ? is_ch * ESZ : OOB;
#pragma unroll
      for (int i = 0; i < APW; ++i) buf_lds16(g.A, g.a_bytes, Ab + (wave * APW + i) * 1024, vA[i], sA);
      if constexpr (TAPS == 3) buf_lds16(g.A, g.a_bytes, halo_wave ? Ab + 16 * 1024 : lds + DUMMY, vH, sA);
      is_ab = is_ab + 1 == NAB ? 0 : is_ab + 1;
    }
    unsigned char* Bs = lds + NAB * ABUF + is_bs * BSL;
    const int sW = valid ? (tapc * C + is_ch) * WSZ : OOB;
#pragma unroll
    for (int i = 0; i < BPW; ++i) buf_lds16(g.W, g.w_bytes, Bs + (wave * BPW + i) * 1024, vW[i], sW);
    is_bs = is_bs + 1 == NSB ? 0 : is_bs + 1;
    ++is_n;
    if (tapc == TAPS - 1) is_ch += BK;
  };
#pragma unroll
  for (int s = 0; s < NSB - 1; ++s) issue(s % TAPS);

  // ------------------------------------------------------------------ fragment addressing
  const int wm = wave / WN, wn = wave % WN;
  const int fi = lane & 31, kh = lane >> 5;
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // byte offset inside an activation buffer of fragment row i at tap t, k-step 0 / 1 (slot (2 kh + s) ^ swizzle); taps that
  // leave the row's clip read the zero row of the halo piece
  int a_ad[2][TAPS][FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int tr = wm * TM + i * 32 + fi;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      int row = tr;
      if constexpr (TAPS == 3) {
        const int q = (m0 + tr) % g.segV;
        const bool ok = !((t == 0 && q == 0) || (t == 2 && q == g.segV - 1));
        const int s = tr + t - 1;
        row = !ok ? ZROW : (s < 0 ? 256 : (s >= BM ? 257 : s));
      }
      const int a0 = row * 64 + (((2 * kh) ^ ((row >> 2) & 3)) << 4);
      a_ad[0][t][i] = a0;
      a_ad[1][t][i] = a0 ^ 16;
    }
  }
  int b_ad[2][FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = wn * TN + j * 32 + fi;
    if constexpr (WF == 0) {
      b_ad[0][j] = n * 64 + (((2 * kh) ^ ((n >> 2) & 3)) << 4);
      b_ad[1][j] = b_ad[0][j] ^ 16;
    } else {
      b_ad[0][j] = b_ad[1][j] = n * 32 + ((kh ^ ((n >> 3) & 1)) << 4);   // one read: the 16 weights of both k-steps
    }
  }
  bf16x8 fa[2][FM], fb[2][FN];
  u32x4 rawb[FN];
  auto rd_a = [&](int s, const unsigned char* Ab, int tap) {
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[s][i] = *(const bf16x8*)(Ab + a_ad[s][tap][i]);
  };
  auto rd_b = [&](int s, const unsigned char* Bs) {   // fp8: the raw bytes of both k-steps with s == 0, nothing with s == 1
    if constexpr (WF == 0) {
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[s][j] = *(const bf16x8*)(Bs + b_ad[s][j]);
    } else if (s == 0) {
#pragma unroll
      for (int j = 0; j < FN; ++j) rawb[j] = *(const u32x4*)(Bs + b_ad[0][j]);
    }
  };
  auto cvtb = [&]() {
    if constexpr (WF != 0) {
#pragma unroll
      for (int j = 0; j < FN; ++j) cvt_fp8x16<WF, T>(rawb[j], fb[0][j], fb[1][j]);
    }
  };
  auto mm = [&](int s) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = mfma16<T>(fa[s][i], fb[s][j], acc[i][j]);
  };
  // own pieces of slice kt (tap `tap`) have landed once only the loads of the NSB-2 younger slices are in flight
  auto wait_landed = [&](int tap) {
    if (tap == 0) wide_wait_vm<wide_inflight(NSB, TAPS, 0, AIW, BPW)>();
    else if (tap == 1) wide_wait_vm<wide_inflight(NSB, TAPS, 1 % TAPS, AIW, BPW)>();
    else wide_wait_vm<wide_inflight(NSB, TAPS, 2 % TAPS, AIW, BPW)>();
  };
#pragma unroll
  for (int sidx = 0; sidx < 2; ++sidx) {   // fragments start as zeros: the late half's first MFMA block adds nothing
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[sidx][i] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[sidx][j] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) rawb[j] = u32x4{0u, 0u, 0u, 0u};

  int ab = 0, bs = 0;
  {
    // two waves per SIMD (w and w + 4): ping-pong - the early wave reads its fragments while the late wave multiplies the
    // slice it read in the previous iteration, then they swap pipes (gemm_ws_impl.h)
    // Strict alternation (the guide's 8-phase idea at this loop's granularity): TWO barriers per slice, so that between them
    // one wave of every SIMD multiplies (16 MFMAs = 512 matrix-pipe cycles) while the other issues its loads and reads its
    // fragments - never matrix || matrix on one SIMD.  Measured on the 256x128 tap-fused kernel first (w1/w3 at M = 4000:
    // 271 -> 262 us, w2 160 -> 152 us; s_setprio on top of it: nothing).  Each half runs its own straight loop.
    if (wave < NW / 2) {
      for (int kt0 = 0; kt0 < nk; kt0 += TAPS) {
        const unsigned char* Ab = lds + ab * ABUF;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const unsigned char* Bs = lds + NAB * ABUF + bs * BSL;
          wait_landed(tap);
          hard_barrier();                          // A: slice kt has landed; the late half is done reading slice kt-1
          if (kt0 + tap == 0) tl_stamp(g, 1);
          issue((tap + NSB - 1) % TAPS);
          rd_a(0, Ab, tap);
          rd_b(0, Bs);
          rd_a(1, Ab, tap);
          rd_b(1, Bs);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          hard_barrier();                          // B: swap pipes
          cvtb();
          mm(0);
          mm(1);
          bs = bs + 1 == NSB ? 0 : bs + 1;
        }
        ab = ab + 1 == NAB ? 0 : ab + 1;
      }
    } else {
      for (int kt0 = 0; kt0 < nk; kt0 += TAPS) {
        const unsigned char* Ab = lds + ab * ABUF;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const unsigned char* Bs = lds + NAB * ABUF + bs * BSL;
          wait_landed(tap);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // its reads of slice kt-1 have returned: the stage may be refilled
          hard_barrier();                          // A
          cvtb();
          mm(0);
          mm(1);
          hard_barrier();                          // B
          issue((tap + NSB - 1) % TAPS);
          rd_a(0, Ab, tap);
          rd_b(0, Bs);
          rd_a(1, Ab, tap);
          rd_b(1, Bs);
          bs = bs + 1 == NSB ? 0 : bs + 1;
        }
        ab = ab + 1 == NAB ? 0 : ab + 1;
      }
      if (nk > 0) {
        cvtb();
        mm(0);
        mm(1);
      }
    }
  }
  wide_wait_vm<0>();     // the zero-fill pieces issued past the end of the K range must not land in the epilogue's tile
  tl_stamp(g, 2);
  // ------------------------------------------------------------------ epilogue: two 256x128 column passes
  constexpr int WNH = WN / 2;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const bool writer = (wn / WNH) == h;
    const int vtid = ((writer ? 0 : NW / 2) + wm * WNH + (wn % WNH)) * 64 + lane;
    if constexpr (EPI == EPI_QKV_SPLIT) gemm_epilogue_qkv<T, BM, 128, WM, WNH, NW / 2, true>(g, acc, lds, m0, n0 + 128 * h, vtid);
    else gemm_epilogue_lds<T, EPI, BM, 128, WM, WNH, NW / 2>(g, acc, lds, m0, n0 + 128 * h, ks, vtid);
  }
  tl_stamp(g, 3);
}

// PAIR: the launch may carry two problems (workgroups from pr.tiles0 on run the second one) - instantiated for the epilogues of
// the two-stream blocks' plain layers only (fc1's GELU, the q/k/v head split, the gated-residual proj / fc2): two bodies double the
// code of a kernel.
template <typename T, int NW, int TAPS, int NSB, int NAB, int EPI, int WF, bool PAIR>
__global__ __launch_bounds__(NW * 64) void gemm_wide_kernel(const GemmPair pr) {
  if constexpr (PAIR) {
    if ((int)blockIdx.x >= pr.tiles0) gemm_wide_body<T, NW, TAPS, NSB, NAB, EPI, WF, 1>(pr);   // workgroup-uniform
    else gemm_wide_body<T, NW, TAPS, NSB, NAB, EPI, WF, 0>(pr);
  } else {
    gemm_wide_body<T, NW, TAPS, NSB, NAB, EPI, WF, 0>(pr);
  }
}

template <typename T, int NW, int TAPS, int EPI, int WF>
int launch_wide_one(const GemmArgs& g, const GemmArgs* g1, hipStream_t st) {
  // conv k=3: 3 (fp8: 4) activation chunks + 6 (8) weight slices; plain: one activation chunk per weight slice, 5 (6) deep
  constexpr int NSB = TAPS == 3 ? (WF ? 8 : 6) : (WF ? 6 : 5);
  constexpr int NAB = TAPS == 3 ? (NSB + 2 + 2) / 3 : NSB;
  constexpr size_t lds_ring = (size_t)NAB * (16 + (TAPS == 3 ? 1 : 0)) * 1024 + (size_t)NSB * 256 * (WF ? 32 : 64) + (TAPS == 3 ? 1024 : 0);
  constexpr size_t lds_epi = (size_t)256 * 128 * 4;
  constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  constexpr bool PAIR = TAPS == 1 && (EPI == EPI_GELU_T || EPI == EPI_QKV_SPLIT || EPI == EPI_GATE_RES);
  if (g1 && !PAIR) return foley_set_err("256x256 GEMM: two-problem launches exist for the GELU, head-split and gated-residual epilogues of plain layers", __FILE__, __LINE__);
  auto ntiles = [](const GemmArgs& q) { return ((q.M + 255) / 256) * ((q.N + 255) / 256) * (EPI == EPI_GATE_RES ? q.ksplit : 1); };
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g1 ? *g1 : g;
  pr.tiles0 = ntiles(g);
  const int tiles = pr.tiles0 + (g1 ? ntiles(*g1) : 0);
  auto k = gemm_wide_kernel<T, NW, TAPS, NSB, NAB, EPI, WF, PAIR>;
  static std::atomic<unsigned long long> raised{0};
  {
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  FOLEY_LAUNCH(k, dim3(tiles), dim3(NW * 64), lds, st, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T, int NW, int TAPS, int WF>
int launch_wide_fmt(const GemmArgs& g, const GemmArgs* g1, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_STORE_F32: return launch_wide_one<T, NW, TAPS, EPI_STORE_F32, WF>(g, g1, st);
    case EPI_GATE_RES: return launch_wide_one<T, NW, TAPS, EPI_GATE_RES, WF>(g, g1, st);
    case EPI_SILUGATE_T: return launch_wide_one<T, NW, TAPS, EPI_SILUGATE_T, WF>(g, g1, st);
  }
  if constexpr (TAPS == 1) {
    switch (epi) {
      case EPI_GELU_T: return launch_wide_one<T, NW, TAPS, EPI_GELU_T, WF>(g, g1, st);
      case EPI_QKV_SPLIT: return launch_wide_one<T, NW, TAPS, EPI_QKV_SPLIT, WF>(g, g1, st);   // each 128-column half of the tile is one head (two epilogue passes)
    }
  }
  return foley_set_err("256x256 GEMM: unsupported epilogue", __FILE__, __LINE__);
}

// tile: 31 = tap-fused conv k=3, 32 = plain linear layer (eight waves of 128x64 each).  g resolved (ksplit, vec_out, operand
// extents) by gemm_impl.h's launcher.
template <typename T>
int launch_gemm_wide_t(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st) {
  const bool conv = tile == 31;
  for (const GemmArgs* q : {&g, g1}) {
    if (!q) continue;
    if (q->wfmt < 0 || q->wfmt > 2 || q->wfmt != g.wfmt) return foley_set_err("256x256 GEMM: bad weight format", __FILE__, __LINE__);
    if (!q->vec_out && epi != EPI_QKV_SPLIT) return foley_set_err("256x256 GEMM: the problem must qualify for the vector epilogue", __FILE__, __LINE__);
    if (conv ? !(q->taps == 3 && q->dil == 1 && q->tap0 == -1 && q->rstride <= 1 && q->segV == q->segS)
             : !(q->taps == 1 && q->segV >= q->M && q->rstride <= 1 && q->tap0 == 0))
      return foley_set_err("256x256 GEMM: operand addressing not supported by this tile", __FILE__, __LINE__);
    if (q->tapC % 32) return foley_set_err("256x256 GEMM: channels must be a multiple of 32", __FILE__, __LINE__);
  }
#define FOLEY_WIDE_CASE(NWV, TAPSV)                                                       \
  do {                                                                                    \
    if (g.wfmt == 0) return launch_wide_fmt<T, NWV, TAPSV, 0>(g, g1, epi, st);           \
    if (g.wfmt == 1) return launch_wide_fmt<T, NWV, TAPSV, 1>(g, g1, epi, st);           \
    return launch_wide_fmt<T, NWV, TAPSV, 2>(g, g1, epi, st);                             \
  } while (0)
  switch (tile) {
    case 31: FOLEY_WIDE_CASE(8, 3);
    case 32: FOLEY_WIDE_CASE(8, 1);
  }
#undef FOLEY_WIDE_CASE
  return foley_set_err("256x256 GEMM: unknown tile", __FILE__, __LINE__);
}

}  // namespace
