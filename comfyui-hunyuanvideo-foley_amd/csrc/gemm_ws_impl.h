// Wave-specialised GEMM mainloop for gfx950 (16-bit operands: T = bf16_t or f16_t; included by gemm_ws_bf16.hip and
// gemm_ws_f16.hip, one translation unit per operand type so that they compile in parallel): LW loader waves do nothing but feed
// the LDS ring with buffer_load_dwordx4 ... lds, WM*WN consumer waves do nothing but read fragments
// and issue MFMAs; one s_barrier per K-slice couples them.
//
// Why (tools/gemm_timeline.py --ablate, tools/ubench/ldsdma_interfere.hip):
//  * in the single-role loop of gemm_impl.h every wave runs barrier -> issue loads -> read fragments ->
//    MFMA back to back, and the slice time is the SUM of the three (0.73 us per 128x128x64 slice vs
//    0.52 us loads only and 0.46 us fragment reads + MFMA only);
//  * a dedicated loader only keeps its rate next to MFMA-saturated SIMDs if issuing a load needs no
//    VALU instruction: with per-lane 64-bit addresses (global_load_lds) a loader drops from 128 to
//    43 GB/s per CU, with an SGPR buffer resource + loop-invariant lane offset + scalar K offset it
//    stays at 118 GB/s.
//
//   loader wave   : wait(slice kt landed) ; barrier ; issue slice kt+NS-1 into the stage freed by kt-1
//   consumer wave : barrier ; read fragments of slice kt ; MFMA   (late half of the waves: MFMA of kt-1 first)
//
// Same operand addressing as gemm_impl.h (virtual rows / taps; zero fill through the buffer range
// check), same source-side XOR swizzle, same epilogues (gemm_common.h).  reference ops: F.linear,
// ChannelLastConv1d (mlp_layers.py:104-110), see include/foley_hip.h foley_op_gemm.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "gemm_common.h"

namespace {

// buffer_load_dwordx4 ... lds: SGPR resource (base, extent) + per-lane byte offset + scalar byte
// offset; out-of-range lanes write zeros.  Not inside the kernel template: the resource type only
// exists in the device pass.
__device__ __forceinline__ void buf_lds16(const void* base, unsigned bytes, unsigned char* lds_wave_base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// the same with cache-policy bits (aux 2 = nt: a weight stream every line of which is read once or twice and never again)
template <int AUX>
__device__ __forceinline__ void buf_lds16_aux(const void* base, unsigned bytes, unsigned char* lds_wave_base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, AUX);
}

// logical tile id -> (tm, tn).  Default order [panel][M tile]: the M tiles of a weight panel are neighbours (one XCD after the remap).
// Large grids (GemmArgs::n_groups > 0): [panel group][M tile][panel inside the group] - the ~32 workgroups an XCD runs at a time then
// cover a near-square block (a few M tiles x the group's panels) instead of all M tiles of one or two panels, so that every line
// the block pulls out of the fabric is shared by more of them.
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int n_groups, int& tm, int& tn) {
  if (n_groups <= 0 || n_groups > tiles_n) {
    tm = id % tiles_m;
    tn = id / tiles_m;
    return;
  }
  const int P = tiles_n / n_groups, r = tiles_n % n_groups;   // the first r groups hold P + 1 panels
  const int big = r * (P + 1) * tiles_m;
  int pg, p0, local;
  if (id < big) {
    pg = P + 1;
    const int gi = id / (pg * tiles_m);
    local = id - gi * pg * tiles_m;
    p0 = gi * pg;
  } else {
    pg = P;
    const int i2 = id - big, gi = i2 / (pg * tiles_m);
    local = i2 - gi * pg * tiles_m;
    p0 = r * (P + 1) + gi * pg;
  }
  tm = local / pg;
  tn = p0 + local - tm * pg;
}

// s_barrier the compiler may not move MFMAs across: they touch no memory, so nothing else orders them against the builtin (left
// alone hipcc sank a whole MFMA block below the second barrier of the strict-alternation loops, see gemm_wide_impl.h)
__device__ __forceinline__ void hard_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// L2 prefetch: one dword per lane into a pinned scratch register (see the consumer loop); out-of-range
// lanes touch nothing.
__device__ __forceinline__ void buf_prefetch4(const void* base, unsigned bytes, int voff, int soff, int& sink) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  const int so = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(sink) : "v"(voff), "s"(r), "s"(so) : "memory");
}

// fp8 -> bf16 (exact: every e4m3fn / e5m2 value is a bf16 value): 16 weights of one ds_read_b128 become
// the B fragments of two consecutive k-steps.  v_cvt_scalef32_pk_bf16_{fp8,bf8} with scale 1.
template <int WF, typename T>
__device__ __forceinline__ void cvt_fp8x16(const u32x4 w, bf16x8& lo, bf16x8& hi) {
  uint32_t p[8];   // packed pairs in the operand type (bf16 or fp16); both conversions are exact
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (__is_same(T, f16_t)) {
      if constexpr (WF == 1) {
        p[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w[i], 1.0f, false));
        p[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w[i], 1.0f, true));
      } else {
        p[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8((int)w[i], 1.0f, false));
        p[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8((int)w[i], 1.0f, true));
      }
    } else {
      if constexpr (WF == 1) {
        p[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w[i], 1.0f, false));
        p[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)w[i], 1.0f, true));
      } else {
        p[2 * i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8((int)w[i], 1.0f, false));
        p[2 * i + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8((int)w[i], 1.0f, true));
      }
    }
  }
  const u32x4 l = {p[0], p[1], p[2], p[3]}, h = {p[4], p[5], p[6], p[7]};
  lo = __builtin_bit_cast(bf16x8, l);
  hi = __builtin_bit_cast(bf16x8, h);
}

// WF: weight storage of the B operand - 0 bf16, 1 fp8 e4m3fn, 2 fp8 e5m2 (reference FP8WeightWrapper,
// utils.py:316-366: storage fp8, `w.to(x.dtype)` per call, no scales).  fp8 rows are 64 bytes per K-slice:
// the loaders move half the weight bytes (HBM -> L2 -> LDS) and the consumers widen to bf16 in registers.
// SEL: which problem of the pair this workgroup runs - a template parameter, so that every GemmArgs field is
// a kernel-argument load at a CONSTANT offset (hoisted and batched into a few wide s_loads at entry).  With
// a run-time index the compiler re-loaded fields one dword at a time at their points of use: 200 scalar
// loads, 169 of them serialised through the epilogue (tools/kernel_resources.py, gemm_timeline.py --epilogue).
template <typename T, int BM, int BN, int WM, int WN, int NS, int LW, int EPI, int WF, int SEL>
__device__ __forceinline__ void gemm_ws_body(const GemmPair& pr) {
  constexpr int sel = SEL;
  const GemmArgs& g = pr.g[SEL];
  constexpr int NW = WM * WN;
  constexpr int BK = 64, ESZ = 2, OOB = 0x7ffffff0;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int WSZ = WF ? 1 : 2;                    // bytes per weight element
  constexpr int BROW = 64 * WSZ;                     // bytes of one W row per K-slice in LDS
  constexpr int AI = BM / 8 / LW, BI = BN * BROW / 1024 / LW;  // 1 KiB pieces per loader wave and K-slice
  constexpr int STAGE = BM * 128 + BN * BROW;
  // Large grids (256-row tiles, eight consumer waves): a SECOND barrier per slice makes the two consumer waves of a SIMD
  // alternate strictly between the matrix pipe and the LDS - never matrix || matrix (w2 at M = 4000: 160 -> 152 us; the
  // M = 500 tiles lose 7 - 13 % to it: their loops wait for memory, not for the pipe).  See gemm_wide_impl.h.
  constexpr bool TWOB = BM == 256 && WM * WN == 8;
  static_assert(BM % (8 * LW) == 0 && (BN * BROW) % (1024 * LW) == 0 && BI >= 1, "bad tile");
  static_assert((NS - 1) * (AI + BI) < 64, "vmcnt is a 6-bit counter");
  static_assert(EPI != EPI_SILUGATE_T || (FN % 2 == 0), "gated epilogue needs fragment pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = (int)blockIdx.x - (sel ? pr.tiles0 : 0);
  {  // bijective XCD remap: consecutive tile ids (same weight panel) share an XCD / L2
    const int nwg = tiles_m * tiles_n * (EPI == EPI_GATE_RES ? g.ksplit : 1);
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int ks = 0;
  if constexpr (EPI == EPI_GATE_RES) {
    // K-range-major order (round 5): the XCD remap above hands consecutive ids to one XCD, so all tiles of a K range sit on one or
    // two XCDs and only those L2s fetch the range's activation columns (range-fastest order: every XCD fetches ALL of A - 8 x 4 MB
    // of fabric reads per w2 launch against 4 MB of activations)
    if (g.ks_major) {
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: role branch, LDS piece addresses (m0) and tile offsets stay on the SALU
  tl_stamp(g, 0);

  int kt_begin = 0, nk = g.K / BK;
  if constexpr (EPI == EPI_GATE_RES) {
    const int tot = nk;
    kt_begin = (int)((long)tot * ks / g.ksplit);
    nk = (int)((long)tot * (ks + 1) / g.ksplit) - kt_begin;
  }

  if (wave >= NW) {
    // ------------------------------------------------------------------ loader wave
    const int lw = wave - NW;
    const int lr = lane >> 3, lp = lane & 7;  // row within the 8-row piece, LDS chunk position
    int a_base[AI], a_q[AI], vA[AI];   // byte offset of the row at tap offset 0 (< 0: row beyond M)
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int rl = (lw * AI + i) * 8 + lr;  // row inside the tile
      const int r = m0 + rl;
      const int rr = r < g.M ? r : 0;
      const int b = g.segV >= g.M ? 0 : rr / g.segV, q = rr - b * g.segV;   // plain GEMM: one segment, no division
      // 32-bit math: the launcher guarantees both operand extents < 2^31 bytes
      const int qs = q * (g.rstride > 1 ? g.rstride : 1);   // source row of tap offset 0 (strided conv)
      a_base[i] = r < g.M ? (int)((unsigned)(b * g.segS + qs) * (unsigned)(g.lda * ESZ) + (unsigned)((lp ^ ((rl >> 1) & 7)) * 16)) : -1;
      a_q[i] = qs;
    }
    int vW[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      if constexpr (WF == 0) {
        const int rl = (lw * BI + i) * 8 + lr;
        const int n = n0 + rl;
        vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)(g.ldw * ESZ) + (unsigned)((lp ^ ((rl >> 1) & 7)) * 16)) : OOB;
      } else {   // fp8: a 1 KiB piece is 16 rows of 64 bytes (4 chunks); same source-side XOR swizzle, 2 bits
        const int rl = (lw * BI + i) * 16 + (lane >> 2);
        const int n = n0 + rl;
        vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)g.ldw + (unsigned)(((lane & 3) ^ ((rl >> 2) & 3)) * 16)) : OOB;
      }
    }
    auto set_tap = [&](int toff) {   // per-lane offsets of the current tap (VALU, once per tap)
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool in = a_base[i] >= 0 && (unsigned)(a_q[i] + toff) < (unsigned)g.segS;
        vA[i] = in ? a_base[i] + toff * (int)(g.lda * ESZ) : OOB;
      }
    };
    // K-origin rotation (GemmArgs::k_rot; plain layers): the walk starts at slice `rot` of the range and wraps at its end
    const int k_lo = kt_begin * BK, k_hi = (kt_begin + nk) * BK;
    const bool rotate = g.k_rot && g.taps == 1 && tiles_m > 1;
    int ld_k0 = k_lo + (rotate ? (int)((long)tm * nk / tiles_m) * BK : 0);
    int ld_c0 = ld_k0, ld_toff = g.tap0;
    if (kt_begin > 0 && !rotate) {
      const int tap = ld_k0 / g.tapC;
      ld_c0 = ld_k0 - tap * g.tapC;
      ld_toff = g.tap0 + tap * g.dil;
    }
    set_tap(ld_toff);
    auto issue = [&](int stage) {
      unsigned char* As = lds + stage * STAGE;
      unsigned char* Bs = As + BM * 128;
      const int sA = ld_c0 * ESZ, sW = ld_k0 * WSZ;   // scalar K offsets: no VALU on the issue path
#pragma unroll
      for (int i = 0; i < AI; ++i) buf_lds16(g.A, g.a_bytes, As + (lw * AI + i) * 1024, vA[i], sA);
#pragma unroll
      for (int i = 0; i < BI; ++i) buf_lds16(g.W, g.w_bytes, Bs + (lw * BI + i) * 1024, vW[i], sW);
      ld_k0 += BK;
      ld_c0 += BK;
      if (rotate) {            // one tap: channel offset == K offset
        if (ld_k0 >= k_hi) ld_k0 = ld_c0 = k_lo;
      } else if (ld_c0 >= g.tapC) {
        ld_c0 = 0;
        ld_toff += g.dil;
        set_tap(ld_toff);
      }
    };
    // dbg_mode 3 (tools/gemm_timeline.py --prologue): where the ring fill spends its time - entry, ring
    // issued, first slice landed, first barrier passed (wall clock, first loader wave)
    const bool pstamp = g.dbg && (g.dbg_mode & 0xff) == 3 && lw == 0 && lane == 0;
    if (pstamp) g.dbg[(long)blockIdx.x * 4 + 0] = wall_clock64();
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nk) issue(s);
    if (pstamp) g.dbg[(long)blockIdx.x * 4 + 1] = wall_clock64();
    int stage = 0;
    // dbg_mode 2 (tools/gemm_timeline.py --waits): the first loader wave accounts where it waits - for
    // memory (s_waitcnt: the oldest slice has not landed) or at the barrier (the consumers are not done)
    const bool acct = g.dbg && (g.dbg_mode & 0xff) == 2 && lw == 0;
    long long t_mem = 0, t_bar = 0, t_start = acct ? (long long)__builtin_readcyclecounter() : 0;
    for (int kt = 0; kt < nk; ++kt) {
      const long long ta = acct ? (long long)__builtin_readcyclecounter() : 0;
      // slice kt has landed once at most the NS-2 younger slices are still in flight
      if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (AI + BI)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long tb = acct ? (long long)__builtin_readcyclecounter() : 0;
      if (pstamp && kt == 0) g.dbg[(long)blockIdx.x * 4 + 2] = wall_clock64();
      __builtin_amdgcn_s_barrier();  // slice kt visible to the consumers; they are done with slice kt-1
      if (pstamp && kt == 0) g.dbg[(long)blockIdx.x * 4 + 3] = wall_clock64();
      if (acct) {
        const long long tc = (long long)__builtin_readcyclecounter();
        if (kt > 0) { t_mem += tb - ta; t_bar += tc - tb; }   // the ring fill (kt == 0) is the prologue, stamped separately
      }
      if (kt + NS - 1 < nk) issue(stage == 0 ? NS - 1 : stage - 1);
      if constexpr (TWOB) __builtin_amdgcn_s_barrier();   // B: the consumer halves swap pipes (below)
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (acct && lane == 0) {
      g.dbg[(long)blockIdx.x * 4 + 0] = t_mem;
      g.dbg[(long)blockIdx.x * 4 + 1] = t_bar;
      g.dbg[(long)blockIdx.x * 4 + 2] = (long long)__builtin_readcyclecounter() - t_start;
      g.dbg[(long)blockIdx.x * 4 + 3] = nk;
    }
    // Eight-consumer tiles: the loaders are done (a finished wave leaves the barrier count).  Four-consumer
    // tiles: the epilogue's LDS -> global passes would run on half the threads, so the loaders stay and
    // take their share of the rows (they hold no accumulators).
    if constexpr (NW == 4) {
      f32x16 none[FM][FN];
      if constexpr (EPI == EPI_QKV_SPLIT) gemm_epilogue_qkv<T, BM, BN, WM, WN, LW>(g, none, lds, m0, n0);
      else if constexpr (EPI == EPI_QKV_ATTN) gemm_epilogue_qkv<T, BM, BN, WM, WN, LW, true, true>(g, none, lds, m0, n0);
      else gemm_epilogue_lds<T, EPI, BM, BN, WM, WN, LW>(g, none, lds, m0, n0, ks);   // vector epilogue only (gemm_impl.h launcher)
    }
    return;
  }

  // -------------------------------------------------------------------- consumer wave
  const int wm = wave / WN, wn = wave % WN;
  const int fi = lane & 31, kh = lane >> 5;
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int a_row[FM], a_sw[FM], b_row[FN], b_sw[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    a_row[i] = (wm * TM + i * 32 + fi) * 128;
    a_sw[i] = ((wm * TM + i * 32 + fi) >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    b_row[j] = (wn * TN + j * 32 + fi) * BROW;
    b_sw[j] = WF ? ((wn * TN + j * 32 + fi) >> 2) & 3 : ((wn * TN + j * 32 + fi) >> 1) & 7;   // 64-bank LDS: 2 bf16 rows / 4 fp8 rows per bank row
  }
  // K permutation inside a slice, shared by both operands (any consistent one is a valid contraction
  // order): MFMA k-step s = 2t+u takes, for lane half kh, the 8 elements [32t + 16kh + 8u, +8) - so the 16
  // fp8 weights a lane gets from ONE ds_read_b128 (chunk 2t+kh of its 64-byte row) feed k-steps 2t and
  // 2t+1, and the bf16 operands read their 16-byte chunk 4t + 2kh + u.  The bf16-weight kernel uses the
  // same order, which makes fp8 storage bit-identical to the same weights widened at load time.
  auto a_chunk = [&](int s) { return 4 * (s >> 1) + 2 * kh + (s & 1); };
  // ---- L2 prefetch stream (GemmArgs::pf_dist > 0).  The direct-to-LDS ring can only keep NS-1 slices
  // in flight (LDS capacity), so with weights arriving cold from HBM (~2 us under load) a loader moves
  // bytes-in-flight / latency = ~64 GB/s per CU - that, not the LDS or the matrix pipe, bounds the K
  // loop at M = 500.  The consumer waves (their vmcnt is otherwise idle) therefore touch one dword of
  // every 128-byte row line `pf_dist` slices beyond the ring: the line is in this XCD's L2 when the
  // ring asks for it.  The loaded dword is never used; the register is pinned ("+v") so that the
  // asynchronous write-back cannot land in a live register.
  const int PF = g.pf_dist;
  int pf_a = OOB, pf_w = OOB, pf_sink = 0;   // every consumer wave covers BM/NW rows of A and BN/NW rows of W
  if (PF > 0) {
    constexpr int RA = BM / NW, RB = BN / NW;
    static_assert(RA <= 64 && RB <= 64, "prefetch: one line per lane");
    const int r = m0 + wave * RA + lane;
    if (lane < RA && r < g.M) {
      const int b = g.segV >= g.M ? 0 : r / g.segV, q = r - b * g.segV;
      pf_a = (int)((unsigned)(b * g.segS + q * (g.rstride > 1 ? g.rstride : 1)) * (unsigned)(g.lda * ESZ));
    }
    const int n = n0 + wave * RB + lane;
    if (lane < RB && n < g.N) pf_w = (int)((unsigned)n * (unsigned)(g.ldw * ESZ));
  }
  int pf_k0 = (kt_begin + NS - 1 + PF) * BK, pf_c0 = 0, pf_toff = g.tap0;
  if (PF > 0) {
    const int tap = pf_k0 / g.tapC;
    pf_c0 = pf_k0 - tap * g.tapC;
    pf_toff = g.tap0 + tap * g.dil;
  }
  const int pf_end = (kt_begin + nk) * BK;
  auto prefetch = [&]() {   // row validity of conv taps is ignored on purpose: a neighbouring row is real memory,
    if (PF > 0) {           // the array ends are range-checked by the buffer resource
      if (pf_k0 < pf_end) {
        buf_prefetch4(g.A, g.a_bytes, pf_a, (pf_toff * (int)g.lda + pf_c0) * ESZ, pf_sink);
        buf_prefetch4(g.W, g.w_bytes, pf_w, pf_k0 * ESZ, pf_sink);
      }
      pf_k0 += BK;
      pf_c0 += BK;
      if (pf_c0 >= g.tapC) {
        pf_c0 = 0;
        pf_toff += g.dil;
      }
    }
  };
  if constexpr (NW == 4) {
    // ---- one consumer wave per SIMD (wave tile TM x TN >= 64 x 64): fewer, larger wave tiles cut the
    // LDS fragment traffic per MFMA (128x128 tile: 64 KiB of reads per K-slice instead of 96 KiB with
    // eight 32x64 waves).  Without a second wave on the SIMD to fill the matrix pipe while fragments
    // are in flight, the loop is software-pipelined by one k-step across the barrier: fragment reads
    // of step s+1 are issued before the MFMAs of step s, and the MFMAs of a slice's last step run
    // after the next barrier, under the first reads of the next slice.
    bf16x8 fa[2][FM], fb[2][FN];
    u32x4 rawb[FN];   // fp8 weights: the 16 values of one step pair, widened by cvtb()
    auto rda = [&](auto set, int s, const unsigned char* As) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[S][i] = *(const bf16x8*)(As + a_row[i] + ((a_chunk(s) ^ a_sw[i]) << 4));
    };
    auto rdb = [&](auto set, int s, const unsigned char* Bs) {   // bf16: the step's fragments; fp8: the pair's raw bytes (even s only)
      constexpr int S = decltype(set)::value;
      if constexpr (WF == 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[S][j] = *(const bf16x8*)(Bs + b_row[j] + ((a_chunk(s) ^ b_sw[j]) << 4));
      } else if (!(s & 1)) {
#pragma unroll
        for (int j = 0; j < FN; ++j) rawb[j] = *(const u32x4*)(Bs + b_row[j] + (((s + kh) ^ b_sw[j]) << 4));   // chunk 2t + kh, s = 2t
      }
    };
    auto cvtb = [&]() {
      if constexpr (WF != 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j) cvt_fp8x16<WF, T>(rawb[j], fb[0][j], fb[1][j]);
      }
    };
    auto mm = [&](auto set) {
      constexpr int S = decltype(set)::value;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<T>(fa[S][i], fb[S][j], acc[i][j]);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of slice kt-1 has returned: its stage may be refilled
      __builtin_amdgcn_s_barrier();
      if (kt == 0) tl_stamp(g, 1);
      const unsigned char* As = lds + stage * STAGE;
      const unsigned char* Bs = As + BM * 128;
      rda(S0{}, 0, As);
      rdb(S0{}, 0, Bs);
      if (kt > 0) mm(S1{});          // last k-step of the previous slice
      cvtb();                        // fp8: steps 0 and 1 (after the last reader of both fragment sets was issued)
      rda(S1{}, 1, As);
      rdb(S1{}, 1, Bs);
      mm(S0{});
      prefetch();
      rda(S0{}, 2, As);
      rdb(S0{}, 2, Bs);
      mm(S1{});
      cvtb();                        // fp8: steps 2 and 3
      rda(S1{}, 3, As);
      rdb(S1{}, 3, Bs);
      mm(S0{});
      stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (nk > 0) mm(S1{});
  } else {
  // Ping-pong between the two consumer waves of a SIMD (waves w and w + NW/2): after each barrier
  // the early wave reads its fragments while the late wave multiplies the slice it read in the
  // previous iteration, then they swap pipes - the LDS and the matrix core are both busy instead of
  // taking turns.  A late wave's fragment reads must have returned before the barrier that lets a
  // loader overwrite their stage (lgkmcnt(0)).
  constexpr bool PP = NW == 8;
  const bool late = PP && wave >= NW / 2;
  bf16x8 fa[4][FM], fb[4][FN];
  u32x4 rawb[2][FN];   // fp8 weights: raw bytes of the two step pairs, widened right before the MFMAs
  auto cvtb = [&]() {
    if constexpr (WF != 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < FN; ++j) cvt_fp8x16<WF, T>(rawb[t][j], fb[2 * t][j], fb[2 * t + 1][j]);
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<T>(fa[s][i], fb[s][j], acc[i][j]);
  };
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (TWOB) hard_barrier();
    else __builtin_amdgcn_s_barrier();
    if (kt == 0) tl_stamp(g, 1);
    prefetch();
    if (late && kt > 0) {
      cvtb();
      mma();
    }
    if constexpr (TWOB) {   // strict alternation: the late half is done multiplying - barrier B, then it reads
      if (late) hard_barrier();
    }
    const unsigned char* As = lds + stage * STAGE;
    const unsigned char* Bs = As + BM * 128;
    auto reads = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[s][i] = *(const bf16x8*)(As + a_row[i] + ((a_chunk(s) ^ a_sw[i]) << 4));
      if constexpr (WF == 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[s][j] = *(const bf16x8*)(Bs + b_row[j] + ((a_chunk(s) ^ b_sw[j]) << 4));
      } else if (!(s & 1)) {
#pragma unroll
        for (int j = 0; j < FN; ++j) rawb[s >> 1][j] = *(const u32x4*)(Bs + b_row[j] + (((s + kh) ^ b_sw[j]) << 4));
      }
    }
    };
    reads();
    if (!late) {
      if constexpr (TWOB) {   // ... the early half has read its fragments - barrier B, then it multiplies
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        hard_barrier();
      }
      cvtb();
      mma();
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  if (late && nk > 0) cvtb();
  if (late && nk > 0) mma();
  }
  if (PF > 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink)::"memory");   // every prefetch has written back: the register is free again
  tl_stamp(g, 2);
  constexpr int XW = NW == 4 ? LW : 0;   // helper waves of the epilogue (see the loader branch)
  if constexpr (EPI == EPI_QKV_SPLIT) {
    gemm_epilogue_qkv<T, BM, BN, WM, WN, XW, NW == 4 && BM <= 128>(g, acc, lds, m0, n0);
  } else if constexpr (EPI == EPI_QKV_ATTN) {
    static_assert(NW == 4 && BM <= 128, "fused cross attention: four-consumer small tiles");
    gemm_epilogue_qkv<T, BM, BN, WM, WN, XW, true, true>(g, acc, lds, m0, n0);
  } else if constexpr (NW == 4) {
    gemm_epilogue_lds<T, EPI, BM, BN, WM, WN, XW>(g, acc, lds, m0, n0, ks);   // the launcher sends scalar-epilogue problems to the eight-consumer twins
  } else {
    if (g.vec_out) gemm_epilogue_lds<T, EPI, BM, BN, WM, WN, XW>(g, acc, lds, m0, n0, ks);
    else gemm_epilogue<T, EPI, FM, FN, TM, TN>(g, acc, m0, n0, wm, wn, fi, kh, ks);
  }
  tl_stamp(g, 3);
  if (g.dbg && (g.dbg_mode & 0xff) == 4 && tid == 0) g.dbg[(long)blockIdx.x * 4 + 3] = wall_clock64();
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int LW, int EPI, int WF>
__global__ __launch_bounds__((WM * WN + LW) * 64) void gemm_ws_kernel(const GemmPair pr) {
  if ((int)blockIdx.x >= pr.tiles0) gemm_ws_body<T, BM, BN, WM, WN, NS, LW, EPI, WF, 1>(pr);   // workgroup-uniform
  else gemm_ws_body<T, BM, BN, WM, WN, NS, LW, EPI, WF, 0>(pr);
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised, TAP-FUSED channels-last conv k=3 (pad 1): ChannelLastConv1d of the single-stream
// blocks (mlp_layers.py:104-110: linear1, ConvMLP w1/w3, w2 - two thirds of the model's FLOPs).
//
// At M = 500 the K loop of the generic mainloop is bound by the bytes the 256 CUs pull out of the
// L2s (~15 TB/s, tools/gemm_timeline.py --waits), and half of those bytes are the activation tile,
// streamed once PER TAP.  Here K is walked channel-chunk major: the BM+2 activation rows of a
// 64-channel chunk are staged ONCE (A buffers, 3 deep) and serve all three taps - the tap is a row
// offset 0/1/2 into the staged rows - while the weights of the three taps stream through a ring
// of 16 KiB (fp8: 8 KiB) slices.  Per MFMA a third fewer bytes leave the L2.  Rows whose neighbour
// lies outside their clip read a zero row instead (the conv's padding).  Loader / consumer roles,
// barrier protocol, K permutation, fp8 weight widening and epilogues are those of gemm_ws_kernel.
// loads of the younger slices kt+1 .. kt+NSB-2 that may still be in flight when slice kt must have landed:
// one weight group each, plus an activation chunk for every one of them that starts a chunk (tap 0)
constexpr int conv3_inflight(int nsb, int tap, int ai, int bi) {
  int n = 0;
  for (int j = 1; j <= nsb - 2; ++j) n += bi + (((tap + j) % 3 == 0) ? ai : 0);
  return n;
}

template <typename T, int BM, int BN, int WM, int WN, int NSB, int NAB, int LW, int EPI, int WF>
__global__ __launch_bounds__((WM * WN + LW) * 64) void gemm_ws_conv3_kernel(const GemmPair pr) {
  const GemmArgs& g = pr.g[0];
  constexpr int NW = WM * WN;
  constexpr int BK = 64, ESZ = 2, OOB = 0x7ffffff0;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int WSZ = WF ? 1 : 2, BROW = 64 * WSZ;
  constexpr int APC = (BM + 2 + 7) / 8;                 // 1 KiB pieces of an activation chunk (BM+2 rows)
  constexpr int AI = (APC + LW - 1) / LW;               // per loader wave (the last wave's surplus pieces stay out of range)
  constexpr int ABUF = AI * LW * 1024;                  // bytes per activation buffer
  constexpr int BI = BN * BROW / 1024 / LW;             // weight pieces per loader wave and tap slice
  constexpr int BSL = BN * BROW;                        // bytes per weight slice
  constexpr int ZOFF = NAB * ABUF + NSB * BSL;          // 128 zero bytes
  constexpr bool TWOB = BM >= 192 && BN == 128;         // large-grid forms: second barrier per slice (see gemm_ws_body; w1/w3 at M = 4000: 271 -> 262 us)
  static_assert(NW == 8 && (BN * BROW) % (1024 * LW) == 0 && BI >= 1, "bad tile");
  static_assert(3 * NAB >= NSB + 2, "an activation buffer would be refilled while its chunk is still being consumed");
  static_assert(conv3_inflight(NSB, 0, AI, BI) < 64 && conv3_inflight(NSB, 1, AI, BI) < 64 && conv3_inflight(NSB, 2, AI, BI) < 64,
                "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = (int)blockIdx.x;
  {  // bijective XCD remap (see gemm_ws_kernel)
    const int nwg = tiles_m * tiles_n * (EPI == EPI_GATE_RES ? g.ksplit : 1);
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int ks = 0;
  if constexpr (EPI == EPI_GATE_RES) {
    // K-range-major order (round 5): the XCD remap above hands consecutive ids to one XCD, so all tiles of a K range sit on one or
    // two XCDs and only those L2s fetch the range's activation columns (range-fastest order: every XCD fetches ALL of A - 8 x 4 MB
    // of fabric reads per w2 launch against 4 MB of activations)
    if (g.ks_major) {
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
  // scalar wave index (see gemm_ws_body) - except in the 256-row fp8 instantiations, which sit at the 168-register
  // cap of a 768-thread kernel and answer the change with 150 spilled registers
  const int wave = (BM == 256 && WF != 0) ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);
  tl_stamp(g, 0);
  const int C = g.tapC;
  int kc_begin = 0, nkc = C / BK;   // channel chunks; K ranges of a split are chunk ranges
  if constexpr (EPI == EPI_GATE_RES) {
    const int tot = nkc;
    kc_begin = (int)((long)tot * ks / g.ksplit);
    nkc = (int)((long)tot * (ks + 1) / g.ksplit) - kc_begin;
  }
  const int nk = 3 * nkc;           // tap slices: slice kt = (chunk kt / 3, tap kt % 3)

  if (wave >= NW) {
    // ------------------------------------------------------------------ loader wave
    const int lw = wave - NW;
    const int lr = lane >> 3, lp = lane & 7;
    int vA[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int j = (lw * AI + i) * 8 + lr;     // staged row j <-> activation row m0 - 1 + j
      const int r = m0 - 1 + j;
      vA[i] = (j < BM + 2 && r >= 0 && r < g.M) ? (int)((unsigned)r * (unsigned)(g.lda * ESZ) + (unsigned)((lp ^ ((j >> 1) & 7)) * 16)) : OOB;
    }
    int vW[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      if constexpr (WF == 0) {
        const int rl = (lw * BI + i) * 8 + lr;
        const int n = n0 + rl;
        vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)(g.ldw * ESZ) + (unsigned)((lp ^ ((rl >> 1) & 7)) * 16)) : OOB;
      } else {
        const int rl = (lw * BI + i) * 16 + (lane >> 2);
        const int n = n0 + rl;
        vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)g.ldw + (unsigned)(((lane & 3) ^ ((rl >> 2) & 3)) * 16)) : OOB;
      }
    }
    // issue slice `sl` (local index): its weight slice, and - for tap 0 - the activation chunk
    auto issue = [&](int sl) {
      const int c = sl / 3, tap = sl - 3 * c;
      const int ch = (kc_begin + c) * BK;
      if (tap == 0) {
        unsigned char* Ab = lds + (c % NAB) * ABUF;
#pragma unroll
        for (int i = 0; i < AI; ++i) buf_lds16(g.A, g.a_bytes, Ab + (lw * AI + i) * 1024, vA[i], ch * ESZ);
      }
      unsigned char* Bs = lds + NAB * ABUF + (sl % NSB) * BSL;
      const int sW = (tap * C + ch) * WSZ;
#pragma unroll
      for (int i = 0; i < BI; ++i) buf_lds16(g.W, g.w_bytes, Bs + (lw * BI + i) * 1024, vW[i], sW);
    };
#pragma unroll
    for (int sl = 0; sl < NSB - 1; ++sl)
      if (sl < nk) issue(sl);
    // slice kt has landed once only the loads of the NSB-2 younger slices are in flight (conv3_inflight)
    for (int kt0 = 0; kt0 < nk; kt0 += 3) {
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int kt = kt0 + tap;
        if (kt + NSB - 2 < nk) {
          if (tap == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(conv3_inflight(NSB, 0, AI, BI)) : "memory");
          else if (tap == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(conv3_inflight(NSB, 1, AI, BI)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(conv3_inflight(NSB, 2, AI, BI)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NSB - 1 < nk) issue(kt + NSB - 1);
        if constexpr (TWOB) __builtin_amdgcn_s_barrier();   // B: the consumer halves swap pipes (below)
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumer wave
  const int wm = wave / WN, wn = wave % WN;
  const int fi = lane & 31, kh = lane >> 5;
  if (tid < 8) *(u32x4*)(lds + ZOFF + tid * 16) = u32x4{0u, 0u, 0u, 0u};   // the zero row (visible after the first barrier)
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // staged row of fragment row i at tap t is j = tile row + t; taps that leave the clip read the zero row
  int a_off[3][FM], a_swz[3][FM];
  bool a_ok[3][FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int tr = wm * TM + i * 32 + fi;
    const int q = (m0 + tr) % g.segV;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      a_ok[t][i] = !((t == 0 && q == 0) || (t == 2 && q == g.segV - 1));
      a_off[t][i] = (tr + t) * 128;
      a_swz[t][i] = ((tr + t) >> 1) & 7;
    }
  }
  int b_row[FN], b_sw[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    b_row[j] = (wn * TN + j * 32 + fi) * BROW;
    b_sw[j] = WF ? ((wn * TN + j * 32 + fi) >> 2) & 3 : ((wn * TN + j * 32 + fi) >> 1) & 7;   // 64-bank LDS: 2 bf16 rows / 4 fp8 rows per bank row
  }
  auto a_chunk = [&](int s) { return 4 * (s >> 1) + 2 * kh + (s & 1); };   // K permutation of gemm_ws_kernel
  const bool late = wave >= NW / 2;
  bf16x8 fa[4][FM], fb[4][FN];
  u32x4 rawb[2][FN];
  auto cvtb = [&]() {
    if constexpr (WF != 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < FN; ++j) cvt_fp8x16<WF, T>(rawb[t][j], fb[2 * t][j], fb[2 * t + 1][j]);
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma16<T>(fa[s][i], fb[s][j], acc[i][j]);
  };
  const unsigned char* zrow = lds + ZOFF;
  for (int kt0 = 0; kt0 < nk; kt0 += 3) {
    const unsigned char* Ab = lds + ((kt0 / 3) % NAB) * ABUF;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int kt = kt0 + tap;
      if (late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (TWOB) hard_barrier();
      else __builtin_amdgcn_s_barrier();
      if (kt == 0) tl_stamp(g, 1);
      if (late && kt > 0) {
        cvtb();
        mma();
      }
      if constexpr (TWOB) {   // strict alternation: the late half is done multiplying - barrier B, then it reads
        if (late) hard_barrier();
      }
      const unsigned char* Bs = lds + NAB * ABUF + (kt % NSB) * BSL;
      auto reads = [&]() {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const unsigned char* src = a_ok[tap][i] ? Ab + a_off[tap][i] + ((a_chunk(s) ^ a_swz[tap][i]) << 4) : zrow;
          fa[s][i] = *(const bf16x8*)src;
        }
        if constexpr (WF == 0) {
#pragma unroll
          for (int j = 0; j < FN; ++j) fb[s][j] = *(const bf16x8*)(Bs + b_row[j] + ((a_chunk(s) ^ b_sw[j]) << 4));
        } else if (!(s & 1)) {
#pragma unroll
          for (int j = 0; j < FN; ++j) rawb[s >> 1][j] = *(const u32x4*)(Bs + b_row[j] + (((s + kh) ^ b_sw[j]) << 4));
        }
      }
      };
      reads();
      if (!late) {
        if constexpr (TWOB) {   // ... the early half has read its fragments - barrier B, then it multiplies
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          hard_barrier();
        }
        cvtb();
        mma();
      }
    }
  }
  if (late && nk > 0) {
    cvtb();
    mma();
  }
  tl_stamp(g, 2);
  if (g.vec_out) gemm_epilogue_lds<T, EPI, BM, BN, WM, WN>(g, acc, lds, m0, n0, ks);
  else gemm_epilogue<T, EPI, FM, FN, TM, TN>(g, acc, m0, n0, wm, wn, fi, kh, ks);
  tl_stamp(g, 3);
}

// ---------------------------------------------------------------------------------------------
// Round 5: the 256x64 tap-fused conv tile with K-SPLIT WAVE PAIRS and slice read-ahead (tile 22 at small M: w1 / w3, w2, linear1
// of the single-stream blocks = 39 % of a bs=1 iteration).  tools/ubench/vmem_paths.hip + mfma_rate.hip (profiles/r05_*):
//  * the matrix pipe alone sustains one s_barrier per 16-MFMA slice at 515 cycles per slice (512 = full), also next to a
//    20 KiB-per-slice LDS-DMA stream - but the early / late consumer halves above take 740 - 850: after the barrier a wave
//    either requests fragments and waits for them before it multiplies, or multiplies and only then requests, so every slice
//    exposes one LDS round trip and the two waves of a SIMD multiply at once half of the time;
//  * here every wave, after the barrier of slice kt, multiplies slice kt-1 from a second fragment register set and requests the
//    fragments of slice kt IN THE SHADOW of those MFMAs (two ds_read_b128 behind each of the first four MFMAs, the last four
//    cover their latency; order pinned with sched_group_barrier - left alone hipcc re-merges the two sets);
//  * two fragment sets of a 32x64 wave tile would be 96 registers (the tile spills at the 168-register cap of a 768-thread
//    kernel), so waves w and w + 4 - the two consumer waves of a SIMD - share the 64x64 tile of rows (w & 3) * 64 and SPLIT THE
//    K-STEPS of every slice (0 / 1 and 2 / 3): 8 fragment reads per 8 MFMAs instead of 12, two sets = 64 registers, 64
//    accumulator registers; the partner's partial tile is added once, through the dead ring, before the epilogue.
// Loader waves, ring protocol, K permutation, zero padding and epilogues are those of gemm_ws_conv3_kernel<256, 64, 8, 1, 6, 3, 4>.
template <typename T, int EPI, int WNT>
__global__ __launch_bounds__(768) void gemm_ws_conv3_ks_kernel(const GemmPair pr) {
  const GemmArgs& g = pr.g[0];
  constexpr int BM = 256, BN = 64, LW = 4, NSB = 6, NAB = 3, NW = 8;
  constexpr int BK = 64, ESZ = 2, OOB = 0x7ffffff0;
  constexpr int APC = (BM + 2 + 7) / 8, AI = (APC + LW - 1) / LW, ABUF = AI * LW * 1024;
  constexpr int BI = BN * 128 / 1024 / LW, BSL = BN * 128;
  constexpr int ZROW = 8 * APC;     // first staged row of the surplus pieces: out of range for every lane, i.e. zero-filled by the DMA
  static_assert(ZROW + 8 <= AI * LW * 8, "the zero rows live in the surplus pieces of an activation buffer");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = (int)blockIdx.x;
  {  // bijective XCD remap (see gemm_ws_kernel)
    const int nwg = tiles_m * tiles_n * (EPI == EPI_GATE_RES ? g.ksplit : 1);
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int ks = 0;
  if constexpr (EPI == EPI_GATE_RES) {
    // K-range-major order (round 5): the XCD remap above hands consecutive ids to one XCD, so all tiles of a K range sit on one or
    // two XCDs and only those L2s fetch the range's activation columns (range-fastest order: every XCD fetches ALL of A - 8 x 4 MB
    // of fabric reads per w2 launch against 4 MB of activations)
    if (g.ks_major) {
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
  const int C = g.tapC;
  int kc_begin = 0, nkc = C / BK;
  if constexpr (EPI == EPI_GATE_RES) {
    const int tot = nkc;
    kc_begin = (int)((long)tot * ks / g.ksplit);
    nkc = (int)((long)tot * (ks + 1) / g.ksplit) - kc_begin;
  }
  const int nk = 3 * nkc;

  if (wave >= NW) {
    // ------------------------------------------------------------------ loader wave (gemm_ws_conv3_kernel's)
    const int lw = wave - NW;
    const int lr = lane >> 3, lp = lane & 7;
    int vA[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int j = (lw * AI + i) * 8 + lr;
      const int r = m0 - 1 + j;
      vA[i] = (j < BM + 2 && r >= 0 && r < g.M) ? (int)((unsigned)r * (unsigned)(g.lda * ESZ) + (unsigned)((lp ^ ((j >> 1) & 7)) * 16)) : OOB;
    }
    int vW[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int rl = (lw * BI + i) * 8 + lr;
      const int n = n0 + rl;
      vW[i] = (n < g.N) ? (int)((unsigned)n * (unsigned)(g.ldw * ESZ) + (unsigned)((lp ^ ((rl >> 1) & 7)) * 16)) : OOB;
    }
    auto issue = [&](int sl) {
      const int c = sl / 3, tap = sl - 3 * c;
      const int ch = (kc_begin + c) * BK;
      if (tap == 0) {
        unsigned char* Ab = lds + (c % NAB) * ABUF;
#pragma unroll
        for (int i = 0; i < AI; ++i) buf_lds16(g.A, g.a_bytes, Ab + (lw * AI + i) * 1024, vA[i], ch * ESZ);
      }
      unsigned char* Bs = lds + NAB * ABUF + (sl % NSB) * BSL;
      const int sW = (tap * C + ch) * ESZ;
#pragma unroll
      for (int i = 0; i < BI; ++i) buf_lds16_aux<WNT ? 2 : 0>(g.W, g.w_bytes, Bs + (lw * BI + i) * 1024, vW[i], sW);
    };
#pragma unroll
    for (int sl = 0; sl < NSB - 1; ++sl)
      if (sl < nk) issue(sl);
    for (int kt0 = 0; kt0 < nk; kt0 += 3) {
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int kt = kt0 + tap;
        if (kt + NSB - 2 < nk) {
          if (tap == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(conv3_inflight(NSB, 0, AI, BI)) : "memory");
          else if (tap == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(conv3_inflight(NSB, 1, AI, BI)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(conv3_inflight(NSB, 2, AI, BI)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + NSB - 1 < nk) issue(kt + NSB - 1);
      }
    }
    return;
  }

  // -------------------------------------------------------------------- consumer wave: 64 rows x 64 columns x half of the k-steps
  const int wq = wave & 3, kq = wave >> 2;
  const int fi = lane & 31, kh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // fragment row i at tap t reads staged row (tile row + t); taps that leave the clip read a zero row of the same buffer
  // LDS byte offsets are kept as ONE register per (tap, row fragment) / per column fragment - k-step 0 of this wave; k-step 1 is
  // the neighbouring 16-byte chunk (offset ^ 16) and the buffer / stage base is a scalar: both are folded in at the point of
  // use through opaque scalars, so that the compiler cannot hoist 24 + 8 finished addresses out of the loop (it did: 178 spilled
  // registers at the 168-register cap of a 768-thread kernel)
  int a_adr[3][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int tr = wq * 64 + i * 32 + fi;
    const int q = (m0 + tr) % g.segV;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool ok = !((t == 0 && q == 0) || (t == 2 && q == g.segV - 1));
      const int j = ok ? tr + t : ZROW;
      a_adr[t][i] = j * 128 + (((4 * kq + 2 * kh) ^ ((j >> 1) & 7)) << 4);   // chunk a_chunk(2 kq), see gemm_ws_body
    }
  }
  int b_adr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = j * 32 + fi;
    b_adr[j] = NAB * ABUF + r * 128 + (((4 * kq + 2 * kh) ^ ((r >> 1) & 7)) << 4);
  }
  bf16x8 fa[2][2][2], fb[2][2][2];   // [set][k-step][fragment]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      fa[1][s][i] = __builtin_bit_cast(bf16x8, z);   // the first step multiplies set 1 before anything was read into it: zeros
      fb[1][s][i] = __builtin_bit_cast(bf16x8, z);
    }
  // one step: barrier of slice kt, then MFMAs of the previous slice with this slice's fragment requests in their shadow
  auto step = [&](auto set, auto tapc, int ab_off, int stage) {
    constexpr int S = decltype(set)::value, TAP = decltype(tapc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's requests of the previous slice have returned: its stage may be refilled
    hard_barrier();
    int ab = ab_off, bs = stage * BSL, sx = 16;
    asm volatile("" : "+s"(ab), "+s"(bs), "+s"(sx));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[S][s][i] = *(const bf16x8*)(lds + (ab + (s ? (a_adr[TAP][i] ^ sx) : a_adr[TAP][i])));
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[S][s][j] = *(const bf16x8*)(lds + (bs + (s ? (b_adr[j] ^ sx) : b_adr[j])));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T>(fa[S ^ 1][s][i], fb[S ^ 1][s][j], acc[i][j]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // (the address arithmetic of the next two requests)
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // ... two fragment requests behind it
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  // slices alternate between the two sets; a chunk is three slices, so chunks alternate between (0,1,0) and (1,0,1); the ring
  // stage of slice kt is kt % 6 = 3 * (chunk parity) + tap
  int c = 0;
  for (; c + 2 <= nkc; c += 2) {
    const int Ab0 = (c % NAB) * ABUF, Ab1 = ((c + 1) % NAB) * ABUF;
    step(S0{}, T0{}, Ab0, 0);
    if (c == 0) tl_stamp(g, 1);
    step(S1{}, T1{}, Ab0, 1);
    step(S0{}, T2{}, Ab0, 2);
    step(S1{}, T0{}, Ab1, 3);
    step(S0{}, T1{}, Ab1, 4);
    step(S1{}, T2{}, Ab1, 5);
  }
  if (c < nkc) {   // odd number of chunks: the tail chunk starts on set 0 / stage 0 again (an even number of chunks came before)
    const int Ab0 = (c % NAB) * ABUF;
    step(S0{}, T0{}, Ab0, 0);
    step(S1{}, T1{}, Ab0, 1);
    step(S0{}, T2{}, Ab0, 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // the last slice sits in set 0: hand it to the common tail below
        fa[1][s][i] = fa[0][s][i];
        fb[1][s][i] = fb[0][s][i];
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < 2; ++s)     // the last slice (zeros if there was none)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<T>(fa[1][s][i], fb[1][s][j], acc[i][j]);
  tl_stamp(g, 2);
  // ---- the partner's partial tile (k-steps 2 / 3 of every slice) joins through the dead ring: lane-major 16-byte records
  __syncthreads();   // the eight consumer waves (the loaders have left): every fragment read of the last slice has returned
  {
    f32x4* xch = (f32x4*)lds + (wq * 16) * 64 + lane;
    if (kq == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const f32x4 t = {acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
            xch[((i * 2 + j) * 4 + v) * 64] = t;
          }
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const f32x4 t = xch[((i * 2 + j) * 4 + v) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * v + e] += t[e];
          }
    }
  }
  // waves 0..3 own the summed 64x64 tiles (WM = 4, WN = 1); waves 4..7 take their share of the LDS -> global passes
  gemm_epilogue_lds<T, EPI, BM, BN, 4, 1, 4>(g, acc, lds, m0, n0, ks);
  tl_stamp(g, 3);
}

template <typename T, int EPI>
int launch_ws_conv3_ks(const GemmArgs& g, hipStream_t st) {
  constexpr int BM = 256, BN = 64, LW = 4, NSB = 6, NAB = 3;
  constexpr size_t ai = ((BM + 2 + 7) / 8 + LW - 1) / LW;
  constexpr size_t lds_ring = NAB * ai * LW * 1024 + (size_t)NSB * BN * 128;
  constexpr size_t lds_epi = (size_t)BM * BN * 4;
  constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g;
  pr.tiles0 = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * (EPI == EPI_GATE_RES ? g.ksplit : 1);
  static const bool nt = []() { const char* e = getenv("FOLEY_W_NT"); return e && e[0] == '1'; }();
  static std::atomic<unsigned long long> raised{0}, raised_nt{0};
  if (nt) {
    auto k = gemm_ws_conv3_ks_kernel<T, EPI, 1>;
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised_nt);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
    FOLEY_LAUNCH(k, dim3(pr.tiles0), dim3(768), lds, st, pr);
  } else {
    auto k = gemm_ws_conv3_ks_kernel<T, EPI, 0>;
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
    FOLEY_LAUNCH(k, dim3(pr.tiles0), dim3(768), lds, st, pr);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T, int BM, int EPI, int WF>
int launch_ws_conv3_one(const GemmArgs& g, hipStream_t st) {
  // 128x128: activation chunks 3 deep + 6 weight slices; 256x128 (large grids): 2 + 4 (fp8 weights: 3 + 6); 192x128 (tile 24: a
  // large grid whose 256-row tiles would leave a quarter of the CUs idle): wave tiles 96x32, 2 + 4
  constexpr int BN = 128, WM = BM == 192 ? 2 : 4, WN = BM == 192 ? 4 : 2, LW = 4;
  constexpr int NSB = BM == 128 ? 6 : (WF ? 6 : 4), NAB = (NSB + 2 + 2) / 3;
  constexpr size_t ai = ((BM + 2 + 7) / 8 + LW - 1) / LW;
  constexpr size_t lds_ring = NAB * ai * LW * 1024 + (size_t)NSB * BN * (WF ? 64 : 128) + 128;
  constexpr size_t lds_epi = (size_t)BM * BN * 4;
  constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g;
  pr.tiles0 = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * (EPI == EPI_GATE_RES ? g.ksplit : 1);
  auto k = gemm_ws_conv3_kernel<T, BM, BN, WM, WN, NSB, NAB, LW, EPI, WF>;
  static std::atomic<unsigned long long> raised{0};
  {
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  FOLEY_LAUNCH(k, dim3(pr.tiles0), dim3((WM * WN + LW) * 64), lds, st, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

// 256x64 form (tile 22): same workgroup count as 128x128 on N = 1536 problems, 57 instead of 65 KiB of operands per
// 64-channel chunk and workgroup (the activation chunk is shared by three taps, so rows are cheaper than columns)
template <typename T, int EPI>
int launch_ws_conv3_tall(const GemmArgs& g, hipStream_t st) {
  constexpr int BM = 256, BN = 64, WM = 8, WN = 1, LW = 4, NSB = 6, NAB = 3;
  constexpr size_t ai = ((BM + 2 + 7) / 8 + LW - 1) / LW;
  constexpr size_t lds_ring = NAB * ai * LW * 1024 + (size_t)NSB * BN * 128 + 128;
  constexpr size_t lds_epi = (size_t)BM * BN * 4;
  constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g;
  pr.tiles0 = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * (EPI == EPI_GATE_RES ? g.ksplit : 1);
  auto k = gemm_ws_conv3_kernel<T, BM, BN, WM, WN, NSB, NAB, LW, EPI, 0>;
  static std::atomic<unsigned long long> raised{0};
  {
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  FOLEY_LAUNCH(k, dim3(pr.tiles0), dim3((WM * WN + LW) * 64), lds, st, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T, int BM, int WF>
int launch_ws_conv3_fmt(const GemmArgs& g, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_STORE_F32: return launch_ws_conv3_one<T, BM, EPI_STORE_F32, WF>(g, st);
    case EPI_GATE_RES: return launch_ws_conv3_one<T, BM, EPI_GATE_RES, WF>(g, st);
    case EPI_SILUGATE_T: return launch_ws_conv3_one<T, BM, EPI_SILUGATE_T, WF>(g, st);
  }
  return foley_set_err("wave-specialised conv3: unsupported epilogue", __FILE__, __LINE__);
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int LW, int EPI, int WF>
int launch_ws_one(const GemmArgs& g, const GemmArgs* g1, hipStream_t st) {
  auto ntiles = [](const GemmArgs& q) {
    return ((q.M + BM - 1) / BM) * ((q.N + BN - 1) / BN) * (EPI == EPI_GATE_RES ? q.ksplit : 1);
  };
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g1 ? *g1 : g;
  pr.tiles0 = ntiles(g);
  const int tiles = pr.tiles0 + (g1 ? ntiles(*g1) : 0);
  constexpr size_t lds_ring = (size_t)NS * (BM * 128 + BN * (WF ? 64 : 128));
  constexpr size_t lds_epi = (size_t)BM * BN * 4 +           // the epilogues transpose the accumulator tile through LDS
                             (EPI == EPI_QKV_ATTN ? (size_t)BM * 256 + 96 * 256 + 2 * 128 * 256 : 0);   // + Q / K / V^T images (two V^T: straddling tiles)
  constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = gemm_ws_kernel<T, BM, BN, WM, WN, NS, LW, EPI, WF>;
  static std::atomic<unsigned long long> raised{0};
  {
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  FOLEY_LAUNCH(k, dim3(tiles), dim3((WM * WN + LW) * 64), lds, st, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int LW, int WF>
int launch_ws_tile(const GemmArgs& g, const GemmArgs* g1, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_STORE_F32: return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_STORE_F32, WF>(g, g1, st);
    case EPI_GELU_T: return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_GELU_T, WF>(g, g1, st);
    case EPI_GATE_RES: return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_GATE_RES, WF>(g, g1, st);
    case EPI_QKV_SPLIT: return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_QKV_SPLIT, WF>(g, g1, st);
    case EPI_SILUGATE_T:
      if constexpr ((BN / WN) % 64 == 0) return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_SILUGATE_T, WF>(g, g1, st);
      else return foley_set_err("gated epilogue needs a 64-wide wave tile", __FILE__, __LINE__);
  }
  if constexpr (WF == 0) {   // epilogues only the bf16-weight embedders use
    switch (epi) {
      case EPI_STORE_T: return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_STORE_T, 0>(g, g1, st);
      case EPI_SILU_T: return launch_ws_one<T, BM, BN, WM, WN, NS, LW, EPI_SILU_T, 0>(g, g1, st);
    }
  }
  return foley_set_err("wave-specialised GEMM: unsupported epilogue for this weight format", __FILE__, __LINE__);
}

// tile: 21 = tap-fused conv k=3 (128x128, 8 consumer + 4 loader waves); 15 = 128x128 (8 consumer + 4 loader waves), 19 = 256x128 (8 + 4), 25 / 29 = the same tiles with 4
// consumer waves (64x64 / 128x64 per wave); g / g1 fully resolved
// (ksplit, vec_out, operand extents) by gemm_impl.h's launcher
template <typename T>
int launch_gemm_ws_t(const GemmArgs& g, const GemmArgs* g1, int epi, int tile, hipStream_t st) {
  if (g.wfmt < 0 || g.wfmt > 2 || (g1 && g1->wfmt != g.wfmt))
    return foley_set_err("wave-specialised GEMM: bad / mixed weight formats", __FILE__, __LINE__);
  if (tile == 22) {   // tap-fused conv k=3, 256x64 (bf16 weights; gated residual / fp32 store)
    if (g1 || g.wfmt) return foley_set_err("wave-specialised conv3 256x64: single problem, bf16 weights", __FILE__, __LINE__);
    // round 5: K-split wave pairs + slice read-ahead (vector epilogue only; FOLEY_CONV3_KS=0 keeps the early / late form)
    static const bool ksp = []() { const char* e = getenv("FOLEY_CONV3_KS"); return !(e && e[0] == '0'); }();
    if (ksp && g.vec_out) {
      if (epi == EPI_GATE_RES) return launch_ws_conv3_ks<T, EPI_GATE_RES>(g, st);
      if (epi == EPI_STORE_F32) return launch_ws_conv3_ks<T, EPI_STORE_F32>(g, st);
      if (epi == EPI_SILUGATE_T) return launch_ws_conv3_ks<T, EPI_SILUGATE_T>(g, st);
    }
    if (epi == EPI_GATE_RES) return launch_ws_conv3_tall<T, EPI_GATE_RES>(g, st);
    if (epi == EPI_STORE_F32) return launch_ws_conv3_tall<T, EPI_STORE_F32>(g, st);
    if (epi == EPI_SILUGATE_T) return launch_ws_conv3_tall<T, EPI_SILUGATE_T>(g, st);
    return foley_set_err("wave-specialised conv3 256x64: unsupported epilogue", __FILE__, __LINE__);
  }
  if (tile == 24) {   // tap-fused conv k=3, 192x128 (bf16 weights; gated residual / fp32 store)
    if (g1 || g.wfmt) return foley_set_err("wave-specialised conv3 192x128: single problem, bf16 weights", __FILE__, __LINE__);
    if (epi == EPI_GATE_RES) return launch_ws_conv3_one<T, 192, EPI_GATE_RES, 0>(g, st);
    if (epi == EPI_STORE_F32) return launch_ws_conv3_one<T, 192, EPI_STORE_F32, 0>(g, st);
    return foley_set_err("wave-specialised conv3 192x128: unsupported epilogue", __FILE__, __LINE__);
  }
  if (tile == 21 || tile == 23) {   // tap-fused conv k=3, 128x128 / 256x128 (the launcher has checked the conv shape)
    if (g1) return foley_set_err("wave-specialised conv3 has no two-problem form", __FILE__, __LINE__);
    if (tile == 21) {
      if (g.wfmt == 0) return launch_ws_conv3_fmt<T, 128, 0>(g, epi, st);
      if (g.wfmt == 1) return launch_ws_conv3_fmt<T, 128, 1>(g, epi, st);
      return launch_ws_conv3_fmt<T, 128, 2>(g, epi, st);
    }
    if (g.wfmt == 0) return launch_ws_conv3_fmt<T, 256, 0>(g, epi, st);
    if (g.wfmt == 1) return launch_ws_conv3_fmt<T, 256, 1>(g, epi, st);
    return launch_ws_conv3_fmt<T, 256, 2>(g, epi, st);
  }
  if (tile == 26) {   // 96x128, four consumer waves of 96x32, 5 x 28 KiB ring: the fused head split when 96-row tiles fill one round
    if (g.wfmt != 0 || epi != EPI_QKV_SPLIT) return foley_set_err("wave-specialised GEMM: tile 26 is a bf16 head-split tile", __FILE__, __LINE__);
    return launch_ws_one<T, 96, 128, 1, 4, 5, 4, EPI_QKV_SPLIT, 0>(g, g1, st);
  }
  if (tile == 28) {   // 192x128, four consumer waves of 96x64, 4 x 40 KiB ring: the fused head split of large grids
    if (g.wfmt != 0 || epi != EPI_QKV_SPLIT) return foley_set_err("wave-specialised GEMM: tile 28 is a bf16 head-split tile", __FILE__, __LINE__);
    return launch_ws_one<T, 192, 128, 2, 2, 4, 4, EPI_QKV_SPLIT, 0>(g, g1, st);
  }
  if (tile == 27) {   // 64x128, four consumer waves of 32x64, 6 x 24 KiB ring: the fused head split of small problems only
    if (g.wfmt != 0 || (epi != EPI_QKV_SPLIT && epi != EPI_QKV_ATTN)) return foley_set_err("wave-specialised GEMM: tile 27 is the bf16 head-split tile", __FILE__, __LINE__);
    if (epi == EPI_QKV_ATTN) return launch_ws_one<T, 64, 128, 2, 2, 6, 4, EPI_QKV_ATTN, 0>(g, g1, st);   // + cross attention in the epilogue
    return launch_ws_one<T, 64, 128, 2, 2, 6, 4, EPI_QKV_SPLIT, 0>(g, g1, st);
  }
  if (g.wfmt == 0) {
    switch (tile) {
      case 15: return launch_ws_tile<T, 128, 128, 4, 2, 5, 4, 0>(g, g1, epi, st);   // 5 x 32 KiB ring = all 160 KiB of LDS
      case 19: return launch_ws_tile<T, 256, 128, 4, 2, 3, 4, 0>(g, g1, epi, st);
      case 25: return launch_ws_tile<T, 128, 128, 2, 2, 5, 4, 0>(g, g1, epi, st);   // 4 consumer waves of 64x64 (one per SIMD) + 4 loaders
      case 29: return launch_ws_tile<T, 256, 128, 2, 2, 3, 4, 0>(g, g1, epi, st);   // 4 consumer waves of 128x64
    }
  } else if (g.wfmt == 1) {   // fp8 e4m3fn weights: 24 / 40 KiB stages
    switch (tile) {
      case 15: return launch_ws_tile<T, 128, 128, 4, 2, 6, 4, 1>(g, g1, epi, st);
      case 19: return launch_ws_tile<T, 256, 128, 4, 2, 4, 4, 1>(g, g1, epi, st);
    }
  } else {
    switch (tile) {
      case 15: return launch_ws_tile<T, 128, 128, 4, 2, 6, 4, 2>(g, g1, epi, st);
      case 19: return launch_ws_tile<T, 256, 128, 4, 2, 4, 4, 2>(g, g1, epi, st);
    }
  }
  return foley_set_err("wave-specialised GEMM: unknown tile for this weight format", __FILE__, __LINE__);
}

}  // namespace
