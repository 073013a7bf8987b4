// MFMA GEMM / conv-as-GEMM engine for gfx950 (see kernels.h for the addressing model).
//
// One workgroup = 4 wavefronts (256 lanes) computes a BM x BN output tile with 32x32 MFMA
// fragments: v_mfma_f32_32x32x2_f32 for fp32 operands (exact fp32 FMA chain - the parity mode)
// and v_mfma_f32_32x32x16_bf16 for bf16 operands, both accumulating in fp32.  Operand tiles are
// 128-byte K-slices (32 fp32 / 64 bf16) staged through a double-buffered LDS tile with a 144-byte
// row pitch, which makes the ds_read_b128 fragment reads bank-conflict free (MI355X_MICROARCH.md,
// LDS table).  A ring of NS register stages keeps NS-1 K-slices of global loads in flight behind
// the MFMA block (one barrier per slice).  Workgroup ids are remapped so tiles that share a
// weight panel run on one XCD (L2).
#pragma once
#include <cstdlib>

#include "gemm_common.h"

// shared by the per-dtype translation units (defined in gemm.hip)
extern long long* g_gemm_dbg;
extern int g_gemm_dbg_mode;
extern int g_gemm_pf_dist;
extern thread_local int g_gemm_krot_ok;

namespace {

template <typename T, int BM, int BN, int WM, int WN, int NS, int EPI, bool CONV>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmPair pr) {
  const int sel = (int)blockIdx.x >= pr.tiles0 ? 1 : 0;  // wave-uniform
  const GemmArgs& g = pr.g[sel];
  constexpr int NT = WM * WN * 64;
  constexpr int EPC = Frag<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int RPP = NT / 8;  // rows covered by one loader pass (8 lanes x 16 B per row)
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int RA = BM / RPP, RB = BN / RPP;
  constexpr int STAGE = (BM + BN) * LDS_PITCH;
  static_assert(BM % RPP == 0 && BN % RPP == 0 && TM % 32 == 0 && TN % 32 == 0, "bad tile");
  static_assert(EPI != EPI_SILUGATE_T || (FN % 2 == 0), "gated epilogue needs fragment pairs");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 stages (double buffer)

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
    ks = bid % g.ksplit;
    bid /= g.ksplit;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  tl_stamp(g, 0);
  const int wm = wave / WN, wn = wave % WN;
  const int chunk = tid & 7, lrow = tid >> 3;

  // Per-thread row descriptors.  Loads are ALWAYS issued (from a clamped, in-bounds address) and
  // masked with a register select when they are written to LDS: a load under a branch would make
  // the compiler wait for each one separately.
  const T* ap[RA];
  int a_q[RA];
  bool a_ok[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int r = m0 + lrow + i * RPP;
    a_ok[i] = r < g.M;
    const int rr = a_ok[i] ? r : 0;
    int b = 0, q = rr;
    if constexpr (CONV) {
      b = rr / g.segV;
      q = rr - b * g.segV;
    }
    const int qs = q * (g.rstride > 1 ? g.rstride : 1);   // source row of tap offset 0 (strided conv)
    ap[i] = (const T*)g.A + ((long)b * g.segS + qs) * g.lda + chunk * EPC;
    a_q[i] = qs;
  }
  const T* wp[RB];
  bool w_ok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int n = n0 + lrow + i * RPP;
    w_ok[i] = n < g.N;
    wp[i] = (const T*)g.W + (long)(w_ok[i] ? n : 0) * g.K + chunk * EPC;
  }

  u32x4 ra[NS][RA], rw[NS][RB];  // register ring: NS-1 K-slices in flight
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  constexpr bool DUAL = (FM * FN == 1) && sizeof(T) == 2;
  f32x16 acc2;
  if constexpr (DUAL) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
  }
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fi = lane & 31, kh = lane >> 5;
  int kt_begin = 0, nk = g.K / BK;
  if constexpr (EPI == EPI_GATE_RES) {  // this workgroup's K range
    const int tot = nk;
    kt_begin = (int)((long)tot * ks / g.ksplit);
    nk = (int)((long)tot * (ks + 1) / g.ksplit) - kt_begin;
  }
  // Loader / LDS-writer cursors over the K axis, advanced incrementally (no per-slice division):
  // channel offset inside the current tap, the tap's source-row offset, and that offset in elements.
  const long tap_step = (long)g.dil * g.lda;
  int ld_k0 = kt_begin * BK;
  int ld_c0 = ld_k0, ld_toff = g.tap0;
  if (kt_begin > 0) {
    const int tap = ld_k0 / g.tapC;
    ld_c0 = ld_k0 - tap * g.tapC;
    ld_toff = g.tap0 + tap * g.dil;
  }
  long ld_roff = (long)ld_toff * g.lda;
  int wr_c0 = ld_c0, wr_toff = ld_toff;

#define FOLEY_GLOAD(slot)                                                                  \
  {                                                                                        \
    _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                       \
      if constexpr (CONV) {                                                                \
        const bool in_ = (unsigned)(a_q[i] + ld_toff) < (unsigned)g.segS;                  \
        ra[slot][i] = *(const u32x4*)(ap[i] + (in_ ? ld_roff : 0L) + ld_c0);               \
      } else {                                                                             \
        ra[slot][i] = *(const u32x4*)(ap[i] + ld_k0);                                      \
      }                                                                                    \
    }                                                                                      \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) rw[slot][i] = *(const u32x4*)(wp[i] + ld_k0); \
    ld_k0 += BK;                                                                           \
    if constexpr (CONV) {                                                                  \
      ld_c0 += BK;                                                                         \
      if (ld_c0 >= g.tapC) {                                                               \
        ld_c0 = 0;                                                                         \
        ld_toff += g.dil;                                                                  \
        ld_roff += tap_step;                                                               \
      }                                                                                    \
    }                                                                                      \
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) FOLEY_GLOAD(s);

  for (int kt0 = 0; kt0 < nk; kt0 += NS) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int kt = kt0 + j;
      if (kt < nk) {
        if (kt + NS - 1 < nk) FOLEY_GLOAD((j + NS - 1) % NS);
        unsigned char* As = lds + (kt & 1) * STAGE;
        unsigned char* Bs = As + BM * LDS_PITCH;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          bool v = a_ok[i];
          if constexpr (CONV) v = v && (unsigned)(a_q[i] + wr_toff) < (unsigned)g.segS;
          *(u32x4*)(As + (lrow + i * RPP) * LDS_PITCH + chunk * 16) = v ? ra[j][i] : zero4;
        }
        if constexpr (CONV) {
          wr_c0 += BK;
          if (wr_c0 >= g.tapC) {
            wr_c0 = 0;
            wr_toff += g.dil;
          }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
          *(u32x4*)(Bs + (lrow + i * RPP) * LDS_PITCH + chunk * 16) = w_ok[i] ? rw[j][i] : zero4;
        // one barrier per K-slice: the stage written now was last read two slices ago, and every
        // wave has passed the previous barrier since
        __syncthreads();
        if (kt == 0) tl_stamp(g, 1);

        if constexpr (sizeof(T) == 4) {
          // lane (fi, kh) owns k = kh*16 .. kh*16+15 of its row; MFMA step s contracts the k pair
          // (s, 16 + s) - any pairing is valid as long as A and B use the same one.
          float a[FM][16], b[FN][16];
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const unsigned char* p = As + (wm * TM + i * 32 + fi) * LDS_PITCH + kh * 64;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 v = *(const f32x4*)(p + c * 16);
              a[i][c * 4 + 0] = v[0]; a[i][c * 4 + 1] = v[1]; a[i][c * 4 + 2] = v[2]; a[i][c * 4 + 3] = v[3];
            }
          }
#pragma unroll
          for (int jj = 0; jj < FN; ++jj) {
            const unsigned char* p = Bs + (wn * TN + jj * 32 + fi) * LDS_PITCH + kh * 64;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 v = *(const f32x4*)(p + c * 16);
              b[jj][c * 4 + 0] = v[0]; b[jj][c * 4 + 1] = v[1]; b[jj][c * 4 + 2] = v[2]; b[jj][c * 4 + 3] = v[3];
            }
          }
#pragma unroll
          for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
              for (int jj = 0; jj < FN; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[jj][s], acc[i][jj], 0, 0, 0);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bf16x8 a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
              a[i] = *(const bf16x8*)(As + (wm * TM + i * 32 + fi) * LDS_PITCH + s * 32 + kh * 16);
#pragma unroll
            for (int jj = 0; jj < FN; ++jj)
              b[jj] = *(const bf16x8*)(Bs + (wn * TN + jj * 32 + fi) * LDS_PITCH + s * 32 + kh * 16);
            if constexpr (DUAL) {
              // single-fragment wave tile: alternate two accumulators so consecutive MFMAs do not
              // wait for each other's 16-pass latency
              if (s & 1) acc2 = mfma16<T>(a[0], b[0], acc2);
              else acc[0][0] = mfma16<T>(a[0], b[0], acc[0][0]);
            } else {
#pragma unroll
              for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int jj = 0; jj < FN; ++jj)
                  acc[i][jj] = mfma16<T>(a[i], b[jj], acc[i][jj]);
            }
          }
        }
      }
    }
  }
#undef FOLEY_GLOAD
  if constexpr (DUAL) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] += acc2[e];
  }

  tl_stamp(g, 2);
  if constexpr (EPI == EPI_QKV_SPLIT) {
    gemm_epilogue_qkv<T, BM, BN, WM, WN>(g, acc, lds, m0, n0);
  } else {
    if (g.vec_out) gemm_epilogue_lds<T, EPI, BM, BN, WM, WN>(g, acc, lds, m0, n0, ks);
    else gemm_epilogue<T, EPI, FM, FN, TM, TN>(g, acc, m0, n0, wm, wn, fi, kh, ks);
  }
  tl_stamp(g, 3);
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS mainloop: K-slices travel HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR
// staging, no ds_write pass), NS LDS stages deep, one raw s_barrier + one counted vmcnt per slice.
// The DMA writes lane l of a wave-instruction at (wave-uniform base + 16*l), i.e. 8 unpadded
// 128-byte rows per instruction, so bank conflicts are avoided by permuting the SOURCE: LDS chunk
// p of row r holds global chunk p ^ ((r >> 1) & 7); fragment reads apply the same XOR.  Rows
// that must read as zero (M/N edge, conv padding) fetch from a zero page instead.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// buffer_load_dwordx4 ... lds: SGPR resource (base, extent) + per-lane byte offset + scalar byte
// offset.  Kept out of the kernel template: the resource type only exists in the device pass.
__device__ __forceinline__ void buf_lds16(const void* base, unsigned bytes, unsigned char* lds_wave_base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_glds_kernel(const GemmPair pr) {
  const int sel = (int)blockIdx.x >= pr.tiles0 ? 1 : 0;
  const GemmArgs& g = pr.g[sel];
  constexpr int NW = WM * WN;
  constexpr int EPC = Frag<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  constexpr int AI = BM / 8 / NW, BI = BN / 8 / NW;  // wave-instructions per wave and K-slice
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "bad tile");
  static_assert(EPI != EPI_SILUGATE_T || (FN % 2 == 0), "gated epilogue needs fragment pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = (int)blockIdx.x - (sel ? pr.tiles0 : 0);
  {
    const int nwg = tiles_m * tiles_n * (EPI == EPI_GATE_RES ? g.ksplit : 1);
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int ks = 0;
  if constexpr (EPI == EPI_GATE_RES) {
    ks = bid % g.ksplit;
    bid /= g.ksplit;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  tl_stamp(g, 0);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane >> 3, lp = lane & 7;  // row within the 8-row group, LDS chunk position

  // Loader addressing without per-slice VALU work: both operands are buffer resources (SGPRs), every
  // lane keeps ONE loop-invariant 32-bit byte offset per 1 KiB piece and the K position travels in a
  // scalar offset.  (With per-lane 64-bit addresses each issue needs VALU adds, and those starve
  // behind the MFMAs of the other waves: tools/ubench/ldsdma_interfere.hip, 118 vs 43 GB/s per CU.)
  // Lanes that must read zeros (M / N edge, conv padding) carry an out-of-range offset: the buffer
  // range check (voffset + soffset >= num_records) makes the DMA write zeros (tools/ubench/buf_oob.hip).
  constexpr int ESZ = (int)sizeof(T);
  constexpr int OOB = 0x7ffffff0;
  int a_base[AI], a_q[AI], vA[AI];   // byte offset of the row at tap offset 0 (< 0: row beyond M)
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int rl = (wave * AI + i) * 8 + lr;  // row inside the tile
    const int r = m0 + rl;
    const int rr = r < g.M ? r : 0;
    const int b = g.segV >= g.M ? 0 : rr / g.segV, q = rr - b * g.segV;   // plain GEMM: one segment, no division
    const int qs = q * (g.rstride > 1 ? g.rstride : 1);   // source row of tap offset 0 (strided conv)
    a_base[i] = r < g.M ? (int)((((long)b * g.segS + qs) * g.lda + (lp ^ ((rl >> 1) & 7)) * EPC) * ESZ) : -1;
    a_q[i] = qs;
  }
  int vW[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int rl = (wave * BI + i) * 8 + lr;
    const int n = n0 + rl;
    vW[i] = (n < g.N) ? (int)(((long)n * g.K + (lp ^ ((rl >> 1) & 7)) * EPC) * ESZ) : OOB;
  }
  auto set_tap = [&](int toff) {   // per-lane offsets of the current tap (VALU, once per tap)
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool in = a_base[i] >= 0 && (unsigned)(a_q[i] + toff) < (unsigned)g.segS;
      vA[i] = in ? a_base[i] + toff * (int)(g.lda * ESZ) : OOB;
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fi = lane & 31, kh = lane >> 5;
  int kt_begin = 0, nk = g.K / BK;
  if constexpr (EPI == EPI_GATE_RES) {
    const int tot = nk;
    kt_begin = (int)((long)tot * ks / g.ksplit);
    nk = (int)((long)tot * (ks + 1) / g.ksplit) - kt_begin;
  }
  int ld_k0 = kt_begin * BK;
  int ld_c0 = ld_k0, ld_toff = g.tap0;
  if (kt_begin > 0) {
    const int tap = ld_k0 / g.tapC;
    ld_c0 = ld_k0 - tap * g.tapC;
    ld_toff = g.tap0 + tap * g.dil;
  }
  set_tap(ld_toff);

  auto issue = [&](int stage) {
    unsigned char* As = lds + stage * STAGE;
    unsigned char* Bs = As + BM * 128;
    const int sA = ld_c0 * ESZ, sW = ld_k0 * ESZ;   // scalar K offsets
#pragma unroll
    for (int i = 0; i < AI; ++i)
      buf_lds16(g.A, g.a_bytes, As + (wave * AI + i) * 1024, vA[i], sA);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      buf_lds16(g.W, g.w_bytes, Bs + (wave * BI + i) * 1024, vW[i], sW);
    ld_k0 += BK;
    ld_c0 += BK;
    if (ld_c0 >= g.tapC) {
      ld_c0 = 0;
      ld_toff += g.dil;
      set_tap(ld_toff);
    }
  };

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s);

  // fragment-read swizzle terms (row-dependent, K-independent)
  int a_row[FM], a_sw[FM], b_row[FN], b_sw[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    a_row[i] = (wm * TM + i * 32 + fi) * 128;
    a_sw[i] = ((wm * TM + i * 32 + fi) >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    b_row[j] = (wn * TN + j * 32 + fi) * 128;
    b_sw[j] = ((wn * TN + j * 32 + fi) >> 1) & 7;
  }

  bf16x8 fa[4][FM], fb[4][FN];
  auto mma_bf16 = [&]() {
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
    // slice kt has landed once at most the NS-2 younger slices are still in flight
    if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (AI + BI)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's part of slice kt is in LDS; stage (kt-1)%NS is free again
    if (kt == 0) tl_stamp(g, 1);
    if (kt + NS - 1 < nk && !(g.dbg_mode & 0x400)) issue(stage == 0 ? NS - 1 : stage - 1);
    const unsigned char* As = lds + stage * STAGE;
    const unsigned char* Bs = As + BM * 128;
    if constexpr (sizeof(T) == 4) {
      float a[FM][16], b[FN][16];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *(const f32x4*)(As + a_row[i] + (((kh * 4 + c) ^ a_sw[i]) << 4));
          a[i][c * 4 + 0] = v[0]; a[i][c * 4 + 1] = v[1]; a[i][c * 4 + 2] = v[2]; a[i][c * 4 + 3] = v[3];
        }
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *(const f32x4*)(Bs + b_row[j] + (((kh * 4 + c) ^ b_sw[j]) << 4));
          b[j][c * 4 + 0] = v[0]; b[j][c * 4 + 1] = v[1]; b[j][c * 4 + 2] = v[2]; b[j][c * 4 + 3] = v[3];
        }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    } else {
      // all fragment reads of the slice are issued up front so the MFMAs run back to back behind
      // counted lgkmcnt waits (LDS latency hidden behind the matrix pipe instead of serialised)
      if (!(g.dbg_mode & 0x200)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[s][i] = *(const bf16x8*)(As + a_row[i] + (((s * 2 + kh) ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[s][j] = *(const bf16x8*)(Bs + b_row[j] + (((s * 2 + kh) ^ b_sw[j]) << 4));
      }
      }
      if (!(g.dbg_mode & 0x100)) mma_bf16();
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  tl_stamp(g, 2);
  if constexpr (EPI == EPI_QKV_SPLIT) {
    gemm_epilogue_qkv<T, BM, BN, WM, WN>(g, acc, lds, m0, n0);
  } else {
    if (g.vec_out) gemm_epilogue_lds<T, EPI, BM, BN, WM, WN>(g, acc, lds, m0, n0, ks);
    else gemm_epilogue<T, EPI, FM, FN, TM, TN>(g, acc, m0, n0, wm, wn, fi, kh, ks);
  }
  tl_stamp(g, 3);
}

template <typename T, int BM, int BN, int WM, int WN, int NS, int EPI, bool GLDS = false>
int launch_one(const GemmArgs& g, const GemmArgs* g1, hipStream_t st) {
  auto ntiles = [](const GemmArgs& q) {
    return ((q.M + BM - 1) / BM) * ((q.N + BN - 1) / BN) * (EPI == EPI_GATE_RES ? q.ksplit : 1);
  };
  GemmPair pr;
  pr.g[0] = g;
  pr.g[1] = g1 ? *g1 : g;
  pr.tiles0 = ntiles(g);
  const int tiles = pr.tiles0 + (g1 ? ntiles(*g1) : 0);
  constexpr size_t lds = GLDS ? (size_t)NS * (BM + BN) * 128 : 2 * (size_t)(BM + BN) * LDS_PITCH;
  void (*k)(const GemmPair);
  // CONV = the row -> (segment, position) mapping is live: taps, a strided walk, or virtual rows that skip source rows
  // (segV != segS: the modulation GEMM over the first P tokens of every (iteration, half) segment)
  auto mapped = [](const GemmArgs& q) { return q.taps > 1 || q.rstride > 1 || (q.segV != q.segS && q.segV < q.M); };
  const bool conv = mapped(g) || (g1 && mapped(*g1));
  if constexpr (GLDS) k = gemm_glds_kernel<T, BM, BN, WM, WN, NS, EPI>;
  else k = conv ? gemm_kernel<T, BM, BN, WM, WN, NS, EPI, true> : gemm_kernel<T, BM, BN, WM, WN, NS, EPI, false>;
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> raised[2];   // per instantiation (plain / conv kernel), per device
    hipError_t e = foley_raise_lds((const void*)k, (int)lds, raised[conv ? 1 : 0]);
    if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  }
  FOLEY_LAUNCH(k, dim3(tiles), dim3(WM * WN * 64), lds, st, pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return foley_set_err(hipGetErrorString(e), __FILE__, __LINE__);
  return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int NS, bool GLDS = false>
int launch_tile(const GemmArgs& g, const GemmArgs* g1, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_STORE_F32: return launch_one<T, BM, BN, WM, WN, NS, EPI_STORE_F32, GLDS>(g, g1, st);
    case EPI_STORE_T: return launch_one<T, BM, BN, WM, WN, NS, EPI_STORE_T, GLDS>(g, g1, st);
    case EPI_SILU_T: return launch_one<T, BM, BN, WM, WN, NS, EPI_SILU_T, GLDS>(g, g1, st);
    case EPI_GELU_T: return launch_one<T, BM, BN, WM, WN, NS, EPI_GELU_T, GLDS>(g, g1, st);
    case EPI_GATE_RES: return launch_one<T, BM, BN, WM, WN, NS, EPI_GATE_RES, GLDS>(g, g1, st);
    case EPI_SILUGATE_T:
      if constexpr ((BN / WN) % 64 == 0) return launch_one<T, BM, BN, WM, WN, NS, EPI_SILUGATE_T, GLDS>(g, g1, st);
      else return foley_set_err("gated epilogue needs a 64-wide wave tile", __FILE__, __LINE__);
    case EPI_QKV_SPLIT:
      if constexpr (BN == 128) return launch_one<T, BM, BN, WM, WN, NS, EPI_QKV_SPLIT, GLDS>(g, g1, st);
      else return foley_set_err("fused head-split epilogue needs a 128-column tile", __FILE__, __LINE__);
    case EPI_DAC:
      if constexpr (sizeof(T) == 4) return launch_one<T, BM, BN, WM, WN, NS, EPI_DAC, GLDS>(g, g1, st);
      else return foley_set_err("DAC epilogue is fp32 only", __FILE__, __LINE__);
  }
  return foley_set_err("unknown GEMM epilogue", __FILE__, __LINE__);
}


inline bool f32_half_tiles() {
  static const bool v = []() { const char* e = getenv("FOLEY_F32_HALF_TILES"); return !(e && e[0] == '0'); }();
  return v;
}

template <typename T>
int check_args(const GemmArgs& g) {
  constexpr int BK = 8 * Frag<T>::EPC;
  if (g.K % BK || g.tapC % BK || g.taps * g.tapC != g.K || g.lda % Frag<T>::EPC)
    return foley_set_err("GEMM: K / tap width / lda must be multiples of the 128-byte K-slice", __FILE__, __LINE__);
  if (((uintptr_t)g.A | (uintptr_t)g.W) & 15)
    return foley_set_err("GEMM: operands must be 16-byte aligned", __FILE__, __LINE__);
  return 0;
}

template <typename T>
int launch_typed(const GemmArgs& g_in, const GemmArgs* g1_in, int epi, int tile, hipStream_t st, int* ksplit_used) {
  GemmArgs g = g_in;
  g.dbg = g_gemm_dbg;
  g.dbg_mode = g_gemm_dbg_mode;
  g.pf_dist = g_gemm_pf_dist;
  {  // round 5: K-range-major workgroup order for the wave-specialised split-K tiles (FOLEY_KS_MAJOR=0: range fastest)
    static const int ksm = []() { const char* e = getenv("FOLEY_KS_MAJOR"); return (e && e[0] == '0') ? 0 : 1; }();
    g.ks_major = ksm;
    g.n_groups = 0;   // decided below, once the tile is known
  }
  if (g.ldw <= 0) g.ldw = g.K;
  GemmArgs g1s;
  const GemmArgs* g1 = nullptr;
  if (g1_in) {
    if (int rc = check_args<T>(*g1_in)) return rc;
    g1s = *g1_in;
    g1s.dbg = nullptr;
    g1s.pf_dist = g_gemm_pf_dist;
    g1s.ks_major = g.ks_major;
    g1s.n_groups = 0;
    if (g1s.ldw <= 0) g1s.ldw = g1s.K;
    g1 = &g1s;
  }
  constexpr int BK = 8 * Frag<T>::EPC;
  if (g.K % BK || g.tapC % BK || g.taps * g.tapC != g.K || g.lda % Frag<T>::EPC)
    return foley_set_err("GEMM: K / tap width / lda must be multiples of the 128-byte K-slice", __FILE__, __LINE__);
  if (((uintptr_t)g.A | (uintptr_t)g.W) & 15)
    return foley_set_err("GEMM: operands must be 16-byte aligned", __FILE__, __LINE__);
  if (g.partial_half && sizeof(T) != 2) return foley_set_err("GEMM: 16-bit partial slabs need 16-bit operands", __FILE__, __LINE__);
  if (g.wfmt && sizeof(T) != 2) return foley_set_err("GEMM: fp8 weight storage needs bf16 operands", __FILE__, __LINE__);
  if (g.wfmt && g1 && g1s.wfmt != g.wfmt) return foley_set_err("GEMM: the two problems of a launch must share the weight format", __FILE__, __LINE__);
  // tap-fused wave-specialised conv3 (tile 21): bf16 operands (any weight storage), dense k=3 'same' conv
  const bool ws_conv3_ok = sizeof(T) == 2 && !g1 && g.taps == 3 && g.dil == 1 && g.tap0 == -1 && g.rstride <= 1 && g.segV == g.segS &&
                           g.osegV >= g.M && (epi == EPI_STORE_F32 || epi == EPI_GATE_RES || epi == EPI_SILUGATE_T);
  if ((tile == 21 || tile == 22 || tile == 23 || tile == 24) && !ws_conv3_ok) return foley_set_err("GEMM: tiles 21 / 22 / 23 / 24 need a bf16 channels-last conv k=3", __FILE__, __LINE__);
  if (tile == 24 && (g.wfmt || epi == EPI_SILUGATE_T)) return foley_set_err("GEMM: tile 24 (192x128 conv) serves bf16 weights, gated-residual / fp32-store epilogues", __FILE__, __LINE__);
  if (tile == 22 && g.wfmt) return foley_set_err("GEMM: tile 22 serves bf16 weights", __FILE__, __LINE__);
  const bool conv3_ok = !g.wfmt && !g1 && g.taps == 3 && g.dil == 1 && g.tap0 == -1 && g.rstride <= 1 && g.segV == g.segS && g.lda == g.tapC &&
                        g.osegV >= g.M && (epi == EPI_STORE_F32 || epi == EPI_GATE_RES || epi == EPI_SILUGATE_T);
  const bool tile_auto = tile == 0;
  // deferred split-K available (bf16 mode, caller provided partial slabs): reductions are cheap
  // vector stores + a few extra row reads in the next LayerNorm
  const bool deferred = epi == EPI_GATE_RES && g.partials && g.partial_cap > 1 && sizeof(T) == 2 && g.ksplit != 1 &&
                        (!g1 || g1->partials);
  // Measured end to end (xxl, 5 s): the tap-fused kernel wins in fp32 (parity mode, -7 % loop time)
  // and for the small-M gated w1/w3 GEMM; elsewhere the generic tiles (+ split-K / 256x128) are as
  // fast or faster in bf16, so it is only auto-selected there.
  const bool small_grid = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) <= 256;
  if (tile == 0 && conv3_ok && (sizeof(T) == 4 || (deferred && small_grid))) {   // other bf16 cases: the wave-specialised generic tiles win (tools/gemm_timeline.py)
    const long b128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    tile = (b128 >= 100 || epi == EPI_SILUGATE_T || deferred) ? 11 : 13;   // the gated epilogue needs 64-wide wave tiles
  }
  if (tile == 0) {
    // Tile choice for 256 CUs (measured on the M=500 / M=4000 shapes of the xxl DiT,
    // tools/gemm_bench.py): the 128x128 / 8-wave tile wins whenever it fills the chip without a
    // ragged last wave of workgroups; otherwise many small 64x64 tiles hide latency better.
    auto nblk = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn); };
    const long b128 = nblk(128, 128);
    const long rem = b128 % 256;
    // (head-split pairs count both problems: the cross-attention q projection of the 30 s clip - 144 + 24 tiles of 256x128 in ONE round -
    // sat on 336 tiles of 128x128 in two ragged ones: 39.8 us per launch)
    static const bool pair_count = []() { const char* e = getenv("FOLEY_QKV_PAIR_COUNT"); return !(e && e[0] == '0'); }();   // A/B switch
    const long pair256 = (pair_count && epi == EPI_QKV_SPLIT && g1) ? (long)((g1s.M + 255) / 256) * ((g1s.N + 127) / 128) : 0;
    if (sizeof(T) == 2 && g.N > 64 && g.M > 128 && nblk(256, 128) + pair256 >= 160) tile = 9;   // big grids: 64x64 per wave (one M tile: 128 rows
                                                                                         // halve the activation DMA, modulation GEMM at M = 16: 258 -> 227 us)
    else if (deferred && g.N > 64 && b128 <= 256) tile = 5;   // 128x128 tiles, K ranges fill the chip (tools/gemm_timeline.py)
    else if (g.N <= 64 && epi != EPI_SILUGATE_T) tile = nblk(128, 64) >= 192 ? 4 : 3;
    // fp32 (the DAC decoder): a long-K GEMM whose 128x128 tiles cover half the chip or less runs at the pace of ONE workgroup
    // (decoder stage 1: M = 2000, N = 1024, K = 7168 -> 128 workgroups, 470 us at 39 % of the fp32 matrix peak); 64x128 tiles
    // double the workgroups (FOLEY_F32_HALF_TILES=0 keeps 128x128)
    else if (sizeof(T) == 4 && f32_half_tiles() && b128 >= 100 && b128 <= 160 && nblk(64, 128) <= 320 && g.K >= 1024 && epi != EPI_SILUGATE_T) tile = 8;
    else if (b128 >= 100 && (b128 <= 256 || rem == 0 || rem >= 128 || b128 >= 2048)) tile = 5;
    else if (epi == EPI_SILUGATE_T) tile = nblk(64, 128) >= 192 ? 2 : 5;
    else tile = 3;
  }
  if (sizeof(T) == 2 && tile_auto) {   // bf16: loader / consumer wave specialisation of the same tiles
    // Four consumer waves (64x64 / 128x64 wave tiles, a third less LDS fragment traffic) with the loader waves
    // helping in the epilogue beat the eight-consumer form wherever the grid does not saturate the L2s
    // (M = 500: q/k/v 22.7 -> 19.1 us, fc1 21.1 -> 16.2, cross-q 21.1 -> 16.5; tools/gemm_timeline.py); the
    // big gated-residual GEMMs at large M keep eight consumers.  fp8 weights exist in the eight-consumer form.
    static const bool ws4 = []() { const char* e = getenv("FOLEY_WS4"); return !(e && e[0] == '0'); }();   // A/B switch for tools/
    if (tile == 5) tile = (g.wfmt || !ws4) ? 15 : 25;
    else if (tile == 9) tile = (g.wfmt || epi == EPI_GATE_RES || !ws4) ? 19 : 29;
  }
  // channels-last conv k=3 on a 128x128-class grid: the tap-fused wave-specialised kernel stages the activation
  // chunk once for the three taps (a third fewer bytes out of the L2s: lin1 16.4 -> 12.1 us, w2 32.4 -> 23.2 us,
  // w1/w3 49.3 -> 40.0 us at M = 500; tools/gemm_timeline.py).  Large grids use its 256x128 form (tile 23).
  if (tile_auto && ws_conv3_ok && (tile == 15 || tile == 25 || tile == 11 || tile == 13 || tile == 5 || tile == 3 || tile == 2)) tile = 21;
  {
    // split-K convs on a one-round grid: the 256x64 form has the same workgroup count on N = 1536 and moves 12 % fewer
    // operand bytes per workgroup (FOLEY_CONV3_TALL=0 keeps 128x128)
    static const bool tall = []() { const char* e = getenv("FOLEY_CONV3_TALL"); return !(e && e[0] == '0'); }();
    if (tile_auto && tile == 21 && tall && deferred && epi == EPI_GATE_RES && !g.wfmt && g.M > 256 && g.N % 128 == 0 &&
        (long)((g.M + 255) / 256) * (g.N / 64) == (long)((g.M + 127) / 128) * (g.N / 128))
      tile = 22;
    // the same form for w1 / w3 (SiLU gate, M = 500: 2 x 128 workgroups instead of 4 x 64): 39.2 -> 37.7 us (FOLEY_CONV3_TALL_GATE=0 keeps 128x128)
    static const bool tall_gate = []() { const char* e = getenv("FOLEY_CONV3_TALL_GATE"); return !(e && e[0] == '0'); }();
    if (tile_auto && tile == 21 && tall_gate && epi == EPI_SILUGATE_T && !g.wfmt && g.M > 256 && g.N % 128 == 0 &&
        (long)((g.M + 255) / 256) * (g.N / 64) == (long)((g.M + 127) / 128) * (g.N / 128))
      tile = 22;
  }
  if (tile_auto && ws_conv3_ok && (tile == 19 || tile == 29 || tile == 9)) tile = 23;   // large grids: the 256x128 tap-fused form (w1/w3 at M = 4000: 329 -> 285 us)
  // 256x256 tiles on the BK = 32 mainloop (gemm_wide_impl.h, tiles 31 / 32) for the large grids, wherever their workgroups
  // fill the last round of 256 CUs about as well as the 256x128 tiles' do (tools/wide_bench.py at M = 4000: w1/w3 298 -> 249 us,
  // w2 158 -> 134, linear2 66 -> 59, fc2 92 -> 79; fc1 - 1.5 rounds of 256x256 against exactly 3 of 256x128 - stays).
  // FOLEY_WIDE=0 keeps the 256x128 tiles.
  // Mid-size grids (M = 3000: the 30 s clip) whose N = 1536 gated-residual GEMMs landed on 128x128 tiles with two K ranges
  // (288 tiles, 1.1 rounds) take the same route: 72 tiles of 256x256 x three K ranges (w2 149 -> 97 us, fc2 99 -> 59 us).
  // (plain layers arrive here on tile 3 / 5 when their 288 tiles of 128x128 fit no rule above - fp8 weights turn that into tile 15 below)
  // (bf16 convs on tile 22, the 256x64 tap-fused form); measured down to M = 2000 (w2 89 -> 70 us, fc2 65 -> 46 with five K ranges)
  const bool mid_split = (tile == 21 || tile == 22 || tile == 15 || tile == 25 || tile == 5 || tile == 3) && g.M >= 1536 && g.N >= 256 &&
                         epi == EPI_GATE_RES && deferred;
  // (round 6) two-problem launches - the audio + visual pair of a two-stream block's gated-residual layers (proj, fc2) - take the same
  // route at mid-size grids: at M = 3000 + 480 they sat on 128x128 tiles (fc2 113 us where the single-problem form of the same shape
  // takes 57 - 59 on 72 tiles x three K ranges)
  static const bool wide_pair_on = []() { const char* e = getenv("FOLEY_WIDE_PAIR"); return !(e && e[0] == '0'); }();   // A/B switch
  // ... where their 128x128 tiles do not fit ONE round of 256 workgroups (M = 3000 + 480: 336).  Where they do (bs = 4, M = 2000 + 320:
  // 228 tiles, no K split, no slabs for the next LayerNorm to read) the 256x256 route with four K ranges LOSES 1.1 % of the loop.
  const long pair_b128 = !g1 ? 0 : (long)((g.M + 127) / 128) * ((g.N + 127) / 128) + (long)((g1s.M + 127) / 128) * ((g1s.N + 127) / 128);
  const bool pair_ok = !g1 || (wide_pair_on && mid_split && pair_b128 > 256 && !ws_conv3_ok && g1s.taps == 1 && g1s.segV >= g1s.M && g1s.rstride <= 1 && g1s.tap0 == 0 &&
                               g1s.tapC % 32 == 0 && g1s.M >= 1 && g1s.N >= 256);
  if (tile_auto && sizeof(T) == 2 && pair_ok && (tile == 23 || tile == 19 || tile == 29 || mid_split)) {
    static const bool wide_on = []() { const char* e = getenv("FOLEY_WIDE"); return !(e && e[0] == '0'); }();
    const bool conv = tile == 23 || tile == 21 || (mid_split && ws_conv3_ok);
    const bool epi_ok = epi == EPI_STORE_F32 || epi == EPI_GATE_RES || epi == EPI_SILUGATE_T || (!conv && epi == EPI_GELU_T);
    // plain layers: long K only - a 256x256 tile pays its two-pass epilogue and 6-slice ring fill once per 24 slices at K = 768
    // (the ViT-B encoders' fc1 at M = 22 000: 223 us on this tile against ~140 on 256x128)
    const bool addr_ok = conv || (g.taps == 1 && g.segV >= g.M && g.rstride <= 1 && g.tap0 == 0 &&
                                  // ... or a weight-streaming panel (the single-block modulation GEMM of a video clip: M = 224 rows against
                                  // N = 331 776 columns - a 256-column tile re-reads the activations half as often: 60.6 -> 51.9 us per eighth)
                                  (g.K >= 2048 || mid_split || (epi == EPI_STORE_F32 && g.N >= 16384 && g.M >= 128)));
    if (wide_on && epi_ok && addr_ok && g.tapC % 32 == 0) {
      long mt = (g.M + 255) / 256, tw = mt * ((g.N + 255) / 256), tb = mt * ((g.N + 127) / 128);
      if (g1) {
        const long mt1 = (g1s.M + 255) / 256;
        tw += mt1 * ((g1s.N + 255) / 256);
        tb += mt1 * ((g1s.N + 127) / 128);
      }
      const int nk64 = (conv ? g.tapC : g.K) / 64;
      auto ksp = [&](long blocks) -> long {   // the K split the deferred rule below will choose for `blocks` tiles
        if (epi != EPI_GATE_RES) return 1;
        if (g.ksplit > 0) return g.ksplit;
        if (!deferred) return 1;
        long w = 256 / blocks;
        if (w > nk64 / 4) w = nk64 / 4;
        if (w > g.partial_cap) w = g.partial_cap;
        return w < 1 ? 1 : (w > 16 ? 16 : w);
      };
      auto eff = [](long wg) { return (double)wg / (double)(((wg + 255) / 256) * 256); };
      const long kw = ksp(tw);
      const double ew = eff(tw * kw), eb = eff(tb * ksp(tb));
      GemmArgs gt = g;
      gt.ksplit = (int)kw;
      // a K split on top of it pays only where the 256x128 tiles leave the chip half empty (M = 3000, N = 1536: 144 workgroups
      // - w2 143 -> 97 us, fc2 99 -> 59 us with three K ranges); at M = 4000 the two slabs cost the next LayerNorm 12 us per launch
      // (pending form 21.6 vs 9.6 us) for 7 us won in the GEMM
      const bool split_ok = kw == 1 || eb < 0.6 || mid_split;
      bool vec1 = true;
      if (g1) {
        GemmArgs gt1 = g1s;
        gt1.ksplit = (int)kw;
        vec1 = gemm_vec_out_ok<T>(gt1, epi);
      }
      if (ew >= 0.74 && ew >= eb - 0.13 && split_ok && gemm_vec_out_ok<T>(gt, epi) && vec1) tile = conv ? 31 : 32;
    }
  }
  // A large conv grid whose 256-row tiles cover clearly less than one round of 256 CUs while 192-row tiles still fit it (w2 / linear1 at
  // M = 4000: 16 x 12 = 192 workgroups against 21 x 12 = 252) takes the 192x128 form: every CU works, and each workgroup streams
  // (192 + 128) instead of (256 + 128) rows per K-slice.  FOLEY_CONV3_192=0 keeps 256x128.
  if (tile_auto && tile == 23 && !g.wfmt && epi != EPI_SILUGATE_T && g.N % 128 == 0) {
    static const bool on192 = []() { const char* e = getenv("FOLEY_CONV3_192"); return !(e && e[0] == '0'); }();
    const long b256 = (long)((g.M + 255) / 256) * (g.N / 128), b192 = (long)((g.M + 191) / 192) * (g.N / 128);
    if (on192 && b256 <= 216 && b192 <= 256 && b192 > b256) tile = 24;
  }
  if (g.wfmt && tile != 15 && tile != 19 && tile != 21 && tile != 23 && tile != 31 && tile != 32) {   // fp8 weights exist only in the wave-specialised mainloops
    if (!tile_auto) return foley_set_err("GEMM: fp8 weights need tile 15, 19 or 21", __FILE__, __LINE__);
    tile = 15;
  }
  if (epi == EPI_QKV_SPLIT) {
    for (const GemmArgs* q : {(const GemmArgs*)&g, g1}) {
      if (!q) continue;
      const QkvSplitArgs& s = q->qs;
      if (s.nK < 1 || s.nK > 3 || s.H < 1 || q->N != s.nK * s.H * 128 || s.L < 1 || q->M % s.L)
        return foley_set_err("fused head split: N must be nK*H*128 and M a multiple of L", __FILE__, __LINE__);
      if (s.out_dtype != DtCode<T>::v)
        return foley_set_err("fused head split: output dtype must equal the operand dtype", __FILE__, __LINE__);
      if (s.vt_pitch && (sizeof(T) != 2 || s.vt_pitch % 8 || (s.tok_off + s.L) > s.vt_pitch))
        return foley_set_err("fused head split: bad transposed-V pitch", __FILE__, __LINE__);
      uintptr_t al = (uintptr_t)s.cos_tab | (uintptr_t)s.sin_tab | (uintptr_t)q->bias;
      for (int i = 0; i < s.nK; ++i) {
        if (!s.dst[i]) return foley_set_err("fused head split: null destination", __FILE__, __LINE__);
        al |= (uintptr_t)s.dst[i] | (uintptr_t)s.gain[i] | (uintptr_t)s.rcos[i] | (uintptr_t)s.rsin[i];
        if (s.pos[i] && (!s.cos_tab || !s.sin_tab)) return foley_set_err("fused head split: RoPE tables missing", __FILE__, __LINE__);
        if ((s.rcos[i] != nullptr) != (s.rsin[i] != nullptr) || (s.rcos[i] && !s.pos[i]))
          return foley_set_err("fused head split: gathered rotation rows need both tables and a position table", __FILE__, __LINE__);
      }
      if (al & 15) return foley_set_err("fused head split: operands must be 16-byte aligned", __FILE__, __LINE__);
    }
    const bool listed = tile == 1 || tile == 2 || tile == 5 || tile == 7 || tile == 8 || tile == 9 || tile == 15 || tile == 19 || tile == 25 ||
                        tile == 26 || tile == 27 || tile == 28 || tile == 29 || tile == 32;
    if (tile_auto && tile == 29 && !g.wfmt) {
      // Large grids: workgroups run in ceil(n / 256) rounds of (BM + 128) * 128 bytes per K-slice each - 192-row tiles
      // win when they save bytes without adding a round (M = 4000 q/k/v: 3 rounds either way, 320 instead of 384 rows
      // per workgroup and slice; FOLEY_WS192=0 keeps 256 rows)
      static const bool ws192 = []() { const char* e = getenv("FOLEY_WS192"); return !(e && e[0] == '0'); }();
      auto cost = [&](int bm) {
        long n = (long)((g.M + bm - 1) / bm) * (g.N / 128);
        if (g1) n += (long)((g1s.M + bm - 1) / bm) * (g1s.N / 128);
        return ((n + 255) / 256) * (bm + 128);
      };
      if (ws192 && cost(192) < cost(256)) tile = 28;
    }
    if (!listed || (tile_auto && tile == 25)) {
      long b128 = (long)((g.M + 127) / 128) * (g.N / 128);
      if (!listed) tile = b128 >= 24 ? (sizeof(T) == 2 ? (g.wfmt ? 15 : 25) : 5) : 2;
      // few 128-row tiles (the cross-attention q projection: 48 + 12 workgroups on 256 CUs): 64-row tiles double the
      // workgroups and move a quarter fewer bytes per workgroup and K-slice (FOLEY_WS64=0 keeps the 128-row tile)
      static const bool ws64 = []() { const char* e = getenv("FOLEY_WS64"); return !(e && e[0] == '0'); }();
      if (g1) b128 += (long)((g1s.M + 127) / 128) * (g1s.N / 128);
      if (tile == 25 && ws64 && b128 <= 100) tile = 27;
      // 96-row tiles when they still fit one round of workgroups (M = 500: 6 x 36 = 216): an eighth fewer bytes per
      // workgroup and K-slice than 128 rows (FOLEY_WS96=0 keeps 128)
      static const bool ws96 = []() { const char* e = getenv("FOLEY_WS96"); return !(e && e[0] == '0'); }();
      long b96 = (long)((g.M + 95) / 96) * (g.N / 128);
      if (g1) b96 += (long)((g1s.M + 95) / 96) * (g1s.N / 128);
      if (tile == 25 && ws96 && b96 <= 256 && b128 > 100 && (g.M % 128 == 0 ? false : (g.M + 95) / 96 * 96 - g.M < 96)) tile = 26;
    }
  }
  // Short-K plain layers of the large grids whose epilogue rules out a K split (fc1's GELU, the q/k/v head split; K = 1536 / 1408;
  // one problem or the audio + visual pair of a two-stream block): such a launch is whole ROUNDS of 256 workgroups, and per round a
  // 256x256 BK = 32 tile costs ~1.64x a 256x128 tile's round at K = 1536 for twice the area (tools/wide_bench.py, M = 4000 / 3000, bf16
  // and fp8 storage: 45 / 43 us against 27.5 / 26 us; a 192x128 round 23.7 us = 0.86).  The 256x256 tile is taken where that count
  // says it wins by more than 5 %: fc1 of the two-stream blocks at bs = 8 (M = 4000 + 640: 2 rounds against 4), q/k/v of the 30 s
  // clip (M = 3000: 216 / 252 tiles = ONE round against two of 256x128).  FOLEY_WIDE_SHORTK=0 keeps the 256x128 / 192x128 tiles.
  if (tile_auto && sizeof(T) == 2 && (epi == EPI_GELU_T || epi == EPI_QKV_SPLIT) && (tile == 19 || tile == 29 || tile == 28)) {
    static const bool on = []() { const char* e = getenv("FOLEY_WIDE_SHORTK"); return !(e && e[0] == '0'); }();
    auto plain = [&](const GemmArgs& q) {
      return q.taps == 1 && q.segV >= q.M && q.rstride <= 1 && q.tap0 == 0 && q.tapC % 32 == 0 && q.K >= 1024 && q.K < 2048 &&
             (epi == EPI_QKV_SPLIT || gemm_vec_out_ok<T>(q, epi));
    };
    auto tiles = [&](int bm, int bn) {
      long n = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
      if (g1) n += (long)((g1s.M + bm - 1) / bm) * ((g1s.N + bn - 1) / bn);
      return n;
    };
    if (on && plain(g) && (!g1 || plain(g1s))) {
      const double cur = (double)((tiles(tile == 28 ? 192 : 256, 128) + 255) / 256) * (tile == 28 ? 0.86 : 1.0);
      const double wide = (double)((tiles(256, 256) + 255) / 256) * 1.64;
      if (wide < 0.95 * cur) tile = 32;
    }
  }
  if (epi != EPI_GATE_RES || g.ksplit == 1 || (g.ksplit == 0 && sizeof(T) == 4)) {
    g.ksplit = 1;   // fp32 (parity) mode keeps a fixed summation order
  } else if (g.ksplit == 0) {
    // fill ~3 workgroups per CU, keep >= 12 K-slices per range
    static const int bm[33] = {0, 128, 64, 64, 128, 128, 64, 128, 64, 256, 0, 128, 0, 64, 0, 128, 0, 0, 0, 256, 0, 128, 256, 256, 192, 128, 0, 0, 0, 256, 0, 256, 256};
    static const int bn[33] = {0, 128, 128, 64, 64, 128, 64, 128, 128, 128, 0, 128, 0, 64, 0, 128, 0, 0, 0, 128, 0, 128, 64, 128, 128, 128, 0, 0, 0, 128, 0, 256, 256};
    if (tile < 0 || tile >= 33 || bm[tile] == 0) return foley_set_err("GEMM: unknown tile", __FILE__, __LINE__);
    const long blocks = (long)((g.M + bm[tile] - 1) / bm[tile]) * ((g.N + bn[tile] - 1) / bn[tile]);
    const int nk = (tile == 11 || tile == 13 || tile == 21 || tile == 22 || tile == 23 || tile == 24 || tile == 31) ? 3 * (g.tapC / BK) / 3 : g.K / BK;   // conv3 splits over channel chunks
    // small tiles want ~3 workgroups per CU; the large, efficient tiles only split when they
    // cannot even cover the chip once (the fp32 atomics are not free)
    const long target = (tile == 1 || tile == 5 || tile == 7 || tile == 9 || tile == 11 || tile == 15 || tile == 19 || tile == 21 || tile == 22 || tile == 23 || tile == 24 || tile == 25 || tile == 29 || tile >= 31) ? 192 : (tile == 13 ? 512 : 768);
    long want = (target + blocks - 1) / blocks;
    if (want > nk / 12) want = nk / 12;
    if (deferred) {   // one resident round of workgroups: as many K ranges as fit on 256 CUs (>= 4 slices each)
      const int per_cu = (tile == 3 || tile == 6 || tile == 13) ? 3 : 1;   // resident workgroups per CU
      long both = blocks;   // a two-problem launch (audio + visual stream) shares the round and the K split
      if (g1) both += (long)((g1->M + bm[tile] - 1) / bm[tile]) * ((g1->N + bn[tile] - 1) / bn[tile]);
      want = 256L * per_cu / both;
      if (tile == 21 && both > 256) want = both < 512 ? 2 : 1;   // mid-size grids: two K ranges beat a ragged second round (M = 3000: 68 -> 58 us)
      if (want > nk / 4) want = nk / 4;
    }
    g.ksplit = (int)(want < 1 ? 1 : (want > 16 ? 16 : want));
  }
  if (epi == EPI_GATE_RES && g.partials) {
    int cap = g.partial_cap;
    if (g1 && g1s.partial_cap < cap) cap = g1s.partial_cap;
    if (g1 && !g1s.partials) cap = 1;
    if (g.ksplit > cap) g.ksplit = cap < 1 ? 1 : cap;
  }
  if (g1) { g1s.ksplit = g.ksplit; g1s.partial_half = g.partial_half; }
  if (ksplit_used) *ksplit_used = g.ksplit;
  {
    // the direct-to-LDS loop addresses its operands through 32-bit buffer offsets
    auto extent = [&](GemmArgs& q) {
      const long rows_src = (long)((q.M + q.segV - 1) / q.segV) * q.segS;
      const long ab = rows_src * q.lda * (long)sizeof(T), wb = (long)q.N * q.ldw * (q.wfmt ? 1L : (long)sizeof(T));
      if (ab >= 0x7fff0000L || wb >= 0x7fff0000L) return false;
      q.a_bytes = (unsigned)ab;
      q.w_bytes = (unsigned)wb;
      return true;
    };
    bool ok = extent(g);
    if (g1) ok = extent(g1s) && ok;
    if (!ok && g.wfmt) return foley_set_err("GEMM: fp8-weight operands exceed the 2 GiB buffer-offset range", __FILE__, __LINE__);
    if (!ok && (tile == 21 || tile == 22 || tile == 23 || tile == 24)) return foley_set_err("GEMM: conv3 operands exceed the 2 GiB buffer-offset range", __FILE__, __LINE__);
    if (!ok && (tile == 31 || tile == 32)) return foley_set_err("GEMM: the 256x256 tiles range every load against 32-bit buffer extents: operands exceed the 2 GiB buffer-offset range", __FILE__, __LINE__);
    if (!ok && ((tile >= 5 && tile <= 9) || tile == 15 || tile == 19 || tile == 25 || tile == 26 || tile == 27 || tile == 28 || tile == 29)) tile = (tile == 6) ? 3 : ((tile == 8 || tile == 27) ? 2 : 1);   // register-staged twins
  }
  {
    // Large grids (several rounds of workgroups, activation panel larger than an L2): tile order [panel group][M tile][panel in group]
    // (tile_coords, gemm_ws_impl.h) so that the ~32 workgroups an XCD runs at a time cover a near-square block of tiles - panels
    // per group ~ sqrt(32 BM / BN) minimises the rows + columns the block pulls out of the fabric per K-slice.  Measured at
    // bs = 8 (one box, both orders twice): q/k/v 85 -> 77 us, fc1 115 -> 111, loop 1459 -> 1436 ms (+1.6 %); C5 +1 %.
    // FOLEY_TILE_GROUPS: unset = automatic, 0 = [panel][M tile] everywhere, n = n groups.
    static const int ngr = []() { const char* e = getenv("FOLEY_TILE_GROUPS"); return e ? atoi(e) : -1; }();
    int tbm = 0, tbn = 0;
    switch (tile) {
      case 15: case 25: case 21: tbm = 128; tbn = 128; break;
      case 19: case 29: case 23: tbm = 256; tbn = 128; break;
      case 28: case 24: tbm = 192; tbn = 128; break;
      case 31: case 32: tbm = 256; tbn = 256; break;
    }
    g.n_groups = 0;
    if (tbm && ngr != 0 && g.M >= 1536 && !g1) {
      const long tm_ = (g.M + tbm - 1) / tbm, tn_ = (g.N + tbn - 1) / tbn;
      if (tm_ * tn_ * (epi == EPI_GATE_RES ? g.ksplit : 1) > 256 && tn_ >= 2) {
        if (ngr > 0) g.n_groups = ngr;
        else {
          int pg = 1;
          while ((pg + 1) * (pg + 1) * tbn <= 32 * tbm) ++pg;   // floor(sqrt(32 BM / BN))
          int n = (int)((tn_ + pg / 2) / pg);
          g.n_groups = n < 1 ? 1 : n;
        }
      }
    }
  }
  {
    // Small grids (one round of workgroups), plain layers on the wave-specialised tiles: K-origin rotation per M tile (GemmArgs::k_rot,
    // gemm_ws_impl.h).  All M tiles of a weight panel start together and walk K in step, so every one of them waits for the SAME cold
    // line (one HBM fill, the others queued behind it in the L2) - each holds a slot of its CU's memory queue for the full HBM latency.
    // Started 1 / tiles_m of the range apart they take turns at the miss and find the other lines in the L2.  Only where the caller
    // opted in (g_gemm_krot_ok, gemm.hip: the summation order of a row then depends on its tile).  FOLEY_K_ROTATE=0: off.
    static const bool krot = []() { const char* e = getenv("FOLEY_K_ROTATE"); return !(e && e[0] == '0'); }();
    const int rbm = tile == 27 ? 64 : tile == 26 ? 96 : (tile == 15 || tile == 25) ? 128 : tile == 28 ? 192 : (tile == 19 || tile == 29) ? 256 : 0;
    g.k_rot = 0;
    if (krot && g_gemm_krot_ok && rbm && sizeof(T) == 2 && g.taps == 1 && (!g1 || g1s.taps == 1)) {
      const long ks_ = epi == EPI_GATE_RES ? g.ksplit : 1;
      long n = (long)((g.M + rbm - 1) / rbm) * ((g.N + 127) / 128) * ks_;
      if (g1) n += (long)((g1s.M + rbm - 1) / rbm) * ((g1s.N + 127) / 128) * ks_;
      if (n <= 256 && (g.M + rbm - 1) / rbm >= 2) g.k_rot = 1;
    }
    if (g1) g1s.k_rot = g.k_rot;
  }
  g.vec_out = gemm_vec_out_ok<T>(g, epi) ? 1 : 0;
  if (g1) g1s.vec_out = gemm_vec_out_ok<T>(g1s, epi) ? 1 : 0;
  // The four-consumer tiles hold 64 / 128 accumulator registers per wave: only the LDS-transposed vector
  // epilogue is instantiated for them (the scalar one made the compiler keep the 256x128 tile's accumulators
  // in scratch memory - 576 bytes per lane, loads / stores inside the K loop: 0.6 -> 0.37 ms for the
  // single-block modulation GEMM once it was gone).  Problems that need the scalar epilogue take the twins.
  if ((tile == 25 || tile == 29) && epi != EPI_QKV_SPLIT && !(g.vec_out && (!g1 || g1s.vec_out))) tile = tile == 25 ? 15 : 19;
  if ((tile == 27 || tile == 26 || tile == 28) && (epi != EPI_QKV_SPLIT || g.wfmt)) return foley_set_err("GEMM: tile 27 (64x128) serves the fused head split with bf16 weights only", __FILE__, __LINE__);
  if ((g.ldw != g.K || (g1 && g1s.ldw != g1s.K)) && tile != 31 && tile != 32 && !(tile == 15 || tile == 19 || tile == 21 || tile == 22 || tile == 23 || tile == 24 || tile == 25 || tile == 26 || tile == 27 || tile == 28 || tile == 29))
    return foley_set_err("GEMM: padded weight rows (ldw != K) need a wave-specialised tile", __FILE__, __LINE__);
  if (tile == 31 || tile == 32) {   // 256x256 tiles on the BK = 32 mainloop (gemm_wide_impl.h)
    if (epi == EPI_QKV_SPLIT) {   // the fused cross attention exists on the 64-row tile only: the caller launches the attention
      if (g.qs.attn_fused) *g.qs.attn_fused = 0;
      if (g1 && g1s.qs.attn_fused) *g1s.qs.attn_fused = 0;
    }
    if constexpr (__is_same(T, bf16_t)) return launch_gemm_wide_bf16(g, g1, epi, tile, st);
    else if constexpr (__is_same(T, f16_t)) return launch_gemm_wide_f16(g, g1, epi, tile, st);
    else return foley_set_err("GEMM: the 256x256 tiles serve 16-bit operands", __FILE__, __LINE__);
  }
  if (tile == 11 || tile == 13) {
    if (g1) return foley_set_err("conv3 kernel has no two-problem form", __FILE__, __LINE__);
    return launch_gemm_conv3(g, DtCode<T>::v, epi, tile == 11 ? 1 : 3, st);
  }
  // Cross-attention q projections may carry the attention in their epilogue (QkvSplitArgs::attn_out): taken when the problem
  // landed on the 64-row head-split tile (small grids, where the attention launch and its boundary cost more than its math);
  // otherwise the plain head split runs and the caller launches the attention (attn_fused tells).  FOLEY_CROSS_FUSED=0: never.
  if (epi == EPI_QKV_SPLIT) {
    static const bool fuse_on = []() { const char* e = getenv("FOLEY_CROSS_FUSED"); return !(e && e[0] == '0'); }();
    auto fits = [](const GemmArgs& q) {
      const QkvSplitArgs& s = q.qs;
      return s.attn_out && s.attn_k && s.attn_vt && s.nK == 1 && s.vt_pitch == 0 && s.attn_skv >= 1 && s.attn_skv <= 96 &&
             s.attn_pitch >= 96 && s.attn_pitch % 8 == 0 && s.attn_bdiv >= 1 &&
             ((long)s.L * s.attn_bdiv >= 64 || q.M <= 2L * s.L * s.attn_bdiv) &&   // a 64-row tile meets at most two text sets
             !(((uintptr_t)s.attn_k | (uintptr_t)s.attn_vt | (uintptr_t)s.attn_out) & 15);
    };
    const bool fuse = fuse_on && sizeof(T) == 2 && tile == 27 && fits(g) && (!g1 || fits(g1s));
    if (g.qs.attn_fused) *g.qs.attn_fused = fuse ? 1 : 0;
    if (g1 && g1s.qs.attn_fused) *g1s.qs.attn_fused = fuse ? 1 : 0;
    if (fuse) epi = EPI_QKV_ATTN;
  }
  if (tile == 15 || tile == 19 || tile == 21 || tile == 22 || tile == 23 || tile == 24 || tile == 25 || tile == 26 || tile == 27 || tile == 28 || tile == 29) {
    if constexpr (__is_same(T, bf16_t)) return launch_gemm_ws_bf16(g, g1, epi, tile, st);
    else if constexpr (__is_same(T, f16_t)) return launch_gemm_ws_f16(g, g1, epi, tile, st);
    else return foley_set_err("GEMM: wave-specialised tiles are bf16 only", __FILE__, __LINE__);
  }
  switch (tile) {
    case 1: return launch_tile<T, 128, 128, 4, 2, 4>(g, g1, epi, st);
    case 2: return launch_tile<T, 64, 128, 2, 2, 3>(g, g1, epi, st);
    case 3: return launch_tile<T, 64, 64, 2, 2, 4>(g, g1, epi, st);
    case 4: return launch_tile<T, 128, 64, 4, 1, 3>(g, g1, epi, st);
    // direct-to-LDS mainloop
    case 5: return launch_tile<T, 128, 128, 4, 2, 4, true>(g, g1, epi, st);
    case 6: return launch_tile<T, 64, 64, 2, 2, 4, true>(g, g1, epi, st);
    case 7:
      if constexpr (sizeof(T) == 2) return launch_tile<T, 128, 128, 2, 2, 4, true>(g, g1, epi, st);
      else return foley_set_err("GEMM: tile 7 is bf16 only", __FILE__, __LINE__);
    case 8: return launch_tile<T, 64, 128, 2, 2, 4, true>(g, g1, epi, st);
    case 9:  // 256x128, 8 waves of 64x64: fewest LDS bytes per MFMA; needs a large grid
      if constexpr (sizeof(T) == 2) return launch_tile<T, 256, 128, 4, 2, 3, true>(g, g1, epi, st);
      else return foley_set_err("GEMM: tile 9 is bf16 only", __FILE__, __LINE__);
  }
  return foley_set_err("GEMM: bad tile id", __FILE__, __LINE__);
}

}  // namespace

