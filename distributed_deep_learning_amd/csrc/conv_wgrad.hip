// Weight-gradient convolution for gfx950.
//
//   dw[k][r][s*C + c] = sum_{m=(n,p,q)} dy[m][k] * x[n, p*stride - pad + r, q*stride - pad + s, c]
//
// Per filter row r this is a GEMM  dY^T [K x M] * Xr [M x (S*C)]  whose reduction index m is the SLOW index of
// both operands in memory (NHWC).  Both tiles are therefore staged in their natural [pixel][channel] layout
// and the MFMA fragments are fetched with ds_read_b64_tr_b16 (the CDNA4 transposing LDS read): each 16-lane
// group pulls a 4-pixel x 16-channel block and receives it channel-major, i.e. 4 consecutive reduction
// elements of one channel per lane.  A and B use the identical read pattern, so the reduction order matches.
//
// Grid: (K tiles) x (S*C tiles) x R x split-M.  Each block walks its slice of M in 64-pixel steps
// (double-buffered LDS, register staging with zero fill for padding); split-M is sized so the grid fills 256 CUs.
// The split-M partial tiles meet in one of two ways:
//   * workspace given (the default, ops/kernels.py): every block STORES its fp32 tile into its split's slab of the
//     workspace ([splits][K][R][S*C], plain coalesced stores) and wgrad_reduce_kernel adds the slabs to dw in a FIXED
//     order - the gradient is bit-reproducible from run to run;
//   * no workspace (or one split only): global float atomics into dw, as in rounds 1-2 (order of the adds, and with
//     it the last bits of the result, differ from run to run).
// Either way dw is ACCUMULATED into (zero it for a plain product).
//
// Cost of the reproducible mode, measured (profiles/r3_wgrad_reproducible.txt): +0.4-0.5 ms per ResNet-50 step (0.6 % at
// batch 1000, 1.3 % at 512).  It is BYTES, not launches: the slabs are written and read once more than the atomics'
// traffic (~1.5 GB per step) in a step that is HBM-bound; running the reduce on a third stream, or inside the first
// microseconds of the NEXT filter gradient's kernel (no extra launch at all), measured the same or worse.
#include "common.h"
#include "host_util.h"
#include <stdlib.h>

struct WgradParams {
  const bf16_t* dy;   // [M][K]
  const bf16_t* x;    // [N][H][W][C]
  float* dw;          // [K][R][S*C]
  float* part;        // null, or the workspace [splits][K][R][S*C]: blocks store their tiles instead of adding atomically
  long long part_stride;   // K*R*S*C
  int plan_only;      // host: stop after sizing the grid (ddl_conv_wgrad_plan)
  float* ws;          // host: caller's workspace (may be null) and its size in floats
  long long ws_elems;
  int N, H, W, C, K, P, Q, R, S, stride, pad;
  int M, SC;
  int tiles_k, tiles_n, splits, m_per_split;
  int plain;          // 1x1, stride 1, pad 0: x row index == m
  FastDiv div_pq, div_q, div_c, div_sc;
  unsigned long long dy_bytes, x_bytes;   // operand sizes (buffer descriptors of the LDS-DMA kernel)
};

// Byte offset of 16-byte chunk `ch` of row `row` inside a [rows][TW channels] bf16 tile.
// 256-byte rows (TW = 128) and 128-byte rows (TW = 64) need different XOR patterns to keep both the
// ds_write_b128 staging stores and the transposed 8-byte reads conflict-free.
template <int TW>
__device__ __forceinline__ int tr_off(int row, int ch) {
  if (TW == 128) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
  return row * 128 + ((ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))) << 4);
}

// G = 4-wave groups per block.  With G = 2 the two groups walk the two halves of the block's M-slice through their
// own double buffers and meet in LDS before the atomics: the same waves and bytes in flight per CU as two
// independent blocks, but HALF the split-M partial tiles, i.e. half the float-atomic traffic (which is ~40 % of
// this kernel's time on the late layers: 512 blocks x 64 KiB = 32 MiB at ~1.3 TB/s = 25 us, all at the very end).
template <int TK, int TN, int G>
__global__ __launch_bounds__(256 * G) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int BKP = 64;                        // pixels per K-step
  constexpr int YCPR = TK / 8, XCPR = TN / 8;    // 16-byte chunks per tile row
  constexpr int YRPP = 256 / YCPR, XRPP = 256 / XCPR;
  constexpr int YI = BKP / YRPP, XI = BKP / XRPP;
  constexpr int YBYTES = BKP * TK * 2, XBYTES = BKP * TN * 2;
  constexpr int WK = TK / 2, WN = TN / 2;        // per-wave sub-tile
  constexpr int FK = WK / 16, FN = WN / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int grp = (G == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  char* const Ys = smem + grp * 2 * (YBYTES + XBYTES);
  char* const Xs = Ys + 2 * YBYTES;

  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  const int wk = wid & 1, wn = wid >> 1;

  // block -> (split, r, k tile, n tile), tiles fastest after the XCD remap: the tiles of one M-slice run
  // together on one XCD, so the dy / x rows they all stream are fetched from HBM once and re-read from that L2.
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = b % p.tiles_n; b /= p.tiles_n;
  const int tk = b % p.tiles_k; b /= p.tiles_k;
  const int r = b % p.R; b /= p.R;
  const int split = (int)b;
  const int k0 = tk * TK, nn0 = tn * TN;

  const int b_begin = split * p.m_per_split;
  const int b_end = min(p.M, b_begin + p.m_per_split);
  // group 0 takes the first half of the slice (rounded up to whole K-steps), group 1 the rest
  const int half = (G == 1) ? (b_end - b_begin) : (((b_end - b_begin + 1) / 2 + BKP - 1) / BKP) * BKP;
  const int m_begin = (grp == 0) ? b_begin : min(b_end, b_begin + half);
  const int m_end = (grp == 0) ? min(b_end, b_begin + half) : b_end;

  // ---- staging assignments ----
  const int ych = tid % YCPR, yrow0 = tid / YCPR;
  const int xch = tid % XCPR, xrow0 = tid / XCPR;
  const bool ycol_ok = (k0 + ych * 8) < p.K;
  const int nn = nn0 + xch * 8;                 // this thread's column inside the S*C axis
  const bool xcol_ok = nn < p.SC;
  const int xs = (int)fastdiv((unsigned)nn, p.div_c);   // filter column s
  const int xc = nn - xs * p.C;                           // channel

  u32x4 yr[YI], xr[XI];

  auto load_tile = [&](int mbase) {
#pragma unroll
    for (int i = 0; i < YI; ++i) {
      const int m = mbase + yrow0 + i * YRPP;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ycol_ok && m < m_end) v = *reinterpret_cast<const u32x4*>(p.dy + (size_t)m * (size_t)p.K + (size_t)(k0 + ych * 8));
      yr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int m = mbase + xrow0 + i * XRPP;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xcol_ok && m < m_end) {
        if (p.plain) {
          v = *reinterpret_cast<const u32x4*>(p.x + (size_t)m * (size_t)p.C + (size_t)xc);
        } else {
          const unsigned n = fastdiv((unsigned)m, p.div_pq);
          const unsigned rem = (unsigned)m - n * p.div_pq.d;
          const unsigned pp = fastdiv(rem, p.div_q);
          const unsigned qq = rem - pp * p.div_q.d;
          const int ih = (int)pp * p.stride - p.pad + r;
          const int iw = (int)qq * p.stride - p.pad + xs;
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
            const size_t off = ((size_t)((int)n * p.H * p.W + ih * p.W + iw)) * (size_t)p.C + (size_t)xc;
            v = *reinterpret_cast<const u32x4*>(p.x + off);
          }
        }
      }
      xr[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < YI; ++i)
      *reinterpret_cast<u32x4*>(Ys + buf * YBYTES + tr_off<TK>(yrow0 + i * YRPP, ych)) = yr[i];
#pragma unroll
    for (int i = 0; i < XI; ++i)
      *reinterpret_cast<u32x4*>(Xs + buf * XBYTES + tr_off<TN>(xrow0 + i * XRPP, xch)) = xr[i];
  };

  f32x4 acc[FK][FN];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane geometry: group g = lane>>4 owns reduction rows 8g..8g+7 of each 32-pixel sub-step;
  // inside the group lane 4q+pp addresses row q, 8 bytes at channel 4*pp of the 16-channel block.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp4 = lane & 3;

  // every group runs the step count of the longer (first) half: block-wide barriers stay uniform, and the steps a
  // shorter half has left over load zeros
  const int nsteps = (half + BKP - 1) / BKP;
  if (nsteps > 0) {
    load_tile(m_begin);
    store_tile(0);
  }
  __syncthreads();

  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1 < nsteps);
    if (more) load_tile(m_begin + (t + 1) * BKP);
    const char* yb = Ys + buf * YBYTES;
    const char* xb = Xs + buf * XBYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[FK], bfr[FN];
      const int rbase = ks * 32 + 8 * g + q4;
#pragma unroll
      for (int i = 0; i < FK; ++i) {
        const int c0 = (wk * WK + i * 16) >> 3;   // first 16-byte chunk of the 16-channel block
        const int ch = c0 + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c0 = (wn * WN + j * 16) >> 3;
        const int ch = c0 + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xb + tr_off<TN>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xb + tr_off<TN>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: D[row = k channel][col = (s,c)] -> fp32 tile in LDS -> row-contiguous float atomics ----
  // (each wave-instruction adds 256 contiguous bytes of one dw row: the shape the memory-side atomic units like)
  if (nsteps > 0 || p.part != nullptr) {                       // (a slab is written even by a block without pixels)
    float* const Ts = reinterpret_cast<float*>(smem);          // [TK][TN] fp32, reuses the staging buffers
#pragma unroll
    for (int gsel = 0; gsel < G; ++gsel) {                      // group 0 stores its tile, group 1 adds its own
      if (grp == gsel) {
#pragma unroll
        for (int i = 0; i < FK; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int col = wn * WN + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = wk * WK + i * 16 + (lane >> 4) * 4 + e;
              if (gsel == 0) Ts[row * TN + col] = acc[i][j][e];
              else           Ts[row * TN + col] += acc[i][j][e];
            }
          }
      }
      __syncthreads();
    }
    constexpr int LPR = TN / 64;                                // 64-lane pieces per row
    float* const out = p.part != nullptr ? p.part + (size_t)split * (size_t)p.part_stride : p.dw;
    for (int it = wid + 4 * grp; it < TK * LPR; it += 4 * G) {
      const int row = it / LPR, col = (it - row * LPR) * 64 + lane;
      const int k = k0 + row, cn = nn0 + col;
      if (k < p.K && cn < p.SC) {
        float* const dst = out + ((size_t)k * (size_t)p.R + (size_t)r) * (size_t)p.SC + (size_t)cn;
        if (p.part != nullptr) *dst = Ts[row * TN + col];
        else atomicAdd(dst, Ts[row * TN + col]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA ring version of the same kernel.  The phase trace of the conv kernels (bench/conv_trace.py) shows that a
// K-step costs (bytes fetched per CU) / (~70 GB/s per CU from L2), i.e. what matters is how many bytes a CU keeps in
// flight.  The register-staged loop above has ONE 64-pixel step (32 KiB per block) in flight and spends VALU on
// address math, zero fill and ds_write; here both operands go global -> LDS with `buffer_load ... lds` (natural
// [pixel][channel] layout, so no transform is needed: the transposing LDS read does the rest) into a 4-stage ring of
// 32-pixel steps, THREE steps in flight, zero fill by out-of-range buffer offsets, one counted-vmcnt barrier per
// step.  64 KiB of LDS per block at 128x128: two blocks per CU, 96 KiB in flight per CU.
// Issued from inline asm ON PURPOSE: hipcc models an LDS-DMA it can see as a pending LDS store and drains it with
// `s_waitcnt vmcnt(0)` before the next transposing LDS read (ds_read_b64_tr_b16 carries no alias information), which
// serialises the whole ring (seen in the .s: one vmcnt(0) per K-step, kernel 11 % SLOWER than register staging).
// Hidden in asm, the only waits are the counted ones below.  M0 = LDS byte address of the 1 KiB destination.
__device__ __forceinline__ void wg_lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");   // M0 has no other user in this kernel
}

__device__ __forceinline__ u32x4 wg_make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);     // stride 0, no swizzle
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);                    // num_records (bytes): range check
  r[3] = 0x00020000u;
  return r;
}

#ifndef WG_RING_STAGES
#define WG_RING_STAGES 4
#endif
template <int N> __device__ __forceinline__ void wg_wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// NSUB = 2 ("fold R"): the X tile is TWO TN-wide sub-tiles side by side and the column index runs over ALL filter rows,
// nn = (r, s, c) in [0, R*S*C): one block accumulates the whole [TK][R*S*C] slice of the filter gradient, so dy is staged
// once instead of once per filter row (grid dimension R collapses to 1).  For small filters on few channels (ResNet's
// space-to-depth stem: 4x4 taps x 16 channels = 256 columns, where the per-row version streams dy four times through L2).
// NSUB = 2 without FOLD: a [TK][2*TN] tile of ONE filter row (25 % less LDS and fabric traffic per MFMA than 128 x 128).
template <int TK, int TN, int NSUB = 1, bool FOLD = false, int NS_ = WG_RING_STAGES, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_ring_kernel(const WgradParams p) {
  constexpr int BKP = 32;                        // pixels per ring stage
  constexpr int NS = NS_;                        // ring stages (NS - 1 in flight)
  constexpr int YROWB = TK * 2, XROWB = TN * 2;  // bytes per tile row
  constexpr int YRW = 1024 / YROWB, XRW = 1024 / XROWB;      // rows per LDS-DMA wave-instruction (4 or 8)
  constexpr int YJ = BKP / (NW * YRW), XJ = BKP / (NW * XRW);
  static_assert(YJ >= 1 && XJ >= 1, "every wave stages at least one 1 KiB slice per operand");  // instructions per thread per stage (2 or 1)
  constexpr int XSUB = BKP * XROWB;                            // bytes of one X sub-tile
  constexpr int YBYTES = BKP * YROWB, XBYTES = XSUB * NSUB, STAGE = YBYTES + XBYTES;
  constexpr int WK = TK / 2, WN = TN * NSUB / (NW / 2);
  constexpr int FK = WK / 16, FN = WN / 16;
  constexpr unsigned OOB = 0xFFFFFFF0u;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wid & 1, wn = wid >> 1;

  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = b % p.tiles_n; b /= p.tiles_n;
  const int tk = b % p.tiles_k; b /= p.tiles_k;
  const int r = FOLD ? 0 : (int)(b % p.R);
  if (!FOLD) b /= p.R;
  const int split = (int)b;
  const int k0 = tk * TK, nn0 = tn * TN * NSUB;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  const u32x4 ysrc = wg_make_rsrc(p.dy, p.dy_bytes);
  const u32x4 xsrc = wg_make_rsrc(p.x, p.x_bytes);
  const unsigned smem_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- staging geometry.  Instruction j of wave w fills tile rows (j*4 + w)*RW .. +RW-1; lane -> (row, physical
  //      chunk); the XOR swizzle of tr_off<> depends only on row bits that are fixed per thread (see tr_off), so the
  //      LOGICAL chunk (= which 8 channels this lane fetches) is one constant per operand.
  const int yrow = lane / (YROWB / 16), yphys = lane % (YROWB / 16);
  const int xrow = lane / (XROWB / 16), xphys = lane % (XROWB / 16);
  int ych, xch;
  if (TK == 128) ych = yphys ^ (((yrow & 3) << 2) | (wid & 3));
  else           ych = yphys ^ ((((yrow >> 1) & 1) << 1) | ((wid & 1) << 2));
  if (TN == 128) xch = xphys ^ (((xrow & 3) << 2) | (wid & 3));
  else           xch = xphys ^ ((((xrow >> 1) & 1) << 1) | ((wid & 1) << 2));
  const bool ycol_ok = (k0 + ych * 8) < p.K;
  const unsigned ycol_off = (unsigned)(k0 + ych * 8);
  // column (r, s, c) of this lane's 8-channel chunk, per sub-tile
  bool xcol_ok[NSUB];
  int xs[NSUB], xc[NSUB], xr[NSUB];
#pragma unroll
  for (int u = 0; u < NSUB; ++u) {
    const int nn = nn0 + u * TN + xch * 8;
    xcol_ok[u] = nn < (FOLD ? p.R * p.SC : p.SC);
    const int rr = FOLD ? (int)fastdiv((unsigned)nn, p.div_sc) : r;
    const int rest = nn - (FOLD ? rr * p.SC : 0);
    xr[u] = rr;
    xs[u] = (int)fastdiv((unsigned)rest, p.div_c);
    xc[u] = rest - xs[u] * p.C;
  }

  // ---- load cursor.  Stages are issued strictly in order, 32 pixels apart, so nothing is decoded per stage: byte offsets
  //      advance by a constant, the row-in-range test is one compare against a scalar, and for stride-1 convolutions the
  //      input pixel of tap (r, s) is  n*H*W + (p - pad + r)*W + (q - pad + s)  with (p, q) carried incrementally
  //      (32 = qa * Q + qb: one conditional wrap of q, worth W - Q input pixels, and one of p, worth (H - P) * W; both
  //      are zero for the "same" 3x3 convs of ResNet / VGG).  The per-stage decode this replaces was two magic-number
  //      divisions and ~8 quarter-rate integer multiplies per DMA, about as many issue cycles as the stage's 16 MFMAs.
  // (the single conditional wrap of p needs qa + 1 <= P: images of at least ~6x6; smaller ones decode per stage)
  const bool s1 = !p.plain && p.stride == 1 && BKP / p.Q + 1 <= p.P;      // wave-uniform
  const int qa = BKP / p.Q, qb = BKP - qa * p.Q;
  int mcur = m_begin;
  unsigned yoff[YJ], xoff[NSUB][XJ];
  int ycm[YJ], xcm[XJ], xp[XJ], xq[XJ];
#pragma unroll
  for (int j = 0; j < YJ; ++j) {
    ycm[j] = (j * NW + wid) * YRW + yrow;
    yoff[j] = ((unsigned)(m_begin + ycm[j]) * (unsigned)p.K + ycol_off) * 2u;
  }
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    xcm[j] = (j * NW + wid) * XRW + xrow;
    const int m = m_begin + xcm[j];
    xp[j] = 0; xq[j] = 0;
    if (p.plain) {
#pragma unroll
      for (int u = 0; u < NSUB; ++u) xoff[u][j] = ((unsigned)m * (unsigned)p.C + (unsigned)xc[u]) * 2u;
    } else {
      const unsigned n = fastdiv((unsigned)m, p.div_pq);
      const unsigned rem = (unsigned)m - n * p.div_pq.d;
      xp[j] = (int)fastdiv(rem, p.div_q);
      xq[j] = (int)rem - xp[j] * (int)p.div_q.d;
      const int rowbase = ((int)n * p.H + xp[j]) * p.W + xq[j];
#pragma unroll
      for (int u = 0; u < NSUB; ++u)
        xoff[u][j] = ((unsigned)(rowbase + (xr[u] - p.pad) * p.W + (xs[u] - p.pad)) * (unsigned)p.C + (unsigned)xc[u]) * 2u;
    }
  }
  const unsigned ystep = (unsigned)(BKP * 2) * (unsigned)p.K;
  const unsigned xstep = p.plain ? (unsigned)(BKP * 2) * (unsigned)p.C : (unsigned)((qa * p.W + qb) * 2) * (unsigned)p.C;
  const unsigned qwrap = (unsigned)((p.W - p.Q) * 2) * (unsigned)p.C;                 // input pixels skipped when q wraps
  const unsigned pwrap = (unsigned)((p.H - p.P) * p.W * 2) * (unsigned)p.C;           // ... when p wraps (next image)

  auto issue = [&](int stage) {
    const unsigned sb = smem_addr + (unsigned)(stage * STAGE);
    const int lim = m_end - mcur;
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
      wg_lds_dma16(ysrc, sb + (unsigned)((j * NW + wid) * 1024), (ycol_ok && ycm[j] < lim) ? yoff[j] : OOB);
      yoff[j] += ystep;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const bool row_ok = xcm[j] < lim;
      if (p.plain) {                              // wave-uniform
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          wg_lds_dma16(xsrc, sb + (unsigned)(YBYTES + u * XSUB + (j * NW + wid) * 1024), (xcol_ok[u] && row_ok) ? xoff[u][j] : OOB);
          xoff[u][j] += xstep;
        }
      } else if (s1) {                            // wave-uniform
        unsigned adv = xstep;
        const int pq = xp[j], qq = xq[j];
        xq[j] += qb; xp[j] += qa;
        if (xq[j] >= p.Q) { xq[j] -= p.Q; ++xp[j]; adv += qwrap; }
        if (xp[j] >= p.P) { xp[j] -= p.P; adv += pwrap; }
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          const int ih = pq - p.pad + xr[u], iw = qq - p.pad + xs[u];
          const bool ok = xcol_ok[u] && row_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
          wg_lds_dma16(xsrc, sb + (unsigned)(YBYTES + u * XSUB + (j * NW + wid) * 1024), ok ? xoff[u][j] : OOB);
          xoff[u][j] += adv;
        }
      } else {                                    // branch-free: every lane decodes, invalid lanes read out of range
        const int m = mcur + xcm[j];
        const unsigned n = fastdiv((unsigned)m, p.div_pq);
        const unsigned rem = (unsigned)m - n * p.div_pq.d;
        const unsigned pp = fastdiv(rem, p.div_q);
        const unsigned qq = rem - pp * p.div_q.d;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          const int ih = (int)pp * p.stride - p.pad + xr[u];
          const int iw = (int)qq * p.stride - p.pad + xs[u];
          const bool ok = xcol_ok[u] && row_ok && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
          const unsigned off = ((unsigned)((int)n * p.H * p.W + ih * p.W + iw) * (unsigned)p.C + (unsigned)xc[u]) * 2u;
          wg_lds_dma16(xsrc, sb + (unsigned)(YBYTES + u * XSUB + (j * NW + wid) * 1024), ok ? off : OOB);
        }
      }
    }
    mcur += BKP;
  };

  f32x4 acc[FK][FN];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp4 = lane & 3;
  const int nsteps = (m_end - m_begin + BKP - 1) / BKP;
  constexpr int PER = YJ + XJ * NSUB;            // loads per thread per stage

  // prologue: NS-1 stages in flight (stages past the end are all-zero fills, so the counted waits stay uniform)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);

  int stage = 0;
  for (int t = 0; t < nsteps; ++t) {
    wg_wait_vmcnt<(NS - 2) * PER>();             // step t has landed; the next NS-2 steps stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // refill the stage that step t-1 vacated (every wave passed the barrier, so nobody reads it any more)
    issue(stage == 0 ? NS - 1 : stage - 1);
    const char* yb = smem + stage * STAGE;
    const char* xb = yb + YBYTES;
    {
      bf16x8 a[FK], bfr[FN];
      const int rbase = 8 * g + q4;
#pragma unroll
      for (int i = 0; i < FK; ++i) {
        const int c0 = (wk * WK + i * 16) >> 3;
        const int ch = c0 + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn * WN + j * 16;                    // column inside the (NSUB x TN)-wide X tile
        const char* const xsb = xb + (col / TN) * XSUB;      // its sub-tile
        const int c0 = (col % TN) >> 3;
        const int ch = c0 + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xsb + tr_off<TN>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xsb + tr_off<TN>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    stage = (stage == NS - 1) ? 0 : stage + 1;
  }
  wg_wait_vmcnt<0>();                            // the trailing zero fills must land before the ring is reused
  __syncthreads();

  if (nsteps > 0 || p.part != nullptr) {
    constexpr int TNW = TN * NSUB;
    float* const out = p.part != nullptr ? p.part + (size_t)split * (size_t)p.part_stride : p.dw;
    // fp32 tile through the ring, in EH passes of TK/EH rows when the whole tile does not fit
    constexpr int EH = (TK * TNW * 4 + NS * STAGE - 1) / (NS * STAGE);
    constexpr int WKH = WK / EH, FKH = FK / EH;
    static_assert(EH >= 1 && FK % EH == 0 && (TK / EH) * TNW * 4 <= NS * STAGE, "epilogue pass must fit in the ring");
    float* const Ts = reinterpret_cast<float*>(smem);          // [TK / EH][TNW] fp32, reuses the ring
#pragma unroll
    for (int h = 0; h < EH; ++h) {
      if (h > 0) __syncthreads();
#pragma unroll
      for (int ii = 0; ii < FKH; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int col = wn * WN + j * 16 + (lane & 15);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int lr = wk * WKH + ii * 16 + (lane >> 4) * 4 + e;
            Ts[lr * TNW + col] = acc[h * FKH + ii][j][e];
          }
        }
      __syncthreads();
      constexpr int LPR = TNW / 64;
      for (int it = wid; it < (TK / EH) * LPR; it += NW) {
        const int lr = it / LPR, col = (it - lr * LPR) * 64 + lane;
        const int row = (lr / WKH) * WK + h * WKH + (lr % WKH);
        const int k = k0 + row, cn = nn0 + col;
        float* dst = nullptr;
        if (FOLD) {                                // cn runs over (r, s, c): the filter row is part of the column
          if (k < p.K && cn < p.R * p.SC) dst = out + (size_t)k * (size_t)p.R * (size_t)p.SC + (size_t)cn;
        } else if (k < p.K && cn < p.SC) {
          dst = out + ((size_t)k * (size_t)p.R + (size_t)r) * (size_t)p.SC + (size_t)cn;
        }
        if (dst != nullptr) {
          if (p.part != nullptr) *dst = Ts[lr * TNW + col];
          else atomicAdd(dst, Ts[lr * TNW + col]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// "Halo" kernel: stride-1 filters with several taps (the 3x3 convs of ResNet / VGG, the 4x4 filter of the
// space-to-depth stem).  The ring kernel above treats every tap as its own GEMM column block, so a block streams x once
// per tap it owns and dy once per (filter row, column block): per 32-pixel step a 128 x 256 tile pulls 24 KiB through the
// L2 -> LDS path for 512 MFMA clocks, 47 B/clk per CU, and that path delivers ~27 B/clk per CU (bench/wgrad_ablate.py: the
// kernel with its MFMAs and fragment reads compiled out runs 60-85 % as long as the full one on the 3x3 layers).
// Here a block owns ALL taps of a [TK filters] x [TC input channels] slice.  A stage is a group of whole output rows
// (TR rows of one image, or G whole small images: <= 224 pixels = 7 MFMA steps of 32) staged ONCE together with the
// input rows they touch INCLUDING the halo ((TR + R - 1) x (Q + S - 1) positions, zero outside the image); every tap
// then reads the same LDS tile at a shifted position - the transposing LDS read takes one address per lane, so a shift
// is just an address offset.  L2 -> LDS bytes per MFMA drop 2.5x (14 x 14 x 256: 65 KiB per 4032 MFMA clocks).
//   dy tile  [224 rows][TK] with tr_off's swizzle; LDS row rho of a 32-row step holds pixel 4*(rho>>3) + (rho&3) +
//            16*((rho>>2)&1) of the step - a permutation of the reduction order that makes the 16 rows one x-read
//            touches CONSECUTIVE pixels (both operands use the same order, so the sum is unchanged);
//   x tile   one sub-tile [positions][16 channels] (32-byte rows, no swizzle needed: a wave-read is 512 contiguous
//            bytes) per 16-channel block.
// NB stages in LDS (2; 3 for the stem, whose stages are shorter than a trip to HBM): the stage NB - 1 ahead is requested
// (LDS-DMA, one instruction per tap from the stage's first tap on) while this one is multiplied; one barrier per stage.
// Waves: WKS x WNS = 4, each 64 filters x (all taps x 16 channels) - or, TAPSPLIT (stem, 16 channels in all), 64
// filters x (one filter row's taps x 16 channels).  One wave per SIMD (268 registers), so the loop is written for an
// in-order machine: everything about a load that does not change per stage is decoded once, the fragment reads run
// three taps ahead of the MFMAs, and the MFMA is issued with its accumulator tied in place (halo_mfma).
// Measured, batch 1000 (bench/wgrad_ablate.py): 3x3 x 64 at 56x56 0.50 -> 0.27 ms, 128 at 28x28 0.29 -> 0.25,
// 256 at 14x14 0.375 -> 0.24, 512 at 7x7 0.42 -> 0.22; the MFMAs alone take 0.157 of those (0.045 of it launch +
// atomic epilogue), i.e. they run at 95 % of peak when nothing else is in the loop.
// -DHALO_ABL=bits compiles parts of the main loop out (1: load requests, 2: MFMAs, 4: fragment reads) for
// bench/wgrad_ablate.py; the library is built with 0.
#ifndef HALO_ABL
#define HALO_ABL 0
#endif
struct HaloGeom {
  int G, TR, XR, Wp, XPOS, NV;        // images / output rows per stage, input rows per image, padded row length, positions, 1 KiB slices per sub-tile
  int nrb, nst, sps;                  // row blocks per image, stages in all, stages per split
  int px;                             // pixels per full stage (G * TR * Q)
  int xsub, stage_bytes;              // bytes of one x sub-tile / of one stage
  FastDiv div_xrwp, div_wp, div_nrb, div_trq;
};

// The MFMA with its accumulator TIED (dst == src C) in the AGPR file.  Through the builtin, hipcc gives the seven
// unrolled steps' results rotating registers and moves 76 of the 144 accumulator registers back at every stage
// boundary (v_accvgpr_mov / _write, ~125 per stage, in a burst the MFMA pipe waits for).  The waits for the fragment
// reads are still the compiler's (it tracks the asm's register operands); no other instruction touches the
// accumulators between the zero fill and the epilogue, which is fenced below.
__device__ __forceinline__ void halo_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int WKS, int WNS, int R_, int S_, bool TAPSPLIT, int XS, int NB>
__global__ __launch_bounds__(256) void conv_wgrad_halo_kernel(const WgradParams p, const HaloGeom hg) {
  constexpr int KS = 7, TK = 64 * WKS, TAPS = R_ * S_;
  constexpr int NCB = TAPSPLIT ? 1 : WNS;               // 16-channel blocks per tile
  constexpr int TC = 16 * NCB;
  constexpr int FK = 4, FN = TAPSPLIT ? TAPS / WNS : TAPS;
  constexpr int YROWB = TK * 2, YRW = 1024 / YROWB;     // dy rows per LDS-DMA instruction (4 or 8)
  constexpr int NYW = KS * 32 / YRW / 4;                // dy instructions per wave per stage (14 or 7)
  constexpr int NDMA = NYW + XS;                        // LDS-DMA instructions per wave per stage (XS: x slices, some may be idle)
  constexpr int PD = 3;                                 // the fragment reads run PD taps ahead of the MFMAs
  constexpr int YBYTES = KS * 32 * YROWB;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  static_assert(WKS * WNS == 4 && NDMA <= (KS - 3) * FN && PD < FN, "four waves; the next stage is requested early in this one");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = (WKS == 2) ? (wid & 1) : 0, wn = (WKS == 2) ? (wid >> 1) : wid;

  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int tc = b % p.tiles_n; b /= p.tiles_n;
  const int tk = b % p.tiles_k; b /= p.tiles_k;
  const int split = (int)b;
  const int k0 = tk * TK, c0 = tc * TC;
  const int st_begin = split * hg.sps;
  const int st_end = min(hg.nst, st_begin + hg.sps);

  const u32x4 ysrc = wg_make_rsrc(p.dy, p.dy_bytes);
  const u32x4 xsrc = wg_make_rsrc(p.x, p.x_bytes);
  const unsigned smem_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- load slots.  One wave, one SIMD, in-order issue: whatever address arithmetic a load needs is time the MFMA pipe
  //      idles, so everything that does not change from stage to stage is decoded ONCE, here.
  // dy: instruction u = e*4 + wid fills LDS rows u*YRW .. +YRW-1 (the row bits tr_off swizzles with are per-thread
  // constants, as in the ring kernel); yrel = byte offset of the lane's 16 bytes from the stage's first pixel.
  const unsigned krow2 = (unsigned)p.K * 2u;
  unsigned yrel[NYW];
  {
    const int yrow = lane / (YROWB / 16), yphys = lane % (YROWB / 16);
    int ych;
    if (TK == 128) ych = yphys ^ (((yrow & 3) << 2) | (wid & 3));
    else           ych = yphys ^ ((((yrow >> 1) & 1) << 1) | ((wid & 1) << 2));
#pragma unroll
    for (int e = 0; e < NYW; ++e) {
      const int row = (e * 4 + wid) * YRW + yrow;
      const int rr = row & 31;
      const int j = (row & ~31) + 4 * (rr >> 3) + (rr & 3) + 16 * ((rr >> 2) & 1);     // pixel of the stage in this LDS row
      yrel[e] = (unsigned)j * krow2 + (unsigned)(k0 + ych * 8) * 2u;
    }
  }
  // x: slice w = wid + 4k of the NCB * NV slices of a stage: 32 positions x 16 channels.  xrel = byte offset from input
  // pixel (n0, p0, 0) (may be negative: halo), xdyn = what the per-stage range check needs (input row, or image of the
  // group when a stage holds whole images), 0x40000000 when the position can never be valid.
  const int NXT = NCB * hg.NV;
  unsigned xrel[XS], xlds[XS];
  int xdyn[XS];
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    const int w = wid + 4 * k;
    int cb = 0;
    if (NCB > 1) cb = (w >= hg.NV) + (w >= 2 * hg.NV) + (w >= 3 * hg.NV);
    const int v = w - cb * hg.NV;
    const unsigned pos = (unsigned)(v * 32 + (lane >> 1));
    const unsigned gi = fastdiv(pos, hg.div_xrwp);
    const unsigned rem = pos - gi * hg.div_xrwp.d;
    const unsigned i = fastdiv(rem, hg.div_wp);
    const int iw = (int)(rem - i * hg.div_wp.d) - p.pad;
    const int ihr = (int)i - p.pad;                                 // input row relative to p0
    const bool st_ok = w < NXT && (int)pos < hg.XPOS && (unsigned)iw < (unsigned)p.W &&
                       (hg.G == 1 || (unsigned)ihr < (unsigned)p.H);
    xrel[k] = (unsigned)((((int)gi * p.H + ihr) * p.W + iw) * p.C + c0 + cb * 16 + (lane & 1) * 8) * 2u;
    xdyn[k] = st_ok ? (hg.G == 1 ? ihr : (int)gi) : 0x40000000;
    // idle slots (w >= NXT) zero-fill a 1 KiB scratch slice behind the two stages
    xlds[k] = w < NXT ? (unsigned)(YBYTES + cb * hg.xsub + v * 1024) : 0xFFFFFFFFu;
  }
  const unsigned dummy_lds = smem_addr + (unsigned)(NB * hg.stage_bytes);

  // per-stage scalars of the stage being fetched
  unsigned l_sb = 0, l_ybase = 0, l_ylim = 0, l_xbase = 0;
  int l_dyn0 = 0, l_dynlim = 0;
  auto begin_stage = [&](int st, int bufi) {
    const int ig = (int)fastdiv((unsigned)st, hg.div_nrb);
    const int rb = st - ig * hg.nrb;
    const int n0 = ig * hg.G, p0 = rb * hg.TR;
    l_sb = smem_addr + (unsigned)(bufi * hg.stage_bytes);
    l_ybase = (unsigned)((n0 * p.P + p0) * p.Q) * krow2;
    l_ylim = (unsigned)(min(hg.G, p.N - n0) * min(hg.TR, p.P - p0) * p.Q) * krow2;
    l_xbase = (unsigned)((n0 * p.H + p0) * p.W * p.C) * 2u;
    l_dyn0 = hg.G == 1 ? p0 : n0;
    l_dynlim = hg.G == 1 ? p.H : p.N;
  };
  auto issue_slot = [&](int slot) {                                 // slot: compile-time after unrolling
    if (slot < NYW) {
      wg_lds_dma16(ysrc, l_sb + (unsigned)((slot * 4 + wid) * 1024), yrel[slot] < l_ylim ? l_ybase + yrel[slot] : OOB);
    } else {
      const int k = slot - NYW;
      const bool ok = (unsigned)(l_dyn0 + xdyn[k]) < (unsigned)l_dynlim;
      const unsigned dst = __builtin_amdgcn_readfirstlane(xlds[k]);
      wg_lds_dma16(xsrc, dst == 0xFFFFFFFFu ? dummy_lds : l_sb + dst, ok ? l_xbase + xrel[k] : OOB);
    }
  };

  f32x4 acc[FK][FN];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment-read geometry.  Reduction slot (g, e = q + 4h) of a step is LDS row 8g + e of dy and pixel 4g + q + 16h.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp4 = lane & 3;
  int posb[KS][2];                                                  // byte offset of the lane's x row, per step and half
  {
    const int trq = hg.TR * p.Q;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int j = ks * 32 + 4 * g + q4 + 16 * h;
        if (j >= hg.px) j = hg.px - 1;                              // slots past the stage: dy is zero there, any finite x does
        const int gi = (int)fastdiv((unsigned)j, hg.div_trq), jr = j - gi * trq;
        const int pr = (int)fastdiv((unsigned)jr, p.div_q), qq = jr - pr * p.Q;
        posb[ks][h] = ((gi * hg.XR + pr) * hg.Wp + qq) * 32 + pp4 * 8;
      }
  }
  const int wp32 = hg.Wp * 32;

  // NB stages in LDS, NB - 1 of them requested ahead (stages past the end are all-zero fills, so the counted wait and
  // the tap loop need no branch).  NB = 3 where a stage is short (the stem: ~1800 MFMA clocks, less than a trip to HBM).
#pragma unroll
  for (int a = 0; a < NB - 1; ++a) {
    begin_stage(min(st_begin + a, hg.nst - 1), a);
    if (st_begin + a >= st_end) { l_ylim = 0; l_dynlim = 0; }
#pragma unroll
    for (int slot = 0; slot < NDMA; ++slot) issue_slot(slot);
  }
  int buf = 0;
  for (int st = st_begin; st < st_end; ++st) {
    wg_wait_vmcnt<(NB - 2) * NDMA>();            // this wave's slices of stage st have landed ...
    __builtin_amdgcn_s_barrier();                // ... everybody's have, and nobody reads the buffer of stage st - 1 any more
    asm volatile("" ::: "memory");
    const bool more = st + NB - 1 < st_end;
    begin_stage(more ? st + NB - 1 : st, buf == 0 ? NB - 1 : buf - 1);
    if (!more) { l_ylim = 0; l_dynlim = 0; }
    const char* const yb = smem + buf * hg.stage_bytes;
    const char* const xb = yb + YBYTES + (TAPSPLIT ? 0 : wn * hg.xsub);
    auto read_a = [&](int ks, bf16x8 (&a)[FK]) {
      const int rbase = ks * 32 + 8 * g + q4;
#pragma unroll
      for (int i = 0; i < FK; ++i) {
        const int cc = (wk * 64 + i * 16) >> 3;
        const int ch = cc + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
    };
    auto read_b = [&](int ks, int t) -> bf16x8 {                    // tap t of this wave: (r, s) = (t / S, t % S), or (wn, t)
      const int toff = TAPSPLIT ? (wn * wp32 + t * 32) : ((t / S_) * wp32 + (t % S_) * 32);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(xb + posb[ks][0] + toff));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(xb + posb[ks][1] + toff));
      s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    // One wave per SIMD: nothing else hides LDS latency, so the fragment reads run PD taps ahead of the MFMAs (the next
    // step's dy fragments ride along with its first tap), pinned in that order for the scheduler; the requests for the
    // NEXT stage go out one per tap from the first tap on, so the last of them has four steps to land.
    bf16x8 a0[FK], a1[FK], bq[PD + 1];
    read_a(0, a0);
    if (HALO_ABL & 4) { read_a(0, a1); bq[PD] = read_b(0, PD); }
#pragma unroll
    for (int t = 0; t < PD; ++t) bq[t] = read_b(0, t);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 (&ac)[FK] = (ks & 1) ? a1 : a0;
      bf16x8 (&an)[FK] = (ks & 1) ? a0 : a1;
#pragma unroll
      for (int t = 0; t < FN; ++t) {
        const int slot = ks * FN + t;
        if (slot < NDMA && !(HALO_ABL & 1)) issue_slot(slot);
        const int nks = (slot + PD) / FN, nt = (slot + PD) % FN;     // the tap PD ahead
        if (nks < KS && !(HALO_ABL & 4)) {
          if (nt == 0) read_a(nks, an);
          bq[(slot + PD) % (PD + 1)] = read_b(nks, nt);
        }
        if (HALO_ABL & 2) {
#pragma unroll
          for (int i = 0; i < FK; ++i) asm volatile("" :: "v"(ac[i]));
          asm volatile("" :: "v"(bq[slot % (PD + 1)]));
        } else {
#pragma unroll
          for (int i = 0; i < FK; ++i) halo_mfma(acc[i][t], ac[i], bq[slot % (PD + 1)]);
        }
        __builtin_amdgcn_sched_barrier(0);      // taps stay in program order: requests, reads PD taps ahead, MFMAs
      }
    }
    buf = (buf == NB - 1) ? 0 : buf + 1;
  }
  wg_wait_vmcnt<0>();                             // the zero fills requested during the last stages
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMAs have written the accumulators before anything reads them
  __syncthreads();                                // all fragment reads done (no load is in flight: the last stage fetched nothing)

  // ---- epilogue: the [TK][TAPS][TC] tile goes through LDS in two passes of TK/2 filters, then out in runs of TC floats
  if (st_begin < st_end || p.part != nullptr) {
    constexpr int TNW = TAPS * TC;
    float* const out = p.part != nullptr ? p.part + (size_t)split * (size_t)p.part_stride : p.dw;
    float* const Ts = reinterpret_cast<float*>(smem);          // [TK / 2][TNW] fp32 (<= 72 KiB), reuses the stages
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (hh > 0) __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int t = 0; t < FN; ++t) {
          const int col = TAPSPLIT ? (wn * FN + t) * 16 + (lane & 15) : t * TC + wn * 16 + (lane & 15);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int lr = wk * 32 + ii * 16 + (lane >> 4) * 4 + e;
            Ts[lr * TNW + col] = acc[hh * 2 + ii][t][e];
          }
        }
      __syncthreads();
      for (int f = tid; f < (TK / 2) * TNW; f += 256) {
        const int lr = f / TNW, col = f - lr * TNW;
        const int k = k0 + (lr >> 5) * 64 + hh * 32 + (lr & 31);
        const int tap = col / TC, c = col - tap * TC;
        float* const dst = out + ((size_t)k * TAPS + (size_t)tap) * (size_t)p.C + (size_t)(c0 + c);
        if (p.part != nullptr) *dst = Ts[f];
        else atomicAdd(dst, Ts[f]);
      }
    }
  }
}

// dw[e] += sum_s part[s][e] with the splits summed in a FIXED order: split-lane ty of a block adds splits ty, ty + NY, ...
// front to back, and the NY partial sums are combined in a fixed order through LDS.  NX lanes of 16 bytes side by side:
// NX = 64 (1 KiB per wave-load) for big filters; NX = 16 with 64 split-lanes for small filters with hundreds of splits
// (a 64 x 64 1x1 filter has 1024 column groups: 16 blocks of the wide shape would leave the chip empty).
template <int NX, int NY>
__global__ __launch_bounds__(NX * NY) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                               long long n4, long long stride, int splits) {
  __shared__ f32x4 red[NY][NX];
  const int tx = threadIdx.x % NX, ty = threadIdx.x / NX;
  const long long e = (long long)blockIdx.x * NX + tx;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e < n4) {
    const float* src = part + e * 4 + (long long)ty * stride;
#pragma unroll 4
    for (int sp = ty; sp < splits; sp += NY, src += (long long)NY * stride) s += *reinterpret_cast<const f32x4*>(src);
  }
  red[ty][tx] = s;
  __syncthreads();
  if constexpr (NY > 16) {                         // two ordered levels: 8 groups of NY/8, then the 8 group sums
    if (ty < 8) {
      f32x4 t = red[ty * (NY / 8)][tx];
#pragma unroll
      for (int y = 1; y < NY / 8; ++y) t += red[ty * (NY / 8) + y][tx];
      red[ty * (NY / 8)][tx] = t;
    }
    __syncthreads();
    if (ty == 0 && e < n4) {
      f32x4 t = red[0][tx];
#pragma unroll
      for (int y = 1; y < 8; ++y) t += red[y * (NY / 8)][tx];
      f32x4* const d = reinterpret_cast<f32x4*>(dw + e * 4);
      *d = *d + t;
    }
  } else if (ty == 0 && e < n4) {
    f32x4 t = red[0][tx];
#pragma unroll
    for (int y = 1; y < NY; ++y) t += red[y][tx];
    f32x4* const d = reinterpret_cast<f32x4*>(dw + e * 4);
    *d = *d + t;
  }
}

// Called by every launcher once the split count is known: picks the meeting mode (see the file header).  Returns false
// when only the plan was wanted.
static bool wgrad_prepare(WgradParams& p) {
  p.part_stride = (long long)p.K * p.R * p.SC;
  p.part = nullptr;
  if (p.plan_only) return false;
  if (p.ws != nullptr && p.splits > 1 && (long long)p.splits * p.part_stride <= p.ws_elems && p.part_stride % 4 == 0)
    p.part = p.ws;
  return true;
}
static int wgrad_finish(const WgradParams& p, hipStream_t stream) {
  const int rc = (int)hipGetLastError();
  if (rc != 0 || p.part == nullptr) return rc;
  const long long n4 = p.part_stride / 4;
  const unsigned grid = (unsigned)((n4 + 63) / 64);
  if (p.splits <= 8)
    hipLaunchKernelGGL((wgrad_reduce_kernel<64, 4>), dim3(grid), dim3(256), 0, stream, p.part, p.dw, n4, p.part_stride, p.splits);
  else if (grid >= 512 || p.splits < 64)
    hipLaunchKernelGGL((wgrad_reduce_kernel<64, 16>), dim3(grid), dim3(1024), 0, stream, p.part, p.dw, n4, p.part_stride, p.splits);
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<16, 64>), dim3((unsigned)((n4 + 15) / 16)), dim3(1024), 0, stream, p.part, p.dw,
                       n4, p.part_stride, p.splits);
  return (int)hipGetLastError();
}

template <int TK, int TN, int G>
static int launch_wgrad_g(WgradParams& p, hipStream_t stream, int smem, unsigned grid) {
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_wgrad_kernel<TK, TN, G>), smem);
  if (rc != 0) return rc;
  hipLaunchKernelGGL((conv_wgrad_kernel<TK, TN, G>), dim3(grid), dim3(256 * G), smem, stream, p);
  return wgrad_finish(p, stream);
}

template <int TK, int TN>
static int launch_wgrad(WgradParams& p, hipStream_t stream) {
  p.tiles_k = (p.K + TK - 1) / TK;
  p.tiles_n = (p.SC + TN - 1) / TN;
  const long long tiles = (long long)p.tiles_k * p.tiles_n * p.R;
  // One full round of resident blocks (2 per CU x 256 CUs), never more: a grid of 1.1 rounds runs as long as 2,
  // and every extra split adds a tile of fp32 atomic traffic (~1.3 TB/s chip-wide).  Splits are never shorter
  // than 256 pixels (4 K-steps).
  // DDL_WGRAD_RING=0 falls back to the register-staged kernel (also used for operands beyond 4 GiB).  Measured on
  // ResNet-50 (bench/conv_layers.py w): the ring is ~15 % faster at batch 512 once its DMA issue is hidden from
  // hipcc's waitcnt insertion (see wg_lds_dma16).
  static const int use_ring = [] { const char* e = getenv("DDL_WGRAD_RING"); return e ? atoi(e) : 1; }();
  constexpr int groups = 2;
  const bool ring = use_ring && p.dy_bytes < 0xFFFF0000ull && p.x_bytes < 0xFFFF0000ull;
  // Measured per layer (bench/conv_layers.py, B=256): pairing the groups pays where the atomic tail dominates, i.e. the
  // plain 1x1 convs with a short reduction per tile (-8..-20 %); 3x3 and the big first-in-stage 1x1 layers lose to the
  // 8-wave barriers (+3..+50 %), so they keep independent 4-wave blocks.
  const double flops = 2.0 * (double)p.M * (double)p.K * (double)p.C * (double)p.R * (double)p.S;
  const int G = (!ring && groups == 2 && p.plain && flops < 40e9) ? 2 : 1;
  const int smem = ring ? WG_RING_STAGES * 32 * (TK + TN) * 2 : G * 2 * 64 * (TK + TN) * 2;
  constexpr int cap_env = 0;
  const int max_cu = 4 / G;                                                   // 16 waves per CU at most
  constexpr int lds_budget = 160;
  int per_cu = (lds_budget * 1024) / smem < max_cu ? (lds_budget * 1024) / smem : max_cu;   // LDS- and register-limited residency
  if (per_cu < 1) per_cu = 1;
  const long long capacity = cap_env > 0 ? cap_env : 256 * per_cu;
  long long splits = capacity / tiles;
  const long long max_splits = (p.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  long long mps = (p.M + splits - 1) / splits;
  mps = ((mps + 64 * G - 1) / (64 * G)) * (64 * G);
  splits = (p.M + mps - 1) / mps;
  p.splits = (int)splits;
  p.m_per_split = (int)mps;
  if (!wgrad_prepare(p)) return 0;
  const unsigned grid = (unsigned)(tiles * splits);
  if (ring) {
    static SmemAttrOnce once_ring;
    const int rc = once_ring.ensure(reinterpret_cast<const void*>(&conv_wgrad_ring_kernel<TK, TN>), smem);
    if (rc != 0) return rc;
    hipLaunchKernelGGL((conv_wgrad_ring_kernel<TK, TN>), dim3(grid), dim3(256), smem, stream, p);
    return wgrad_finish(p, stream);
  }
  if (G == 2) return launch_wgrad_g<TK, TN, 2>(p, stream, smem, grid);
  return launch_wgrad_g<TK, TN, 1>(p, stream, smem, grid);
}

// Fold-R launch: one tile [64][R*S*C <= 256], every resident block slot is a split of M.
static int launch_wgrad_fold(WgradParams& p, hipStream_t stream) {
  constexpr int TK = 64, TN = 128, NSUB = 2;
  constexpr bool FOLD = true;
  p.tiles_k = 1; p.tiles_n = 1;
  const int smem = WG_RING_STAGES * 32 * (TK + TN * NSUB) * 2;          // 2 blocks per CU at 4 stages
  constexpr int cap_env = 512;
  long long splits = cap_env;
  const long long max_splits = (p.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  long long mps = (p.M + splits - 1) / splits;
  mps = ((mps + 63) / 64) * 64;
  splits = (p.M + mps - 1) / mps;
  p.splits = (int)splits;
  p.m_per_split = (int)mps;
  if (!wgrad_prepare(p)) return 0;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_wgrad_ring_kernel<TK, TN, NSUB, FOLD>), smem);
  if (rc != 0) return rc;
  hipLaunchKernelGGL((conv_wgrad_ring_kernel<TK, TN, NSUB, FOLD>), dim3((unsigned)splits), dim3(256), smem, stream, p);
  return wgrad_finish(p, stream);
}

// Wide launch: [128][256] tiles of one filter row, three ring stages (72 KiB: two blocks per CU, as the 128 x 128 tile).
static int launch_wgrad_wide(WgradParams& p, hipStream_t stream, int capacity) {
  constexpr int TK = 128, TN = 128, NSUB = 2, NS = 3;
  p.tiles_k = (p.K + TK - 1) / TK;
  p.tiles_n = (p.SC + TN * NSUB - 1) / (TN * NSUB);
  const long long tiles = (long long)p.tiles_k * p.tiles_n * p.R;
  const int smem = NS * 32 * (TK + TN * NSUB) * 2;
  long long splits = capacity / tiles;
  const long long max_splits = (p.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  long long mps = (p.M + splits - 1) / splits;
  mps = ((mps + 63) / 64) * 64;
  splits = (p.M + mps - 1) / mps;
  p.splits = (int)splits;
  p.m_per_split = (int)mps;
  if (!wgrad_prepare(p)) return 0;
  // (eight waves per block, 64 x 64 each, measured 0.1 % slower in the step than four waves of 64 x 128)
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_wgrad_ring_kernel<TK, TN, NSUB, false, NS>), smem);
  if (rc != 0) return rc;
  hipLaunchKernelGGL((conv_wgrad_ring_kernel<TK, TN, NSUB, false, NS>), dim3((unsigned)(tiles * splits)), dim3(256), smem, stream, p);
  return wgrad_finish(p, stream);
}

static int g_halo_min_stages = 20;
static long long g_halo_launches = 0;
// Test hook: launches of the halo kernel so far (a parity test that silently ran the ring kernels proves nothing).
extern "C" long long ddl_wgrad_halo_launches() { return g_halo_launches; }
// Test hook: the smallest run of stages per block for which the halo kernel is chosen (0: whenever the shape fits).
extern "C" void ddl_wgrad_halo_min_stages(int n) { g_halo_min_stages = n; }

// Halo launch (see conv_wgrad_halo_kernel).  Returns -100 when the problem does not fit the kernel's assumptions (the
// caller then takes the ring kernels).
template <int WKS, int WNS, int R_, int S_, bool TAPSPLIT, int XS, int NB>
static int launch_wgrad_halo(WgradParams& p, hipStream_t stream) {
  constexpr int TK = 64 * WKS, NCB = TAPSPLIT ? 1 : WNS, TC = 16 * NCB, KS = 7;
  if (p.K % TK != 0 || p.C % TC != 0 || p.Q > KS * 32) return -100;
  HaloGeom hg;
  const int pq = p.P * p.Q;
  if (pq <= KS * 32) { hg.TR = p.P; hg.G = (KS * 32) / pq; if (hg.G > p.N) hg.G = p.N; }
  else               { hg.G = 1; hg.TR = (KS * 32) / p.Q; }
  hg.px = hg.G * hg.TR * p.Q;
  if (hg.px < 5 * 32) return -100;                              // more than two of the seven steps would be padding
  hg.XR = hg.TR + R_ - 1;
  hg.Wp = p.Q + S_ - 1;
  hg.XPOS = hg.G * hg.XR * hg.Wp;
  hg.NV = (hg.XPOS + 31) / 32;
  hg.xsub = hg.NV * 1024;
  hg.stage_bytes = KS * 32 * TK * 2 + NCB * hg.xsub;
  const int smem = NB * hg.stage_bytes + 1024;                  // + the scratch slice idle load slots zero-fill
  if (smem > 160 * 1024 || smem < (TK / 2) * R_ * S_ * TC * 4 || (NCB * hg.NV + 3) / 4 > XS) return -100;
  hg.nrb = (p.P + hg.TR - 1) / hg.TR;
  hg.nst = ((p.N + hg.G - 1) / hg.G) * hg.nrb;
  hg.div_xrwp = make_fastdiv((unsigned)(hg.XR * hg.Wp));
  hg.div_wp = make_fastdiv((unsigned)hg.Wp);
  hg.div_nrb = make_fastdiv((unsigned)hg.nrb);
  hg.div_trq = make_fastdiv((unsigned)(hg.TR * p.Q));
  p.tiles_k = p.K / TK;
  p.tiles_n = p.C / TC;
  const long long tiles = (long long)p.tiles_k * p.tiles_n;
  long long splits = 256 / tiles;
  if (splits < 1) splits = 1;
  if (splits > hg.nst) splits = hg.nst;
  hg.sps = (int)((hg.nst + splits - 1) / splits);
  splits = (hg.nst + hg.sps - 1) / hg.sps;
  // A block needs a run of stages to pay for its prologue (per-lane read / request geometry) and the cold first stage:
  // below ~20 stages per block the ring kernels win (swept 0 / 12 / 20 / never on ResNet-18-CIFAR at batch 32 / 128 / 512
  // and ResNet-50 at batch 256, profiles/r3_wgrad_halo_kernel.txt).
  if (hg.sps < g_halo_min_stages) return -100;
  p.splits = (int)splits;
  p.m_per_split = hg.sps * hg.px;
  if (!wgrad_prepare(p)) return 0;
  ++g_halo_launches;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_wgrad_halo_kernel<WKS, WNS, R_, S_, TAPSPLIT, XS, NB>), 160 * 1024);
  if (rc != 0) return rc;
  hipLaunchKernelGGL((conv_wgrad_halo_kernel<WKS, WNS, R_, S_, TAPSPLIT, XS, NB>), dim3((unsigned)(tiles * splits)), dim3(256), smem,
                     stream, p, hg);
  return wgrad_finish(p, stream);
}

// The work of ddl_conv_wgrad / ddl_conv_wgrad_plan: fills the problem, picks the tile family, sizes the grid, launches.
static int wgrad_dispatch(WgradParams& p, hipStream_t st) {
  const int K = p.K, R = p.R;
  // stride-1 filters with several taps: all taps from one staged tile (DDL_WGRAD_HALO=0: the ring kernels below)
  static const int halo_on = [] { const char* e = getenv("DDL_WGRAD_HALO"); return e ? atoi(e) : 1; }();
  if (halo_on && p.stride == 1 && p.dy_bytes < 0xFFFF0000ull && p.x_bytes < 0xFFFF0000ull) {
    int rc = -100;
    if (R == 3 && p.S == 3) {
      if (K % 128 == 0 && p.C % 32 == 0) rc = launch_wgrad_halo<2, 2, 3, 3, false, 6, 2>(p, st);
      else if (K % 64 == 0 && p.C % 64 == 0) rc = launch_wgrad_halo<1, 4, 3, 3, false, 11, 2>(p, st);
    } else if (R == 4 && p.S == 4 && p.C == 16 && K % 64 == 0) {
      rc = launch_wgrad_halo<1, 4, 4, 4, true, 5, 3>(p, st);
    }
    if (rc != -100) return rc;
  }
  // small filters on few channels (the space-to-depth stem: 4 x 4 x 16): all filter rows in one block, dy staged once
  static const int fold_on = [] { const char* e = getenv("DDL_WGRAD_FOLD"); return e ? atoi(e) : 1; }();
  if (fold_on && K <= 64 && R > 1 && R * p.SC <= 256 && R * p.SC > 128 && p.dy_bytes < 0xFFFF0000ull && p.x_bytes < 0xFFFF0000ull)
    return launch_wgrad_fold(p, st);
  const bool k64 = K <= 64, n64 = p.SC <= 64 || (p.SC % 128 != 0 && p.SC % 64 == 0 && p.SC < 512);
  // Wide [128][256] tiles, ONE block per CU (256 resident blocks): the same bytes in flight as two 128 x 128 blocks with
  // 25 % less LDS and fabric traffic per MFMA.  Alone the filter gradients are no faster this way (8.3 vs 8.2 ms per
  // ResNet-50 step); next to the data-gradient / BatchNorm chain of the main stream, which is where they run, the step
  // gains 0.8 ms of 37.4 (docs/STATUS.md, round 2).  DDL_WGRAD_WIDE=0 restores 128 x 128 everywhere.
  static const int wide_on = [] { const char* e = getenv("DDL_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
  constexpr int wide_pad = 0;
  constexpr int wide_cap = 256;
  const int sc_pad = (p.SC + 255) / 256 * 256;
  const bool wide_fit = K % 128 == 0 && (p.SC % 256 == 0 || (wide_pad && p.SC >= 256 && sc_pad * 100 <= p.SC * 115));
  if (wide_on && wide_fit && p.dy_bytes < 0xFFFF0000ull && p.x_bytes < 0xFFFF0000ull)
    return launch_wgrad_wide(p, st, wide_cap);
  if (k64 && n64) return launch_wgrad<64, 64>(p, st);
  if (k64) return launch_wgrad<64, 128>(p, st);
  if (n64) return launch_wgrad<128, 64>(p, st);
  return launch_wgrad<128, 128>(p, st);
}

static int wgrad_fill(WgradParams& p, const void* dy, const void* x, float* dw, int N, int H, int W, int C, int K, int P,
                      int Q, int R, int S, int stride, int pad) {
  if (C % 8 != 0 || K % 8 != 0) return -1;
  const long long M = (long long)N * P * Q;
  if (M >= (1ll << 31) || (long long)N * H * W >= (1ll << 31) || M <= 0) return -3;
  p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.P = P; p.Q = Q; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.M = (int)M; p.SC = S * C;
  p.plain = (R == 1 && S == 1 && stride == 1 && pad == 0 && P == H && Q == W) ? 1 : 0;
  p.div_pq = make_fastdiv((unsigned)(P * Q));
  p.div_q = make_fastdiv((unsigned)Q);
  p.div_c = make_fastdiv((unsigned)C);
  p.div_sc = make_fastdiv((unsigned)(S * C));
  p.dy_bytes = (unsigned long long)M * K * 2ull;
  p.x_bytes = (unsigned long long)N * H * W * C * 2ull;
  p.part = nullptr; p.part_stride = 0; p.plan_only = 0; p.ws = nullptr; p.ws_elems = 0; p.splits = 1;
  return 0;
}

// dw (fp32, [K][R][S][C]) is ACCUMULATED into.  ws: null (float atomics) or a scratch buffer of ws_elems floats, at least
// ddl_conv_wgrad_plan(...) of them (split-M partial slabs + ordered reduce: reproducible to the bit).
extern "C" int ddl_conv_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, int C, int K, int P,
                              int Q, int R, int S, int stride, int pad, float* ws, long long ws_elems, void* stream) {
  WgradParams p;
  const int rc = wgrad_fill(p, dy, x, dw, N, H, W, C, K, P, Q, R, S, stride, pad);
  if (rc != 0) return rc;
  p.ws = ws; p.ws_elems = ws_elems;
  return wgrad_dispatch(p, (hipStream_t)stream);
}

// Floats of workspace the reproducible mode needs for this problem (0: one split, nothing to reduce); < 0: bad problem.
extern "C" long long ddl_conv_wgrad_plan(int N, int H, int W, int C, int K, int P, int Q, int R, int S, int stride, int pad) {
  WgradParams p;
  const int rc = wgrad_fill(p, nullptr, nullptr, nullptr, N, H, W, C, K, P, Q, R, S, stride, pad);
  if (rc != 0) return rc;
  p.plan_only = 1;
  const int rc2 = wgrad_dispatch(p, nullptr);
  if (rc2 != 0) return rc2 < 0 ? rc2 : -rc2;
  return p.splits > 1 ? (long long)p.splits * p.part_stride : 0;
}
