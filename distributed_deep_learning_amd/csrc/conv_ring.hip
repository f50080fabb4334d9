// Implicit-GEMM convolution, LDS-DMA ring version (gfx950): forward and data-gradient.
//
// Same math, operand roles, LDS swizzle and epilogue as conv_igemm.hip; what changes is the staging pipeline.
// The register-staged kernel keeps ONE K-step of loads in flight and pays a full memory round trip per 64-deep
// K-step — fatal for the short reductions of ResNet's 1x1 convs (4..32 K-steps).  Here:
//
//   * tile 256 pixels x BN channels x BK (64 or 32), 512 threads = 8 waves (4 x 2).  BK=64: 144 KiB ring, one block
//     per CU, fewest barriers - for long reductions.  BK=32: 72 KiB ring, TWO blocks per CU, so one block's
//     prologue / epilogue overlaps the other's MFMA loop - for the short reductions of 1x1 convs, which are otherwise
//     latency-serialised (measured: 6x off the MFMA floor with one resident block);
//   * both operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, no
//     VGPRs), into a 3-stage ring, so TWO K-steps of loads are always in flight behind the one being multiplied;
//   * zero fill (conv padding, M / N / K tails) costs nothing: lanes that must read zero use an out-of-range
//     buffer offset, for which the LDS-DMA writes zeros (probe: bench/probes/lds_dma_oob.hip);
//   * the XOR swizzle that keeps ds_read_b128 conflict-free is applied on the per-lane SOURCE address (the LDS
//     destination of an LDS-DMA is lane-linear) and on the fragment read;
//   * one raw s_barrier per K-step with a COUNTED s_waitcnt vmcnt(6) (= the next stage's six loads stay in
//     flight across the barrier); the loads are issued from inline asm so hipcc neither counts nor drains them.
#include "common.h"
#include "host_util.h"
#include "conv_params.h"
#include "conv_epilogue.h"
#include <cstdlib>

namespace {

constexpr int RBM = 256;      // pixels per tile
constexpr int RNS = 3;        // ring stages

// One LDS-DMA: 64 lanes x 16 bytes -> 1 KiB at `lds_byte` (wave-uniform), each lane from rsrc + voff (bytes).
__device__ __forceinline__ void lds_dma16(const __amdgpu_buffer_rsrc_t rsrc, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}
// (The non-temporal hint, aux bit 1, was tried for activation tiles that only one channel tile reads: 25-40 % SLOWER
// on the 1x1 layers at 56x56 — the tile's lines come from the Infinity Cache, where the producer just left them.)

// EPI bits: 1 = bias / ReLU epilogue, 2 = BN partial sums, 4 = addend, 8 = BN(+ReLU) of the producer applied to the
// input fragments (compile-time, so unused paths cost nothing)
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// BK = 32 stages are sized for TWO blocks per CU (4 waves per SIMD = 128 VGPRs); every variant fits on its own.
// (Round 2 had a variant that applied the PRODUCER's BatchNorm + ReLU to the input fragments after the LDS read, so the
//  normalised activation was never written: 0.92 x at 56x56, 1.05-1.2 x elsewhere - the transform competes with the MFMA
//  issue slots - and the filter gradient still needs the normalised tensor.  Measured, rejected, removed: docs/STATUS.md.)
// UNI: every K-step lies inside ONE filter tap for all lanes (IC % BK == 0; forward, or dgrad at stride 1).  The tap
// cursor then lives in scalar registers, each pixel row carries a precomputed byte offset and a bit mask of its in-bounds
// taps, and a stage's address work shrinks from ~40 vector instructions (six quarter-rate integer multiplies among
// them, all in front of the MFMA block) to an add, a bit test and a select per LDS-DMA.
template <int BN, int BK, int MODE, int EPI, bool UNI>
__global__ __launch_bounds__(512) void conv_ring_kernel(const ConvParams p) {
  constexpr int RP = BK * 2;                // LDS row pitch in bytes
  constexpr int CH = BK / 8;                // 16-byte chunks per row
  constexpr int RW = 1024 / RP;             // rows per LDS-DMA wave-instruction
  constexpr int XBYTES = RBM * RP, WBYTES = BN * RP, STAGE = XBYTES + WBYTES;
  constexpr int XJ = RBM / (8 * RW);        // X LDS-DMA instructions per thread per stage
  constexpr int WI = BN / RW;               // W wave-instructions per stage (spread over the 8 waves)
  constexpr int WJ = (WI + 7) / 8;
  constexpr bool WHALF = (WI < 8);          // only waves 0 .. WI-1 stage W
  constexpr int WM = 64, WN = BN / 2;       // per-wave sub-tile: 64 pixels x WN channels
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr unsigned OOB = 0xFFFFFFF0u;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 3, wn = wid >> 2;

  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;
  const int m0 = tile_m * RBM, n0 = tile_n * BN;
  const bool tracing = p.trace != nullptr && tid == 0;
  if (tracing) p.trace[blockIdx.x * 4 + 0] = wall_clock64();

  // buffer descriptors (wave-uniform: built from kernel arguments only)
  const __amdgpu_buffer_rsrc_t xsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wmat, 0, (int)p.w_bytes, 0x00020000);

  // ---- staging geometry: wave w, instruction j writes LDS rows (j*8 + w)*8 .. +7, lane -> (row lane>>3, physical
  //      chunk lane&7); the logical chunk is the same for every row of a thread.
  const int lrow = lane / CH;
  // BK=32 (64-byte rows): ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X
  // guide, LDS table), i.e. rows 0-3 and 12-15 of one chunk column together with rows 4-11 of the NEXT one; the four
  // rows that share a bank set (r, r+4, r+8, r+12) then need phys = chunk ^ {0,0,3,3}[row>>2] to land on four
  // different 16-byte slots (PMC: SQ_LDS_BANK_CONFLICT 50 % -> 0 of LDS-active cycles).
  const int lchunk = (BK == 64) ? ((lane & 7) ^ ((4 * wid + (lane >> 4)) & 7)) : ((lane & 3) ^ (3 * ((lane >> 5) & 1)));
  const bool wwave = !WHALF || wid < WI;

  int nb[XJ], bh[XJ], bw[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int m = m0 + (j * 8 + wid) * RW + lrow;
    if (m < p.M) {
      const int n = (int)fastdiv((unsigned)m, p.div_ohw);
      const int rem = m - n * (int)p.div_ohw.d;
      const int oh = (int)fastdiv((unsigned)rem, p.div_ow);
      const int ow = rem - oh * p.OW;
      nb[j] = n * p.IH * p.IW;
      if (MODE == 0) { bh[j] = oh * p.stride - p.pad; bw[j] = ow * p.stride - p.pad; }
      else           { bh[j] = oh * p.o_step + p.o_ph + p.pad; bw[j] = ow * p.o_step + p.o_pw + p.pad; }
    } else {
      nb[j] = 0; bh[j] = -(1 << 28); bw[j] = -(1 << 28);
    }
  }
  unsigned wrow_off[WJ];
  bool wrow_ok[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int n = n0 + (j * 8 + wid) * RW + lrow;
    wrow_ok[j] = n < p.OC;
    wrow_off[j] = (unsigned)n * (unsigned)p.KredFull;
  }
  // UNI: per-row byte offset of tap (0, 0) at this lane's channel chunk, and the bit mask of taps that fall inside the image
  unsigned xoff[XJ], vmask[XJ];
  if constexpr (UNI) {
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      xoff[j] = ((unsigned)(nb[j] + bh[j] * p.IW + bw[j]) * (unsigned)p.IC + (unsigned)(lchunk * 8)) * 2u;
      unsigned mk = 0u;
      if (bh[j] > -(1 << 27)) {
        for (int tr = 0, bit = 0; tr < p.nR; ++tr)
          for (int ts = 0; ts < p.nS; ++ts, ++bit) {
            const int r = p.r0 + tr * p.t_step, sx = p.s0 + ts * p.t_step;
            const int ih = MODE == 0 ? bh[j] + r : bh[j] - r, iw = MODE == 0 ? bw[j] + sx : bw[j] - sx;
            if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) mk |= 1u << bit;
          }
      }
      vmask[j] = mk;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) wrow_off[j] = (wrow_off[j] + (unsigned)(lchunk * 8)) * 2u;
  }
  int u_tr = 0, u_ts = 0, u_bit = 0, u_kc = 0;          // UNI: scalar tap cursor
  // reduction cursor of this thread's chunk: (tap row, tap col, channel) inside the tap subset
  int kr, ks_, kc;
  {
    const int kk = lchunk * 8;
    const int tap = kk / p.IC;
    kc = kk - tap * p.IC;
    kr = tap / p.nS;
    ks_ = tap - kr * p.nS;
  }

  auto issue = [&](int stage) {
    if constexpr (UNI) {
      const int r = p.r0 + u_tr * p.t_step, sx = p.s0 + u_ts * p.t_step;
      const int tap_el = (r * p.IW + sx) * p.IC;
      const unsigned xdelta = (unsigned)((MODE == 0 ? tap_el : -tap_el) + u_kc) * 2u;
      const unsigned wdelta = (unsigned)((r * p.S + sx) * p.IC + u_kc) * 2u;
      char* const sbase = smem + stage * STAGE;
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        lds_dma16(xsrc, sbase + (j * 8 + wid) * 1024, ((vmask[j] >> u_bit) & 1u) ? xoff[j] + xdelta : OOB);
      if (wwave) {
#pragma unroll
        for (int j = 0; j < WJ; ++j)
          lds_dma16(wsrc, sbase + XBYTES + (j * 8 + wid) * 1024, wrow_ok[j] ? wrow_off[j] + wdelta : OOB);
      }
      u_kc += BK;
      if (u_kc >= p.IC) {
        u_kc = 0; ++u_bit;
        if (++u_ts == p.nS) { u_ts = 0; ++u_tr; }
      }
      return;
    }
    const int r = p.r0 + kr * p.t_step, sx = p.s0 + ks_ * p.t_step;
    const bool kok = kr < p.nR;
    char* const sbase = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      int ih, iw; bool ok;
      if (MODE == 0) {
        ih = bh[j] + r; iw = bw[j] + sx;
        ok = ((unsigned)ih < (unsigned)p.IH) && ((unsigned)iw < (unsigned)p.IW);
      } else {
        const int th = bh[j] - r, tw = bw[j] - sx;
        if (p.stride == 1)      { ih = th; iw = tw; }
        else if (p.stride == 2) { ih = th >> 1; iw = tw >> 1; }
        else                    { ih = th / p.stride; iw = tw / p.stride; }
        ok = (th >= 0) && (tw >= 0) && (ih < p.IH) && (iw < p.IW);
      }
      const unsigned off = ((unsigned)(nb[j] + ih * p.IW + iw) * (unsigned)p.IC + (unsigned)kc) * 2u;
      lds_dma16(xsrc, sbase + (j * 8 + wid) * 1024, (ok && kok) ? off : OOB);
    }
    const unsigned woff = (unsigned)((r * p.S + sx) * p.IC + kc);
    if (wwave) {
#pragma unroll
      for (int j = 0; j < WJ; ++j)
        lds_dma16(wsrc, sbase + XBYTES + (j * 8 + wid) * 1024,
                  (wrow_ok[j] && kok) ? (wrow_off[j] + woff) * 2u : OOB);
    }
    // advance the cursor by one K-step
    kc += BK;
    while (kc >= p.IC) {
      kc -= p.IC;
      if (++ks_ == p.nS) { ks_ = 0; ++kr; }
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ksteps = (p.Kred + BK - 1) / BK;

  issue(0);
  if (ksteps > 1) issue(1);

  if (tracing) p.trace[blockIdx.x * 4 + 1] = wall_clock64();
  int stage = 0;
  for (int t = 0; t < ksteps; ++t) {
    // tile t has landed once at most the NEXT tile's loads (XJ + WJ per thread) are still outstanding
    if (t + 1 < ksteps) {
      if constexpr (WHALF) {
        if (wwave) wait_vmcnt<XJ + WJ>(); else wait_vmcnt<XJ>();
      } else {
        wait_vmcnt<XJ + WJ>();
      }
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();           // everyone's part of tile t is in LDS; everyone is done with tile t-1
    asm volatile("" ::: "memory");
    const char* xb = smem + stage * STAGE;
    const char* wb = xb + XBYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 a[TN], b[TM];
      const int ch = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * WN + i * 16 + (lane & 15);
        const int sw = (BK == 64) ? ((row >> 1) & 7) : (3 * ((row >> 3) & 1));
        a[i] = *reinterpret_cast<const bf16x8*>(wb + row * RP + ((ch ^ sw) << 4));
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm * WM + j * 16 + (lane & 15);
        const int sw = (BK == 64) ? ((row >> 1) & 7) : (3 * ((row >> 3) & 1));
        b[j] = *reinterpret_cast<const bf16x8*>(xb + row * RP + ((ch ^ sw) << 4));
      }
      // refill the stage tile t-1 just vacated — AFTER this step's fragment reads are in flight, so the address
      // arithmetic of the refill overlaps the LDS latency instead of delaying the first MFMA behind the barrier
      if (ks == 0 && t + 2 < ksteps) issue(stage == 0 ? 2 : stage - 1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    stage = (stage == RNS - 1) ? 0 : stage + 1;
  }
  __syncthreads();     // all fragment reads done (nothing is in flight any more): the ring becomes epilogue space
  if (tracing) p.trace[blockIdx.x * 4 + 2] = wall_clock64();

  conv_epilogue_256<BN, MODE, EPI, RNS * STAGE>(p, smem, acc, tile_m, m0, n0, tid, lane, wm, wn);
  if (tracing) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // include the drain of this wave's output stores
    p.trace[blockIdx.x * 4 + 3] = wall_clock64();
  }
}

template <int BN, int BK, int MODE, int EPI, bool UNI>
int launch_ring_u(ConvParams& p, hipStream_t stream) {
  constexpr int smem = RNS * (RBM + BN) * BK * 2;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_ring_kernel<BN, BK, MODE, EPI, UNI>), smem);
  if (rc != 0) return rc;
  p.tiles_m = (p.M + RBM - 1) / RBM;
  p.tiles_n = (p.OC + BN - 1) / BN;
  const unsigned grid = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  hipLaunchKernelGGL((conv_ring_kernel<BN, BK, MODE, EPI, UNI>), dim3(grid), dim3(512), smem, stream, p);
  return (int)hipGetLastError();
}

// Uniform-tap fast path whenever a K-step never straddles two taps (and never runs past the reduction): IC % BK == 0,
// forward or stride-1 dgrad, at most 32 taps for the per-row mask.  DDL_RING_UNI=0 switches it off (A/B).
template <int BN, int BK, int MODE, int EPI>
int launch_ring(ConvParams& p, hipStream_t stream) {
  static const int uni_on = [] { const char* e = getenv("DDL_RING_UNI"); return e ? atoi(e) : 1; }();
  if (uni_on && p.IC % BK == 0 && (MODE == 0 || p.stride == 1) && p.nR * p.nS <= 32)
    return launch_ring_u<BN, BK, MODE, EPI, true>(p, stream);
  return launch_ring_u<BN, BK, MODE, EPI, false>(p, stream);
}

template <int BN, int BK>
int dispatch_ring(ConvParams& p, int mode, hipStream_t stream) {
  if (mode == 0) {
    if (p.stats != nullptr) {
      if (p.bias != nullptr || p.relu) return -6;               // statistics are taken of the raw conv output
      if (p.fin.counter != nullptr) {                            // + BatchNorm finalize by the last block of each tile column
        p.fin_done = 1;
        return launch_ring<BN, BK, 0, 18>(p, stream);
      }
      return launch_ring<BN, BK, 0, 2>(p, stream);
    }
    if (p.bias != nullptr || p.relu) return launch_ring<BN, BK, 0, 1>(p, stream);
    return launch_ring<BN, BK, 0, 0>(p, stream);
  }
  if (p.stats != nullptr)      // BN-backward sums of the consumer BatchNorm (needs bnz), after the addend
    return p.addend != nullptr ? launch_ring<BN, BK, 1, 6>(p, stream) : launch_ring<BN, BK, 1, 2>(p, stream);
  if (p.addend != nullptr) return launch_ring<BN, BK, 1, 4>(p, stream);
  return launch_ring<BN, BK, 1, 0>(p, stream);
}

// Reduction depth per stage.  Measured over ResNet-50's layers at per-GPU batch 256 and 512 (bench/conv_layers.py):
// BK=32 (two resident blocks per CU) wins whenever the grid can give every CU a second block, for short and long
// reductions alike; a grid of at most one block per CU runs faster with the deeper BK=64 stages (fewer barriers).
int ring_bk(const ConvParams& p) {
  static const int forced = [] { const char* e = getenv("DDL_RING_BK"); return e ? atoi(e) : 0; }();
  if (forced == 32 || forced == 64) return forced;
  const long long tiles = (long long)((p.M + RBM - 1) / RBM) * ((p.OC + 127) / 128);
  return tiles <= 256 ? 64 : 32;
}

}  // namespace

// 32-bit byte offsets inside both operands, and room for the out-of-range sentinel above them.
bool ddl_conv_ring_supported(const ConvParams& p) {
  if (p.mode_hint == 1 && (p.bias != nullptr || p.relu || (p.stats != nullptr && p.bnz == nullptr))) return false;
  if (p.mode_hint == 0 && p.addend != nullptr) return false;
  if (p.stats != nullptr && (p.bias != nullptr || p.relu)) return false;
  return p.in_bytes < 0xFFFF0000ull && p.w_bytes < 0xFFFF0000ull;
}

int ddl_launch_conv_ring(ConvParams& p, int mode, hipStream_t stream) {
  p.div_ohw = make_fastdiv((unsigned)(p.OH * p.OW));
  p.div_ow = make_fastdiv((unsigned)p.OW);
  p.div_fhw = make_fastdiv((unsigned)(p.FOH * p.FOW));
  p.div_fw = make_fastdiv((unsigned)p.FOW);
  const int bk = ring_bk(p);
  int rc;
  if (p.OC <= 64) rc = bk == 32 ? dispatch_ring<64, 32>(p, mode, stream) : dispatch_ring<64, 64>(p, mode, stream);
  else            rc = bk == 32 ? dispatch_ring<128, 32>(p, mode, stream) : dispatch_ring<128, 64>(p, mode, stream);
  return rc;
}
