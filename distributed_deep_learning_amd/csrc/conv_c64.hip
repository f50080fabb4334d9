// 3 x 3, stride 1, pad 1 convolution over 64 -> 64 channels on gfx950, forward and data gradient: ResNet's first stage
// (three bottlenecks at 56 x 56, i.e. 3 + 3 launches per ResNet-50 step; ResNet-18/34's 64-channel blocks).
//
// On the implicit-GEMM ring kernels this shape is the worst of the network: with only 64 output channels a 256 x 64
// tile stages 20 KiB per 128 MFMA clocks (160 B/clk per block against the ~27 B/clk a CU gets from its L2) because the
// input tile is fetched again for every one of the nine taps: 0.43 ms forward / 0.37 ms data gradient at batch 1000
// for 0.15 ms of compulsory traffic.  Here both operands stay put:
//   * the whole filter (9 taps x 64 x 64 bf16 = 72 KiB) is copied into LDS ONCE per (persistent) block, laid out so
//     that an A fragment (16 filters x 32 reduction elements) is 256 contiguous bytes per 16-lane group;
//   * a stage is TR output rows (TR * Q = 448 pixels = 28 MFMA column blocks, 7 per wave) and the TR + 2 input rows
//     under them WITH their halo, fetched once by LDS-DMA as four 16-channel sub-tiles [position][32 bytes]; every tap
//     reads them at a shifted position (one ds_read_b128 per B fragment, 512 contiguous bytes per wave-read);
//   * the patch comes in two halves of 32 channels, each requested while the other is being multiplied (the filter
//     leaves room for exactly one patch).
// One block per CU (158 KiB of LDS), one wave per SIMD: written for an in-order machine like the halo filter gradient
// (csrc/conv_wgrad.hip): loads decoded once per kernel, one request per MFMA step, fragments one step ahead, MFMAs with
// the accumulator tied in place.  Outputs leave through wave-private LDS windows as full 128-byte pixel rows; forward
// BatchNorm sums of the ROUNDED values about per-lane pivots in registers, folded to the fp64 slab once per block.
// Data gradient = the same convolution over dy with the taps mirrored: the filter copy does the mirroring.
#include "common.h"
#include "conv_params.h"
#include "host_util.h"
#include <stdlib.h>

// -DC64_ABL=bits compiles parts of the kernel out for timing (1: patch requests inside the loop, 2: MFMAs, 4: fragment
// reads after the first step's, 8: output stores and statistics); the library is built with 0.
#ifndef C64_ABL
#define C64_ABL 0
#endif

namespace {

__device__ __forceinline__ void c64_lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");   // M0 has no other user in this kernel
}
__device__ __forceinline__ u32x4 c64_make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
  r[3] = 0x00020000u;
  return r;
}
// accumulator tied in place (see halo_mfma in conv_wgrad.hip); `first` starts a stage from zero without a clear pass
__device__ __forceinline__ void c64_mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void c64_mfma_first(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

struct C64Geom {
  int TR, Wp, XPOS, NV;           // output rows per stage, padded row length, positions per patch, 1 KiB slices per sub-tile
  int nrb, nst, sps;              // row blocks per image, stages in all, stages per block
  int chunk_bytes, half_bytes;    // one 16-channel sub-tile / one 32-channel half of the patch
  FastDiv div_nrb, div_wp, div_q;
};

constexpr int C64_WBYTES = 9 * 64 * 64 * 2;      // the filter in LDS
constexpr int C64_XS = 10;                       // LDS-DMA slots per wave and half patch
constexpr int C64_MT = 7;                        // 16-pixel blocks per wave and stage
constexpr int C64_OPITCH = 144;                  // bytes per pixel row of a wave's output window (conflict-free 8-byte writes)
constexpr int C64_WIN = 16 * C64_OPITCH;

template <int MODE, bool STATS>
__global__ __launch_bounds__(256) void conv3x3_c64_kernel(const ConvParams p, const C64Geom cg) {
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int XS = C64_XS, MT = C64_MT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, col = lane & 15;

  const int st_begin = blockIdx.x * cg.sps;
  const int st_end = min(cg.nst, st_begin + cg.sps);
  const u32x4 xsrc = c64_make_rsrc(p.in, p.in_bytes);
  const unsigned smem_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  const unsigned dummy_lds = smem_addr + (unsigned)(C64_WBYTES + 2 * cg.half_bytes);

  // ---- the filter -> LDS, once.  Global: [filter 64][tap 9][channel 64].  LDS: [tap][32-channel half][g 4][filter][8
  //      channels]: the 16 lanes of a fragment group (same g, consecutive filters) read 256 contiguous bytes.
  //      Data gradient: physical tap (r, s) serves the mirrored offset, i.e. lands at logical tap (2 - r, 2 - s).
  for (int idx = tid; idx < 9 * 64 * 8; idx += 256) {
    const int j = idx & 7, t = (idx >> 3) % 9, f = idx / 72;            // 16-byte chunk j of tap t of filter f
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.wmat + (size_t)f * 576 + t * 64 + j * 8);
    const int tl = MODE == 1 ? 8 - t : t;
    *reinterpret_cast<u32x4*>(smem + ((((tl * 2 + (j >> 2)) * 4 + (j & 3)) * 64 + f) << 4)) = v;
  }

  // ---- load slots of a HALF patch (two 16-channel sub-tiles x NV slices, dealt round-robin to the waves); everything
  //      that does not change from stage to stage is decoded here (the second half is the same plus 64 bytes)
  const int NXT = 2 * cg.NV;
  unsigned xrel[XS], xlds[XS];
  int xdyn[XS];
#pragma unroll
  for (int k = 0; k < XS; ++k) {
    const int w = wid + 4 * k;
    const int c2 = w >= cg.NV ? 1 : 0;
    const int v = w - c2 * cg.NV;
    const unsigned pos = (unsigned)(v * 32 + (lane >> 1));
    const unsigned i = fastdiv(pos, cg.div_wp);
    const int iw = (int)(pos - i * cg.div_wp.d) - 1;
    const int ihr = (int)i - 1;                                     // input row relative to the stage's first output row
    const bool ok = w < NXT && (int)pos < cg.XPOS && (unsigned)iw < (unsigned)p.IW;
    xrel[k] = (unsigned)((ihr * p.IW + iw) * 64 + c2 * 16 + (lane & 1) * 8) * 2u;
    xdyn[k] = ok ? ihr : 0x40000000;
    xlds[k] = w < NXT ? (unsigned)(c2 * cg.chunk_bytes + v * 1024) : 0xFFFFFFFFu;
  }
  unsigned l_base = 0, l_dst = 0;
  int l_p0 = 0, l_lim = 0;
  auto begin_half = [&](int st, int h, bool live) {
    const int n = (int)fastdiv((unsigned)st, cg.div_nrb);
    const int p0 = (st - n * cg.nrb) * cg.TR;
    l_base = (unsigned)((n * p.IH + p0) * p.IW * 64 + h * 32) * 2u;
    l_dst = smem_addr + (unsigned)(C64_WBYTES + h * cg.half_bytes);
    l_p0 = p0;
    l_lim = live ? p.IH : 0;                                        // past the block's last stage every slot zero-fills
  };
  auto issue_slot = [&](int k) {
    const bool ok = (unsigned)(l_p0 + xdyn[k]) < (unsigned)l_lim;
    const unsigned dst = __builtin_amdgcn_readfirstlane(xlds[k]);
    c64_lds_dma16(xsrc, dst == 0xFFFFFFFFu ? dummy_lds : l_dst + dst, ok ? l_base + xrel[k] : OOB);
  };

  // ---- fragment-read geometry: this wave's MT pixel blocks are pixels wid*112 .. +111 of the stage's 448
  int posb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const unsigned m = (unsigned)((wid * MT + i) * 16 + col);
    const unsigned pr = fastdiv(m, cg.div_q);
    posb[i] = (int)((pr * cg.Wp + (m - pr * cg.div_q.d)) * 32u) + (g & 1) * 16 + (g >> 1) * cg.chunk_bytes;
  }
  const int wp32 = cg.Wp * 32;
  const char* const wbase = smem + g * 1024 + col * 16;

  f32x4 acc[MT][4];
  float s1[8], s2[8], kf[8];
  if (STATS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; kf[q] = 0.f; }
  }
  char* const win = smem + C64_WBYTES + 2 * cg.half_bytes + 1024 + wid * C64_WIN;
  int npix = 0;
  bool have_pivot = false;

  // one half of a stage: nine MFMA steps (taps) over 32 channels; fragments one step ahead; the requests of the NEXT
  // half (other buffer) go out with the first two steps
  auto half = [&](int h, bool first) {
    const char* const ph = smem + C64_WBYTES + h * cg.half_bytes;
    bf16x8 a0[4], a1[4], b0[MT], b1[MT];
    auto load = [&](int t, bf16x8 (&a)[4], bf16x8 (&b)[MT]) {
      const char* const wa = wbase + (t * 2 + h) * 4096;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) a[nb] = *reinterpret_cast<const bf16x8*>(wa + nb * 256);
      const int toff = (t / 3) * wp32 + (t % 3) * 32;
#pragma unroll
      for (int i = 0; i < MT; ++i) b[i] = *reinterpret_cast<const bf16x8*>(ph + posb[i] + toff);
    };
    load(0, a0, b0);
    if (C64_ABL & 4) load(1, a1, b1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      bf16x8 (&ac)[4] = (t & 1) ? a1 : a0;
      bf16x8 (&bc)[MT] = (t & 1) ? b1 : b0;
      if (t < 2 && !(C64_ABL & 1)) {     // all ten requests behind the first two steps: the rest of the half is their time to land
#pragma unroll
        for (int k = 0; k < XS / 2; ++k) issue_slot(t * (XS / 2) + k);
      }
      if (t + 1 < 9 && !(C64_ABL & 4)) load(t + 1, (t & 1) ? a0 : a1, (t & 1) ? b0 : b1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if ((C64_ABL & 2) && !(first && t == 0)) { asm volatile("" :: "v"(ac[nb]), "v"(bc[i])); continue; }
          if (first && t == 0) c64_mfma_first(acc[i][nb], ac[nb], bc[i]);
          else                 c64_mfma(acc[i][nb], ac[nb], bc[i]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  __syncthreads();                                              // the filter is in LDS
  begin_half(min(st_begin, cg.nst - 1), 0, st_begin < st_end);
#pragma unroll
  for (int k = 0; k < XS; ++k) issue_slot(k);
  int stores = -1;                                              // output stores issued behind the last request (-1: unknown)
  for (int st = st_begin; st < st_end; ++st) {
    const int n = (int)fastdiv((unsigned)st, cg.div_nrb);
    const int p0 = (st - n * cg.nrb) * cg.TR;
    // ---- first half: its requests went out during the previous stage's second half, BEFORE that stage's output stores;
    //      vector memory operations retire in issue order, so the stores (2 per pixel block) may stay in flight
    if (stores == 2 * MT) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(2 * C64_MT == 14, "the counted wait above");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    begin_half(st, 1, true);
    half(0, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // second half (requested during the first)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    begin_half(min(st + 1, cg.nst - 1), 0, st + 1 < st_end);
    half(1, false);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs have written the accumulators
    // ---- outputs: each 16-pixel block goes through the wave's private LDS window ([pixel][64 channels], no block
    //      barrier) and leaves as two full-line stores (8 pixels x 128 bytes each).  Storing straight from the
    //      accumulator layout (a lane holds 4 channels of one pixel: four 8-byte stores per block) was measured at
    //      2.3x the cost per byte (0.09 ms of a 0.28 ms layer, bench/c64_abl.sh).
    //      Leaving this phase to the NEXT stage's first half (outputs rounded into 56 registers, one block per MFMA
    //      step, stores and statistics dealt out between the MFMAs) was built and measured: forward 0.284 -> 0.288 ms, data
    //      gradient 0.258 -> 0.245 ms - the layer sits at 0.9 GB of traffic (0.17 ms), not at its issue slots; not kept.
    stores = 0;
    const int trv_px = min(cg.TR, p.OH - p0) * p.OW;            // pixels of the stage inside the image
    const size_t mrow0 = (size_t)(n * p.OH + p0) * p.OW;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mb = (wid * MT + i) * 16;                        // first pixel of the block
      if (mb < trv_px && !(C64_ABL & 8)) {                       // wave-uniform
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[i][nb][0], acc[i][nb][1]);
          pk[1] = pack_bf16x2(acc[i][nb][2], acc[i][nb][3]);
          *reinterpret_cast<u32x2*>(win + col * C64_OPITCH + (nb * 16 + g * 4) * 2) = pk;
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int px = pass * 8 + (lane >> 3);
          const u32x4 v = *reinterpret_cast<const u32x4*>(win + px * C64_OPITCH + (lane & 7) * 16);
          if (mb + px < trv_px) {                                 // per lane: a block may straddle the last valid row
            *reinterpret_cast<u32x4*>(p.out + (mrow0 + (size_t)(mb + px)) * 64 + (lane & 7) * 8) = v;
            if (STATS) {
              float f[8];
              unpack8(v, f);
              if (!have_pivot) {
#pragma unroll
                for (int q = 0; q < 8; ++q) kf[q] = f[q];
                have_pivot = true;
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) { const float d = f[q] - kf[q]; s1[q] += d; s2[q] += d * d; }
              ++npix;
            }
          }
        }
        stores += 2;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the zero fills requested during the last stage

  if (STATS) {
    // per-lane shifted fp32 sums -> plain fp64 sums (own pivot, own count); lanes with the same 8 channels meet in LDS
    // (over the filter); one slab add per channel and block
    __syncthreads();
    double* const red = reinterpret_cast<double*>(smem);        // [256 threads][2][8] fp64 = 32 KiB
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      double sx, sxx;
      unshift_sums(s1[q], s2[q], kf[q], (double)npix, sx, sxx);
      red[(tid * 2 + 0) * 8 + q] = sx;
      red[(tid * 2 + 1) * 8 + q] = sxx;
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, ch = tid & 63;
      double s = 0.0;
      for (int t = ch >> 3; t < 256; t += 8) s += red[(t * 2 + which) * 8 + (ch & 7)];   // threads with lane % 8 == ch / 8
      const size_t sbase = (size_t)(blockIdx.x % DDL_BN_SLOTS) * 2;
      slab_add(p.stats, (sbase + which) * (size_t)p.OC + ch, s);
    }
  }
}

bool c64_geom(const ConvParams& p, C64Geom& cg) {
  if (p.OW <= 0 || 448 % p.OW != 0) return false;
  cg.TR = 448 / p.OW;
  cg.Wp = p.OW + 2;
  cg.XPOS = (cg.TR + 2) * cg.Wp;
  cg.NV = (cg.XPOS + 31) / 32;
  if ((2 * cg.NV + 3) / 4 > C64_XS) return false;
  cg.chunk_bytes = cg.NV * 1024;
  cg.half_bytes = 2 * cg.chunk_bytes;
  if (C64_WBYTES + 2 * cg.half_bytes + 1024 + 4 * C64_WIN > 160 * 1024) return false;
  cg.nrb = (p.OH + cg.TR - 1) / cg.TR;
  cg.nst = p.N * cg.nrb;
  cg.div_nrb = make_fastdiv((unsigned)cg.nrb);
  cg.div_wp = make_fastdiv((unsigned)cg.Wp);
  cg.div_q = make_fastdiv((unsigned)p.OW);
  return true;
}

template <int MODE, bool STATS>
int c64_launch(ConvParams& p, C64Geom& cg, int blocks, int smem, hipStream_t stream) {
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv3x3_c64_kernel<MODE, STATS>), 160 * 1024);
  if (rc != 0) return rc;
  hipLaunchKernelGGL((conv3x3_c64_kernel<MODE, STATS>), dim3((unsigned)blocks), dim3(256), smem, stream, p, cg);
  return (int)hipGetLastError();
}

}  // namespace

static int g_c64_min_stages = 256 * 4;
static long long g_c64_launches = 0;
// Test hook: launches of this kernel so far (a parity test that silently ran the ring kernels proves nothing).
extern "C" long long ddl_conv_c64_launches() { return g_c64_launches; }
// Test hook: the smallest number of stages (all blocks together) for which the kernel is chosen.
extern "C" void ddl_conv_c64_min_stages(int n) { g_c64_min_stages = n; }

// mode 0: forward (BatchNorm sums optional); mode 1: stride-1 data gradient without epilogue extras
bool ddl_conv_c64_supported(const ConvParams& p, int mode) {
  static const int on = [] { const char* e = getenv("DDL_CONV_C64"); return e ? atoi(e) : 1; }();
  if (!on) return false;
  if (p.R != 3 || p.S != 3 || p.IC != 64 || p.OC != 64 || p.stride != 1 || p.pad != 1) return false;
  if (p.OH != p.IH || p.OW != p.IW || p.o_step != 1 || p.nR != 3 || p.nS != 3) return false;
  if (p.bias != nullptr || p.addend != nullptr || p.relu || p.out_ld != p.OC || p.fin.counter != nullptr) return false;
  if (p.bnz != nullptr || p.bnz_mask != nullptr || p.trace != nullptr) return false;
  if (mode == 1 && p.stats != nullptr) return false;
  if (p.in_bytes >= 0xFFFF0000ull) return false;
  C64Geom cg;
  if (!c64_geom(p, cg)) return false;
  return cg.nst >= g_c64_min_stages;             // a block needs a run of stages to pay for its 72 KiB filter copy
}

int ddl_launch_conv_c64(ConvParams& p, int mode, hipStream_t stream) {
  C64Geom cg;
  if (!c64_geom(p, cg)) return -1;
  int blocks = 256;
  cg.sps = (cg.nst + blocks - 1) / blocks;
  blocks = (cg.nst + cg.sps - 1) / cg.sps;
  const int smem = C64_WBYTES + 2 * cg.half_bytes + 1024 + 4 * C64_WIN;
  ++g_c64_launches;
  if (mode == 1) return c64_launch<1, false>(p, cg, blocks, smem, stream);
  if (p.stats != nullptr) return c64_launch<0, true>(p, cg, blocks, smem, stream);
  return c64_launch<0, false>(p, cg, blocks, smem, stream);
}
