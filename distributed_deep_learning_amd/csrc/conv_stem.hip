// Forward convolution of the space-to-depth ResNet stem on gfx950: 4 x 4 taps x 16 channels -> 64 filters, stride 1,
// no padding (the 7x7/2 stem after nn/modules.py's regrouping), + BatchNorm partial sums.
//
// As an implicit GEMM through the ring kernels this layer stages 512 bytes of im2col per 128-byte output pixel and sits
// at the L2 -> LDS ingest limit (1.5 ms at batch 1000 against ~0.4 ms of compulsory traffic, docs/STATUS.md).  Here the
// INPUT PATCH is what lives in LDS: a stage is TR = 4 output rows of one image and the 7 input rows under them - one
// contiguous 25 KiB span of the NHWC input (all 115 columns are needed), fetched by LDS-DMA exactly as it lies in memory.
// The 32 reduction elements of an MFMA step are two horizontally adjacent taps x 16 channels = 64 CONTIGUOUS bytes of
// the patch, so a B fragment (32 k x 16 pixels) is one ds_read_b128 per lane at  pixel * 32 + (lane >> 4) * 16 + tap
// offset: no im2col, no transpose, and 7x fewer staged bytes.  The whole 32 KiB filter sits in registers as the A
// fragments (8 steps x 4 blocks of 16 filters), loaded once per block.
// Block = 4 waves, two blocks per CU (<= 256 registers, 61 KiB of LDS); a wave takes 16-pixel tiles of the stage, runs
// the 32 MFMAs of a tile, and stores it through a wave-private LDS window (no block barrier): bf16 rounding, full-line
// stores, and the BatchNorm sums of the ROUNDED values about a per-lane pivot, kept in registers across the block's
// whole (persistent) run and folded to the fp64 slab once at its end.
#include "common.h"
#include "conv_params.h"
#include "host_util.h"
#include <stdlib.h>

// -DSTEM_ABL=bits compiles parts of the kernel out for timing (1: output stores, 2: statistics, 4: all MFMAs but the
// first step's, 8: patch requests after the first); the library is built with 0.
#ifndef STEM_ABL
#define STEM_ABL 0
#endif

namespace {

__device__ __forceinline__ void stem_lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");   // M0 has no other user in this kernel
}

__device__ __forceinline__ u32x4 stem_make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
  r[3] = 0x00020000u;
  return r;
}

struct StemGeom {
  int nrb, nst, sps;          // row blocks per image, stages in all, stages per block
  int ni;                     // 1 KiB LDS-DMA slices per patch
  int patch_bytes;            // ni * 1024
  FastDiv div_nrb, div_mtrow;
};

constexpr int STEM_TR = 4;
constexpr int STEM_OPITCH = 144;                 // bytes per pixel row of a wave's output window (128 + 16: conflict-free 8-byte writes)
constexpr int STEM_WIN = 16 * STEM_OPITCH;

template <bool STATS>
__global__ __launch_bounds__(256, 2) void conv_stem_fwd_kernel(const ConvParams p, const StemGeom sg) {
  constexpr int TR = STEM_TR;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, col = lane & 15;

  const int st_begin = blockIdx.x * sg.sps;
  const int st_end = min(sg.nst, st_begin + sg.sps);
  const u32x4 xsrc = stem_make_rsrc(p.in, p.in_bytes);
  const unsigned smem_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  char* const win = smem + 2 * sg.patch_bytes + wid * STEM_WIN;

  // the filter, as A fragments: step kk covers taps (r = kk >> 1, s = 2 * (kk & 1) + {0, 1}) x 16 channels = elements
  // kk * 32 .. + 31 of a filter's 256; lane (row = filter, g) holds 8 of them
  bf16x8 wf[8][4];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
      wf[kk][nb] = *reinterpret_cast<const bf16x8*>(p.wmat + (size_t)(nb * 16 + col) * 256 + kk * 32 + g * 8);

  const int row32 = p.IW * 32;                    // bytes per input row of the patch
  const int mt_row = p.OW >> 4;                   // 16-pixel tiles per output row
  const int n_mt = TR * mt_row;

  auto issue_patch = [&](int st, int bufi) {
    const int n = (int)fastdiv((unsigned)st, sg.div_nrb);
    const int p0 = (st - n * sg.nrb) * TR;
    const unsigned base = (unsigned)((n * p.IH + p0) * p.IW) * 32u;
    const unsigned lim = (unsigned)(min(TR + 3, p.IH - p0) * row32);      // bytes of this image under the stage
    for (int i = wid; i < sg.ni; i += 4) {
      const unsigned rel = (unsigned)(i * 1024 + lane * 16);
      stem_lds_dma16(xsrc, __builtin_amdgcn_readfirstlane(smem_addr + (unsigned)(bufi * sg.patch_bytes + i * 1024)),
                     rel < lim ? base + rel : OOB);
    }
  };

  float s1[8], s2[8], kf[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; kf[q] = 0.f; }
  int npix = 0;                                    // pixels this lane has summed
  bool have_pivot = false;

  // A tile in two halves, software-pipelined inside the wave: `front` = fragment reads, 32 MFMAs, bf16 rounding into the
  // wave's window; `back` = window read-back, the two full-line stores and the statistics.  The back half of tile t-1 is
  // ISSUED BETWEEN the MFMAs of tile t (it only needs the window, which tile t rewrites after its own MFMAs), so its
  // ~70 VALU instructions and its stores run in the matrix pipe's shadow instead of after it.
  auto tile_addr = [&](const char* xb, int pr, int qt) { return xb + (pr * p.IW + qt * 16 + col) * 32 + g * 16; };
  auto read_b = [&](const char* xa, bf16x8 (&bfr)[8]) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      bfr[kk] = *reinterpret_cast<const bf16x8*>(xa + (kk >> 1) * row32 + (kk & 1) * 64);
  };
  auto mfmas = [&](const bf16x8 (&bfr)[8], f32x4 (&acc)[4]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < ((STEM_ABL & 4) ? 1 : 8); ++kk)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][nb], bfr[kk], acc[nb], 0, 0, 0);
    if (STEM_ABL & 4) {
#pragma unroll
      for (int kk = 1; kk < 8; ++kk) asm volatile("" :: "v"(bfr[kk]));
    }
  };
  auto to_window = [&](const f32x4 (&acc)[4]) {   // D: row = filter nb*16 + g*4 + e, column = pixel `col` -> [pixel][64 filters] bf16
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      u32x2 pk;
      pk[0] = pack_bf16x2(acc[nb][0], acc[nb][1]);
      pk[1] = pack_bf16x2(acc[nb][2], acc[nb][3]);
      *reinterpret_cast<u32x2*>(win + col * STEM_OPITCH + (nb * 16 + g * 4) * 2) = pk;
    }
  };
  auto window_rows = [&](u32x4 (&v)[2]) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
      v[pass] = *reinterpret_cast<const u32x4*>(win + (pass * 8 + (lane >> 3)) * STEM_OPITCH + (lane & 7) * 16);
  };
  auto back = [&](const u32x4 (&v)[2], size_t m0) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (!(STEM_ABL & 1))
        *reinterpret_cast<u32x4*>(p.out + (m0 + pass * 8 + (lane >> 3)) * 64 + (lane & 7) * 8) = v[pass];
      if (STATS && !(STEM_ABL & 2)) {
        float f[8];
        unpack8(v[pass], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float d = f[q] - kf[q]; s1[q] += d; s2[q] += d * d; }
        ++npix;
      }
    }
  };

  if (st_begin < st_end) issue_patch(st_begin, 0);
  int buf = 0;
  for (int st = st_begin; st < st_end; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's slices of the patch (and its output stores) are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st + 1 < st_end && !(STEM_ABL & 8)) issue_patch(st + 1, buf ^ 1);
    const int n = (int)fastdiv((unsigned)st, sg.div_nrb);
    const int p0 = (st - n * sg.nrb) * TR;
    const int trv = min(TR, p.OH - p0);
    const char* const xb = smem + buf * sg.patch_bytes;
    const size_t mrow0 = (size_t)(n * p.OH + p0) * p.OW;
    int mt = wid;
    int pr = (int)fastdiv((unsigned)mt, sg.div_mtrow);
    if (mt < n_mt && pr < trv) {                             // wave-uniform
      // first tile of the stage: nothing to overlap with
      bf16x8 bfr[8];
      f32x4 acc[4];
      int qt = mt - pr * mt_row;
      read_b(tile_addr(xb, pr, qt), bfr);
      mfmas(bfr, acc);
      to_window(acc);
      size_t m_prev = mrow0 + (size_t)pr * p.OW + (size_t)qt * 16;
      if (STATS && !have_pivot) {                            // the lane's pivot: its first pixel (once per block)
        u32x4 v[2];
        window_rows(v);
        unpack8(v[0], kf);
        have_pivot = true;
      }
      for (mt += 4; mt < n_mt; mt += 4) {
        pr = (int)fastdiv((unsigned)mt, sg.div_mtrow);
        if (pr >= trv) break;                                // rows past the image
        qt = mt - pr * mt_row;
        u32x4 v[2];
        read_b(tile_addr(xb, pr, qt), bfr);
        window_rows(v);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);  // the 8 fragment reads and the 2 window reads first
        mfmas(bfr, acc);
        back(v, m_prev);
        // one MFMA, then up to three of the back half's VALU / store instructions, 32 times
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x042, 3, 0);
        }
        to_window(acc);
        m_prev = mrow0 + (size_t)pr * p.OW + (size_t)qt * 16;
      }
      u32x4 v[2];
      window_rows(v);
      back(v, m_prev);
    }
    buf ^= 1;
  }

  if (STATS) {
    // every lane turns its shifted fp32 sums into plain fp64 sums (its own pivot, its own count); lanes with the same
    // 8 channels meet in LDS; one slab add per channel and block
    __syncthreads();
    double* const red = reinterpret_cast<double*>(smem);          // [256 threads][2][8] fp64 = 32 KiB
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

}  // namespace

bool ddl_conv_stem_supported(const ConvParams& p) {
  static const int on = [] { const char* e = getenv("DDL_CONV_STEM"); return e ? atoi(e) : 1; }();
  if (!on) return false;
  if (p.R != 4 || p.S != 4 || p.IC != 16 || p.OC != 64 || p.stride != 1 || p.pad != 0) return false;
  if (p.OH != p.IH - 3 || p.OW != p.IW - 3 || p.OW % 16 != 0 || p.OW / 16 > 64) return false;
  if (p.bias != nullptr || p.addend != nullptr || p.relu || p.out_ld != p.OC) return false;
  if (p.in_bytes >= 0xFFFF0000ull) return false;
  const int patch = ((STEM_TR + 3) * p.IW * 32 + 1023) / 1024 * 1024;
  if (2 * patch + 4 * STEM_WIN > 80 * 1024 || 2 * patch < 256 * 2 * 8 * 8) return false;   // two blocks per CU; room for the final fold
  return true;
}

int ddl_launch_conv_stem(ConvParams& p, hipStream_t stream) {
  StemGeom sg;
  sg.nrb = (p.OH + STEM_TR - 1) / STEM_TR;
  sg.nst = p.N * sg.nrb;
  sg.ni = ((STEM_TR + 3) * p.IW * 32 + 1023) / 1024;
  sg.patch_bytes = sg.ni * 1024;
  int blocks = 512;
  if (blocks > sg.nst) blocks = sg.nst;
  sg.sps = (sg.nst + blocks - 1) / blocks;
  blocks = (sg.nst + sg.sps - 1) / sg.sps;
  sg.div_nrb = make_fastdiv((unsigned)sg.nrb);
  sg.div_mtrow = make_fastdiv((unsigned)(p.OW / 16));
  const int smem = 2 * sg.patch_bytes + 4 * STEM_WIN;
  if (p.stats != nullptr) {
    static SmemAttrOnce once;
    const int rc = once.ensure(reinterpret_cast<const void*>(&conv_stem_fwd_kernel<true>), 80 * 1024);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(conv_stem_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), smem, stream, p, sg);
  } else {
    static SmemAttrOnce once;
    const int rc = once.ensure(reinterpret_cast<const void*>(&conv_stem_fwd_kernel<false>), 80 * 1024);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(conv_stem_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), smem, stream, p, sg);
  }
  return (int)hipGetLastError();
}
