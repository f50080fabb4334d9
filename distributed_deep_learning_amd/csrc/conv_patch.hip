// 3x3 / stride 1 / pad 1 convolution (forward and data-gradient) with an LDS-RESIDENT INPUT PATCH (gfx950).
//
// The ring kernel (conv_ring.hip) re-fetches the activation tile from L2 for every one of the nine filter taps, and
// its K-steps cost (bytes fetched per CU) / (~70 GB/s per CU), not MFMA time (bench/conv_trace.py).  In the flat
// [N*H*W][C] pixel matrix the nine taps of a tile of 256 consecutive output pixels read rows p + dr*W + ds, i.e. ONE
// contiguous row range [p0 - W - 1, p0 + 256 + W + 1) — also across image boundaries, where the rows that do not
// belong to the tap are masked to zero per lane.  So per 32-channel slice the kernel
//
//   * loads that patch ONCE (LDS-DMA, double-buffered, prefetched a whole slice ahead),
//   * streams only the nine 128 x 32 weight slices through a 3-slot ring (two K-steps in flight),
//   * reads the B fragments of tap (dr, ds) from the patch at a row SHIFT, zeroing fragments whose pixel falls off
//     the image (9-bit validity mask per pixel, computed once per tile).
//
// L2 -> LDS traffic per K-step drops from 24 KiB to ~10.7 KiB (2.2x).  MEASURED VERDICT (one MI355X, bench/conv_layers.py
// and bench.py): no gain — ResNet-50's 3x3 layers run within +-3 % of the ring kernel (12,516 vs 12,538 img/s end to
// end), VGG-16 is 9 % slower (6,114 vs 6,708 img/s).  The nine taps' re-reads of the activation tile were evidently
// served from L1/L2 cheaply already; what limits the ring kernel on these layers is the per-K-step barrier cadence
// (16 MFMAs per wave between barriers) and grid quantisation, which this kernel shares, and its shifted fragment reads
// add 2-way LDS conflicts.  It therefore stays OPT-IN (DDL_CONV_PATCH=1); numerics are covered by the conv tests.
// Same tile (256 pixels x 128 channels, 512 threads, 8 waves 4 x 2, two blocks per CU), MFMA
// layout, swizzle and epilogue as the ring kernel.  dgrad = the same kernel on the flipped taps with the CRSK filter.
#include "common.h"
#include "host_util.h"
#include "conv_params.h"
#include "conv_epilogue.h"
#include <cstdlib>

namespace {

constexpr int PBM = 256;        // pixels per tile
constexpr int PWR = 128;        // weight rows staged per slot (tiles of 64 channels leave the upper half zero)
constexpr int PBK = 32;         // channels per K-step (one tap of one 32-channel slice)
constexpr int PNW = 3;          // weight ring slots
constexpr unsigned POOB = 0xFFFFFFF0u;

__device__ __forceinline__ void pt_lds_dma16(const u32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc) : "memory");
}

__device__ __forceinline__ u32x4 pt_make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
  r[3] = 0x00020000u;
  return r;
}

// wait until at most `n` of this wave's loads are outstanding (n is wave-uniform, small)
__device__ __forceinline__ void pt_wait(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;      // conservative
  }
}

// 64-byte LDS rows (four 16-byte chunks): physical chunk = logical ^ 3*((row >> 3) & 1)   (see conv_ring.hip)
__device__ __forceinline__ int pt_sw(int row) { return 3 * ((row >> 3) & 1); }

template <int BN, int MODE, int EPI>
__global__ __launch_bounds__(512) void conv_patch_kernel(const ConvParams p, const int PJ) {
  constexpr int WM = 64, WN = BN / 2;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int WBYTES = PWR * 64;                   // one weight slot
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 3, wn = wid >> 2;

  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;
  const int m0 = tile_m * PBM, n0 = tile_n * BN;

  const u32x4 xsrc = pt_make_rsrc(p.in, p.in_bytes);
  const u32x4 wsrc = pt_make_rsrc(p.wmat, p.w_bytes);
  const unsigned smem_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // LDS map: [weight ring: PNW x 8 KiB][patch buffer 0][patch buffer 1], patch buffer = PJ*128 rows x 64 B
  const int PROWS = PJ * 128;
  const int PBYTES = PROWS * 64;
  const unsigned wring = smem_addr;
  const unsigned pbuf = smem_addr + PNW * WBYTES;
  const int W = p.IW;                                // input == output geometry (stride 1, pad 1, 3x3)
  const int halo = W + 1;
  const int PR = PBM + 2 * halo;                     // patch rows actually needed

  // ---- staging geometry: one wave-instruction = 16 rows x 64 B; lane -> (row lane>>2, physical chunk lane&3)
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ (3 * ((lane >> 5) & 1));
  // weights: 8 instructions per slot, one per wave: rows wid*16 .. +15 of the channel tile
  const int wrow = n0 + wid * 16 + lrow;
  const bool wrow_ok = (wid * 16 + lrow) < BN && wrow < p.OC;
  const unsigned wrow_off = (unsigned)wrow * (unsigned)p.KredFull;

  auto issue_w = [&](int t) {                        // K-step t = (slice t / 9, tap t % 9)
    const int cs = t / 9, tap = t - cs * 9;
    const unsigned off = (wrow_off + (unsigned)(tap * p.IC + cs * PBK + lchunk * 8)) * 2u;
    pt_lds_dma16(wsrc, wring + (unsigned)((t % PNW) * WBYTES + wid * 1024), wrow_ok ? off : POOB);
  };
  auto issue_patch = [&](int cs) {                   // PJ instructions per wave: patch rows (j*8 + wid)*16 .. +15
    const unsigned base = pbuf + (unsigned)((cs & 1) * PBYTES);
    for (int j = 0; j < PJ; ++j) {
      const int pr = (j * 8 + wid) * 16 + lrow;
      const long long pix = (long long)m0 - halo + pr;
      const bool ok = pr < PR && pix >= 0 && pix < (long long)p.M;
      const unsigned off = ((unsigned)pix * (unsigned)p.IC + (unsigned)(cs * PBK + lchunk * 8)) * 2u;
      pt_lds_dma16(xsrc, base + (unsigned)((j * 8 + wid) * 1024), ok ? off : POOB);
    }
  };

  // ---- per-lane tap validity of the four pixels this lane feeds into its B fragments
  unsigned vmask[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * WM + j * 16 + (lane & 15);
    unsigned mk = 0;
    if (m < p.M) {
      const int n = (int)fastdiv((unsigned)m, p.div_ohw);
      const int rem = m - n * (int)p.div_ohw.d;
      const int h = (int)fastdiv((unsigned)rem, p.div_ow);
      const int w = rem - h * p.OW;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dr = tap / 3 - 1, ds = tap % 3 - 1;
        // forward reads x[h+dr][w+ds]; dgrad reads dy[h-dr][w-ds] for filter tap (dr, ds): the mirrored position
        const int hh = MODE == 0 ? h + dr : h - dr, ww = MODE == 0 ? w + ds : w - ds;
        if ((unsigned)hh < (unsigned)p.IH && (unsigned)ww < (unsigned)p.IW) mk |= 1u << tap;
      }
    }
    vmask[j] = mk;
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nslices = p.IC / PBK;
  const int T = nslices * 9;

  // prologue: patch(0), W(0), W(1)   [patch(1) follows at step 0]
  issue_patch(0);
  issue_w(0);
  if (T > 1) issue_w(1);

  for (int t = 0; t < T; ++t) {
    const int cs = t / 9, tap = t - cs * 9;
    // Everything up to W(t) must have landed.  Issued after W(t): W(t+1) [at step t-1] and, just before it when step
    // t-1 opened a slice, patch(cs+1) (PJ loads).  The patch of THIS slice was issued nine steps ago.
    {
      int allow = (t + 1 < T) ? 1 : 0;
      if (t >= 1 && (t - 1) % 9 == 0 && (t - 1) / 9 + 1 < nslices) allow += PJ;
      pt_wait(allow);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (tap == 0 && cs + 1 < nslices) issue_patch(cs + 1);       // buffer (cs+1)&1 was last read in slice cs-1
    if (t + 2 < T) issue_w(t + 2);                               // slot (t+2)%3 == (t-1)%3 was read at step t-1

    const char* wb = smem + (t % PNW) * WBYTES;
    const char* pb = smem + PNW * WBYTES + (cs & 1) * PBYTES;
    const int dr = tap / 3 - 1, ds = tap - (tap / 3) * 3 - 1;
    const int shift = halo + (MODE == 0 ? dr * W + ds : -(dr * W + ds));
    bf16x8 a[TN], b[TM];
    const int ch = lane >> 4;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int row = wn * WN + i * 16 + (lane & 15);
      a[i] = *reinterpret_cast<const bf16x8*>(wb + row * 64 + ((ch ^ pt_sw(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int row = wm * WM + j * 16 + (lane & 15) + shift;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(pb + row * 64 + ((ch ^ pt_sw(row)) << 4));
      if (!((vmask[j] >> tap) & 1u)) v = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
      b[j] = v;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  __syncthreads();

  conv_epilogue_256<BN, MODE, EPI, PNW * WBYTES + 2 * 3 * 128 * 64>(p, smem, acc, tile_m, m0, n0, tid, lane, wm, wn);
}

template <int BN, int MODE, int EPI>
int launch_patch(ConvParams& p, hipStream_t stream) {
  const int PR = PBM + 2 * (p.IW + 1);
  const int PJ = (((PR + 15) / 16) + 7) / 8;
  const int smem = PNW * PWR * 64 + 2 * PJ * 128 * 64;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_patch_kernel<BN, MODE, EPI>),
                             PNW * PWR * 64 + 2 * 7 * 128 * 64);
  if (rc != 0) return rc;
  p.tiles_m = (p.M + PBM - 1) / PBM;
  p.tiles_n = (p.OC + BN - 1) / BN;
  const unsigned grid = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  hipLaunchKernelGGL((conv_patch_kernel<BN, MODE, EPI>), dim3(grid), dim3(512), smem, stream, p, PJ);
  return (int)hipGetLastError();
}

template <int BN>
int dispatch_patch(ConvParams& p, int mode, hipStream_t stream) {
  if (mode == 0) {
    if (p.stats != nullptr) {
      if (p.fin.counter != nullptr) {
        p.fin_done = 1;
        return launch_patch<BN, 0, 18>(p, stream);
      }
      return launch_patch<BN, 0, 2>(p, stream);
    }
    if (p.bias != nullptr || p.relu) return launch_patch<BN, 0, 1>(p, stream);
    return launch_patch<BN, 0, 0>(p, stream);
  }
  if (p.addend != nullptr) return launch_patch<BN, 1, 4>(p, stream);
  return launch_patch<BN, 1, 0>(p, stream);
}

}  // namespace

// 3x3, stride 1, pad 1, channel counts in whole 32-slices, wide enough output tile, 32-bit offsets, and a patch
// (2 buffers of PJ*8 KiB) + weight ring that still leaves room for two blocks per CU only matters for speed: the
// launcher accepts up to PJ = 7 (W <= 318).
bool ddl_conv_patch_supported(const ConvParams& p) {
  static const int enabled = [] { const char* e = getenv("DDL_CONV_PATCH"); return e ? atoi(e) : 0; }();
  if (!enabled) return false;
  if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1) return false;
  if (p.o_step != 1 || p.t_step != 1 || p.nR != 3 || p.nS != 3) return false;
  if (p.IC % PBK != 0 || p.OC < 64) return false;
  if (p.IH != p.OH || p.IW != p.OW) return false;
  const int PR = PBM + 2 * (p.IW + 1);
  const int PJ = (((PR + 15) / 16) + 7) / 8;
  if (PJ > 7) return false;
  if (p.mode_hint == 1 && (p.bias != nullptr || p.relu || p.stats != nullptr)) return false;
  if (p.mode_hint == 0 && p.addend != nullptr) return false;
  if (p.stats != nullptr && (p.bias != nullptr || p.relu)) return false;
  return p.in_bytes < 0xFFFF0000ull && p.w_bytes < 0xFFFF0000ull;
}

int ddl_launch_conv_patch(ConvParams& p, int mode, hipStream_t stream) {
  p.div_ohw = make_fastdiv((unsigned)(p.OH * p.OW));
  p.div_ow = make_fastdiv((unsigned)p.OW);
  p.div_fhw = make_fastdiv((unsigned)(p.FOH * p.FOW));
  p.div_fw = make_fastdiv((unsigned)p.FOW);
  const int rc = p.OC <= 64 ? dispatch_patch<64>(p, mode, stream) : dispatch_patch<128>(p, mode, stream);
  return rc;
}
