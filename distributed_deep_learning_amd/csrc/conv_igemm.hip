// Implicit-GEMM convolution for gfx950: forward and data-gradient share one gather-GEMM kernel.
//
//   out[m][oc] = sum_{r,s,ic} in[n, ih(m,r), iw(m,s), ic] * wmat[oc][(r*S+s)*IC + ic]
//
// All tensors are NHWC bf16, accumulation is fp32 on the MFMA pipe (v_mfma_f32_16x16x32_bf16).
//   MODE 0 (fwd)  : rows m = output pixels (n,p,q); ih = p*stride - pad + r.
//   MODE 1 (dgrad): rows m = input pixels (n,h,w); the gathered tensor is dy, ih = (h + pad - r)/stride
//                   when divisible; wmat is the [C][R][S][K] repack of the filter.  For stride > 1 the launcher
//                   splits the input pixels into stride^2 parity classes (h % stride, w % stride): inside a
//                   class only the taps r = r0 + t*stride contribute and every one of them divides exactly, so
//                   each class is a dense GEMM over its own tap subset (no MFMA work on structural zeros).
//
// Tile: BM pixels x BN out-channels x BK=64 reduction, 256 threads = 4 waves (2 x 2), each wave owns a
// (BM/2) x (BN/2) sub-tile.  The weight tile is the MFMA A operand and the pixel tile the B operand, so every
// lane ends up with 4 consecutive output channels of one pixel (8-byte packed LDS store in the epilogue).
// Staging is global -> registers -> LDS (the gather needs zero fill), double-buffered, one barrier per K-step;
// the LDS image is XOR-swizzled in 16-byte chunks so both the ds_write_b128 and the ds_read_b128 fragment
// reads are bank-conflict free.  The epilogue stages the bf16 tile through LDS for full-line stores and can
// fuse bias+ReLU and the per-channel sum / sum-of-squares partials that BatchNorm needs (no extra read pass).
#include "common.h"
#include "host_util.h"
#include "conv_params.h"
#include <stdlib.h>


template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int XI = BM / 32;
  constexpr int WI = BN / 32;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;
  char* const Ws = smem + 2 * XBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;

  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / p.tiles_n, tile_n = lid - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int chunk = tid & 7, row0 = tid >> 3;

  // ---- per-thread gather rows ----
  int nb[XI], bh[XI], bw[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int m = m0 + row0 + 32 * i;
    if (m < p.M) {
      const int ohw = p.OH * p.OW;
      const int n = m / ohw;
      const int rem = m - n * ohw;
      const int oh = rem / p.OW;
      const int ow = rem - oh * p.OW;
      nb[i] = n * p.IH * p.IW;
      if (MODE == 0) { bh[i] = oh * p.stride - p.pad; bw[i] = ow * p.stride - p.pad; }
      else           { bh[i] = oh * p.o_step + p.o_ph + p.pad; bw[i] = ow * p.o_step + p.o_pw + p.pad; }
    } else {
      nb[i] = 0; bh[i] = -(1 << 28); bw[i] = -(1 << 28);
    }
  }
  // ---- per-thread reduction cursor: kk = k0 + chunk*8 -> (tap row, tap col, c) inside the tap subset ----
  int kr, ks_, kc;
  {
    const int kk = chunk * 8;
    const int tap = kk / p.IC;
    kc = kk - tap * p.IC;
    kr = tap / p.nS;
    ks_ = tap - kr * p.nS;
  }

  u32x4 xr[XI], wr[WI];

  auto load_tile = [&]() {
    const int r = p.r0 + kr * p.t_step, sx = p.s0 + ks_ * p.t_step;   // actual filter tap of this chunk
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int ih, iw; bool ok;
      if (MODE == 0) {
        ih = bh[i] + r; iw = bw[i] + sx;
        ok = ((unsigned)ih < (unsigned)p.IH) && ((unsigned)iw < (unsigned)p.IW);
      } else {
        const int th = bh[i] - r, tw = bw[i] - sx;          // multiples of stride by construction of the class
        if (p.stride == 1)      { ih = th; iw = tw; }
        else if (p.stride == 2) { ih = th >> 1; iw = tw >> 1; }
        else                    { ih = th / p.stride; iw = tw / p.stride; }
        ok = (th >= 0) && (tw >= 0) && (ih < p.IH) && (iw < p.IW);
      }
      ok = ok && (kr < p.nR);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        const size_t off = ((size_t)(nb[i] + ih * p.IW + iw)) * (size_t)p.IC + (size_t)kc;
        v = *reinterpret_cast<const u32x4*>(p.in + off);
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int n = n0 + row0 + 32 * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (n < p.OC && kr < p.nR) {
        const int woff = (r * p.S + sx) * p.IC + kc;
        v = *reinterpret_cast<const u32x4*>(p.wmat + (size_t)n * (size_t)p.KredFull + (size_t)woff);
      }
      wr[i] = v;
    }
  };
  auto advance = [&]() {
    kc += 64;
    while (kc >= p.IC) {
      kc -= p.IC;
      if (++ks_ == p.nS) { ks_ = 0; ++kr; }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = row0 + 32 * i;
      const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
      *reinterpret_cast<u32x4*>(Xs + buf * XBYTES + off) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = row0 + 32 * i;
      const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
      *reinterpret_cast<u32x4*>(Ws + buf * WBYTES + off) = wr[i];
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ksteps = (p.Kred + 63) >> 6;

  load_tile();
  advance();
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < ksteps; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1 < ksteps);
    if (more) { load_tile(); advance(); }
    const char* xb = Xs + buf * XBYTES;
    const char* wb = Ws + buf * WBYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[TN], b[TM];
      const int ch = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * WN + i * 16 + (lane & 15);
        a[i] = *reinterpret_cast<const bf16x8*>(wb + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm * WM + j * 16 + (lane & 15);
        b[j] = *reinterpret_cast<const bf16x8*>(xb + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc -> (bias, relu) -> bf16 tile in LDS -> full-line global stores (+ BN partial sums) ----
  constexpr int OPITCH = (BN + 8) * 2;
  char* const Os = smem;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int ch = wn * WN + i * 16 + (lane >> 4) * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = (n0 + ch + q < p.OC) ? p.bias[n0 + ch + q] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int px = wm * WM + j * 16 + (lane & 15);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = acc[i][j][q] + bv[q];
        if (p.relu) v[q] = fmaxf(v[q], 0.f);
      }
      u32x2 pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(Os + px * OPITCH + ch * 2) = pk;
    }
  }
  __syncthreads();

  constexpr int CPR = BN / 8;       // 16-byte chunks per output row
  constexpr int RPP = 256 / CPR;    // rows per pass
  const int oc = tid % CPR, orow0 = tid / CPR;
  const int ncol = n0 + oc * 8;
  float s1[8], s2[8], kf[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; kf[q] = 0.f; }
  if (p.stats != nullptr) unpack8(*reinterpret_cast<const u32x4*>(Os + oc * 16), kf);     // pivot: the tile's first row
#pragma unroll
  for (int pass = 0; pass < BM / RPP; ++pass) {
    const int row = orow0 + pass * RPP;
    const int m = m0 + row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(Os + row * OPITCH + oc * 16);
    if (m < p.M && ncol < p.OC) {
      size_t opix = (size_t)m;
      if (p.o_step != 1) {                       // parity-class sub-grid -> pixel of the full output tensor
        const int ohw = p.OH * p.OW;
        const int n = m / ohw;
        const int rem = m - n * ohw;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        opix = ((size_t)n * p.FOH + (size_t)(oh * p.o_step + p.o_ph)) * p.FOW + (size_t)(ow * p.o_step + p.o_pw);
      }
      u32x4 vo = v;
      size_t apix = opix;
      bool take = p.addend != nullptr;
      if (take && p.addend_step > 1) {
        const size_t fhw = (size_t)p.FOH * p.FOW;
        const size_t n = opix / fhw, rem = opix - n * fhw;
        const int h = (int)(rem / p.FOW), w = (int)(rem - (size_t)h * p.FOW);
        take = (h % p.addend_step == 0) && (w % p.addend_step == 0);
        apix = (n * p.AH + (size_t)(h / p.addend_step)) * p.AW + (size_t)(w / p.addend_step);
      }
      if (take) {
        const u32x4 av = *reinterpret_cast<const u32x4*>(p.addend + apix * (size_t)p.OC + (size_t)ncol);
        float f[8], a[8];
        unpack8(v, f);
        unpack8(av, a);
        if (p.addend_mask != nullptr) {          // addend = (upstream gradient) x (ReLU mask of the block output)
          const unsigned mk = p.addend_mask[opix * (size_t)(p.OC >> 3) + (size_t)(ncol >> 3)];
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] = ((mk >> q) & 1u) ? a[q] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] += a[q];
        vo = pack8(f);
      }
      *reinterpret_cast<u32x4*>(p.out + opix * (size_t)p.out_ld + (size_t)ncol) = vo;
      if (p.stats != nullptr) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float d = f[q] - kf[q]; s1[q] += d; s2[q] += d * d; }
      }
    }
  }
  if (p.stats != nullptr) {
    constexpr int OS_BYTES = ((BM * OPITCH + 15) / 16) * 16;
    float* const red = reinterpret_cast<float*>(smem + OS_BYTES);   // [RPP][2][BN] + pivots [BN]
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      red[(orow0 * 2 + 0) * BN + oc * 8 + q] = s1[q];
      red[(orow0 * 2 + 1) * BN + oc * 8 + q] = s2[q];
    }
    if (orow0 == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) red[RPP * 2 * BN + oc * 8 + q] = kf[q];
    }
    __syncthreads();
    if (tid < BN) {
      const int col = tid;
      float a = 0.f, b = 0.f;
#pragma unroll 4
      for (int g = 0; g < RPP; ++g) { a += red[(g * 2) * BN + col]; b += red[(g * 2 + 1) * BN + col]; }
      if (n0 + col < p.OC) {
        const int nrows = min(BM, p.M - m0);
        double sx, sxx;
        unshift_sums(a, b, red[RPP * 2 * BN + col], (double)nrows, sx, sxx);
        const size_t sbase = (size_t)(tile_m % DDL_BN_SLOTS) * 2;
        slab_add(p.stats, sbase * (size_t)p.OC + n0 + col, sx);
        slab_add(p.stats, (sbase + 1) * (size_t)p.OC + n0 + col, sxx);
      }
    }
  }
}

template <int BM, int BN, int MODE>
static int launch_conv(const ConvParams& p, hipStream_t stream) {
  const int smem = 2 * (BM + BN) * 128;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, MODE>), smem);
  if (rc != 0) return rc;
  const unsigned grid = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, MODE>), dim3(grid), dim3(256), smem, stream, p);
  return (int)hipGetLastError();
}

// mode: 0 = forward gather, 1 = dgrad gather.  Returns a hipError_t code (0 = ok), or negative for bad arguments.
// `stats`, when not null, must hold DDL_BN_SLOTS * 2 * OC floats and be ZERO on entry (partials are accumulated
// with atomics; ddl_bn_finalize folds the slab and leaves it zero again).

// Debug: per-block phase stamps of the ring kernel (bench/conv_trace.py).  Not thread-safe; null in normal use.
static long long* g_conv_trace = nullptr;
extern "C" void ddl_conv_set_trace(long long* buf) { g_conv_trace = buf; }

template <int MODE>
static int launch_any(ConvParams& p, hipStream_t st) {
  // default: the 3-stage LDS-DMA ring kernel (conv_ring.hip); DDL_CONV_RING=0 selects the register-staged kernel
  static const int use_ring = [] { const char* e = getenv("DDL_CONV_RING"); return e ? atoi(e) : 1; }();
  p.mode_hint = MODE;
  p.trace = g_conv_trace;
  const bool bwd_sums = MODE == 1 && p.stats != nullptr;
  if (MODE == 0 && use_ring && p.fin.counter == nullptr && ddl_conv_stem_supported(p)) return ddl_launch_conv_stem(p, st);
  if (use_ring && ddl_conv_c64_supported(p, MODE)) return ddl_launch_conv_c64(p, MODE, st);
  if (use_ring && ddl_conv_big_shape(p) != 0) {
    const int n0 = ddl_conv_big_split(p);
    if (n0 >= p.N) return ddl_launch_conv_big(p, MODE, st);
    // whole tile rounds on the big-tile kernel, the remaining images on the ring kernels (conv_big.hip)
    ConvParams a = p;
    a.N = n0; a.M = n0 * p.OH * p.OW;
    a.in_bytes = 2ull * (unsigned long long)n0 * p.IH * p.IW * p.IC;
    const int rc = ddl_launch_conv_big(a, MODE, st);
    if (rc != 0) return rc;
    ConvParams b = p;
    const size_t opix = (size_t)n0 * p.OH * p.OW;
    b.N = p.N - n0; b.M = b.N * p.OH * p.OW;
    b.in = p.in + (size_t)n0 * p.IH * p.IW * p.IC;
    b.in_bytes = 2ull * (unsigned long long)b.N * p.IH * p.IW * p.IC;
    b.out = p.out + opix * (size_t)p.out_ld;
    if (p.addend != nullptr) b.addend = p.addend + opix * (size_t)p.OC;
    if (p.addend_mask != nullptr) b.addend_mask = p.addend_mask + opix * (size_t)(p.OC >> 3);
    if (ddl_conv_pring_supported(b)) return ddl_launch_conv_pring(b, MODE, st);
    if (ddl_conv_ring_supported(b)) return ddl_launch_conv_ring(b, MODE, st);
    return ddl_launch_conv_big(b, MODE, st);
  }
  if (use_ring && ddl_conv_pring_supported(p)) return ddl_launch_conv_pring(p, MODE, st);
  if (use_ring && ddl_conv_ring_supported(p)) return ddl_launch_conv_ring(p, MODE, st);
  if (bwd_sums) return -9;                                     // only the ring kernel's epilogue takes these sums
  p.tiles_m = (p.M + 127) / 128;
  static const int force64 = [] { const char* e = getenv("DDL_CONV_BN64"); return e ? atoi(e) : 0; }();
  if (p.OC <= 64 || force64) {
    p.tiles_n = (p.OC + 63) / 64;
    return launch_conv<128, 64, MODE>(p, st);
  }
  p.tiles_n = (p.OC + 127) / 128;
  return launch_conv<128, 128, MODE>(p, st);
}

extern "C" int ddl_bn_finalize(float* slab, int nslots, int C, float count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* mean,
                               float* invstd, float* scale, float* shift, void* stream);     // bn.hip

static int conv_igemm_impl(int mode, const void* in, const void* wmat, void* out, const float* bias, float* stats,
                           const void* addend, const void* addend_mask, int addend_step, const void* bnz,
                           const void* bnz_mask, int bn_flags, int N, int OH, int OW, int IH, int IW, int IC, int OC, int R, int S, int stride,
                           int pad, int relu, void* stream, int out_ld = 0,
                           const BnFin* fin = nullptr) {
  if (IC % 8 != 0 || OC % 8 != 0) return -1;
  if (out_ld != 0 && (mode != 0 || out_ld < OC || out_ld % 8 != 0 || ((uintptr_t)out & 15) != 0)) return -12;
  if (N <= 0 || OH <= 0 || OW <= 0 || IH <= 0 || IW <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0) return -2;
  const long long M = (long long)N * OH * OW;
  if (M >= (1ll << 31) || (long long)N * IH * IW >= (1ll << 31)) return -3;
  ConvParams p;
  p.in = (const bf16_t*)in; p.wmat = (const bf16_t*)wmat; p.out = (bf16_t*)out; p.bias = bias; p.stats = stats;
  p.addend = (const bf16_t*)addend;
  p.addend_mask = (const unsigned char*)addend_mask;
  if (addend_mask != nullptr && addend == nullptr) return -7;
  if (addend_step < 1 || (addend_step > 1 && addend_mask != nullptr)) return -8;
  p.addend_step = addend != nullptr ? addend_step : 1;
  p.AH = (OH + addend_step - 1) / addend_step; p.AW = (OW + addend_step - 1) / addend_step;
  if (mode == 0 && addend != nullptr && stats != nullptr) return -5;   // forward statistics: of the raw conv output
  if (mode == 0 && (bnz != nullptr || bnz_mask != nullptr)) return -10;
  if (mode == 1 && ((stats != nullptr) != (bnz != nullptr) || (bnz_mask != nullptr && bnz == nullptr))) return -10;
  if (bn_flags < 0 || bn_flags > 3 || (bn_flags != 0 && (bnz == nullptr || bnz_mask != nullptr))) return -10;
  p.bnz = (const bf16_t*)bnz; p.bnz_mask = (const unsigned char*)bnz_mask; p.bn_flags = bn_flags;
  p.out_ld = out_ld != 0 ? out_ld : OC;
  p.fin = BnFin{};
  p.fin_done = 0;
  if (fin != nullptr) {
    if (mode != 0 || stats == nullptr || fin->vec == nullptr) return -13;
    p.fin = *fin;
    p.fin.count = (float)M;
    // arrival counters live in the slab's fp32 scratch rows, behind the three coefficient vectors of the backward pass
    // Measured (docs/STATUS.md): the in-kernel finalize wins where launches are latency-bound (small tensors: -3..4 %
    // step time on the reference CNN at batch 32 and ResNet-18/CIFAR) and LOSES on a saturated chip (+1.5 % on
    // ResNet-50 at batch 512: every block drains its atomics before taking a ticket and the last block's serial fold
    // runs at loaded-memory latency).  1 (default): fuse when the output is at most 16 MiB; 2: always; 0: never.
    static const int fused = [] { const char* e = getenv("DDL_FUSED_FINALIZE"); return e ? atoi(e) : 1; }();
    const bool fuse = fused == 2 || (fused == 1 && (size_t)M * (size_t)OC * 2 <= ((size_t)16 << 20));
    p.fin.counter = fuse ? reinterpret_cast<unsigned*>(stats + (size_t)DDL_BN_SLOTS * 2 * OC * 2 + 3 * (size_t)OC) : nullptr;
  }
  p.N = N; p.OH = OH; p.OW = OW; p.IH = IH; p.IW = IW; p.IC = IC; p.OC = OC; p.R = R; p.S = S;
  p.stride = stride; p.pad = pad; p.relu = relu;
  p.KredFull = R * S * IC;
  p.in_bytes = 2ull * (unsigned long long)N * IH * IW * IC;
  p.w_bytes = 2ull * (unsigned long long)OC * p.KredFull;
  p.FOH = OH; p.FOW = OW;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0 || stride == 1) {
    p.M = (int)M; p.Kred = p.KredFull;
    p.o_step = 1; p.o_ph = 0; p.o_pw = 0;
    p.r0 = 0; p.s0 = 0; p.t_step = 1; p.nR = R; p.nS = S;
    if (mode != 0) return launch_any<1>(p, st);
    const int rc = launch_any<0>(p, st);
    if (rc != 0 || fin == nullptr || p.fin_done) return rc;
    // the kernel that ran has no fused finalize (register-staged fallback, or fusion switched off): separate launch
    return ddl_bn_finalize(stats, DDL_BN_SLOTS, OC, (float)M, fin->gamma, fin->beta, fin->running_mean, fin->running_var,
                           fin->momentum, fin->eps, fin->vec, fin->vec + OC, fin->vec + 2 * (size_t)OC,
                           fin->vec + 3 * (size_t)OC, stream);
  }
  // dgrad with stride > 1: one dense launch per parity class of the input pixels
  bool any_empty = false;
  for (int ph = 0; ph < stride && !any_empty; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      const int r0 = (ph + pad) % stride, s0 = (pw + pad) % stride;
      if (r0 >= R || s0 >= S || ph >= OH || pw >= OW) { any_empty = true; break; }
    }
  if (any_empty && (addend != nullptr || stats != nullptr)) return -4;   // zero-filled pixels would miss the epilogue
  if (any_empty) {   // classes without a contributing tap (e.g. 1x1 stride-2) are plain zeros
    hipError_t e = ddl_zero_async(out, sizeof(bf16_t) * (size_t)M * (size_t)OC, st);
    if (e != hipSuccess) return (int)e;
  }
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      const int r0 = (ph + pad) % stride, s0 = (pw + pad) % stride;
      if (r0 >= R || s0 >= S || ph >= OH || pw >= OW) continue;
      ConvParams q = p;
      q.OH = (OH - ph + stride - 1) / stride;
      q.OW = (OW - pw + stride - 1) / stride;
      q.M = N * q.OH * q.OW;
      q.o_step = stride; q.o_ph = ph; q.o_pw = pw;
      q.r0 = r0; q.s0 = s0; q.t_step = stride;
      q.nR = (R - r0 + stride - 1) / stride;
      q.nS = (S - s0 + stride - 1) / stride;
      q.Kred = q.nR * q.nS * IC;
      const int rc = launch_any<1>(q, st);
      if (rc != 0) return rc;
    }
  return 0;
}

extern "C" int ddl_conv_igemm(int mode, const void* in, const void* wmat, void* out, const float* bias, float* stats,
                              const void* addend, const void* addend_mask, int addend_step, const void* bnz,
                              const void* bnz_mask, int bn_flags, int N, int OH, int OW, int IH, int IW, int IC, int OC, int R,
                              int S, int stride, int pad, int relu, void* stream) {
  return conv_igemm_impl(mode, in, wmat, out, bias, stats, addend, addend_mask, addend_step, bnz, bnz_mask, bn_flags, N, OH,
                         OW, IH, IW, IC, OC, R, S, stride, pad, relu, stream);
}

// Forward convolution writing its [pixels][OC] result into a channel window of a wider [pixels][out_ld] buffer
// (`out` = address of the window's first channel; out_ld % 8 == 0).  The window's statistics go to `stats`.
extern "C" int ddl_conv_fwd_ld(const void* in, const void* wmat, void* out, int out_ld, float* stats, int N, int OH, int OW,
                               int IH, int IW, int IC, int OC, int R, int S, int stride, int pad, void* stream) {
  return conv_igemm_impl(0, in, wmat, out, nullptr, stats, nullptr, nullptr, 1, nullptr, nullptr, 0, N, OH, OW, IH, IW, IC, OC,
                         R, S, stride, pad, 0, stream, out_ld);
}

// Forward convolution + BatchNorm statistics + BatchNorm finalize as ONE launch where the kernel supports it: the last
// block of every channel tile turns the slab into vec[4][OC] = (mean, invstd, scale, shift) and updates the running
// statistics (conv_epilogue.h).  `slab` must be a zeroed [2*DDL_BN_SLOTS + 2][2][OC] fp32 workspace (kernels.new_slab)
// and is left zero.  out_ld: 0 (dense) or the row pitch of a wider output buffer.
extern "C" int ddl_conv_fwd_bn(const void* in, const void* wmat, void* out, int out_ld, float* slab, int N, int OH, int OW,
                               int IH, int IW, int IC, int OC, int R, int S, int stride, int pad, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* vec, void* stream) {
  BnFin fin{};
  fin.gamma = gamma; fin.beta = beta; fin.running_mean = running_mean; fin.running_var = running_var;
  fin.momentum = momentum; fin.eps = eps; fin.vec = vec;
  return conv_igemm_impl(0, in, wmat, out, nullptr, slab, nullptr, nullptr, 1, nullptr, nullptr, 0, N, OH, OW, IH, IW, IC, OC,
                         R, S, stride, pad, 0, stream, out_ld, &fin);
}
