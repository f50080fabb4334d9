// Shared epilogue of the 256 x BN implicit-GEMM convolution kernels (conv_ring.hip):
// accumulators -> (bias, ReLU) -> bf16 tile in LDS -> full-line global stores (+ addend, + BN partial sums).
// 512 threads = 8 waves (4 along pixels x 2 along channels), each wave 64 pixels x BN/2 channels.
// `smem` must offer LDS_BYTES bytes that no wave reads any more (callers __syncthreads() first).
#pragma once
#include "common.h"
#include "conv_params.h"

// Last-arriver finalize of the forward statistics of channel tile [n0, n0 + BN).  Every block of the tile column has
// added its partial sums to the slab with memory-side fp64 atomics; the hand-off is the counter form of the guide's
// inter-workgroup protocol: each wave drains its atomics (vmcnt 0) -> workgroup barrier -> ONE relaxed agent-scope
// ticket add; the block that draws the last ticket runs ONE agent-scope acquire (invalidates this CU's L1), waits for
// it, barriers, and only then reads the slab.  No block ever waits on another: the grid always drains.
template <int BN, int LDS_BYTES>
__device__ __forceinline__ void conv_fused_bn_finalize(const ConvParams& p, char* smem, int n0, int tid) {
  constexpr int G = 512 / BN;                   // slot groups (4 or 8), PER slots each
  constexpr int PER = DDL_BN_SLOTS / G;
  constexpr int PART_BYTES = G * 2 * BN * 8;
  static_assert(PART_BYTES + 16 <= LDS_BYTES, "finalize scratch must fit in the epilogue space");
  const int tile_n = n0 / BN;
  unsigned* const flag = reinterpret_cast<unsigned*>(smem + PART_BYTES);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's slab atomics have been performed
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(p.fin.counter + tile_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (t == (unsigned)p.tiles_m - 1u) ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0u) return;
  double* const slab = reinterpret_cast<double*>(p.stats);
  const int c = tid % BN, g = tid / BN, ch = n0 + c;
  double a = 0.0, b = 0.0;
  if (ch < p.OC) {
    // 8 slots (16 loads) in flight per thread and round trip: more would push the kernel past the 128 VGPRs that let
    // two blocks share a CU
    constexpr int CH = PER < 8 ? PER : 8;
#pragma unroll 1
    for (int i0 = 0; i0 < PER; i0 += CH) {
      double va[CH], vb[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const size_t s = (size_t)(g + G * (i0 + i));
        va[i] = slab[(s * 2 + 0) * p.OC + ch];
        vb[i] = slab[(s * 2 + 1) * p.OC + ch];
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const size_t s = (size_t)(g + G * (i0 + i));
        a += va[i]; b += vb[i];
        slab[(s * 2 + 0) * p.OC + ch] = 0.0;      // self-cleaning, like the stand-alone finalize
        slab[(s * 2 + 1) * p.OC + ch] = 0.0;
      }
    }
  }
  double* const part = reinterpret_cast<double*>(smem);          // [G][2][BN]
  part[(g * 2 + 0) * BN + c] = a;
  part[(g * 2 + 1) * BN + c] = b;
  __syncthreads();
  if (tid == 0) p.fin.counter[tile_n] = 0u;                      // ready for the next launch
  if (tid < BN && ch < p.OC) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int q = 0; q < G; ++q) { s1 += part[(q * 2 + 0) * BN + tid]; s2 += part[(q * 2 + 1) * BN + tid]; }
    const float count = p.fin.count;
    const double mud = s1 / (double)count;
    const float mu = (float)mud;
    const float var = fmaxf((float)(s2 / (double)count - mud * mud), 0.f);
    const float is = rsqrtf(var + p.fin.eps);
    const float gm = p.fin.gamma ? p.fin.gamma[ch] : 1.f, bt = p.fin.beta ? p.fin.beta[ch] : 0.f;
    float* const vec = p.fin.vec;
    vec[ch] = mu;
    vec[(size_t)p.OC + ch] = is;
    vec[2 * (size_t)p.OC + ch] = gm * is;
    vec[3 * (size_t)p.OC + ch] = bt - mu * gm * is;
    if (p.fin.running_mean != nullptr) {
      const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
      p.fin.running_mean[ch] = (1.f - p.fin.momentum) * p.fin.running_mean[ch] + p.fin.momentum * mu;
      p.fin.running_var[ch] = (1.f - p.fin.momentum) * p.fin.running_var[ch] + p.fin.momentum * unbiased;
    }
  }
}

template <int BN, int MODE, int EPI, int LDS_BYTES>
__device__ __forceinline__ void conv_epilogue_256(const ConvParams& p, char* smem, f32x4 (&acc)[BN / 32][4], int tile_m,
                                                  int m0, int n0, int tid, int lane, int wm, int wn) {
  constexpr int RBM = 256;
  constexpr int WM = 64, WN = BN / 2;
  constexpr int TM = WM / 16, TN = WN / 16;
  // ---- epilogue: acc -> (bias, relu) -> bf16 tile in LDS -> full-line global stores (+ addend, + BN partial sums) ----
  constexpr int OPITCH = (BN + 8) * 2;
  char* const Os = smem;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int ch = wn * WN + i * 16 + (lane >> 4) * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if ((EPI & 1) && p.bias != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[q] = (n0 + ch + q < p.OC) ? p.bias[n0 + ch + q] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int px = wm * WM + j * 16 + (lane & 15);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[i][j][q];
      if (EPI & 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += bv[q];
        if (p.relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
      }
      u32x2 pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(Os + px * OPITCH + ch * 2) = pk;
    }
  }
  __syncthreads();

  constexpr int CPR = BN / 8;       // 16-byte chunks per output row
  constexpr int RPP = 512 / CPR;    // rows per pass
  const int oc = tid % CPR, orow0 = tid / CPR;
  const int ncol = n0 + oc * 8;
  float s1[8], s2[8], kf[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; kf[q] = 0.f; }
  // forward statistics are summed about a pivot — the tile's first row (always a valid pixel) — so fp32 is enough
  // inside the tile; the plain sums are rebuilt in fp64 before they go to the slab (common.h)
  if ((EPI & 2) && MODE != 1) unpack8(*reinterpret_cast<const u32x4*>(Os + oc * 16), kf);
#pragma unroll
  for (int pass = 0; pass < RBM / RPP; ++pass) {
    const int row = orow0 + pass * RPP;
    const int m = m0 + row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(Os + row * OPITCH + oc * 16);
    if (m < p.M && ncol < p.OC) {
      size_t opix = (size_t)m;
      if (MODE == 1 && p.o_step != 1) {
        const int n = (int)fastdiv((unsigned)m, p.div_ohw);
        const int rem = m - n * (int)p.div_ohw.d;
        const int oh = (int)fastdiv((unsigned)rem, p.div_ow);
        const int ow = rem - oh * p.OW;
        opix = ((size_t)n * p.FOH + (size_t)(oh * p.o_step + p.o_ph)) * p.FOW + (size_t)(ow * p.o_step + p.o_pw);
      }
      u32x4 vo = v;
      size_t apix = opix;
      bool take = (EPI & 4) && p.addend != nullptr;
      if ((EPI & 4) && p.addend_step > 1) {       // compact addend of a strided 1x1 shortcut: only every step-th pixel
        const unsigned n = fastdiv((unsigned)opix, p.div_fhw);
        const unsigned rem = (unsigned)opix - n * p.div_fhw.d;
        const unsigned h = fastdiv(rem, p.div_fw);
        const unsigned w = rem - h * p.div_fw.d;
        const unsigned hs = h / (unsigned)p.addend_step, ws = w / (unsigned)p.addend_step;
        take = take && (hs * p.addend_step == h) && (ws * p.addend_step == w);
        apix = ((size_t)n * p.AH + hs) * p.AW + ws;
      }
      if (take) {
        const u32x4 av = __builtin_nontemporal_load(                 // read exactly once: keep it out of L2
            reinterpret_cast<const u32x4*>(p.addend + apix * (size_t)p.OC + (size_t)ncol));
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
      if ((EPI & 2) && MODE == 1) {
        // Sums for the layer behind this gradient: g = (bf16-rounded) dx behind its ReLU gate; sum(g), sum(g*z).
        // BatchNorm consumer: gate = bit mask, z = BN input.  conv+bias+ReLU consumer (bn_flags): gate = (z > 0) with
        // z the ReLU output, and the GATED gradient is what gets stored.
        float g[8], z[8];
        unpack8(vo, g);
        const u32x4 zv = *reinterpret_cast<const u32x4*>(p.bnz + opix * (size_t)p.OC + (size_t)ncol);
        unpack8(zv, z);
        if (p.bnz_mask != nullptr) {
          const unsigned mk = p.bnz_mask[opix * (size_t)(p.OC >> 3) + (size_t)(ncol >> 3)];
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = ((mk >> q) & 1u) ? g[q] : 0.f;
        } else if (p.bn_flags & 1) {
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = z[q] > 0.f ? g[q] : 0.f;
        }
        if (p.bn_flags & 2) vo = pack8(g);
#pragma unroll
        for (int q = 0; q < 8; ++q) { s1[q] += g[q]; s2[q] += g[q] * z[q]; }
      }
      *reinterpret_cast<u32x4*>(p.out + opix * (size_t)p.out_ld + (size_t)ncol) = vo;
      if ((EPI & 2) && MODE != 1) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float d = f[q] - kf[q]; s1[q] += d; s2[q] += d * d; }
      }
    }
  }
  if (EPI & 2) {
    static_assert(RBM * OPITCH <= LDS_BYTES && (RPP * 2 + 1) * BN * 4 <= LDS_BYTES, "epilogue must fit in the ring");
    __syncthreads();                                                // the output tile has been read out
    float* const red = reinterpret_cast<float*>(smem);              // [RPP][2][BN] + pivots [BN], over the output tile
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      red[(orow0 * 2 + 0) * BN + oc * 8 + q] = s1[q];
      red[(orow0 * 2 + 1) * BN + oc * 8 + q] = s2[q];
    }
    if (MODE != 1 && orow0 == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) red[RPP * 2 * BN + oc * 8 + q] = kf[q];
    }
    __syncthreads();
    const size_t sbase = (size_t)(tile_m % DDL_BN_SLOTS) * 2;
    if (MODE == 1) {
      if (tid < 2 * BN) {
        const int which = tid / BN, col = tid - which * BN;
        float s = 0.f;
#pragma unroll 4
        for (int g = 0; g < RPP; ++g) s += red[(g * 2 + which) * BN + col];
        if (n0 + col < p.OC) slab_add(p.stats, (sbase + which) * (size_t)p.OC + n0 + col, (double)s);
      }
    } else if (tid < BN) {
      const int col = tid;
      float a = 0.f, b = 0.f;
#pragma unroll 4
      for (int g = 0; g < RPP; ++g) { a += red[(g * 2) * BN + col]; b += red[(g * 2 + 1) * BN + col]; }
      if (n0 + col < p.OC) {
        const int nrows = min(RBM, p.M - m0);
        double sx, sxx;
        unshift_sums(a, b, red[RPP * 2 * BN + col], (double)nrows, sx, sxx);
        slab_add(p.stats, sbase * (size_t)p.OC + n0 + col, sx);
        slab_add(p.stats, (sbase + 1) * (size_t)p.OC + n0 + col, sxx);
      }
    }
    // EPI & 16: its own instantiation, so layers that do not fuse (large tensors, docs/STATUS.md) run the lean epilogue
    if constexpr (MODE != 1 && (EPI & 16) != 0) conv_fused_bn_finalize<BN, LDS_BYTES>(p, smem, n0, tid);
  }
}
