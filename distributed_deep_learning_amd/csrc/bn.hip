// Fused BatchNorm(+residual add)(+ReLU) for NHWC bf16 activations on gfx950.  All kernels are HBM-bound:
// 16-byte (8 x bf16) accesses per lane, every thread keeps its channel chunk fixed so the per-channel
// coefficients live in registers, and reductions go wave -> LDS -> a small [nslots][2][C] fp64 slab with atomics
// (the slab spreads same-address contention; a finalize kernel folds it).  Forward statistics are accumulated in
// fp32 about a per-block pivot and converted to plain fp64 sums before they leave the block (common.h), so the
// variance survives |mean| >> std.
//
// forward  (train): stats (or the conv epilogue's fused partials) -> finalize (mean, invstd, scale, shift,
//                   running stats) -> apply  y = act(x*scale + shift [+ res])
// backward        : reduce  (sum g, sum g*xhat with g = dy * [y > 0]) -> finalize (dgamma, dbeta, A, B, Cc)
//                   -> apply dx = A*g + B*x + Cc  [dres = g]
// The ReLU mask travels as 1 bit per element (written by the forward apply), so the backward never re-reads y.
#include "common.h"
#include "host_util.h"
#include <cstdlib>

// Streaming 16-byte load: these passes touch every byte once, so the lines need not stay in L2 (measured on the
// HBM probe, bench/probes/hbm_stream.hip: read-only streams 5.3-5.5 -> 6.0 TB/s with the nt hint).
__device__ __forceinline__ u32x4 ld_stream(const bf16_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}

#define BN_SLOTS DDL_BN_SLOTS

struct RowMap {
  int cpb;    // chunk columns handled per block (<= 256)
  int rp;     // rows per block pass
  int c8;     // C / 8
  int M;      // rows
  int C;
  // Channel WINDOWS of a wider buffer (dense blocks, csrc/eltwise.hip copy_channels): the BN input x may have a row
  // pitch ldx > C, the input gradient dx a pitch ldo > C and may be ACCUMULATED into what is there (acc).  Everything
  // else (dy, y, masks, residuals) is dense [M][C].
  int ldx, ldo, acc;
  const bf16_t* acc_src;   // acc: dx = acc_src[same window position] + (this BN's input gradient); may alias dx
};

static inline RowMap make_rowmap(int M, int C) {
  RowMap r;
  r.c8 = C / 8;
  r.cpb = r.c8 <= 256 ? r.c8 : 256;
  r.rp = 256 / r.cpb;
  r.M = M;
  r.C = C;
  r.ldx = C; r.ldo = C; r.acc = 0; r.acc_src = nullptr;
  return r;
}

static inline int apply_grid_cap() {
  constexpr int v = 1024;   // 512..8192 within 0.6 %
  return v;
}

static inline dim3 rowmap_grid(const RowMap& r, int max_x = 2048) {
  const int gy = (r.c8 + r.cpb - 1) / r.cpb;
  long long gx = ((long long)r.M + r.rp * 4 - 1) / (r.rp * 4);   // >= 4 rows per thread when possible
  if (gx < 1) gx = 1;
  const int cap = max_x / gy > 0 ? max_x / gy : 1;
  if (gx > cap) gx = cap;
  return dim3((unsigned)gx, (unsigned)gy, 1);
}

// ---------------------------------------------------------------- forward: statistics
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ slab,
                                                       RowMap rm) {
  __shared__ float red[256 * 17];       // per thread: 8 shifted sums, 8 shifted sums of squares, row count
  __shared__ float piv[256 * 8];        // per chunk column: the block's pivot (its first row)
  const int tid = threadIdx.x;
  const int col = tid % rm.cpb, rg = tid / rm.cpb;
  const int chunk = blockIdx.y * rm.cpb + col;
  const bool active = (rg < rm.rp) && (chunk < rm.c8);
  float s1[8], s2[8], k[8];
  float cnt = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; k[q] = 0.f; }
  if (active) {
    // pivot: the first row this block visits (the same for every row group of a column)
    const int prow = min((int)(blockIdx.x * rm.rp), rm.M - 1);
    unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)prow * rm.ldx + (size_t)chunk * 8), k);
    for (int row = blockIdx.x * rm.rp + rg; row < rm.M; row += gridDim.x * rm.rp) {
      const u32x4 v = ld_stream(x + (size_t)row * rm.ldx + (size_t)chunk * 8);
      float f[8];
      unpack8(v, f);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = f[q] - k[q]; s1[q] += d; s2[q] += d * d; }
      cnt += 1.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[tid * 17 + q] = s1[q]; red[tid * 17 + 8 + q] = s2[q]; }
  red[tid * 17 + 16] = cnt;
  if (rg == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) piv[col * 8 + q] = k[q];
  }
  __syncthreads();
  // thread (col, q) folds the row groups and hands plain fp64 sums to the slab
  const int nout = rm.cpb * 8;
  for (int o = tid; o < nout; o += 256) {
    const int c = o / 8, q = o % 8;
    const int ch = blockIdx.y * rm.cpb + c;
    if (ch >= rm.c8) continue;
    float a = 0.f, b = 0.f, n = 0.f;
    for (int g = 0; g < rm.rp; ++g) {
      const float* r = red + (g * rm.cpb + c) * 17;
      a += r[q]; b += r[8 + q]; n += r[16];
    }
    double sx, sxx;
    unshift_sums(a, b, piv[c * 8 + q], (double)n, sx, sxx);
    const int slot = blockIdx.x % BN_SLOTS;
    slab_add(slab, ((size_t)slot * 2 + 0) * rm.C + (size_t)ch * 8 + q, sx);
    slab_add(slab, ((size_t)slot * 2 + 1) * rm.C + (size_t)ch * 8 + q, sxx);
  }
}

// Fold the [nslots][2][C] slab for 64 channels per block: wave w sums slots w, w+4, ... (coalesced 256-byte rows),
// the four partials meet in LDS.  Returns the two sums for channel c (valid in threads with threadIdx.x < 64).
// The slab is SELF-CLEANING: every entry is zeroed right after it is read, so a slab that starts at zero (one
// memset at allocation) can be reused by the next accumulate -> finalize pair without a memset launch.
__device__ __forceinline__ void slab_fold(float* __restrict__ slab_f, int nslots, int C, double& s1, double& s2) {
  __shared__ double part[4][2][64];
  double* __restrict__ slab = reinterpret_cast<double*>(slab_f);
  const int cc = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cc;
  double a = 0.0, b = 0.0;
  if (c < C) {
    // ALL loads of a thread are issued before the first is consumed: these kernels sit on the critical path between two
    // bandwidth-bound passes while the side stream saturates HBM, where a dependent read costs 2-3 us - four
    // serialised batches of loads made this 13-19 us per call (rocprof), one batch makes it one round trip.
    constexpr int PER = BN_SLOTS / 4;
    double va[PER], vb[PER];
    if (nslots == BN_SLOTS) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const size_t s = (size_t)(w + 4 * i);
        va[i] = __builtin_nontemporal_load(slab + (s * 2 + 0) * C + c);
        vb[i] = __builtin_nontemporal_load(slab + (s * 2 + 1) * C + c);
      }
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const size_t s = (size_t)(w + 4 * i);
        a += va[i]; b += vb[i];
        slab[(s * 2 + 0) * C + c] = 0.0;
        slab[(s * 2 + 1) * C + c] = 0.0;
      }
    } else {
      for (int s = w; s < nslots; s += 4) {
        double* p0 = slab + ((size_t)s * 2 + 0) * C + c;
        double* p1 = slab + ((size_t)s * 2 + 1) * C + c;
        a += *p0; b += *p1;
        *p0 = 0.0; *p1 = 0.0;
      }
    }
  }
  part[w][0][cc] = a;
  part[w][1][cc] = b;
  __syncthreads();
  s1 = part[0][0][cc] + part[1][0][cc] + part[2][0][cc] + part[3][0][cc];
  s2 = part[0][1][cc] + part[1][1][cc] + part[2][1][cc] + part[3][1][cc];
}

// slab [nslots][2][C] -> mean/invstd/scale/shift (+ running stats, PyTorch semantics: unbiased running var)
__global__ __launch_bounds__(256) void bn_finalize_kernel(float* __restrict__ slab, int nslots, int C, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
  double s1, s2;
  slab_fold(slab, nslots, C, s1, s2);
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x >= 64 || c >= C) return;
  const double mud = s1 / (double)count;
  const float mu = (float)mud;
  float var = (float)(s2 / (double)count - mud * mud);      // the subtraction that must not happen in fp32
  var = fmaxf(var, 0.f);
  const float is = rsqrtf(var + eps);
  mean[c] = mu;
  invstd[c] = is;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - mu * g * is;
  if (running_mean) {
    const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// ---------------------------------------------------------------- forward: apply
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                       bf16_t* __restrict__ y, unsigned char* __restrict__ mask,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ res_scale,
                                                       const float* __restrict__ res_shift, RowMap rm) {
  const int tid = threadIdx.x;
  const int col = tid % rm.cpb, rg = tid / rm.cpb;
  const int chunk = blockIdx.y * rm.cpb + col;
  if (rg >= rm.rp || chunk >= rm.c8) return;
  float sc[8], sh[8], rsc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { sc[q] = scale[chunk * 8 + q]; sh[q] = shift[chunk * 8 + q]; rsc[q] = 1.f; }
  // Projection shortcut: the residual is the RAW output of the shortcut conv and its BN affine is applied here
  // (res_scale / res_shift), so the normalised shortcut tensor is never written or re-read.
  if (RES && res_scale != nullptr) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { rsc[q] = res_scale[chunk * 8 + q]; sh[q] += res_shift[chunk * 8 + q]; }
  }
  // Rows are walked from the END of the tensor: the producer (conv epilogue / reduce pass) touched the tail last, so
  // that part is still in the 256 MB memory-side cache when this pass starts.
  for (int r0 = blockIdx.x * rm.rp + rg; r0 < rm.M; r0 += gridDim.x * rm.rp) {
    const int row = rm.M - 1 - r0;
    const size_t off = (size_t)row * rm.C + (size_t)chunk * 8;
    const u32x4 v = ld_stream(x + (size_t)row * rm.ldx + (size_t)chunk * 8);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = f[q] * sc[q] + sh[q];
    if (RES) {
      const u32x4 rv = ld_stream(res + off);
      float r[8];
      unpack8(rv, r);
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] += r[q] * rsc[q];
    }
    if (RELU) {
      if (mask != nullptr) {                  // 1 bit per element: the backward reads this instead of y
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) m |= (f[q] > 0.f ? 1u : 0u) << q;
        mask[(size_t)row * rm.c8 + chunk] = (unsigned char)m;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = fmaxf(f[q], 0.f);
    }
    *reinterpret_cast<u32x4*>(y + off) = pack8(f);
  }
}

// ---------------------------------------------------------------- BN + ReLU + max-pool as one unit (ResNet stem)
// Forward: y_pool = maxpool(relu(x*scale + shift)) straight from the conv output: the full-resolution activation is
// never written.  idx = argmax position inside the window (first max wins), 255 when the pooled value is <= 0 (ReLU
// dead: no gradient).  Backward: the gradient of the (virtual) activation at a pixel is gathered from the <= 4 pooled
// windows that contain it — g = sum of dp over windows whose argmax is this pixel — inside the two BN-backward passes,
// so neither the un-pooled gradient nor a ReLU mask ever exists in memory.
struct PoolMap {
  int H, W, P, Q, k, stride, pad;
  FastDiv div_hw, div_w;
};

__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  bf16_t* __restrict__ y, unsigned char* __restrict__ idx,
                                                                  bf16_t* __restrict__ zmax, unsigned char* __restrict__ pmask,
                                                                  PoolMap pm, int N, int C, long long total) {
  const int c8 = C >> 3;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % c8);
    long long r = t / c8;
    const int q = (int)(r % pm.Q); r /= pm.Q;
    const int p = (int)(r % pm.P);
    const int n = (int)(r / pm.P);
    float sc[8], sh[8], best[8], bz[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[ch * 8 + e]; sh[e] = shift[ch * 8 + e]; best[e] = -INFINITY; bi[e] = 255; bz[e] = 0.f; }
    const int h0 = p * pm.stride - pm.pad, w0 = q * pm.stride - pm.pad;
    for (int i = 0; i < pm.k; ++i) {
      const int h = h0 + i;
      if ((unsigned)h >= (unsigned)pm.H) continue;
      for (int j = 0; j < pm.k; ++j) {
        const int w = w0 + j;
        if ((unsigned)w >= (unsigned)pm.W) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * pm.H + h) * pm.W + w) * C + (size_t)ch * 8);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = f[e] * sc[e] + sh[e];
          if (a > best[e] || a != a) { best[e] = a; bi[e] = i * pm.k + j; bz[e] = f[e]; }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (!(best[e] > 0.f) && best[e] == best[e]) { best[e] = 0.f; bi[e] = 255; }     // ReLU (NaN passes through)
    const size_t o = (((size_t)n * pm.P + p) * pm.Q + q) * C + (size_t)ch * 8;
    *reinterpret_cast<u32x4*>(y + o) = pack8(best);
    u32x2 pk;
    pk[0] = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    pk[1] = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
    *reinterpret_cast<u32x2*>(idx + o) = pk;
    if (zmax != nullptr) {
      // raw conv output AT the argmax and the 1-bit "has a gradient" gate: the backward SUMS then run over the pooled
      // tensors alone (a pixel that is nobody's argmax contributes nothing to sum(g) or sum(g * xhat))
      *reinterpret_cast<u32x4*>(zmax + o) = pack8(bz);
      unsigned m = 0u;
#pragma unroll
      for (int e = 0; e < 8; ++e) m |= (bi[e] != 255 ? 1u : 0u) << e;
      pmask[o >> 3] = (unsigned char)m;
    }
  }
}

// The same unit for the 3x3 / stride 2 / pad 1 pool on an even-sized image (ResNet's stem): one thread owns TWO
// horizontally adjacent pooled outputs of one 8-channel chunk.  Their windows share a column, so 15 instead of 18 input
// chunks are loaded, all of them issued before the first compare (the generic kernel's run-time loops serialise them).
// Candidate order (rows outer, columns inner, first maximum wins, NaN propagates) is the generic kernel's.
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_321_kernel(const bf16_t* __restrict__ x,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift,
                                                                      bf16_t* __restrict__ y, unsigned char* __restrict__ idx,
                                                                      bf16_t* __restrict__ zmax, unsigned char* __restrict__ pmask,
                                                                      int N, int H, int W, int C, int P, int Q, long long total) {
  const int c8 = C >> 3, Q2 = Q >> 1;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % c8);
    long long r = t / c8;
    const int b = (int)(r % Q2); r /= Q2;
    const int p = (int)(r % P);
    const int n = (int)(r / P);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[ch * 8 + e]; sh[e] = shift[ch * 8 + e]; }
    const int h0 = 2 * p - 1, w0 = 4 * b - 1;              // rows h0 .. h0+2, columns w0 .. w0+4 (only h0 / w0 can be -1)
    u32x4 v[3][5];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int h = h0 + i;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int w = w0 + j;
        const bool ok = h >= 0 && w >= 0;
        const size_t off = (((size_t)n * H + (size_t)(ok ? h : 0)) * W + (size_t)(ok ? w : 0)) * C + (size_t)ch * 8;
        v[i][j] = *reinterpret_cast<const u32x4*>(x + off);
      }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      float best[8], bz[8];
      int bi[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 255; bz[e] = 0.f; }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (h0 + i < 0) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (w0 + 2 * o + j < 0) continue;
          float f[8];
          unpack8(v[i][2 * o + j], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float a = f[e] * sc[e] + sh[e];
            if (a > best[e] || a != a) { best[e] = a; bi[e] = i * 3 + j; bz[e] = f[e]; }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (!(best[e] > 0.f) && best[e] == best[e]) { best[e] = 0.f; bi[e] = 255; }     // ReLU (NaN passes through)
      const size_t ob = (((size_t)n * P + p) * Q + (size_t)(2 * b + o)) * C + (size_t)ch * 8;
      *reinterpret_cast<u32x4*>(y + ob) = pack8(best);
      u32x2 pk;
      pk[0] = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      pk[1] = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
      *reinterpret_cast<u32x2*>(idx + ob) = pk;
      if (zmax != nullptr) {
        *reinterpret_cast<u32x4*>(zmax + ob) = pack8(bz);
        unsigned m = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) m |= (bi[e] != 255 ? 1u : 0u) << e;
        pmask[ob >> 3] = (unsigned char)m;
      }
    }
  }
}

// gradient of the virtual relu(bn(x)) activation at (row = (n,h,w), 8-channel chunk) from the pooled gradient
__device__ __forceinline__ void pooled_grad8(const bf16_t* __restrict__ dp, const unsigned char* __restrict__ idx,
                                             const PoolMap& pm, int row, int chunk, int C, float g[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = 0.f;
  const int n = (int)fastdiv((unsigned)row, pm.div_hw);
  const int rem = row - n * (int)pm.div_hw.d;
  const int h = (int)fastdiv((unsigned)rem, pm.div_w);
  const int w = rem - h * pm.W;
  const int hp = h + pm.pad, wp = w + pm.pad;
  int p_lo = (hp - pm.k + pm.stride) / pm.stride; if (hp - pm.k + 1 <= 0) p_lo = 0;
  int q_lo = (wp - pm.k + pm.stride) / pm.stride; if (wp - pm.k + 1 <= 0) q_lo = 0;
  const int p_hi = min(hp / pm.stride, pm.P - 1), q_hi = min(wp / pm.stride, pm.Q - 1);
  for (int p = p_lo; p <= p_hi; ++p) {
    const int i = hp - p * pm.stride;
    for (int q = q_lo; q <= q_hi; ++q) {
      const int j = wp - q * pm.stride;
      const unsigned me = (unsigned)(i * pm.k + j);
      const size_t o = (((size_t)n * pm.P + p) * pm.Q + q) * C + (size_t)chunk * 8;
      const u32x2 pk = *reinterpret_cast<const u32x2*>(idx + o);
      const u32x4 gv = *reinterpret_cast<const u32x4*>(dp + o);
      float gf[8];
      unpack8(gv, gf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned b = ((e < 4 ? pk[0] : pk[1]) >> ((e & 3) * 8)) & 0xffu;
        if (b == me) g[e] += gf[e];
      }
    }
  }
}

// ---------------------------------------------------------------- backward: reductions
// Fused finalize of the backward sums (C <= 512, one column group): the LAST block to finish folds the slab and writes
// dgamma / dbeta and the three coefficient vectors of the apply pass - no finalize launch between the two passes.
// Same last-arriver hand-off as conv_epilogue.h (drain atomics -> barrier -> relaxed agent ticket; the last arriver runs
// one agent-scope acquire, waits, barriers, reads).  `nslots` slots are used by this launch (fewer than BN_SLOTS, so
// that the fold is ONE batch of at most 16 loads per thread: bwd_fin_slots); counter == null: plain reduction into BN_SLOTS slots.
struct BwdFin {
  unsigned* counter;
  int nslots;
  float count;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  int accumulate;
  float* coefA;
  float* coefB;
  float* coefC;
};

__device__ __forceinline__ void bn_bwd_fused_finalize(float* __restrict__ slab_f, float* red, const BwdFin& f, int C,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd) {
  const int tid = threadIdx.x;
  unsigned* const flag = reinterpret_cast<unsigned*>(red + 256 * 16 - 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's slab atomics have been performed
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (t == gridDim.x - 1u) ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0u) return;
  double* const slab = reinterpret_cast<double*>(slab_f);
  double* const part = reinterpret_cast<double*>(red);           // [G][2][C] (C <= 256) - 16 KB of `red` at most
  int G = C < 256 ? 256 / C : 1;
  if (G > f.nslots) G = f.nslots;
  for (int c0 = 0; c0 < C; c0 += 256) {                          // one trip for C <= 256, two for C = 512 (G == 1)
    const int item = c0 + tid;
    const int c = C < 256 ? tid % C : item, g = C < 256 ? tid / C : 0;
    double a = 0.0, b = 0.0;
    if (c < C && g < G) {
      double va[8], vb[8];                                       // nslots / G <= 8 (see bwd_fin_slots)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sl = g + G * i;
        const bool ok = sl < f.nslots;
        va[i] = ok ? slab[((size_t)sl * 2 + 0) * C + c] : 0.0;
        vb[i] = ok ? slab[((size_t)sl * 2 + 1) * C + c] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sl = g + G * i;
        if (sl < f.nslots) {
          a += va[i]; b += vb[i];
          slab[((size_t)sl * 2 + 0) * C + c] = 0.0;
          slab[((size_t)sl * 2 + 1) * C + c] = 0.0;
        }
      }
    }
    if (G > 1) {
      if (c < C && g < G) { part[(g * 2 + 0) * C + c] = a; part[(g * 2 + 1) * C + c] = b; }
      __syncthreads();
      if (tid < C) {
        a = 0.0; b = 0.0;
        for (int q = 0; q < G; ++q) { a += part[(q * 2 + 0) * C + tid]; b += part[(q * 2 + 1) * C + tid]; }
      }
    }
    if (c < C && g == 0) {
      const float sg = (float)a, sgx = (float)b;
      if (f.accumulate) { f.dgamma[c] += sgx; f.dbeta[c] += sg; }      // straight into the arena
      else              { f.dgamma[c] = sgx;  f.dbeta[c] = sg; }
      const float gm = f.gamma ? f.gamma[c] : 1.f;
      const float sc = gm * invstd[c];
      const float mg = sg / f.count, mgx = sgx / f.count;
      f.coefA[c] = sc;
      f.coefB[c] = -sc * mgx * invstd[c];
      f.coefC[c] = -sc * mg + sc * mgx * invstd[c] * mean[c];
    }
  }
  if (tid == 0) *f.counter = 0u;                                 // ready for the next launch
}

// POOL: dy / mask are the POOLED gradient and the argmax bytes of bn_relu_maxpool_fwd_kernel (see pooled_grad8)
// FUSE: the last block finalizes (bn_bwd_fused_finalize); a separate instantiation so that the plain reduction keeps its
// register budget (8 waves per SIMD - with the fold inlined it dropped to 7 and the large layers ran 16 % slower).
template <bool RELU, bool POOL = false, bool FUSE = false, bool WIN = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                            const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* __restrict__ slab,
                                                            RowMap rm, BwdFin fin, PoolMap pm = PoolMap()) {
  // 17-float pitch per thread: with 16 every lane of a wave stored to bank 0 or 16 (PMC round 2: 60 % of this kernel's LDS
  // cycles were bank conflicts - harmless next to its HBM time, since it happens once per block, but free to avoid)
  __shared__ float red[256 * 17];
  const int tid = threadIdx.x;
  const int col = tid % rm.cpb, rg = tid / rm.cpb;
  const int chunk = blockIdx.y * rm.cpb + col;
  const bool active = (rg < rm.rp) && (chunk < rm.c8);
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
  if (active) {
    float mu[8], is[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { mu[q] = mean[chunk * 8 + q]; is[q] = invstd[chunk * 8 + q]; }
    for (int row = blockIdx.x * rm.rp + rg; row < rm.M; row += gridDim.x * rm.rp) {
      const size_t off = (size_t)row * rm.C + (size_t)chunk * 8;
      const u32x4 xv = ld_stream(x + (WIN ? (size_t)row * rm.ldx + (size_t)chunk * 8 : off));
      float g[8], xf[8];
      unpack8(xv, xf);
      if (POOL) {
        pooled_grad8(dy, mask, pm, row, chunk, rm.C, g);
      } else {
        const u32x4 gv = ld_stream(dy + off);
        unpack8(gv, g);
        if (RELU) {
          const unsigned m = mask[(size_t)row * rm.c8 + chunk];
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = ((m >> q) & 1u) ? g[q] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) { s1[q] += g[q]; s2[q] += g[q] * (xf[q] - mu[q]) * is[q]; }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[tid * 17 + q] = s1[q]; red[tid * 17 + 8 + q] = s2[q]; }
  __syncthreads();
  const int nout = rm.cpb * 16;
  for (int o = tid; o < nout; o += 256) {
    const int c = o / 16, e = o % 16;
    const int ch = blockIdx.y * rm.cpb + c;
    if (ch >= rm.c8) continue;
    float s = 0.f;
    for (int g = 0; g < rm.rp; ++g) s += red[(g * rm.cpb + c) * 17 + e];
    const int which = e >> 3, q = e & 7;
    const int slot = FUSE ? (int)(blockIdx.x % (unsigned)fin.nslots) : (int)(blockIdx.x % BN_SLOTS);
    slab_add(slab, ((size_t)slot * 2 + which) * rm.C + (size_t)ch * 8 + q, (double)s);
  }
  if constexpr (FUSE) bn_bwd_fused_finalize(slab, red, fin, rm.C, mean, invstd);
}

// slab -> dgamma, dbeta and the three per-channel coefficients of dx = A*g + B*x + Cc
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(float* __restrict__ slab, int nslots, int C, float count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int accumulate, float* __restrict__ coefA,
                                       float* __restrict__ coefB, float* __restrict__ coefC, int raw = 0) {
  double sgd, sgxd;
  slab_fold(slab, nslots, C, sgd, sgxd);
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x >= 64 || c >= C) return;
  float sg = (float)sgd, sgx = (float)sgxd;
  // raw: the slab holds sum(g), sum(g*x) from a dgrad epilogue (conv_epilogue.h) instead of sum(g*xhat)
  if (raw) sgx = invstd[c] * (float)(sgxd - (double)mean[c] * sgd);
  if (accumulate) { dgamma[c] += sgx; dbeta[c] += sg; }      // straight into the arena gradient
  else            { dgamma[c] = sgx;  dbeta[c] = sg; }
  const float g = gamma ? gamma[c] : 1.f;
  const float sc = g * invstd[c];
  const float mg = sg / count, mgx = sgx / count;
  coefA[c] = sc;
  coefB[c] = -sc * mgx * invstd[c];
  coefC[c] = -sc * mg + sc * mgx * invstd[c] * mean[c];
}

// WIN: x and dx are channel windows of wider buffers (pitches rm.ldx / rm.ldo) and dx may be accumulated (rm.acc_src);
// its own instantiation so the dense kernel keeps one offset for all four streams (and 8 waves per SIMD).
template <bool RELU, bool DRES, bool POOL = false, bool WIN = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                           const bf16_t* __restrict__ x, bf16_t* __restrict__ dx,
                                                           bf16_t* __restrict__ dres, const float* __restrict__ coefA,
                                                           const float* __restrict__ coefB,
                                                           const float* __restrict__ coefC, RowMap rm,
                                                           PoolMap pm = PoolMap()) {
  const int tid = threadIdx.x;
  const int col = tid % rm.cpb, rg = tid / rm.cpb;
  const int chunk = blockIdx.y * rm.cpb + col;
  if (rg >= rm.rp || chunk >= rm.c8) return;
  float a[8], b[8], c[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = coefA[chunk * 8 + q]; b[q] = coefB[chunk * 8 + q]; c[q] = coefC[chunk * 8 + q]; }
  for (int r0 = blockIdx.x * rm.rp + rg; r0 < rm.M; r0 += gridDim.x * rm.rp) {
    const int row = rm.M - 1 - r0;          // tail first: what the reduce pass read last is still cached
    const size_t off = (size_t)row * rm.C + (size_t)chunk * 8;
    const u32x4 xv = ld_stream(x + (WIN ? (size_t)row * rm.ldx + (size_t)chunk * 8 : off));
    float g[8], xf[8];
    unpack8(xv, xf);
    if (POOL) {
      pooled_grad8(dy, mask, pm, row, chunk, rm.C, g);
    } else {
      const u32x4 gv = ld_stream(dy + off);
      unpack8(gv, g);
      if (RELU) {
        const unsigned m = mask[(size_t)row * rm.c8 + chunk];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = ((m >> q) & 1u) ? g[q] : 0.f;
      }
    }
    if (DRES) *reinterpret_cast<u32x4*>(dres + off) = pack8(g);
    float o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = a[q] * g[q] + b[q] * xf[q] + c[q];
    if constexpr (WIN) {
      const size_t doff = (size_t)row * rm.ldo + (size_t)chunk * 8;
      if (rm.acc) {                          // gradient of a shared feature map: add to what later layers left there
        float old[8];
        unpack8(*reinterpret_cast<const u32x4*>(rm.acc_src + doff), old);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += old[q];
      }
      *reinterpret_cast<u32x4*>(dx + doff) = pack8(o);
    } else {
      *reinterpret_cast<u32x4*>(dx + off) = pack8(o);
    }
  }
}

// Apply pass of the fused BN + ReLU + max-pool unit for the 3x3 / stride 2 / pad 1 pool on an even-sized image (ResNet's
// stem): one thread owns a 2x2 QUAD of full-resolution pixels (one 8-channel chunk).  The four pooled windows that can
// have picked any of them - (a, b), (a, b+1), (a+1, b), (a+1, b+1) for the quad at (2a, 2b) - are loaded ONCE (gradient +
// argmax bytes) instead of 1, 2, 2 and 4 times by four independent pixels: 1.0 instead of 2.25 window loads per pixel.
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_quad_kernel(const bf16_t* __restrict__ dp,
                                                                    const unsigned char* __restrict__ idx,
                                                                    const bf16_t* __restrict__ x, bf16_t* __restrict__ dx,
                                                                    const float* __restrict__ coefA,
                                                                    const float* __restrict__ coefB,
                                                                    const float* __restrict__ coefC, int N, int H, int W,
                                                                    int C, int P, int Q, long long total) {
  const int c8 = C >> 3, H2 = H >> 1, W2 = W >> 1;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int chunk = (int)(t % c8);
    long long r = t / c8;
    const int b = (int)(r % W2); r /= W2;
    const int a = (int)(r % H2);
    const int n = (int)(r / H2);
    float ca[8], cb[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = coefA[chunk * 8 + e]; cb[e] = coefB[chunk * 8 + e]; cc[e] = coefC[chunk * 8 + e]; }
    float g[2][2][8];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw)
#pragma unroll
        for (int e = 0; e < 8; ++e) g[dh][dw][e] = 0.f;
#pragma unroll
    for (int wp = 0; wp < 2; ++wp) {
      const int p = a + wp;
      if (p >= P) continue;
#pragma unroll
      for (int wq = 0; wq < 2; ++wq) {
        const int q = b + wq;
        if (q >= Q) continue;
        const size_t o = (((size_t)n * P + p) * Q + q) * C + (size_t)chunk * 8;
        const u32x2 pk = *reinterpret_cast<const u32x2*>(idx + o);
        float gf[8];
        unpack8(*reinterpret_cast<const u32x4*>(dp + o), gf);
        // pixel (2a + dh, 2b + dw) sits at window position i = dh + 1 - 2 wp, j = dw + 1 - 2 wq (inside the 3x3 window iff 0..2)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          const int i = dh + 1 - 2 * wp;
          if (i < 0) continue;
#pragma unroll
          for (int dw = 0; dw < 2; ++dw) {
            const int j = dw + 1 - 2 * wq;
            if (j < 0) continue;
            const unsigned me = (unsigned)(i * 3 + j);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const unsigned bsel = ((e < 4 ? pk[0] : pk[1]) >> ((e & 3) * 8)) & 0xffu;
              if (bsel == me) g[dh][dw][e] += gf[e];
            }
          }
        }
      }
    }
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const size_t off = (((size_t)n * H + (size_t)(2 * a + dh)) * W + (size_t)(2 * b + dw)) * C + (size_t)chunk * 8;
        float xf[8], ov[8];
        unpack8(ld_stream(x + off), xf);
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = ca[e] * g[dh][dw][e] + cb[e] * xf[e] + cc[e];
        *reinterpret_cast<u32x4*>(dx + off) = pack8(ov);
      }
  }
}

// ---------------------------------------------------------------- backward of TWO BatchNorms behind one gradient
// A projection block ends in  out = relu(BN_a(z_a) + BN_b(z_b))  (last main-path BN + shortcut BN): both backward
// passes consume the same dy and the same ReLU mask.  These variants read them ONCE per pass for both normalisations.
template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                             const bf16_t* __restrict__ xa, const bf16_t* __restrict__ xb,
                                                             const float* __restrict__ mean_a, const float* __restrict__ invstd_a,
                                                             const float* __restrict__ mean_b, const float* __restrict__ invstd_b,
                                                             float* __restrict__ slab_a, float* __restrict__ slab_b, RowMap rm) {
  __shared__ float red[256 * 24];
  const int tid = threadIdx.x;
  const int col = tid % rm.cpb, rg = tid / rm.cpb;
  const int chunk = blockIdx.y * rm.cpb + col;
  const bool active = (rg < rm.rp) && (chunk < rm.c8);
  float s1[8], sa[8], sb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { s1[q] = 0.f; sa[q] = 0.f; sb[q] = 0.f; }
  if (active) {
    float mua[8], isa[8], mub[8], isb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mua[q] = mean_a[chunk * 8 + q]; isa[q] = invstd_a[chunk * 8 + q];
      mub[q] = mean_b[chunk * 8 + q]; isb[q] = invstd_b[chunk * 8 + q];
    }
    for (int row = blockIdx.x * rm.rp + rg; row < rm.M; row += gridDim.x * rm.rp) {
      const size_t off = (size_t)row * rm.C + (size_t)chunk * 8;
      const u32x4 av = ld_stream(xa + off);
      const u32x4 bv = ld_stream(xb + off);
      const u32x4 gv = ld_stream(dy + off);
      float g[8], fa[8], fb[8];
      unpack8(av, fa); unpack8(bv, fb); unpack8(gv, g);
      if (RELU) {
        const unsigned m = mask[(size_t)row * rm.c8 + chunk];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = ((m >> q) & 1u) ? g[q] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        s1[q] += g[q];
        sa[q] += g[q] * (fa[q] - mua[q]) * isa[q];
        sb[q] += g[q] * (fb[q] - mub[q]) * isb[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) { red[tid * 24 + q] = s1[q]; red[tid * 24 + 8 + q] = sa[q]; red[tid * 24 + 16 + q] = sb[q]; }
  __syncthreads();
  const int nout = rm.cpb * 24;
  for (int o = tid; o < nout; o += 256) {
    const int c = o / 24, e = o % 24;
    const int ch = blockIdx.y * rm.cpb + c;
    if (ch >= rm.c8) continue;
    float s = 0.f;
    for (int g = 0; g < rm.rp; ++g) s += red[(g * rm.cpb + c) * 24 + e];
    const int which = e >> 3, q = e & 7;
    const int slot = blockIdx.x % BN_SLOTS;
    const size_t cidx = (size_t)ch * 8 + q;
    if (which == 0) {                                   // sum(g) belongs to both
      slab_add(slab_a, ((size_t)slot * 2) * rm.C + cidx, (double)s);
      slab_add(slab_b, ((size_t)slot * 2) * rm.C + cidx, (double)s);
    } else if (which == 1) {
      slab_add(slab_a, ((size_t)slot * 2 + 1) * rm.C + cidx, (double)s);
    } else {
      slab_add(slab_b, ((size_t)slot * 2 + 1) * rm.C + cidx, (double)s);
    }
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const bf16_t* __restrict__ dy, const unsigned char* __restrict__ mask,
                                                            const bf16_t* __restrict__ xa, const bf16_t* __restrict__ xb,
                                                            bf16_t* __restrict__ dxa, bf16_t* __restrict__ dxb,
                                                            const float* __restrict__ coef_a, const float* __restrict__ coef_b,
                                                            RowMap rm) {
  const int tid = threadIdx.x;
  const int col = tid % rm.cpb, rg = tid / rm.cpb;
  const int chunk = blockIdx.y * rm.cpb + col;
  if (rg >= rm.rp || chunk >= rm.c8) return;
  float aa[8], ab[8], ac[8], ba[8], bb[8], bc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = chunk * 8 + q;
    aa[q] = coef_a[c]; ab[q] = coef_a[rm.C + c]; ac[q] = coef_a[2 * rm.C + c];
    ba[q] = coef_b[c]; bb[q] = coef_b[rm.C + c]; bc[q] = coef_b[2 * rm.C + c];
  }
  for (int r0 = blockIdx.x * rm.rp + rg; r0 < rm.M; r0 += gridDim.x * rm.rp) {
    const int row = rm.M - 1 - r0;
    const size_t off = (size_t)row * rm.C + (size_t)chunk * 8;
    const u32x4 av = ld_stream(xa + off);
    const u32x4 bv = ld_stream(xb + off);
    const u32x4 gv = ld_stream(dy + off);
    float g[8], fa[8], fb[8];
    unpack8(av, fa); unpack8(bv, fb); unpack8(gv, g);
    if (RELU) {
      const unsigned m = mask[(size_t)row * rm.c8 + chunk];
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = ((m >> q) & 1u) ? g[q] : 0.f;
    }
    float oa[8], ob[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      oa[q] = aa[q] * g[q] + ab[q] * fa[q] + ac[q];
      ob[q] = ba[q] * g[q] + bb[q] * fb[q] + bc[q];
    }
    *reinterpret_cast<u32x4*>(dxa + off) = pack8(oa);
    *reinterpret_cast<u32x4*>(dxb + off) = pack8(ob);
  }
}

// Row 0 of a reduction slab -> out[c] (+)= sum over slots; the slab is left zero (both rows).
__global__ __launch_bounds__(256) void slab_fold_kernel(float* __restrict__ slab, int nslots, int C, float* __restrict__ out,
                                                        int accumulate) {
  double s1, s2;
  slab_fold(slab, nslots, C, s1, s2);
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x >= 64 || c >= C) return;
  if (accumulate) out[c] += (float)s1; else out[c] = (float)s1;
}

// ================================================================ C ABI
extern "C" int ddl_slab_fold(float* slab, int C, float* out, int accumulate, void* stream) {
  if (C <= 0 || slab == nullptr || out == nullptr) return -1;
  hipLaunchKernelGGL(slab_fold_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, slab, BN_SLOTS, C, out,
                     accumulate);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- statistics of a growing feature map (dense blocks)
// Every channel of a dense block's buffer is produced once (block input, then one conv window per layer) and read by
// every later layer's BatchNorm.  Its batch statistics do not depend on the reader, so they are taken ONCE - in the
// epilogue of the producing conv - folded into a per-block totals table tot[2][Ctot] (fp64, NOT self-cleaning), and each
// reader finalizes its own (gamma, beta, running stats) from the prefix of that table: no statistics pass per layer.
__global__ __launch_bounds__(256) void slab_fold_totals_kernel(float* __restrict__ slab, int C, double* __restrict__ tot,
                                                               int Ctot, int c0) {
  double s1, s2;
  slab_fold(slab, BN_SLOTS, C, s1, s2);
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && c < C) { tot[c0 + c] = s1; tot[(size_t)Ctot + c0 + c] = s2; }
}

__global__ __launch_bounds__(256) void bn_finalize_totals_kernel(const double* __restrict__ tot, int Ctot, int C, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double mud = tot[c] / (double)count;
  const float mu = (float)mud;
  const float var = fmaxf((float)(tot[(size_t)Ctot + c] / (double)count - mud * mud), 0.f);
  const float is = rsqrtf(var + eps);
  mean[c] = mu;
  invstd[c] = is;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - mu * g * is;
  if (running_mean) {
    const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

extern "C" int ddl_slab_fold_totals(float* slab, int C, void* tot, int Ctot, int c0, void* stream) {
  if (C <= 0 || c0 < 0 || c0 + C > Ctot || slab == nullptr || tot == nullptr) return -1;
  hipLaunchKernelGGL(slab_fold_totals_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, slab, C,
                     (double*)tot, Ctot, c0);
  return (int)hipGetLastError();
}

extern "C" int ddl_bn_finalize_totals(const void* tot, int Ctot, int C, float count, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                      float* invstd, float* scale, float* shift, void* stream) {
  if (C <= 0 || C > Ctot || tot == nullptr) return -1;
  hipLaunchKernelGGL(bn_finalize_totals_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const double*)tot, Ctot, C, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                     scale, shift);
  return (int)hipGetLastError();
}

extern "C" int ddl_bn_slots() { return BN_SLOTS; }

// slab: [BN_SLOTS][2][C] fp64 workspace (passed as float*), must be ZERO on entry (finalize re-zeroes it: self-cleaning).
static inline bool pitch_ok(const void* p, int ld, int C) { return ld >= C && ld % 8 == 0 && ((uintptr_t)p & 15) == 0; }

// ldx: row pitch of x in elements (a channel window [M][C] of a wider [M][ldx] buffer; ldx == C is the dense case)
extern "C" int ddl_bn_stats_ld(const void* x, int ldx, float* slab, int M, int C, void* stream) {
  if (C % 8 != 0 || M <= 0 || !pitch_ok(x, ldx, C)) return -1;
  hipStream_t st = (hipStream_t)stream;
  RowMap rm = make_rowmap(M, C);
  rm.ldx = ldx;
  hipLaunchKernelGGL(bn_stats_kernel, rowmap_grid(rm), dim3(256), 0, st, (const bf16_t*)x, slab, rm);
  return (int)hipGetLastError();
}

extern "C" int ddl_bn_stats(const void* x, float* slab, int M, int C, void* stream) {
  return ddl_bn_stats_ld(x, C, slab, M, C, stream);
}

extern "C" int ddl_bn_finalize(float* slab, int nslots, int C, float count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum,
                               float eps, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, slab, nslots, C,
                     count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  return (int)hipGetLastError();
}

// mask (optional, only written when relu): [M][C/8] bytes, bit q of byte (row, chunk) = (y[row][chunk*8+q] > 0)
static int bn_apply_impl(const void* x, int ldx, const void* res, void* y, void* mask, const float* scale,
                         const float* shift, const float* res_scale, const float* res_shift, int M, int C, int relu,
                         void* stream) {
  if (C % 8 != 0 || M <= 0 || !pitch_ok(x, ldx, C)) return -1;
  if ((res_scale != nullptr) != (res_shift != nullptr) || (res_scale != nullptr && res == nullptr)) return -2;
  const float* rs = res_scale; const float* rh = res_shift;
  RowMap rm = make_rowmap(M, C);
  rm.ldx = ldx;
  const dim3 grid = rowmap_grid(rm, apply_grid_cap());
  hipStream_t st = (hipStream_t)stream;
  const bf16_t* xp = (const bf16_t*)x; const bf16_t* rp = (const bf16_t*)res; bf16_t* yp = (bf16_t*)y;
  unsigned char* mp = (unsigned char*)mask;
  if (res != nullptr) {
    if (relu) hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, dim3(256), 0, st, xp, rp, yp, mp, scale, shift, rs, rh, rm);
    else      hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, dim3(256), 0, st, xp, rp, yp, mp, scale, shift, rs, rh, rm);
  } else {
    if (relu) hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, dim3(256), 0, st, xp, rp, yp, mp, scale, shift, rs, rh, rm);
    else      hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, dim3(256), 0, st, xp, rp, yp, mp, scale, shift, rs, rh, rm);
  }
  return (int)hipGetLastError();
}

extern "C" int ddl_bn_apply(const void* x, const void* res, void* y, void* mask, const float* scale,
                            const float* shift, const float* res_scale, const float* res_shift, int M, int C, int relu,
                            void* stream) {
  return bn_apply_impl(x, C, res, y, mask, scale, shift, res_scale, res_shift, M, C, relu, stream);
}

// y[M][C] (dense) = act(x * scale + shift) for a channel window x of a [M][ldx] buffer
extern "C" int ddl_bn_apply_ld(const void* x, int ldx, void* y, void* mask, const float* scale, const float* shift, int M,
                               int C, int relu, void* stream) {
  return bn_apply_impl(x, ldx, nullptr, y, mask, scale, shift, nullptr, nullptr, M, C, relu, stream);
}

// Full backward.  slab must be ZERO on entry (it is left zero again).  mask: the [M][C/8] ReLU bit mask written by ddl_bn_apply (required when relu).  slab: [BN_SLOTS][2][C] workspace; coef: [3][C] workspace.  dres may be null.
// Slots for a fused backward reduce: 16 / 16 / 8 / 4 for C <= 64 / 128 / 256 / 512 - with 256 / C slot groups of
// threads (one for C >= 256) every thread folds at most 8 slots.  0: not fused (wide layers: one block folding 1024+
// channels would take longer than the launch it saves).
// DDL_FUSED_FINALIZE: 1 (default) fuses for tensors of at most 16 MiB - where launches are latency-bound; on a saturated
// chip the drained atomics and the serial fold cost more than the launch they replace (conv_igemm.hip, docs/STATUS.md).
static inline int bwd_fin_slots(int C, const dim3& grid, long long M) {
  static const int fused = [] { const char* e = getenv("DDL_FUSED_FINALIZE"); return e ? atoi(e) : 1; }();
  if (!fused || C > 512 || grid.y != 1) return 0;
  if (fused == 1 && M * C * 2 > (16ll << 20)) return 0;
  int n = 2048 / C;
  if (n > 16) n = 16;
  if (n < 4) n = 4;
  const int G = C < 256 ? 256 / C : 1;
  while (n > 8 * G) n >>= 1;
  return n;
}

static int bn_backward_impl(const void* dy, const void* mask, const void* x, int ldx, const float* gamma, const float* mean,
                            const float* invstd, void* dx, int ldo, const void* acc_src, void* dres, float* dgamma, float* dbeta,
                            float* slab, float* coef, int M, int C, int relu, int accumulate, int presummed, void* stream) {
  if (C % 8 != 0 || M <= 0 || !pitch_ok(x, ldx, C) || !pitch_ok(dx, ldo, C)) return -1;
  hipStream_t st = (hipStream_t)stream;
  RowMap rm = make_rowmap(M, C);
  if (acc_src != nullptr && ((uintptr_t)acc_src & 15) != 0) return -1;
  rm.ldx = ldx; rm.ldo = ldo; rm.acc = acc_src != nullptr; rm.acc_src = (const bf16_t*)acc_src;
  const dim3 grid = rowmap_grid(rm);
  const bf16_t* dyp = (const bf16_t*)dy; const unsigned char* yp = (const unsigned char*)mask; const bf16_t* xp = (const bf16_t*)x;
  if (relu && mask == nullptr) return -2;
  // presummed: the producer of dy (a dgrad epilogue) already left sum(g), sum(g*x) in the slab: no reduction pass
  float* cA = coef; float* cB = coef + C; float* cC = coef + 2 * (size_t)C;
  BwdFin fin{};
  fin.nslots = BN_SLOTS;
  const int fslots = presummed ? 0 : bwd_fin_slots(C, grid, (long long)M);
  if (fslots > 0) {       // the arrival counter lives behind the three coefficient vectors (kernels.new_slab: 4C floats)
    fin.counter = reinterpret_cast<unsigned*>(coef + 3 * (size_t)C);
    fin.nslots = fslots; fin.count = (float)M; fin.gamma = gamma; fin.dgamma = dgamma; fin.dbeta = dbeta;
    fin.accumulate = accumulate; fin.coefA = cA; fin.coefB = cB; fin.coefC = cC;
  }
  const bool win = ldx != C || ldo != C || acc_src != nullptr;
  if (win && dres != nullptr) return -3;
#define DDL_REDUCE(RELU_, FUSE_, WIN_)                                                                                   \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<RELU_, false, FUSE_, WIN_>), grid, dim3(256), 0, st, dyp, yp, xp, mean, invstd, \
                     slab, rm, fin)
  if (!presummed) {
    const int sel = (relu ? 4 : 0) | (fslots > 0 ? 2 : 0) | (win ? 1 : 0);
    switch (sel) {
      case 0: DDL_REDUCE(false, false, false); break;
      case 1: DDL_REDUCE(false, false, true); break;
      case 2: DDL_REDUCE(false, true, false); break;
      case 3: DDL_REDUCE(false, true, true); break;
      case 4: DDL_REDUCE(true, false, false); break;
      case 5: DDL_REDUCE(true, false, true); break;
      case 6: DDL_REDUCE(true, true, false); break;
      default: DDL_REDUCE(true, true, true); break;
    }
  }
#undef DDL_REDUCE
  if (fslots == 0)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st, slab, BN_SLOTS, C, (float)M, gamma,
                       mean, invstd, dgamma, dbeta, accumulate, cA, cB, cC, presummed ? 1 : 0);
  bf16_t* dxp = (bf16_t*)dx; bf16_t* drp = (bf16_t*)dres;
  const dim3 agrid = rowmap_grid(rm, apply_grid_cap());
  if (win) {
    if (relu) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, false, true>), agrid, dim3(256), 0, st, dyp, yp, xp, dxp, drp, cA, cB, cC, rm);
    else      hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false, false, true>), agrid, dim3(256), 0, st, dyp, yp, xp, dxp, drp, cA, cB, cC, rm);
  } else if (relu) {
    if (dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), agrid, dim3(256), 0, st, dyp, yp, xp, dxp, drp, cA, cB, cC, rm);
    else      hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false>), agrid, dim3(256), 0, st, dyp, yp, xp, dxp, drp, cA, cB, cC, rm);
  } else {
    if (dres) hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), agrid, dim3(256), 0, st, dyp, yp, xp, dxp, drp, cA, cB, cC, rm);
    else      hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false>), agrid, dim3(256), 0, st, dyp, yp, xp, dxp, drp, cA, cB, cC, rm);
  }
  return (int)hipGetLastError();
}

extern "C" int ddl_bn_backward(const void* dy, const void* mask, const void* x, const float* gamma, const float* mean,
                               const float* invstd, void* dx, void* dres, float* dgamma, float* dbeta, float* slab,
                               float* coef, int M, int C, int relu, int accumulate, int presummed, void* stream) {
  return bn_backward_impl(dy, mask, x, C, gamma, mean, invstd, dx, C, nullptr, dres, dgamma, dbeta, slab, coef, M, C, relu,
                          accumulate, presummed, stream);
}

// Backward for a channel window: x is a window of a [M][ldx] buffer, dx goes to a window of a [M][ldo] buffer.  With
// acc_src (same window position in a buffer of the same pitch, may BE dx) the result is acc_src + gradient: the shared
// feature map of a dense block collects one gradient per reader, the first reader taking the upstream gradient as source.
extern "C" int ddl_bn_backward_ld(const void* dy, const void* mask, const void* x, int ldx, const float* gamma,
                                  const float* mean, const float* invstd, void* dx, int ldo, const void* acc_src, float* dgamma,
                                  float* dbeta, float* slab, float* coef, int M, int C, int relu, int accumulate,
                                  void* stream) {
  return bn_backward_impl(dy, mask, x, ldx, gamma, mean, invstd, dx, ldo, acc_src, nullptr, dgamma, dbeta, slab, coef, M, C,
                          relu, accumulate, 0, stream);
}

// Backward of two BatchNorms sharing dy (+ mask): see bn_bwd_reduce2_kernel.  Every per-BN argument comes as a / b.
extern "C" int ddl_bn_backward_pair(const void* dy, const void* mask, const void* xa, const void* xb,
                                    const float* gamma_a, const float* mean_a, const float* invstd_a,
                                    const float* gamma_b, const float* mean_b, const float* invstd_b, void* dxa, void* dxb,
                                    float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, float* slab_a,
                                    float* slab_b, float* coef_a, float* coef_b, int M, int C, int relu, int accumulate,
                                    void* stream) {
  if (C % 8 != 0 || M <= 0) return -1;
  if (relu && mask == nullptr) return -2;
  if (slab_a == slab_b || coef_a == coef_b || dxa == dxb) return -3;
  hipStream_t st = (hipStream_t)stream;
  const RowMap rm = make_rowmap(M, C);
  const dim3 grid = rowmap_grid(rm);
  const bf16_t* dyp = (const bf16_t*)dy; const unsigned char* mp = (const unsigned char*)mask;
  const bf16_t* ap = (const bf16_t*)xa; const bf16_t* bp = (const bf16_t*)xb;
  if (relu) hipLaunchKernelGGL((bn_bwd_reduce2_kernel<true>), grid, dim3(256), 0, st, dyp, mp, ap, bp, mean_a, invstd_a, mean_b, invstd_b, slab_a, slab_b, rm);
  else      hipLaunchKernelGGL((bn_bwd_reduce2_kernel<false>), grid, dim3(256), 0, st, dyp, mp, ap, bp, mean_a, invstd_a, mean_b, invstd_b, slab_a, slab_b, rm);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st, slab_a, BN_SLOTS, C, (float)M, gamma_a,
                     mean_a, invstd_a, dgamma_a, dbeta_a, accumulate, coef_a, coef_a + C, coef_a + 2 * (size_t)C);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st, slab_b, BN_SLOTS, C, (float)M, gamma_b,
                     mean_b, invstd_b, dgamma_b, dbeta_b, accumulate, coef_b, coef_b + C, coef_b + 2 * (size_t)C);
  const dim3 agrid = rowmap_grid(rm, apply_grid_cap());
  if (relu) hipLaunchKernelGGL((bn_bwd_apply2_kernel<true>), agrid, dim3(256), 0, st, dyp, mp, ap, bp, (bf16_t*)dxa, (bf16_t*)dxb, coef_a, coef_b, rm);
  else      hipLaunchKernelGGL((bn_bwd_apply2_kernel<false>), agrid, dim3(256), 0, st, dyp, mp, ap, bp, (bf16_t*)dxa, (bf16_t*)dxb, coef_a, coef_b, rm);
  return (int)hipGetLastError();
}

static PoolMap make_poolmap(int H, int W, int P, int Q, int k, int stride, int pad) {
  PoolMap pm;
  pm.H = H; pm.W = W; pm.P = P; pm.Q = Q; pm.k = k; pm.stride = stride; pm.pad = pad;
  pm.div_hw = make_fastdiv((unsigned)(H * W));
  pm.div_w = make_fastdiv((unsigned)W);
  return pm;
}

// y_pool [N][P][Q][C], idx [N][P][Q][C] bytes  <-  maxpool_k/stride/pad(relu(x*scale + shift)),  x [N][H][W][C]
// zmax (bf16, shape of y) / pmask (bytes [N*P*Q][C/8]), both or neither: the raw conv output at each argmax and the bit
// mask of pooled elements that carry a gradient, for the pooled-domain reduction of ddl_bn_relu_maxpool_bwd.
extern "C" int ddl_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, void* idx, void* zmax,
                                       void* pmask, int N, int H, int W, int C, int P, int Q, int k, int stride, int pad,
                                       void* stream) {
  if ((zmax == nullptr) != (pmask == nullptr)) return -2;
  if (C % 8 != 0 || k * k > 254 || N <= 0) return -1;
  const PoolMap pm = make_poolmap(H, W, P, Q, k, stride, pad);
  const long long total = (long long)N * P * Q * (C / 8);
  static const int pair_on = [] { const char* e = getenv("DDL_POOL_FWD_PAIR"); return e ? atoi(e) : 1; }();
  if (pair_on && k == 3 && stride == 2 && pad == 1 && H % 2 == 0 && W % 4 == 0 && P == H / 2 && Q == W / 2) {
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_321_kernel, dim3(grid_for(total / 2, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, scale, shift, (bf16_t*)y, (unsigned char*)idx, (bf16_t*)zmax, (unsigned char*)pmask,
                       N, H, W, C, P, Q, total / 2);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, scale, shift, (bf16_t*)y, (unsigned char*)idx, (bf16_t*)zmax, (unsigned char*)pmask, pm,
                     N, C, total);
  return (int)hipGetLastError();
}

// BN backward of the fused unit: dp / idx are the pooled gradient and argmax bytes, x the conv output [N][H][W][C].
// zmax / pmask (from the forward, both or neither): the reduction runs over the POOLED tensors - dp, zmax and one mask byte
// per chunk instead of the full-resolution conv output plus a gather of 1-4 (dp, idx) pairs per pixel (4x fewer bytes
// through the caches; measured on ResNet-50's stem at batch 512: 0.48 -> 0.12 ms).
extern "C" int ddl_bn_relu_maxpool_bwd(const void* dp, const void* idx, const void* x, const void* zmax, const void* pmask,
                                       const float* gamma, const float* mean, const float* invstd, void* dx, float* dgamma,
                                       float* dbeta, float* slab, float* coef, int N, int H, int W, int C, int P, int Q, int k,
                                       int stride, int pad, int accumulate, void* stream) {
  const long long M = (long long)N * H * W;
  if (C % 8 != 0 || M <= 0 || M >= (1ll << 31)) return -1;
  if ((zmax == nullptr) != (pmask == nullptr)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const RowMap rm = make_rowmap((int)M, C);
  const dim3 grid = rowmap_grid(rm);
  const PoolMap pm = make_poolmap(H, W, P, Q, k, stride, pad);
  const bf16_t* dpp = (const bf16_t*)dp; const unsigned char* ip = (const unsigned char*)idx; const bf16_t* xp = (const bf16_t*)x;
  float* cA = coef; float* cB = coef + C; float* cC = coef + 2 * (size_t)C;
  BwdFin fin{};
  fin.nslots = BN_SLOTS;
  const int fslots = bwd_fin_slots(C, grid, (long long)M);
  if (fslots > 0) {
    fin.counter = reinterpret_cast<unsigned*>(coef + 3 * (size_t)C);
    fin.nslots = fslots; fin.count = (float)M; fin.gamma = gamma; fin.dgamma = dgamma; fin.dbeta = dbeta;
    fin.accumulate = accumulate; fin.coefA = cA; fin.coefB = cB; fin.coefC = cC;
  }
  if (zmax != nullptr) {
    const RowMap rp = make_rowmap((int)((long long)N * P * Q), C);
    BwdFin none{};
    none.nslots = BN_SLOTS;
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, false>), rowmap_grid(rp), dim3(256), 0, st, dpp, (const unsigned char*)pmask,
                       (const bf16_t*)zmax, mean, invstd, slab, rp, none, PoolMap());
  } else if (fslots > 0) {
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, true, true>), grid, dim3(256), 0, st, dpp, ip, xp, mean, invstd, slab, rm, fin, pm);
  } else {
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, true>), grid, dim3(256), 0, st, dpp, ip, xp, mean, invstd, slab, rm, fin, pm);
  }
  if (fslots == 0 || zmax != nullptr)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st, slab, BN_SLOTS, C, (float)M, gamma,
                       mean, invstd, dgamma, dbeta, accumulate, cA, cB, cC);
  static const int quad_on = [] { const char* e = getenv("DDL_POOL_QUAD"); return e ? atoi(e) : 1; }();
  if (quad_on && k == 3 && stride == 2 && pad == 1 && H % 2 == 0 && W % 2 == 0 && P == H / 2 && Q == W / 2) {
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(bn_pool_bwd_apply_quad_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, dpp, ip, xp,
                       (bf16_t*)dx, cA, cB, cC, N, H, W, C, P, Q, total);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, true>), grid, dim3(256), 0, st, dpp, ip, xp, (bf16_t*)dx,
                     (bf16_t*)nullptr, cA, cB, cC, rm, pm);
  return (int)hipGetLastError();
}
