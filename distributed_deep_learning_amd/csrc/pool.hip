// Pooling kernels for NHWC bf16 activations (gfx950).  One thread = one pixel x 8 channels (16-byte accesses).
//   max pool   : forward also records the argmax position inside the window (uint8) so the backward is an
//                atomic-free gather with PyTorch's first-max tie rule.
//   avg pool   : k x k / stride (no padding) and global average pool, forward + backward.
#include "common.h"
#include "host_util.h"

struct PoolGeom {
  int N, H, W, C, P, Q, k, stride, pad;
  int c8;
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          unsigned char* __restrict__ idx, PoolGeom g, long long total) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % g.c8);
    long long r = t / g.c8;
    const int q = (int)(r % g.Q); r /= g.Q;
    const int p = (int)(r % g.P);
    const int n = (int)(r / g.P);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    const int h0 = p * g.stride - g.pad, w0 = q * g.stride - g.pad;
    for (int i = 0; i < g.k; ++i) {
      const int h = h0 + i;
      if ((unsigned)h >= (unsigned)g.H) continue;
      for (int j = 0; j < g.k; ++j) {
        const int w = w0 + j;
        if ((unsigned)w >= (unsigned)g.W) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * g.H + h) * g.W + w) * g.C + (size_t)ch * 8);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // first max wins (strict >); NaN propagates like PyTorch
          if (f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; bi[e] = i * g.k + j; }
        }
      }
    }
    const size_t o = (((size_t)n * g.P + p) * g.Q + q) * g.C + (size_t)ch * 8;
    *reinterpret_cast<u32x4*>(y + o) = pack8(best);
    u32x2 pk;
    pk[0] = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    pk[1] = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
    *reinterpret_cast<u32x2*>(idx + o) = pk;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const unsigned char* __restrict__ idx,
                                                          bf16_t* __restrict__ dx, PoolGeom g, long long total) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % g.c8);
    long long r = t / g.c8;
    const int w = (int)(r % g.W); r /= g.W;
    const int h = (int)(r % g.H);
    const int n = (int)(r / g.H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // windows (p,q) with p*stride - pad <= h < p*stride - pad + k
    const int hp = h + g.pad, wp = w + g.pad;
    int p_lo = (hp - g.k + g.stride) / g.stride; if (hp - g.k + 1 <= 0) p_lo = 0;
    int q_lo = (wp - g.k + g.stride) / g.stride; if (wp - g.k + 1 <= 0) q_lo = 0;
    const int p_hi = min(hp / g.stride, g.P - 1), q_hi = min(wp / g.stride, g.Q - 1);
    for (int p = p_lo; p <= p_hi; ++p) {
      const int i = hp - p * g.stride;
      for (int q = q_lo; q <= q_hi; ++q) {
        const int j = wp - q * g.stride;
        const unsigned me = (unsigned)(i * g.k + j);
        const size_t o = (((size_t)n * g.P + p) * g.Q + q) * g.C + (size_t)ch * 8;
        const u32x2 pk = *reinterpret_cast<const u32x2*>(idx + o);
        const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + o);      // re-read by up to four pixels: default policy
        float gf[8];
        unpack8(gv, gf);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned b = ((e < 4 ? pk[0] : pk[1]) >> ((e & 3) * 8)) & 0xffu;
          if (b == me) acc[e] += gf[e];
        }
      }
    }
    *reinterpret_cast<u32x4*>(dx + (((size_t)n * g.H + h) * g.W + w) * g.C + (size_t)ch * 8) = pack8(acc);
  }
}

__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          PoolGeom g, long long total) {
  const float inv = 1.f / (float)(g.k * g.k);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % g.c8);
    long long r = t / g.c8;
    const int q = (int)(r % g.Q); r /= g.Q;
    const int p = (int)(r % g.P);
    const int n = (int)(r / g.P);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int i = 0; i < g.k; ++i)
      for (int j = 0; j < g.k; ++j) {
        const int h = p * g.stride + i, w = q * g.stride + j;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * g.H + h) * g.W + w) * g.C + (size_t)ch * 8);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    *reinterpret_cast<u32x4*>(y + (((size_t)n * g.P + p) * g.Q + q) * g.C + (size_t)ch * 8) = pack8(acc);
  }
}

// requires stride >= k (non-overlapping windows): each input pixel belongs to at most one window
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                                          PoolGeom g, long long total) {
  const float inv = 1.f / (float)(g.k * g.k);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % g.c8);
    long long r = t / g.c8;
    const int w = (int)(r % g.W); r /= g.W;
    const int h = (int)(r % g.H);
    const int n = (int)(r / g.H);
    const int p = h / g.stride, q = w / g.stride;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    if (p < g.P && q < g.Q && (h - p * g.stride) < g.k && (w - q * g.stride) < g.k) {
      const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + (((size_t)n * g.P + p) * g.Q + q) * g.C + (size_t)ch * 8);
      unpack8(gv, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] *= inv;
    }
    *reinterpret_cast<u32x4*>(dx + (((size_t)n * g.H + h) * g.W + w) * g.C + (size_t)ch * 8) = pack8(o);
  }
}

// global average pool: x [N][HW][C] -> y [N][C]; 256 threads split HW for one (n, 32-chunk group)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int HW,
                                                      int C, int c8) {
  __shared__ float red[256 * 8];
  const int n = blockIdx.y;
  const int cpb = c8 < 32 ? c8 : 32;            // chunk columns per block
  const int rgs = 256 / cpb;                    // row groups
  const int col = threadIdx.x % cpb, rg = threadIdx.x / cpb;
  const int chunk = blockIdx.x * cpb + col;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (rg < rgs && chunk < c8) {
    for (int i = rg; i < HW; i += rgs) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + ((size_t)n * HW + i) * C + (size_t)chunk * 8);
      float f[8];
      unpack8(v, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < cpb && chunk < c8) {
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int g = 0; g < rgs; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += red[(g * cpb + col) * 8 + e];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] *= inv;
    *reinterpret_cast<u32x4*>(y + (size_t)n * C + (size_t)chunk * 8) = pack8(s);
  }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int HW,
                                                      int C, int c8, long long total) {
  const float inv = 1.f / (float)HW;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % c8);
    const long long r = t / c8;
    const int n = (int)(r / HW);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + (size_t)n * C + (size_t)ch * 8);
    float f[8];
    unpack8(gv, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] *= inv;
    *reinterpret_cast<u32x4*>(dx + (size_t)r * C + (size_t)ch * 8) = pack8(f);
  }
}

static PoolGeom make_geom(int N, int H, int W, int C, int P, int Q, int k, int stride, int pad) {
  PoolGeom g;
  g.N = N; g.H = H; g.W = W; g.C = C; g.P = P; g.Q = Q; g.k = k; g.stride = stride; g.pad = pad; g.c8 = C / 8;
  return g;
}

extern "C" int ddl_maxpool_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, int P, int Q, int k,
                               int stride, int pad, void* stream) {
  if (C % 8 != 0 || k * k > 255) return -1;
  const PoolGeom g = make_geom(N, H, W, C, P, Q, k, stride, pad);
  const long long total = (long long)N * P * Q * g.c8;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, (unsigned char*)idx, g, total);
  return (int)hipGetLastError();
}

extern "C" int ddl_maxpool_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, int P, int Q,
                               int k, int stride, int pad, void* stream) {
  if (C % 8 != 0) return -1;
  const PoolGeom g = make_geom(N, H, W, C, P, Q, k, stride, pad);
  const long long total = (long long)N * H * W * g.c8;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const unsigned char*)idx, (bf16_t*)dx, g, total);
  return (int)hipGetLastError();
}

extern "C" int ddl_avgpool_fwd(const void* x, void* y, int N, int H, int W, int C, int P, int Q, int k, int stride,
                               void* stream) {
  if (C % 8 != 0) return -1;
  const PoolGeom g = make_geom(N, H, W, C, P, Q, k, stride, 0);
  const long long total = (long long)N * P * Q * g.c8;
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, g, total);
  return (int)hipGetLastError();
}

extern "C" int ddl_avgpool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int P, int Q, int k, int stride,
                               void* stream) {
  if (C % 8 != 0 || stride < k) return -1;
  const PoolGeom g = make_geom(N, H, W, C, P, Q, k, stride, 0);
  const long long total = (long long)N * H * W * g.c8;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, g, total);
  return (int)hipGetLastError();
}

extern "C" int ddl_gap_fwd(const void* x, void* y, int N, int HW, int C, void* stream) {
  if (C % 8 != 0) return -1;
  const int c8 = C / 8;
  const int cpb = c8 < 32 ? c8 : 32;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3((c8 + cpb - 1) / cpb, N), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, HW, C, c8);
  return (int)hipGetLastError();
}

extern "C" int ddl_gap_bwd(const void* dy, void* dx, int N, int HW, int C, void* stream) {
  if (C % 8 != 0) return -1;
  const int c8 = C / 8;
  const long long total = (long long)N * HW * c8;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, HW, C, c8, total);
  return (int)hipGetLastError();
}
