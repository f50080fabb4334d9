// Element-wise and layout kernels (gfx950): all 16-byte vectorized, grid-stride, <= 2048 blocks.
#include "common.h"
#include <stdlib.h>
#include "host_util.h"

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                            long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(in + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(in + i * 8 + 4);
    u32x4 o;
    o[0] = pack_bf16x2(a[0], a[1]); o[1] = pack_bf16x2(a[2], a[3]);
    o[2] = pack_bf16x2(b[0], b[1]); o[3] = pack_bf16x2(b[2], b[3]);
    *reinterpret_cast<u32x4*>(out + i * 8) = o;
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out,
                                                            long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(in + i * 8);
    float f[8];
    unpack8(v, f);
    *reinterpret_cast<f32x4*>(out + i * 8) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(out + i * 8 + 4) = f32x4{f[4], f[5], f[6], f[7]};
  }
}

// y = max(x, 0)
__global__ __launch_bounds__(256) void relu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    *reinterpret_cast<u32x4*>(y + i * 8) = pack8(f);
  }
}

// dx = dy * [y > 0]
__global__ __launch_bounds__(256) void relu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                       bf16_t* __restrict__ dx, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 g = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dy + i * 8));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + i * 8));
    float gf[8], yf[8];
    unpack8(g, gf);
    unpack8(v, yf);
#pragma unroll
    for (int e = 0; e < 8; ++e) gf[e] = yf[e] > 0.f ? gf[e] : 0.f;
    *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(gf);
  }
}

// out = a + b, optional ReLU
template <bool RELU>
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                  bf16_t* __restrict__ out, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const u32x4 va = *reinterpret_cast<const u32x4*>(a + i * 8);
    const u32x4 vb = *reinterpret_cast<const u32x4*>(b + i * 8);
    float fa[8], fb[8];
    unpack8(va, fa);
    unpack8(vb, fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) { fa[e] += fb[e]; if (RELU) fa[e] = fmaxf(fa[e], 0.f); }
    *reinterpret_cast<u32x4*>(out + i * 8) = pack8(fa);
  }
}

// Per-channel bias gradient: db[c] += sum_m g[m][c]  (bf16 [M][C] in, fp32 atomics out).
// GATE: g = dy * [y > 0] is formed on the fly AND written to dx — ReLU backward and bias gradient in ONE pass over
// the activation gradient (VGG-style conv+bias+ReLU blocks: saves a full re-read of the gated gradient).
template <bool GATE>
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                        bf16_t* __restrict__ dx, float* __restrict__ db, int M, int C,
                                                        int c8) {
  __shared__ float red[256 * 8];
  const int cpb = c8 <= 256 ? c8 : 256;
  const int rp = 256 / cpb;
  const int col = threadIdx.x % cpb, rg = threadIdx.x / cpb;
  const int chunk = blockIdx.y * cpb + col;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (rg < rp && chunk < c8) {
    for (int row = blockIdx.x * rp + rg; row < M; row += gridDim.x * rp) {
      const size_t off = (size_t)row * C + (size_t)chunk * 8;
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dy + off));
      float f[8];
      unpack8(v, f);
      if (GATE) {
        const u32x4 yv = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + off));
        float yf[8];
        unpack8(yv, yf);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = yf[e] > 0.f ? f[e] : 0.f;
        *reinterpret_cast<u32x4*>(dx + off) = pack8(f);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int o = threadIdx.x; o < cpb * 8; o += 256) {
    const int c = o / 8, e = o % 8;
    const int ch = blockIdx.y * cpb + c;
    if (ch >= c8) continue;
    float s = 0.f;
    for (int g = 0; g < rp; ++g) s += red[(g * cpb + c) * 8 + e];
    atomicAdd(&db[(size_t)ch * 8 + e], s);
  }
}

// Filter repack for dgrad: in fp32 or bf16 [K][RS][C]  ->  out bf16 [C][RS][K]  (+ optional bf16 copy [K][RS][C]).
// 32x32 LDS-tiled transpose per filter tap; blockIdx.z = tap.
template <typename TIN>
__global__ __launch_bounds__(256) void filter_repack_kernel(const TIN* __restrict__ in, bf16_t* __restrict__ copy,
                                                            bf16_t* __restrict__ outT, int K, int RS, int C) {
  __shared__ bf16_t tile[32][33];
  const int tap = blockIdx.z;
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + i * 8, c = c0 + tx;
    bf16_t v = 0;
    if (k < K && c < C) {
      const size_t off = ((size_t)k * RS + tap) * C + c;
      if constexpr (sizeof(TIN) == 4) v = f32_to_bf16(in[off]); else v = (bf16_t)in[off];
      if (copy != nullptr) copy[off] = v;
    }
    tile[ty + i * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, k = k0 + tx;
    if (c < C && k < K) outT[((size_t)c * RS + tap) * K + k] = tile[tx][ty + i * 8];
  }
}

// All conv filters of an arena in ONE launch: bf16 shadow [K][RS][C] at in_off -> [C][RS][K] at out_off.
struct RepackDesc {
  long long in_off, out_off;
  int K, RS, C, tile_begin, tiles_c, tiles_k;
};

__global__ __launch_bounds__(256) void filter_transpose_batched_kernel(const bf16_t* __restrict__ src,
                                                                       bf16_t* __restrict__ dst,
                                                                       const RepackDesc* __restrict__ descs,
                                                                       int ndesc) {
  __shared__ bf16_t tile[32][33];
  int lo = 0, hi = ndesc - 1;                 // last descriptor with tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RepackDesc d = descs[lo];
  int local = blockIdx.x - d.tile_begin;
  const int per_tap = d.tiles_c * d.tiles_k;
  const int tap = local / per_tap;
  local -= tap * per_tap;
  const int tk = local / d.tiles_c, tc = local - tk * d.tiles_c;
  const int c0 = tc * 32, k0 = tk * 32;
  const bf16_t* in = src + d.in_off;
  bf16_t* out = dst + d.out_off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + i * 8, c = c0 + tx;
    bf16_t v = 0;
    if (k < d.K && c < d.C) v = in[((size_t)k * d.RS + tap) * d.C + c];
    tile[ty + i * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, k = k0 + tx;
    if (c < d.C && k < d.K) out[((size_t)c * d.RS + tap) * d.K + k] = tile[tx][ty + i * 8];
  }
}

// NCHW fp32 -> NHWC bf16 with the channel axis zero-padded to Cp (input images: 3 -> 8)
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                               int N, int C, int HW, int Cp, long long total) {
  // one thread = one output pixel x 8 channels
  const int cp8 = Cp / 8;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ch = (int)(t % cp8);
    const long long pix = t / cp8;
    const int n = (int)(pix / HW);
    const int hw = (int)(pix % HW);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = ch * 8 + e;
      f[e] = c < C ? in[((size_t)n * C + c) * HW + hw] : 0.f;
    }
    *reinterpret_cast<u32x4*>(out + (size_t)pix * Cp + (size_t)ch * 8) = pack8(f);
  }
}

// Space-to-depth input for a 7x7 / stride 2 / pad 3 stem on 3-channel images (the ResNet / DenseNet stem):
//   out[n][h2][w2][(p*2 + q)*3 + c] = in[n][c][2*(h2-1) + p - 1][2*(w2-1) + q - 1]      (0 outside the image)
// with h2 in [0, H/2 + 3), channels 12..15 zero.  A 4x4 / stride 1 / pad 0 convolution over this tensor with the
// filter regrouped the same way (tap r = 2a + p, r = 7 -> 0) IS the 7x7 stem: reduction length 256 instead of 392
// (8-channel padded taps), no strided taps, and the padding is materialised so no bounds logic runs in the conv.
__global__ __launch_bounds__(256) void nchw_to_s2d_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                           int N, int H, int W, int H2, int W2, long long total) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int w2 = (int)(t % W2);
    const long long r = t / W2;
    const int h2 = (int)(r % H2);
    const int n = (int)(r / H2);
    float f[16];
#pragma unroll
    for (int e = 12; e < 16; ++e) f[e] = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int ih = 2 * (h2 - 1) + p - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int iw = 2 * (w2 - 1) + q - 1;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          f[(p * 2 + q) * 3 + c] = ok ? in[(((size_t)n * 3 + c) * H + ih) * W + iw] : 0.f;
      }
    }
    bf16_t* o = out + (size_t)t * 16;
    *reinterpret_cast<u32x4*>(o) = pack8(f);
    *reinterpret_cast<u32x4*>(o + 8) = pack8(f + 8);
  }
}

extern "C" int ddl_nchw_to_s2d(const float* in, void* out, int N, int H, int W, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return -1;
  const int H2 = H / 2 + 3, W2 = W / 2 + 3;
  const long long total = (long long)N * H2 * W2;
  hipLaunchKernelGGL(nchw_to_s2d_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     (bf16_t*)out, N, H, W, H2, W2, total);
  return (int)hipGetLastError();
}

extern "C" int ddl_cast_f32_bf16(const float* in, void* out, long long n, void* stream) {
  if (n % 8 != 0) return -1;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     (bf16_t*)out, n / 8);
  return (int)hipGetLastError();
}
extern "C" int ddl_cast_bf16_f32(const void* in, float* out, long long n, void* stream) {
  if (n % 8 != 0) return -1;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, out, n / 8);
  return (int)hipGetLastError();
}
extern "C" int ddl_relu_fwd(const void* x, void* y, long long n, void* stream) {
  if (n % 8 != 0) return -1;
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, n / 8);
  return (int)hipGetLastError();
}
extern "C" int ddl_relu_bwd(const void* dy, const void* y, void* dx, long long n, void* stream) {
  if (n % 8 != 0) return -1;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n / 8);
  return (int)hipGetLastError();
}
extern "C" int ddl_add(const void* a, const void* b, void* out, long long n, int relu, void* stream) {
  if (n % 8 != 0) return -1;
  if (relu) hipLaunchKernelGGL((add_kernel<true>), dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / 8);
  else      hipLaunchKernelGGL((add_kernel<false>), dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / 8);
  return (int)hipGetLastError();
}
static int launch_bias_grad(const void* dy, const void* y, void* dx, float* db, int M, int C, int accumulate,
                            hipStream_t st) {
  if (C % 8 != 0 || M <= 0) return -1;
  if (!accumulate) {
    hipError_t e = ddl_zero_async(db, sizeof(float) * (size_t)C, st);
    if (e != hipSuccess) return (int)e;
  }
  const int c8 = C / 8, cpb = c8 <= 256 ? c8 : 256, rp = 256 / cpb;
  const int gy = (c8 + cpb - 1) / cpb;
  long long gx = ((long long)M + rp * 8 - 1) / (rp * 8);
  if (gx < 1) gx = 1;
  const long long cap = y != nullptr ? 2048 : 1024;
  if (gx > cap) gx = cap;
  if (y != nullptr)
    hipLaunchKernelGGL((bias_grad_kernel<true>), dim3((unsigned)gx, gy), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)y, (bf16_t*)dx, db, M, C, c8);
  else
    hipLaunchKernelGGL((bias_grad_kernel<false>), dim3((unsigned)gx, gy), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)nullptr, (bf16_t*)nullptr, db, M, C, c8);
  return (int)hipGetLastError();
}
extern "C" int ddl_bias_grad(const void* dy, float* db, int M, int C, void* stream) {
  return launch_bias_grad(dy, nullptr, nullptr, db, M, C, 0, (hipStream_t)stream);
}
// dx = dy * [y > 0] and db (+)= column sums of dx, one pass.  accumulate: add into db (arena gradient) instead of
// overwriting it.
extern "C" int ddl_relu_bwd_bias_grad(const void* dy, const void* y, void* dx, float* db, int M, int C,
                                      int accumulate, void* stream) {
  if (y == nullptr || dx == nullptr) return -2;
  return launch_bias_grad(dy, y, dx, db, M, C, accumulate, (hipStream_t)stream);
}
// in_is_f32: 1 -> `in` is fp32 (master weights), 0 -> bf16.  copy may be null.
extern "C" int ddl_filter_repack(const void* in, int in_is_f32, void* copy, void* outT, int K, int RS, int C,
                                 void* stream) {
  const dim3 grid((C + 31) / 32, (K + 31) / 32, RS);
  if (in_is_f32)
    hipLaunchKernelGGL((filter_repack_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)in,
                       (bf16_t*)copy, (bf16_t*)outT, K, RS, C);
  else
    hipLaunchKernelGGL((filter_repack_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                       (bf16_t*)copy, (bf16_t*)outT, K, RS, C);
  return (int)hipGetLastError();
}
// descs: device array of RepackDesc (48 bytes each, see ops/kernels.py), total_tiles = sum RS*tiles_c*tiles_k
extern "C" int ddl_filter_transpose_batched(const void* src, void* dst, const void* descs, int ndesc, int total_tiles,
                                            void* stream) {
  if (ndesc <= 0 || total_tiles <= 0) return 0;
  hipLaunchKernelGGL(filter_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, (const RepackDesc*)descs, ndesc);
  return (int)hipGetLastError();
}
extern "C" int ddl_repack_desc_bytes() { return (int)sizeof(RepackDesc); }
extern "C" int ddl_nchw_to_nhwc_pad(const float* in, void* out, int N, int C, int HW, int Cp, void* stream) {
  if (Cp % 8 != 0 || Cp < C) return -1;
  const long long total = (long long)N * HW * (Cp / 8);
  hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     (bf16_t*)out, N, C, HW, Cp, total);
  return (int)hipGetLastError();
}

// One wave that stays resident for ~`ticks` wall-clock ticks (100 MHz), bounded by an iteration cap as well.  Used by
// ops/streams.py to find out whether two HIP streams really run concurrently (= sit on different hardware queues).
__global__ __launch_bounds__(64) void spin_kernel(long long ticks, int max_iter, int* __restrict__ sink) {
  const long long t0 = wall_clock64();
  int it = 0;
  while (it < max_iter && wall_clock64() - t0 < ticks) {
    ++it;
    __builtin_amdgcn_s_sleep(32);
  }
  if (sink != nullptr && it < 0) *sink = it;       // never true: keeps the loop observable
}
extern "C" int ddl_spin(long long microseconds, void* stream) {
  if (microseconds < 0 || microseconds > 100000) return -1;
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, microseconds * 100, 1 << 22, (int*)nullptr);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- stand-in for a ring collective's kernel (diagnostics)
// What an RCCL ring all-reduce does to the GPU that hosts it: a FEW workgroups (one per channel) that stream the bucket
// through registers - read, combine, write - for the whole duration of the transfer, on the communicator's stream, beside
// the compute kernels.  On one GPU the real collective is an identity; this kernel reproduces its footprint (workgroups,
// bytes read + written, `passes` sweeps ~ the 2 (N-1)/N traversals of a ring) so that bench.py --comm-proxy can measure
// what that many resident workgroups cost the compute streams.  Values are written back unchanged.
__global__ __launch_bounds__(256) void stream_proxy_kernel(u32x4* __restrict__ p, long long n16, int passes) {
  for (int ps = 0; ps < passes; ++ps)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
      u32x4 v = p[i];
      asm volatile("" : "+v"(v));              // keep the load and the store: the data goes through registers
      p[i] = v;
    }
}
extern "C" int ddl_stream_proxy(void* buf, long long bytes, int workgroups, int passes, void* stream) {
  if (workgroups < 1 || workgroups > 1024 || passes < 1 || passes > 16 || bytes < 0 || ((uintptr_t)buf & 15) != 0) return -1;
  const long long n16 = bytes / 16;
  if (n16 == 0) return 0;
  // DDL_PROXY_LDS_KB: LDS the proxy's blocks hold while they run (a communication kernel's blocks carry tens of KiB of
  // staging buffers: a CU that hosts one cannot host a compute block that needs (almost) all 160 KiB)
  static const int lds_kb = [] { const char* e = getenv("DDL_PROXY_LDS_KB"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
  hipLaunchKernelGGL(stream_proxy_kernel, dim3(workgroups), dim3(256), (size_t)lds_kb * 1024, (hipStream_t)stream, (u32x4*)buf,
                     n16, passes);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- PS / ZeRO-1 pull: masters of the foreign ranges
// After the bf16 shadow of every shard has been gathered, the fp32 masters of the ranges OTHER ranks own are refreshed:
// (1) one launch casts every foreign range from the shadow (ranges = [lo, hi) pairs, multiples of 8, blockIdx.y = range);
// (2) the parameters that kernels consume in fp32 travel exactly: pack (owned values, zero elsewhere) -> one small
//     all-reduce -> scatter.  idx = arena positions, own = 1 where this rank owns the position.
__global__ __launch_bounds__(256) void cast_ranges_kernel(const bf16_t* __restrict__ shadow, float* __restrict__ master,
                                                          const long long* __restrict__ ranges) {
  const long long lo = ranges[2 * blockIdx.y], hi = ranges[2 * blockIdx.y + 1];
  const long long n8 = (hi - lo) / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(shadow + lo + i * 8), f);
    *reinterpret_cast<f32x4*>(master + lo + i * 8) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(master + lo + i * 8 + 4) = f32x4{f[4], f[5], f[6], f[7]};
  }
}
__global__ __launch_bounds__(256) void pack_owned_kernel(const float* __restrict__ master, const long long* __restrict__ idx,
                                                         const unsigned char* __restrict__ own, float* __restrict__ out,
                                                         long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = own[i] ? master[idx[i]] : 0.f;
}
__global__ __launch_bounds__(256) void scatter_values_kernel(float* __restrict__ master, const long long* __restrict__ idx,
                                                             const float* __restrict__ vals, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    master[idx[i]] = vals[i];
}
// ranges: device table of nranges [lo, hi) pairs (int64, multiples of 8, max_len = the longest one)
extern "C" int ddl_cast_ranges(const void* shadow, float* master, const long long* ranges, int nranges, long long max_len,
                               void* stream) {
  if (nranges <= 0 || max_len <= 0) return 0;
  if (nranges > 65535 || max_len % 8 != 0) return -1;
  hipLaunchKernelGGL(cast_ranges_kernel, dim3(grid_for(max_len / 8, 256), nranges), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)shadow, master, ranges);
  return (int)hipGetLastError();
}
extern "C" int ddl_pack_owned(const float* master, const long long* idx, const void* own, float* out, long long n,
                              void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pack_owned_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, master, idx,
                     (const unsigned char*)own, out, n);
  return (int)hipGetLastError();
}
extern "C" int ddl_scatter_values(float* master, const long long* idx, const float* vals, long long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scatter_values_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, master, idx, vals, n);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- channel windows of NHWC buffers
// dst[row][0..C) = src[row][0..C) for two [M][*] bf16 buffers with independent row pitches (elements): packs a dense
// tensor into a channel window of a wider buffer (dense-block feature map) or extracts a window into a dense tensor.
// One 16-byte chunk per thread step; base pointers already point at the window's first channel.
__global__ __launch_bounds__(256) void copy_channels_kernel(const bf16_t* __restrict__ src, long long src_ld,
                                                            bf16_t* __restrict__ dst, long long dst_ld, long long M,
                                                            FastDiv div_c8) {
  const long long total = M * (long long)div_c8.d;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long row;
    int ch;
    if (total < (1ll << 31)) {
      row = (long long)fastdiv((unsigned)i, div_c8);
      ch = (int)(i - row * (long long)div_c8.d);
    } else {
      row = i / (long long)div_c8.d;
      ch = (int)(i % (long long)div_c8.d);
    }
    *reinterpret_cast<u32x4*>(dst + row * dst_ld + ch * 8) =
        *reinterpret_cast<const u32x4*>(src + row * src_ld + ch * 8);
  }
}

extern "C" int ddl_copy_channels(const void* src, int src_ld, void* dst, int dst_ld, long long M, int C, void* stream) {
  if (C <= 0 || C % 8 != 0 || M <= 0 || src_ld < C || dst_ld < C || src_ld % 8 != 0 || dst_ld % 8 != 0) return -1;
  if ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) return -2;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(M * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (long long)src_ld, (bf16_t*)dst, (long long)dst_ld, M, make_fastdiv((unsigned)(C / 8)));
  return (int)hipGetLastError();
}
