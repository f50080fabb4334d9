// HBM streaming ceilings on MI355X: which access form gets closest to the ~6.3 TB/s the guide quotes?
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip ; run: ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) out[i] = in[i];
}

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ in, unsigned* __restrict__ sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}

// contiguous chunk per block instead of a grid-stride interleave
template <bool NT>
__global__ __launch_bounds__(256) void copy_chunk_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const u32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}

template <typename F>
static float time_ms(F f, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const size_t bytes = 1ull << 30;        // 1 GiB per buffer: far beyond the 256 MiB Infinity Cache
  const size_t n = bytes / 16;
  u32x4 *in, *out; unsigned* sink;
  hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMalloc(&sink, 4);
  hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
  const double gb = bytes / 1e9;
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    float t;
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<1, false>), dim3(grid), dim3(256), 0, 0, in, out, n); });
    printf("grid %5d copy plain u1   : %6.2f TB/s\n", grid, 2 * gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3(grid), dim3(256), 0, 0, in, out, n); });
    printf("grid %5d copy plain u4   : %6.2f TB/s\n", grid, 2 * gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, true>), dim3(grid), dim3(256), 0, 0, in, out, n); });
    printf("grid %5d copy nt    u4   : %6.2f TB/s\n", grid, 2 * gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_kernel<8, true>), dim3(grid), dim3(256), 0, 0, in, out, n); });
    printf("grid %5d copy nt    u8   : %6.2f TB/s\n", grid, 2 * gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_chunk_kernel<false>), dim3(grid), dim3(256), 0, 0, in, out, n); });
    printf("grid %5d copy chunked    : %6.2f TB/s\n", grid, 2 * gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(grid), dim3(256), 0, 0, in, sink, n); });
    printf("grid %5d read plain u4   : %6.2f TB/s\n", grid, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(grid), dim3(256), 0, 0, in, sink, n); });
    printf("grid %5d read nt    u4   : %6.2f TB/s\n", grid, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((write_kernel<false>), dim3(grid), dim3(256), 0, 0, out, n); });
    printf("grid %5d write plain     : %6.2f TB/s\n", grid, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((write_kernel<true>), dim3(grid), dim3(256), 0, 0, out, n); });
    printf("grid %5d write nt        : %6.2f TB/s\n", grid, gb / t);
    fflush(stdout);
  }
  return 0;
}
