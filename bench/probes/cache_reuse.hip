// Does the Infinity Cache (256 MiB, memory side) serve the TAIL of a tensor a kernel has just written or read, when the
// next kernel sweeps the same tensor in the OPPOSITE direction?  write(ascending) then read ascending / descending, and
// read(ascending) then read ascending / descending, for tensors below and above the cache size.
// Build: hipcc --offload-arch=gfx950 -O3 -o cache_reuse cache_reuse.hip ; run: ./cache_reuse
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// all blocks sweep the tensor together (grid-stride), from the front (DESC = false) or from the back
template <bool DESC>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ in, unsigned* __restrict__ sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const u32x4 v = in[DESC ? n - 1 - i : i];
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <bool DESC>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[DESC ? n - 1 - i : i] = v;
}

template <typename F> float time_ms(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  float tot = 0.f;
  for (int i = 0; i < iters; ++i) { tot += f(); }
  return tot / iters;
}

int main() {
  unsigned* sink; hipMalloc(&sink, 64);
  const int grid = 2048;
  for (size_t mb : {102ul, 205ul, 410ul, 822ul, 1644ul}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    u32x4* buf; hipMalloc(&buf, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](auto first, auto second) {
      float best = 1e9f;
      for (int it = 0; it < 5; ++it) {
        first();
        hipEventRecord(e0);
        second();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      return best;
    };
    auto w_asc = [&] { hipLaunchKernelGGL(write_kernel<false>, dim3(grid), dim3(256), 0, 0, buf, n); };
    auto r_asc = [&] { hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(256), 0, 0, buf, sink, n); };
    auto r_desc = [&] { hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(256), 0, 0, buf, sink, n); };
    const float wa_ra = timed(w_asc, r_asc), wa_rd = timed(w_asc, r_desc);
    const float ra_ra = timed(r_asc, r_asc), ra_rd = timed(r_asc, r_desc);
    printf("%5zu MiB | after WRITE asc: read asc %.3f ms (%.2f TB/s)  read desc %.3f ms (%.2f TB/s) | after READ asc: read asc %.3f ms (%.2f)  "
           "read desc %.3f ms (%.2f)\n", mb, wa_ra, bytes / wa_ra / 1e9, wa_rd, bytes / wa_rd / 1e9, ra_ra, bytes / ra_ra / 1e9, ra_rd,
           bytes / ra_rd / 1e9);
    hipFree(buf);
  }
  return 0;
}
