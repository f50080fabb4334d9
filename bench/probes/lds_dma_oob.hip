// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` lane write zeros to LDS or leave it untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned short* in, unsigned bytes, u32x4* o) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4 fill = {0xABABABABu, 0xABABABABu, 0xABABABABu, 0xABABABABu};
  *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16) = fill;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, bytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (threadIdx.x >> 6) * 1024), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  o[threadIdx.x] = *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16);
}
int main() {
  const int T = 256;
  std::vector<unsigned short> h(T * 8);
  for (int i = 0; i < T * 8; ++i) h[i] = (unsigned short)(i + 1);
  unsigned short* d; u32x4* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, T * 16);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(T), T * 16, 0, d, (unsigned)(h.size() * 2), o);
  std::vector<unsigned> r(T * 4);
  hipMemcpy(r.data(), o, T * 16, hipMemcpyDeviceToHost);
  int zeros = 0, stale = 0, good = 0, other = 0;
  for (int t = 0; t < T; ++t) {
    unsigned v = r[t * 4];
    if (t & 1) { if (v == 0) zeros++; else if (v == 0xABABABABu) stale++; else other++; }
    else { unsigned want = (unsigned)(t * 8 + 1) | ((unsigned)(t * 8 + 2) << 16); if (v == want) good++; else other++; }
  }
  printf("in-range lanes correct: %d/128; OOB lanes: zero=%d stale=%d other=%d\n", good, zeros, stale, other);
  return 0;
}
