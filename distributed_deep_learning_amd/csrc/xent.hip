// Fused softmax-cross-entropy (gfx950): one wave64 per row computes max / log-sum-exp, the mean loss
// contribution, the top-1 hit and d(logits) = (softmax - target) / B in a single launch.  Logits are bf16 or
// fp32 [B][V]; targets are int64 class ids or fp32 probability rows (the reference's one-hot float targets).
#include "common.h"
#include "host_util.h"

template <typename T> __device__ __forceinline__ float ld(const T* p, size_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void st(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }

// out[0] += mean loss contribution, out[1] += top-1 hits
template <typename T, bool SOFT>
__global__ __launch_bounds__(256) void xent_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                   const float* __restrict__ probs, T* __restrict__ dlogits,
                                                   float* __restrict__ out, int B, int V, float inv_b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const size_t base = (size_t)row * V;
  // pass 1: max + argmax (first max wins)
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int j = lane; j < V; j += 64) {
    const float z = ld<T>(logits, base + j);
    if (z > mx) { mx = z; mi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omx = __shfl_xor(mx, o, 64);
    const int omi = __shfl_xor(mi, o, 64);
    if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
  }
  // pass 2: sum exp, target terms
  float se = 0.f, tsum = 0.f, tz = 0.f, tmx = -INFINITY;
  int tmi = 0x7fffffff;
  for (int j = lane; j < V; j += 64) {
    const float z = ld<T>(logits, base + j);
    se += __expf(z - mx);
    if (SOFT) {
      const float t = probs[base + j];
      tsum += t;
      tz += t * z;
      if (t > tmx) { tmx = t; tmi = j; }
    }
  }
  se = wave_reduce_sum(se);
  const float lse = mx + __logf(se);
  float loss;
  int target;
  if (SOFT) {
    tsum = wave_reduce_sum(tsum);
    tz = wave_reduce_sum(tz);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float omx = __shfl_xor(tmx, o, 64);
      const int omi = __shfl_xor(tmi, o, 64);
      if (omx > tmx || (omx == tmx && omi < tmi)) { tmx = omx; tmi = omi; }
    }
    loss = tsum * lse - tz;
    target = tmi;
  } else {
    target = (int)labels[row];
    const float zt = (target >= 0 && target < V) ? ld<T>(logits, base + target) : 0.f;
    loss = (target >= 0 && target < V) ? (lse - zt) : 0.f;
    tsum = (target >= 0 && target < V) ? 1.f : 0.f;
  }
  if (lane == 0) {
    atomicAdd(&out[0], loss * inv_b);
    if (mi == target) atomicAdd(&out[1], 1.f);
  }
  // pass 3: gradient
  if (dlogits != nullptr) {
    const float inv_se = 1.f / se;
    for (int j = lane; j < V; j += 64) {
      const float z = ld<T>(logits, base + j);
      const float pj = __expf(z - mx) * inv_se;
      float t;
      if (SOFT) t = probs[base + j]; else t = (j == target) ? 1.f : 0.f;
      st<T>(dlogits, base + j, (pj * tsum - t) * inv_b);
    }
  }
}

// is_bf16: logits/dlogits dtype; labels (int64) or probs (fp32) — exactly one non-null.  out: 2 floats, zeroed here.
extern "C" int ddl_softmax_xent(const void* logits, int is_bf16, const long long* labels, const float* probs,
                                void* dlogits, float* out, int B, int V, void* stream) {
  if (B <= 0 || V <= 0 || ((labels == nullptr) == (probs == nullptr))) return -1;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = ddl_zero_async(out, 2 * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((B + 3) / 4), block(256);
  const float inv_b = 1.f / (float)B;
  if (is_bf16) {
    if (probs) hipLaunchKernelGGL((xent_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)logits, labels, probs, (bf16_t*)dlogits, out, B, V, inv_b);
    else       hipLaunchKernelGGL((xent_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)logits, labels, probs, (bf16_t*)dlogits, out, B, V, inv_b);
  } else {
    if (probs) hipLaunchKernelGGL((xent_kernel<float, true>), grid, block, 0, s, (const float*)logits, labels, probs, (float*)dlogits, out, B, V, inv_b);
    else       hipLaunchKernelGGL((xent_kernel<float, false>), grid, block, 0, s, (const float*)logits, labels, probs, (float*)dlogits, out, B, V, inv_b);
  }
  return (int)hipGetLastError();
}
