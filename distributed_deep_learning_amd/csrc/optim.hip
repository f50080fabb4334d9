// Fused multi-tensor optimizer step over a flat parameter arena (gfx950).
//
// Every parameter tensor lives in one contiguous fp32 master buffer (tensor starts aligned to DDL_CHUNK
// elements), with matching flat gradient / momentum buffers and a bf16 shadow copy the conv kernels read.  A
// "multi-tensor" update is therefore ONE launch: each block owns one 1024-element chunk, looks up its tensor
// (segment) for the per-tensor weight decay / LARS trust ratio, updates master + momentum, rewrites the bf16
// shadow and optionally clears the gradient in the same pass.  Hyper-parameters are read from a small device
// array so the launch is hipGraph-capturable and LR schedules need no re-capture.
//
//   SGD :  g' = trust * (gscale*g + wd*w);  v = mu*v + (1-damp)*g';  w -= lr * (nesterov ? g' + mu*v : v)
//   LARS:  trust = eta*||w|| / (||gscale*g|| + wd*||w|| + eps)   per tensor (segments flagged `lars`), else 1
//   Adam:  m = b1*m + (1-b1)*g'; v = b2*v + (1-b2)*g'^2; w -= lr * (m/bc1) / (sqrt(v/bc2) + eps)
#include "common.h"
#include "host_util.h"

#define DDL_CHUNK 1024

// hyper layout (floats): SGD  [0]=lr [1]=momentum [2]=dampening [3]=nesterov [4]=gscale [5]=lars_eta [6]=lars_eps
//                        Adam [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=gscale [5]=bias_corr1 [6]=bias_corr2

__global__ __launch_bounds__(256) void flat_sumsq_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                         const int* __restrict__ chunk_seg, float* __restrict__ norms,
                                                         int chunk_begin) {
  __shared__ float red[8];
  const int chunk = chunk_begin + blockIdx.x;
  const int seg = chunk_seg[chunk];
  if (seg < 0) return;
  const size_t i = (size_t)chunk * DDL_CHUNK + (size_t)threadIdx.x * 4;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + i);
  const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
  float sw = wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2] + wv[3] * wv[3];
  float sg = gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2] + gv[3] * gv[3];
  sw = wave_reduce_sum(sw);
  sg = wave_reduce_sum(sg);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[wid * 2] = sw; red[wid * 2 + 1] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&norms[seg * 2 + 0], red[0] + red[2] + red[4] + red[6]);
    atomicAdd(&norms[seg * 2 + 1], red[1] + red[3] + red[5] + red[7]);
  }
}

__global__ __launch_bounds__(256) void flat_sgd_kernel(float* __restrict__ w, float* __restrict__ g,
                                                       float* __restrict__ mom, bf16_t* __restrict__ wb,
                                                       const int* __restrict__ chunk_seg,
                                                       const float* __restrict__ seg_wd,
                                                       const int* __restrict__ seg_lars,
                                                       const float* __restrict__ norms,
                                                       const float* __restrict__ hyper, int chunk_begin,
                                                       int zero_grad) {
  const int chunk = chunk_begin + blockIdx.x;
  const int seg = chunk_seg[chunk];
  if (seg < 0) return;                       // padding-only chunk
  const float lr = hyper[0], mu = hyper[1], damp = hyper[2], nesterov = hyper[3], gscale = hyper[4];
  const float wd = seg_wd[seg];
  float trust = 1.f;
  if (norms != nullptr && seg_lars[seg]) {
    const float wn = sqrtf(norms[seg * 2 + 0]);
    const float gn = sqrtf(norms[seg * 2 + 1]) * fabsf(gscale);
    if (wn > 0.f && gn > 0.f) trust = hyper[5] * wn / (gn + wd * wn + hyper[6]);
  }
  const size_t i = (size_t)chunk * DDL_CHUNK + (size_t)threadIdx.x * 4;
  f32x4 wv = *reinterpret_cast<const f32x4*>(w + i);
  const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
  f32x4 mv = {0.f, 0.f, 0.f, 0.f};
  if (mom != nullptr) mv = *reinterpret_cast<const f32x4*>(mom + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gp = trust * (gscale * gv[e] + wd * wv[e]);
    float step = gp;
    if (mom != nullptr) {
      mv[e] = mu * mv[e] + (1.f - damp) * gp;
      step = nesterov != 0.f ? gp + mu * mv[e] : mv[e];
    }
    wv[e] -= lr * step;
  }
  *reinterpret_cast<f32x4*>(w + i) = wv;
  if (mom != nullptr) *reinterpret_cast<f32x4*>(mom + i) = mv;
  if (wb != nullptr) {
    u32x2 pk;
    pk[0] = pack_bf16x2(wv[0], wv[1]);
    pk[1] = pack_bf16x2(wv[2], wv[3]);
    *reinterpret_cast<u32x2*>(wb + i) = pk;
  }
  if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ __launch_bounds__(256) void flat_adam_kernel(float* __restrict__ w, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        bf16_t* __restrict__ wb, const int* __restrict__ chunk_seg,
                                                        const float* __restrict__ seg_wd,
                                                        const float* __restrict__ hyper, int chunk_begin,
                                                        int zero_grad) {
  const int chunk = chunk_begin + blockIdx.x;
  const int seg = chunk_seg[chunk];
  if (seg < 0) return;
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], gscale = hyper[4];
  const float bc1 = hyper[5], bc2 = hyper[6];
  const float wd = seg_wd[seg];
  const size_t i = (size_t)chunk * DDL_CHUNK + (size_t)threadIdx.x * 4;
  f32x4 wv = *reinterpret_cast<const f32x4*>(w + i);
  const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
  f32x4 mv = *reinterpret_cast<const f32x4*>(m + i);
  f32x4 vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float gp = gscale * gv[e] + wd * wv[e];
    mv[e] = b1 * mv[e] + (1.f - b1) * gp;
    vv[e] = b2 * vv[e] + (1.f - b2) * gp * gp;
    const float denom = sqrtf(vv[e] / bc2) + eps;
    wv[e] -= lr * (mv[e] / bc1) / denom;
  }
  *reinterpret_cast<f32x4*>(w + i) = wv;
  *reinterpret_cast<f32x4*>(m + i) = mv;
  *reinterpret_cast<f32x4*>(v + i) = vv;
  if (wb != nullptr) {
    u32x2 pk;
    pk[0] = pack_bf16x2(wv[0], wv[1]);
    pk[1] = pack_bf16x2(wv[2], wv[3]);
    *reinterpret_cast<u32x2*>(wb + i) = pk;
  }
  if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}

extern "C" int ddl_chunk_elems() { return DDL_CHUNK; }

// norms: [nseg][2] fp32, zeroed here; accumulates sum w^2 and sum g^2 over chunks [chunk_begin, chunk_end).
extern "C" int ddl_flat_sumsq(const float* w, const float* g, const int* chunk_seg, float* norms, int nseg,
                              int chunk_begin, int chunk_end, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = ddl_zero_async(norms, sizeof(float) * 2 * (size_t)nseg, st);
  if (e != hipSuccess) return (int)e;
  if (chunk_end <= chunk_begin) return 0;
  hipLaunchKernelGGL(flat_sumsq_kernel, dim3(chunk_end - chunk_begin), dim3(256), 0, st, w, g, chunk_seg, norms,
                     chunk_begin);
  return (int)hipGetLastError();
}

extern "C" int ddl_flat_sgd(float* w, float* g, float* mom, void* wb, const int* chunk_seg, const float* seg_wd,
                            const int* seg_lars, const float* norms, const float* hyper, int chunk_begin,
                            int chunk_end, int zero_grad, void* stream) {
  if (chunk_end <= chunk_begin) return 0;
  hipLaunchKernelGGL(flat_sgd_kernel, dim3(chunk_end - chunk_begin), dim3(256), 0, (hipStream_t)stream, w, g, mom,
                     (bf16_t*)wb, chunk_seg, seg_wd, seg_lars, norms, hyper, chunk_begin, zero_grad);
  return (int)hipGetLastError();
}

extern "C" int ddl_flat_adam(float* w, float* g, float* m, float* v, void* wb, const int* chunk_seg,
                             const float* seg_wd, const float* hyper, int chunk_begin, int chunk_end, int zero_grad,
                             void* stream) {
  if (chunk_end <= chunk_begin) return 0;
  hipLaunchKernelGGL(flat_adam_kernel, dim3(chunk_end - chunk_begin), dim3(256), 0, (hipStream_t)stream, w, g, m, v,
                     (bf16_t*)wb, chunk_seg, seg_wd, hyper, chunk_begin, zero_grad);
  return (int)hipGetLastError();
}
