// Fully connected layers on our own GEMM (gfx950).
//
// The three GEMMs of a Linear layer all run on the split-M "TN" MFMA kernel of conv_wgrad.hip
//     OUT[Ka][Cb] (fp32, accumulated) += sum_m A[m][Ka] * B[m][Cb]          (A, B bf16, reduction index m slowest)
// which splits the reduction over the whole chip — what skinny fc shapes need (VGG-16's 128 x 25088 x 4096 has 32
// output tiles; a tile-per-block GEMM would stream its 205 MB weight through 32 CUs):
//     forward   y [B][out]  = sum_i  xT[i][B]  * WT[i][out]     (WT = the arena's transposed bf16 shadow)
//     dgrad     dx[B][in]   = sum_o  gT[o][B]  * W [o][in]      (W  = the arena's bf16 shadow)
//     wgrad     dW[out][in] = sum_b  g [b][out] * x [b][in]     (straight into the arena gradient)
// This file holds what surrounds those GEMMs: the activation transposes (zero-padded to 8 columns), the forward
// epilogue (fp32 accumulator + bias -> ReLU -> dropout -> bf16), the gated backward pass (ReLU / dropout gate, bias
// gradient, scale) and the L1 loss of the reference's LSTM workload (/root/reference/src/pytorch/LSTM/main.py:163).
#include "common.h"
#include "host_util.h"

// ---------------------------------------------------------------- counter-based RNG for dropout
// state[0] = seed, state[1] = launch counter (device resident: a captured hipGraph draws fresh masks on every replay).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {        // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void rng_advance_kernel(unsigned long long* __restrict__ state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += 1ull;
}

// ---------------------------------------------------------------- forward epilogue
// y[r][c] = drop(relu(acc[r][c] + bias[c])) as bf16; rows >= M_valid and the accumulator's padding are not written.
// With n8_clear > 0 the accumulator ([n8_clear * 8] floats, >= the rows read) is ZEROED behind the read: a persistent
// accumulator is handed back ready for the next split-M GEMM, with no memset launch between them.
// Dropout keeps an element with probability 1 - p and scales it by 1/(1-p); the mask is NOT stored: a kept, live
// element is exactly one with y > 0, which is all the backward needs.
__global__ __launch_bounds__(256) void linear_epilogue_kernel(float* __restrict__ acc, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ y, long long n8, long long n8_clear, int N,
                                                              int relu, float drop_p,
                                                              const unsigned long long* __restrict__ rng) {
  const int n8row = N / 8;
  unsigned long long key = 0ull;
  unsigned thresh = 0u;
  float scale = 1.f;
  if (drop_p > 0.f) {
    key = mix64(rng[0] ^ (rng[1] * 0x9E3779B97F4A7C15ull));
    thresh = (unsigned)(drop_p * 65536.f);                                       // 16-bit uniform per element
    scale = 1.f / (1.f - drop_p);
  }
  const long long n8_all = n8 > n8_clear ? n8 : n8_clear;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8_all; i += (long long)gridDim.x * 256) {
    if (i >= n8) {                                                               // padding rows: only cleared
      *reinterpret_cast<f32x4*>(acc + i * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(acc + i * 8 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const int c = (int)(i % n8row) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(acc + i * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(acc + i * 8 + 4);
    if (i < n8_clear) {
      *reinterpret_cast<f32x4*>(acc + i * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(acc + i * 8 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (bias != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += bias[c + e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    if (drop_p > 0.f) {
      const unsigned long long r0 = mix64(key + 2ull * (unsigned long long)i);
      const unsigned long long r1 = mix64(key + 2ull * (unsigned long long)i + 1ull);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned u = (unsigned)(((e < 4 ? r0 : r1) >> (16 * (e & 3))) & 0xffffull);
        f[e] = u >= thresh ? f[e] * scale : 0.f;
      }
    }
    *reinterpret_cast<u32x4*>(y + i * 8) = pack8(f);
  }
}

// ---------------------------------------------------------------- backward gate
// g = dy * gate(y) * scale or dy * scale; db[c] (+)= sum_r g[r][c].  One pass over the gradient.
// GATE 1: gate = [y > 0] - ReLU, with or without dropout behind it (a dropped element is 0).  GATE 2: gate = [y != 0] -
// dropout WITHOUT ReLU, where a kept element may be negative (a kept element that is exactly 0 has measure zero).
template <int GATE>
__global__ __launch_bounds__(256) void linear_bwd_gate_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                              bf16_t* __restrict__ g, float* __restrict__ db, int M, int C,
                                                              int c8, float scale) {
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
      float f[8];
      unpack8(*reinterpret_cast<const u32x4*>(dy + off), f);
      if (GATE != 0) {
        float yf[8];
        unpack8(*reinterpret_cast<const u32x4*>(y + off), yf);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (GATE == 1 ? yf[e] > 0.f : yf[e] != 0.f) ? f[e] * scale : 0.f;
      } else if (scale != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= scale;
      }
      if (g != nullptr) *reinterpret_cast<u32x4*>(g + off) = pack8(f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
  }
  if (db == nullptr) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int o = threadIdx.x; o < cpb * 8; o += 256) {
    const int c = o / 8, e = o % 8;
    const int ch = blockIdx.y * cpb + c;
    if (ch >= c8) continue;
    float s = 0.f;
    for (int gi = 0; gi < rp; ++gi) s += red[(gi * cpb + c) * 8 + e];
    atomicAdd(&db[(size_t)ch * 8 + e], s);
  }
}

// ---------------------------------------------------------------- transpose [R][C] -> [C][Rp] (bf16, rows zero-padded)
template <typename TIN>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const TIN* __restrict__ in, bf16_t* __restrict__ out, int R, int C,
                                                            int Rp, long long in_pitch) {
  __shared__ bf16_t tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    bf16_t v = 0;
    if (r < R && c < C) {
      if constexpr (sizeof(TIN) == 4) v = f32_to_bf16(in[(size_t)r * in_pitch + c]); else v = (bf16_t)in[(size_t)r * in_pitch + c];
    }
    tile[ty + i * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (c < C && r < Rp) out[(size_t)c * Rp + r] = tile[tx][ty + i * 8];
  }
}

// ---------------------------------------------------------------- fp32 [M][Np] -> bf16 [M][N] (drop padding columns)
// n8_clear > 0 (needs acc_pitch == N): the accumulator's first n8_clear * 8 floats are zeroed behind the read.
__global__ __launch_bounds__(256) void cast_rows_kernel(float* __restrict__ acc, bf16_t* __restrict__ out, long long n8,
                                                        long long n8_clear, int N8, long long acc_pitch) {
  const long long n8_all = n8 > n8_clear ? n8 : n8_clear;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8_all; i += (long long)gridDim.x * 256) {
    const long long r = i / N8;
    const int c = (int)(i - r * N8) * 8;
    float* src = acc + r * acc_pitch + c;
    if (i >= n8) {
      *reinterpret_cast<f32x4*>(src) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(src + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
    if (i < n8_clear) {
      *reinterpret_cast<f32x4*>(src) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(src + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *reinterpret_cast<u32x4*>(out + i * 8) = pack8(f);
  }
}

// ---------------------------------------------------------------- L1 loss (mean absolute error) forward + gradient
// out[0] = mean |pred - target|;  grad = sign(pred - target) / n.  One block: the workload's outputs are [B][5].
template <typename T>
__global__ __launch_bounds__(1024) void l1_loss_kernel(const T* __restrict__ pred, const float* __restrict__ target,
                                                       float* __restrict__ grad, float* __restrict__ out, long long n) {
  __shared__ float part[16];
  float s = 0.f;
  const float inv = 1.f / (float)n;
  for (long long i = threadIdx.x; i < n; i += 1024) {
    float p;
    if constexpr (sizeof(T) == 4) p = pred[i]; else p = bf16_to_f32(pred[i]);
    const float d = p - target[i];
    s += fabsf(d);
    if (grad != nullptr) grad[i] = d > 0.f ? inv : (d < 0.f ? -inv : 0.f);
  }
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w];
    out[0] = t * inv;
  }
}

// ================================================================ C ABI
// clear_rows > 0: zero the accumulator's first clear_rows rows (>= M: its padding rows too) behind the read.
extern "C" int ddl_linear_epilogue(float* acc, const float* bias, void* y, int M, int N, int relu, float drop_p,
                                   const void* rng_state, int clear_rows, void* stream) {
  if (M <= 0 || N <= 0 || N % 8 != 0 || drop_p < 0.f || drop_p >= 1.f) return -1;
  if (drop_p > 0.f && rng_state == nullptr) return -2;
  if (clear_rows != 0 && clear_rows < M) return -3;
  const long long n8 = (long long)M * (N / 8), n8c = (long long)clear_rows * (N / 8);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_epilogue_kernel, dim3(grid_for(n8 > n8c ? n8 : n8c, 256)), dim3(256), 0, st, acc, bias,
                     (bf16_t*)y, n8, n8c, N, relu, drop_p, (const unsigned long long*)rng_state);
  if (drop_p > 0.f)
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, st, (unsigned long long*)rng_state);
  return (int)hipGetLastError();
}

// g may be null (only the bias gradient is wanted), y null = no gate, db null = no bias.  accumulate: add into db.
extern "C" int ddl_linear_bwd_gate(const void* dy, const void* y, void* g, float* db, int M, int C, float scale,
                                   int accumulate, int gate_nonzero, void* stream) {
  if (C % 8 != 0 || M <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (db != nullptr && !accumulate) {
    hipError_t e = ddl_zero_async(db, sizeof(float) * (size_t)C, st);
    if (e != hipSuccess) return (int)e;
  }
  const int c8 = C / 8, cpb = c8 <= 256 ? c8 : 256, rp = 256 / cpb;
  const int gy = (c8 + cpb - 1) / cpb;
  long long gx = ((long long)M + rp * 4 - 1) / (rp * 4);
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  if (y != nullptr && gate_nonzero)
    hipLaunchKernelGGL((linear_bwd_gate_kernel<2>), dim3((unsigned)gx, gy), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)y, (bf16_t*)g, db, M, C, c8, scale);
  else if (y != nullptr)
    hipLaunchKernelGGL((linear_bwd_gate_kernel<1>), dim3((unsigned)gx, gy), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)y, (bf16_t*)g, db, M, C, c8, scale);
  else
    hipLaunchKernelGGL((linear_bwd_gate_kernel<0>), dim3((unsigned)gx, gy), dim3(256), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)nullptr, (bf16_t*)g, db, M, C, c8, scale);
  return (int)hipGetLastError();
}

// in: [R][C] (row pitch in_pitch elements; fp32 when in_is_f32) -> out bf16 [C][Rp], rows R..Rp-1 zero.
extern "C" int ddl_transpose_pad(const void* in, int in_is_f32, void* out, int R, int C, int Rp, long long in_pitch,
                                 void* stream) {
  if (R <= 0 || C <= 0 || Rp < R || in_pitch < C) return -1;
  const dim3 grid((C + 31) / 32, (Rp + 31) / 32);
  if (in_is_f32)
    hipLaunchKernelGGL((transpose_pad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)in,
                       (bf16_t*)out, R, C, Rp, in_pitch);
  else
    hipLaunchKernelGGL((transpose_pad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in,
                       (bf16_t*)out, R, C, Rp, in_pitch);
  return (int)hipGetLastError();
}

// acc fp32 [M][acc_pitch] -> out bf16 [M][N] (N % 8 == 0, N <= acc_pitch)
// clear_rows > 0 (needs acc_pitch == N, clear_rows >= M): zero the accumulator's first clear_rows rows behind the read.
extern "C" int ddl_cast_rows(float* acc, void* out, int M, int N, long long acc_pitch, int clear_rows, void* stream) {
  if (M <= 0 || N <= 0 || N % 8 != 0 || acc_pitch < N || acc_pitch % 4 != 0) return -1;
  if (clear_rows != 0 && (clear_rows < M || acc_pitch != N)) return -3;
  const long long n8 = (long long)M * (N / 8), n8c = (long long)clear_rows * (N / 8);
  hipLaunchKernelGGL(cast_rows_kernel, dim3(grid_for(n8 > n8c ? n8 : n8c, 256)), dim3(256), 0, (hipStream_t)stream, acc,
                     (bf16_t*)out, n8, n8c, N / 8, acc_pitch);
  return (int)hipGetLastError();
}

// pred: fp32 or bf16 [n]; target fp32 [n]; grad fp32 [n] or null; out fp32 [1]
extern "C" int ddl_l1_loss(const void* pred, int pred_is_bf16, const float* target, float* grad, float* out, long long n,
                           void* stream) {
  if (n <= 0) return -1;
  if (pred_is_bf16)
    hipLaunchKernelGGL((l1_loss_kernel<bf16_t>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)pred, target,
                       grad, out, n);
  else
    hipLaunchKernelGGL((l1_loss_kernel<float>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)pred, target,
                       grad, out, n);
  return (int)hipGetLastError();
}

// ================================================================ kernel-size-1 Conv1d + ReLU (the LSTM workload's front)
// y[b][o][l] = relu(bias[o] + sum_c w[o][c] * x[b][c][l]):  /root/reference/src/pytorch/LSTM/model.py:70-71 applies
// Conv1d(history=10 -> 64, k=1) + ReLU to a [B, 10, 32] window, and the LSTM then walks the 64 output channels as
// time steps.  The reduction is 10 deep: far too short for MFMA (K = 32 per instruction), so this is a VALU kernel;
// its output is written as bf16 in exactly the [B][T=64][I=32] layout the fused LSTM kernel reads.
// One thread = one (b, o) row x 8 consecutive l.
__global__ __launch_bounds__(256) void conv1d_k1_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                                 int B, int C, int O, int L8, long long total) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int lc = (int)(t % L8);
    const long long r = t / L8;
    const int o = (int)(r % O);
    const long long b = r / O;
    const float bo = bias != nullptr ? bias[o] : 0.f;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bo;
    const float* xb = x + (b * C) * (long long)(L8 * 8) + lc * 8;
    for (int c = 0; c < C; ++c) {
      const float wv = w[o * C + c];
      const f32x4 a = *reinterpret_cast<const f32x4*>(xb + (long long)c * L8 * 8);
      const f32x4 d = *reinterpret_cast<const f32x4*>(xb + (long long)c * L8 * 8 + 4);
      acc[0] += wv * a[0]; acc[1] += wv * a[1]; acc[2] += wv * a[2]; acc[3] += wv * a[3];
      acc[4] += wv * d[0]; acc[5] += wv * d[1]; acc[6] += wv * d[2]; acc[7] += wv * d[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
    *reinterpret_cast<u32x4*>(y + t * 8) = pack8(acc);
  }
}

// dw[o][c] += sum_{b,l} g x[b][c][l],  db[o] += sum_{b,l} g,  g = dy * [y > 0].   Block = (O rows) x (L8 chunks) threads
// (<= 1024), grid-strided over b; per-thread partial sums for all c (C <= 16) in registers, folded over the l-chunks
// through LDS, one atomic per (o, c) per block.
template <int CMAX>
__global__ __launch_bounds__(1024) void conv1d_k1_relu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                                  const float* __restrict__ x, float* __restrict__ dw,
                                                                  float* __restrict__ db, int B, int C, int O, int L8) {
  extern __shared__ float red[];                       // [O * L8][CMAX + 1]
  const int tid = threadIdx.x;
  const int nthr = O * L8;
  float acc[CMAX + 1];
#pragma unroll
  for (int c = 0; c <= CMAX; ++c) acc[c] = 0.f;
  if (tid < nthr) {
    const int lc = tid % L8, o = tid / L8;
    for (long long b = blockIdx.x; b < B; b += gridDim.x) {
      const long long off = ((b * O + o) * (long long)L8 + lc) * 8;
      float g[8], yf[8];
      unpack8(*reinterpret_cast<const u32x4*>(dy + off), g);
      unpack8(*reinterpret_cast<const u32x4*>(y + off), yf);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { g[e] = yf[e] > 0.f ? g[e] : 0.f; s += g[e]; }
      acc[CMAX] += s;
      const float* xb = x + (b * C) * (long long)(L8 * 8) + lc * 8;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(xb + (long long)c * L8 * 8);
          const f32x4 d = *reinterpret_cast<const f32x4*>(xb + (long long)c * L8 * 8 + 4);
          acc[c] += g[0] * a[0] + g[1] * a[1] + g[2] * a[2] + g[3] * a[3] + g[4] * d[0] + g[5] * d[1] + g[6] * d[2] + g[7] * d[3];
        }
      }
    }
#pragma unroll
    for (int c = 0; c <= CMAX; ++c) red[tid * (CMAX + 1) + c] = acc[c];
  }
  __syncthreads();
  for (int i = tid; i < O * (CMAX + 1); i += blockDim.x) {
    const int o = i / (CMAX + 1), c = i % (CMAX + 1);
    if (c < C || c == CMAX) {
      float s = 0.f;
      for (int lc = 0; lc < L8; ++lc) s += red[(o * L8 + lc) * (CMAX + 1) + c];
      if (c == CMAX) { if (db != nullptr) atomicAdd(&db[o], s); }
      else atomicAdd(&dw[o * C + c], s);
    }
  }
}

extern "C" int ddl_conv1d_k1_relu_fwd(const float* x, const float* w, const float* bias, void* y, int B, int C, int O, int L,
                                      void* stream) {
  if (B <= 0 || C <= 0 || O <= 0 || L <= 0 || L % 8 != 0) return -1;
  const long long total = (long long)B * O * (L / 8);
  hipLaunchKernelGGL(conv1d_k1_relu_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     (bf16_t*)y, B, C, O, L / 8, total);
  return (int)hipGetLastError();
}

// dw [O][C] and db [O] (may be null) are ACCUMULATED into (arena gradients); C <= 16, O * L/8 <= 1024.
extern "C" int ddl_conv1d_k1_relu_bwd(const void* dy, const void* y, const float* x, float* dw, float* db, int B, int C, int O,
                                      int L, void* stream) {
  if (B <= 0 || C <= 0 || C > 16 || O <= 0 || L <= 0 || L % 8 != 0 || O * (L / 8) > 1024) return -1;
  const int L8 = L / 8, nthr = O * L8;
  const int block = ((nthr + 63) / 64) * 64;
  const int smem = nthr * 17 * (int)sizeof(float);
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv1d_k1_relu_bwd_kernel<16>), smem);
  if (rc != 0) return rc;
  const int grid = B < 512 ? B : 512;
  hipLaunchKernelGGL((conv1d_k1_relu_bwd_kernel<16>), dim3(grid), dim3(block), smem, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)y, x, dw, db, B, C, O, L8);
  return (int)hipGetLastError();
}
