// Whole small MLP (Linear + ReLU stacks up to 64 wide, the reference's MLP workload:
// /root/reference/src/pytorch/MLP/model.py:51-59, 48 -> 38 -> 38 -> 5) as TWO launches per step instead of ~20:
//
//   mlp_fwd_kernel  x -> every layer's activation (saved for the backward) and the logits
//   mlp_bwd_kernel  dlogits -> all weight / bias gradients (accumulated straight into the fp32 arena) and dx
//
// A network of 3.5k parameters is pure launch latency on a 256-CU GPU, so the design goal is "no intermediate
// kernel boundary", not FLOP/s: one workgroup of 256 threads owns 32 samples, the parameters of ALL layers sit in LDS (14 KB for the
// reference net), activations live in LDS as [unit][sample] (lane-contiguous, conflict free), math is fp32 (the
// reference trains in fp32; MFMA tiles would be >75 % padding at these widths).  Per-block weight gradients are
// reduced over the block's samples in LDS and leave with one float atomic per weight and block.
#include "common.h"
#include "host_util.h"

#define MLP_MAX_LAYERS 8
#define MLP_MAX_WIDTH 64
#define MLP_SAMPLES 32                      // samples per workgroup
#define MLP_THREADS 256
#define MLP_PITCH (MLP_SAMPLES + 1)         // row pitch of the [unit][sample] tiles

struct MlpDesc {
  const float* W[MLP_MAX_LAYERS];           // [dims[l+1]][dims[l]] row-major (torch.nn.Linear layout)
  const float* b[MLP_MAX_LAYERS];           // [dims[l+1]] or null
  float* dW[MLP_MAX_LAYERS];                // accumulated into (same layouts), may be null for a frozen layer
  float* db[MLP_MAX_LAYERS];
  int dims[MLP_MAX_LAYERS + 1];
  int woff[MLP_MAX_LAYERS + 1];             // offset of layer l's weights inside the LDS parameter block (floats)
  int boff[MLP_MAX_LAYERS];                 // offset of layer l's bias
  int aoff[MLP_MAX_LAYERS + 1];             // column offset of layer l's INPUT inside a saved-activation row
  int L;
  int relu_last;                            // ReLU after the last layer too
  int act_width;                            // floats per saved-activation row = sum of dims[0..L-1]
  int param_floats;
};

namespace {

__device__ __forceinline__ void load_params(const MlpDesc& d, float* P) {
  for (int l = 0; l < d.L; ++l) {
    const int n = d.dims[l + 1] * d.dims[l];
    for (int i = threadIdx.x; i < n; i += MLP_THREADS) P[d.woff[l] + i] = d.W[l][i];
    for (int i = threadIdx.x; i < d.dims[l + 1]; i += MLP_THREADS) P[d.boff[l] + i] = d.b[l] ? d.b[l][i] : 0.f;
  }
}

// Work items of a layer are (unit, sample) pairs, sample fastest: consecutive lanes read consecutive LDS words of one
// activation row and share (broadcast) the weight.  One barrier per layer.
// acts [B][act_width] fp32: the input of every layer (x, a_1, ..., a_{L-1});  out [B][dims[L]]
__global__ __launch_bounds__(MLP_THREADS) void mlp_fwd_kernel(const float* __restrict__ x, MlpDesc d,
                                                              float* __restrict__ acts, float* __restrict__ out, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* P = smem;                                           // parameters of all layers
  float* A = P + d.param_floats;                             // [MLP_MAX_WIDTH][MLP_PITCH] current input
  float* Z = A + MLP_MAX_WIDTH * MLP_PITCH;                  // next
  const int t = threadIdx.x;
  const int s0 = blockIdx.x * MLP_SAMPLES;
  const int ns = min(MLP_SAMPLES, B - s0);
  load_params(d, P);
  for (int it = t; it < MLP_SAMPLES * d.dims[0]; it += MLP_THREADS) {      // coalesced over the row-major input
    const int s = it / d.dims[0], i = it - s * d.dims[0];
    A[i * MLP_PITCH + s] = s < ns ? x[(size_t)(s0 + s) * d.dims[0] + i] : 0.f;
  }
  __syncthreads();
  for (int l = 0; l < d.L; ++l) {
    const int din = d.dims[l], dout = d.dims[l + 1];
    if (acts != nullptr) {
      for (int it = t; it < MLP_SAMPLES * din; it += MLP_THREADS) {
        const int s = it / din, i = it - s * din;
        if (s < ns) acts[(size_t)(s0 + s) * d.act_width + d.aoff[l] + i] = A[i * MLP_PITCH + s];
      }
    }
    const float* W = P + d.woff[l];
    const float* bb = P + d.boff[l];
    const bool relu = (l + 1 < d.L) || d.relu_last;
    for (int it = t; it < dout * MLP_SAMPLES; it += MLP_THREADS) {
      const int j = it / MLP_SAMPLES, s = it - j * MLP_SAMPLES;
      const float* w = W + j * din;
      float acc = bb[j];
#pragma unroll 4
      for (int i = 0; i < din; ++i) acc += w[i] * A[i * MLP_PITCH + s];
      Z[j * MLP_PITCH + s] = relu ? fmaxf(acc, 0.f) : acc;
    }
    __syncthreads();
    float* tmp = A; A = Z; Z = tmp;
  }
  const int dl = d.dims[d.L];
  for (int it = t; it < MLP_SAMPLES * dl; it += MLP_THREADS) {
    const int s = it / dl, j = it - s * dl;
    if (s < ns) out[(size_t)(s0 + s) * dl + j] = A[j * MLP_PITCH + s];
  }
}

// dout [B][dims[L]] gradient of the output;  out = the forward output (for the last ReLU, when relu_last)
__global__ __launch_bounds__(MLP_THREADS) void mlp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                              const float* __restrict__ acts, MlpDesc d,
                                                              float* __restrict__ dx, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* P = smem;
  float* D = P + d.param_floats;                             // [width][PITCH] delta of the current layer's output
  float* A = D + MLP_MAX_WIDTH * MLP_PITCH;                  // [width][PITCH] input of the current layer
  float* N = A + MLP_MAX_WIDTH * MLP_PITCH;                  // [width][PITCH] delta of the layer below
  const int t = threadIdx.x;
  const int s0 = blockIdx.x * MLP_SAMPLES;
  const int ns = min(MLP_SAMPLES, B - s0);
  load_params(d, P);
  {
    const int dl = d.dims[d.L];
    for (int it = t; it < MLP_SAMPLES * dl; it += MLP_THREADS) {
      const int s = it / dl, j = it - s * dl;
      float g = 0.f;
      if (s < ns) {
        g = dout[(size_t)(s0 + s) * dl + j];
        if (d.relu_last && out[(size_t)(s0 + s) * dl + j] <= 0.f) g = 0.f;
      }
      D[j * MLP_PITCH + s] = g;                               // samples beyond the batch carry zero gradient
    }
  }
  for (int l = d.L - 1; l >= 0; --l) {
    const int din = d.dims[l], dout_ = d.dims[l + 1];
    for (int it = t; it < MLP_SAMPLES * din; it += MLP_THREADS) {
      const int s = it / din, i = it - s * din;
      A[i * MLP_PITCH + s] = s < ns ? acts[(size_t)(s0 + s) * d.act_width + d.aoff[l] + i] : 0.f;
    }
    __syncthreads();                                         // D and A of every sample are in place
    // ---- parameter gradients: this block's samples reduced here, one atomic per weight and block
    if (d.dW[l] != nullptr) {
      for (int p = t; p < dout_ * din; p += MLP_THREADS) {
        const int j = p / din, i = p - j * din;
        const float* dj = D + j * MLP_PITCH;
        const float* ai = A + i * MLP_PITCH;
        float acc = 0.f;
#pragma unroll 8
        for (int b = 0; b < MLP_SAMPLES; ++b) acc += dj[b] * ai[b];
        atomicAdd(d.dW[l] + p, acc);
      }
    }
    if (d.db[l] != nullptr) {
      for (int j = t; j < dout_; j += MLP_THREADS) {
        const float* dj = D + j * MLP_PITCH;
        float acc = 0.f;
#pragma unroll 8
        for (int b = 0; b < MLP_SAMPLES; ++b) acc += dj[b];
        atomicAdd(d.db[l] + j, acc);
      }
    }
    // ---- delta of the layer below (its output is this layer's input a_l; ReLU behind every hidden layer)
    if (l > 0 || dx != nullptr) {
      const float* W = P + d.woff[l];
      for (int it = t; it < din * MLP_SAMPLES; it += MLP_THREADS) {
        const int i = it / MLP_SAMPLES, s = it - i * MLP_SAMPLES;
        float acc = 0.f;
#pragma unroll 4
        for (int j = 0; j < dout_; ++j) acc += W[j * din + i] * D[j * MLP_PITCH + s];
        if (l > 0 && A[i * MLP_PITCH + s] <= 0.f) acc = 0.f;
        N[i * MLP_PITCH + s] = acc;
      }
    }
    __syncthreads();                                         // everyone is done with D / A of this layer; N complete
    if (l == 0 && dx != nullptr) {
      for (int it = t; it < MLP_SAMPLES * din; it += MLP_THREADS) {
        const int s = it / din, i = it - s * din;
        if (s < ns) dx[(size_t)(s0 + s) * din + i] = N[i * MLP_PITCH + s];
      }
    }
    float* tmp = D; D = N; N = tmp;
  }
}

int fill_offsets(MlpDesc& d) {
  if (d.L < 1 || d.L > MLP_MAX_LAYERS) return -1;
  int off = 0, a = 0;
  for (int l = 0; l < d.L; ++l) {
    if (d.dims[l] < 1 || d.dims[l] > MLP_MAX_WIDTH || d.dims[l + 1] < 1 || d.dims[l + 1] > MLP_MAX_WIDTH) return -1;
    if (d.W[l] == nullptr) return -2;
    d.woff[l] = off; off += d.dims[l + 1] * d.dims[l];
    d.boff[l] = off; off += d.dims[l + 1];
    d.aoff[l] = a; a += d.dims[l];
  }
  d.param_floats = (off + 3) & ~3;
  d.act_width = a;
  return 0;
}

}  // namespace

extern "C" {

int ddl_mlp_max_width() { return MLP_MAX_WIDTH; }
int ddl_mlp_max_layers() { return MLP_MAX_LAYERS; }

// Wptr / bptr / dims: host arrays of L (L, L + 1) entries; device pointers inside.
int ddl_mlp_fwd(const float* x, const void* const* Wptr, const void* const* bptr, const int* dims, int L, int relu_last,
                float* acts, float* out, int B, void* stream) {
  if (B <= 0 || x == nullptr || out == nullptr || Wptr == nullptr || dims == nullptr) return -2;
  MlpDesc d = {};
  d.L = L; d.relu_last = relu_last;
  if (L < 1 || L > MLP_MAX_LAYERS) return -1;
  for (int l = 0; l <= L; ++l) d.dims[l] = dims[l];
  for (int l = 0; l < L; ++l) { d.W[l] = (const float*)Wptr[l]; d.b[l] = bptr ? (const float*)bptr[l] : nullptr; }
  const int rc = fill_offsets(d);
  if (rc != 0) return rc;
  const int smem = (d.param_floats + 2 * MLP_MAX_WIDTH * MLP_PITCH) * 4;
  static SmemAttrOnce once;
  const int ra = once.ensure(reinterpret_cast<const void*>(&mlp_fwd_kernel), 160 * 1024);
  if (ra != 0) return ra;
  if (smem > 160 * 1024) return -3;
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3((B + MLP_SAMPLES - 1) / MLP_SAMPLES), dim3(MLP_THREADS), smem, (hipStream_t)stream, x, d,
                     acts, out, B);
  return (int)hipGetLastError();
}

// dWptr / dbptr entries may be null (no gradient wanted); gradients are ACCUMULATED.  dx may be null.
int ddl_mlp_bwd(const float* dout, const float* out, const float* acts, const void* const* Wptr, void* const* dWptr,
                void* const* dbptr, const int* dims, int L, int relu_last, float* dx, int B, void* stream) {
  if (B <= 0 || dout == nullptr || acts == nullptr || Wptr == nullptr || dims == nullptr) return -2;
  if (relu_last && out == nullptr) return -2;
  MlpDesc d = {};
  d.L = L; d.relu_last = relu_last;
  if (L < 1 || L > MLP_MAX_LAYERS) return -1;
  for (int l = 0; l <= L; ++l) d.dims[l] = dims[l];
  for (int l = 0; l < L; ++l) {
    d.W[l] = (const float*)Wptr[l];
    d.dW[l] = dWptr ? (float*)dWptr[l] : nullptr;
    d.db[l] = dbptr ? (float*)dbptr[l] : nullptr;
  }
  const int rc = fill_offsets(d);
  if (rc != 0) return rc;
  const int smem = (d.param_floats + 3 * MLP_MAX_WIDTH * MLP_PITCH) * 4;
  static SmemAttrOnce once;
  const int ra = once.ensure(reinterpret_cast<const void*>(&mlp_bwd_kernel), 160 * 1024);
  if (ra != 0) return ra;
  if (smem > 160 * 1024) return -3;
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3((B + MLP_SAMPLES - 1) / MLP_SAMPLES), dim3(MLP_THREADS), smem, (hipStream_t)stream, dout,
                     out, acts, d, dx, B);
  return (int)hipGetLastError();
}

}  // extern "C"
