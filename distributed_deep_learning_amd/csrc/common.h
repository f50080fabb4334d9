// Shared device helpers for the gfx950 (MI355X / CDNA4) kernel library.
// Everything here assumes wave64 and bf16 storage with fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;     // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_t;                                 // raw bf16 bits in global memory

#define DDL_WAVE 64
// Reduction slabs ([DDL_BN_SLOTS][2][C] fp64) spread same-address atomics over this many copies.  The sums are kept
// in DOUBLE: BatchNorm's variance is E[x^2] - E[x]^2, which cancels catastrophically in fp32 once |mean| >> std;
// producers accumulate fp32 partials about a per-tile pivot, convert them to plain sums in fp64 (exact to ~1e-16) and
// add those (global_atomic_add_f64), the finalize kernels subtract in fp64.  The C ABI still passes `float*`.
#define DDL_BN_SLOTS 64
typedef double slab_t;
__device__ __forceinline__ void slab_add(float* slab, size_t idx, double v) {
  atomicAdd(reinterpret_cast<double*>(slab) + idx, v);
}
// fp32 sums about a pivot k over n elements -> plain sums:  sum x = s1 + n k ;  sum x^2 = s2 + 2 k s1 + n k^2
__device__ __forceinline__ void unshift_sums(float s1, float s2, float k, double n, double& sx, double& sxx) {
  const double kd = (double)k, a = (double)s1;
  sx = a + n * kd;
  sxx = (double)s2 + 2.0 * kd * a + n * kd * kd;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned int)v) << 16);
}

// Round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// Two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved).
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned int, b);
}

__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
  f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
  f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
  f[4] = __uint_as_float(v[2] << 16); f[5] = __uint_as_float(v[2] & 0xffff0000u);
  f[6] = __uint_as_float(v[3] << 16); f[7] = __uint_as_float(v[3] & 0xffff0000u);
}

__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
  v[0] = pack_bf16x2(f[0], f[1]); v[1] = pack_bf16x2(f[2], f[3]);
  v[2] = pack_bf16x2(f[4], f[5]); v[3] = pack_bf16x2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact unsigned division by a runtime constant for n < 2^31:
//   q = umulhi(n, magic) >> shift   with  magic = ceil(2^(31+s)/d), shift = s-1, s = ceil(log2 d).
// d == 1 is encoded as magic = 0 (identity).  Host side: make_fastdiv() in host_util.h.
struct FastDiv {
  unsigned int magic;
  unsigned int shift;
  unsigned int d;
};

__device__ __forceinline__ unsigned int fastdiv(unsigned int n, const FastDiv& f) {
  return f.magic == 0u ? n : (__umulhi(n, f.magic) >> f.shift);
}

// XCD-aware block remap (8 XCDs, each with a private 4 MiB L2): blocks b and b+8 share an XCD
// under round-robin dispatch, so give each XCD a contiguous run of logical tile ids.  Bijective
// for every grid size.  Speed only, never correctness.
__device__ __forceinline__ unsigned int xcd_remap(unsigned int bid, unsigned int nwg) {
  const unsigned int q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, idx = bid >> 3;
  const unsigned int base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + idx;
}
