// Fused LSTM layer for gfx950: the WHOLE time loop of one layer runs inside one kernel.
//
// One workgroup owns 16 batch rows for all T steps.  Wave w owns hidden units [16w, 16w+16): its slices of W_ih and
// W_hh (all four gates) live in registers as MFMA A-operands for the whole kernel, h_{t-1} is the B operand read from
// a double-buffered 4 KB LDS tile, the cell state never leaves registers.  Per step: (I+H)/32 x 4 MFMAs
// (v_mfma_f32_16x16x32_bf16) per wave, the gate non-linearities on the accumulator fragments (a lane holds
// i, f, g, o of the same (unit, batch) pair), ONE barrier.  Nothing but x_t is read from memory inside the loop.
//
// The backward kernel runs the recurrence in reverse: gate gradients are formed in registers, exchanged through LDS,
// and dh_{t-1} = W_hh^T dG_t is again an MFMA product with the (transposed) weights resident in registers.  It
// writes dG [B][T][4H]; the three non-recurrent products (dW_ih, dW_hh, dx) are plain GEMMs over all B*T rows and
// are left to the BLAS library.
//
// Replaces torch.nn.LSTM of the reference's LSTM workload (/root/reference/src/pytorch/LSTM/model.py:70-94:
// 64 steps, 32 features, 128 hidden units).  Gate order i, f, g, o as in PyTorch.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 2.f * __fdividef(1.f, 1.f + __expf(-2.f * x)) - 1.f; }

__device__ __forceinline__ u32x2 pack4(const float* v) {
  u32x2 r;
  r[0] = pack_bf16x2(v[0], v[1]);
  r[1] = pack_bf16x2(v[2], v[3]);
  return r;
}

__device__ __forceinline__ void unpack4(const u32x2& v, float* f) {
  f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
  f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
}

// x      [B][T][I]   bf16, strides x_sb / x_st (elements), innermost contiguous
// w_ih   [4H][I]     bf16          w_hh [4H][H] bf16          bias [4H] fp32 (b_ih + b_hh)
// hseq   [B][T+1][H] bf16, row 0 holds h_0 (read), rows 1..T are written
// cseq   [B][T+1][H] fp32, row 0 holds c_0 (read), rows 1..T are written when save != 0
// gates  [B][T][4][H] bf16 post-activation i, f, g, o (written when save != 0)
template <int H, int I>
__global__ __launch_bounds__(H / 16 * 64) void lstm_fwd_kernel(
    const bf16_t* __restrict__ x, long long x_sb, long long x_st, const bf16_t* __restrict__ w_ih,
    const bf16_t* __restrict__ w_hh, const float* __restrict__ bias, bf16_t* __restrict__ hseq,
    float* __restrict__ cseq, bf16_t* __restrict__ gates, int B, int T, int save) {
  constexpr int NT = H / 16 * 64;
  constexpr int KH = H / 32, KI = I / 32;
  // h tile: 16 rows (batch) x H bf16, no padding; 16-byte chunks XOR-swizzled by the row so that the ds_read_b128
  // fragment reads (row = lane & 15, chunk = 4*ks + lane/16 — the conv kernels' pattern) are bank-conflict free:
  // 256-byte rows: chunk ^ row; 128-byte rows: chunk ^ ((row >> 1) & 7).  (PMC before: 50 % conflict cycles.)
  constexpr int ROWB = H * 2;
  __shared__ __attribute__((aligned(16))) char hbuf[2 * 16 * ROWB];
  auto hoff = [](int buf, int row, int chunk) {
    const int sw = (ROWB >= 256) ? row : ((row >> 1) & 7);
    return (buf * 16 + row) * ROWB + ((chunk ^ sw) << 4);
  };

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, q4 = lane >> 4;
  const int b0 = blockIdx.x * 16;
  const int b = b0 + n;
  const bool live = b < B;
  const size_t bl = (size_t)(live ? b : B - 1);     // clamped row for loads
  const int u0 = 16 * w + 4 * q4;                   // first of this lane's 4 hidden units

  bf16x8 a_ih[4][KI], a_hh[4][KH];
  float bv[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const size_t row = (size_t)g * H + 16 * w + n;
#pragma unroll
    for (int ks = 0; ks < KI; ++ks)
      a_ih[g][ks] = *reinterpret_cast<const bf16x8*>(w_ih + row * I + ks * 32 + 8 * q4);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
      a_hh[g][ks] = *reinterpret_cast<const bf16x8*>(w_hh + row * H + ks * 32 + 8 * q4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + g * H + u0);
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[g][i] = bb[i];
  }

  for (int idx = tid; idx < 16 * (H / 8); idx += NT) {
    const int r = idx / (H / 8), ch = idx % (H / 8);
    const size_t br = (size_t)((b0 + r < B) ? b0 + r : B - 1);
    *reinterpret_cast<u32x4*>(hbuf + hoff(0, r, ch)) =
        *reinterpret_cast<const u32x4*>(hseq + br * (size_t)(T + 1) * H + ch * 8);
  }
  float c[4];
  {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cseq + bl * (size_t)(T + 1) * H + u0);
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = c0[i];
  }
  const bf16_t* xrow = x + bl * x_sb + 8 * q4;
  bf16x8 xc[KI];
#pragma unroll
  for (int ks = 0; ks < KI; ++ks) xc[ks] = *reinterpret_cast<const bf16x8*>(xrow + ks * 32);
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < T; ++t) {
    bf16x8 xn[KI];
    const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
    for (int ks = 0; ks < KI; ++ks) xn[ks] = *reinterpret_cast<const bf16x8*>(xrow + tn * x_st + ks * 32);

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{bv[g][0], bv[g][1], bv[g][2], bv[g][3]};
#pragma unroll
    for (int ks = 0; ks < KI; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ih[g][ks], xc[ks], acc[g], 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hbuf + hoff(cur, n, ks * 4 + q4));
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hh[g][ks], hf, acc[g], 0, 0, 0);
    }

    float gi[4], gf[4], gg[4], go[4], h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gi[i] = sigmoid_f(acc[0][i]);
      gf[i] = sigmoid_f(acc[1][i]);
      gg[i] = tanh_f(acc[2][i]);
      go[i] = sigmoid_f(acc[3][i]);
      c[i] = gf[i] * c[i] + gi[i] * gg[i];
      h[i] = go[i] * tanh_f(c[i]);
    }
    const u32x2 hp = pack4(h);
    *reinterpret_cast<u32x2*>(hbuf + hoff(cur ^ 1, n, 2 * w + (q4 >> 1)) + (q4 & 1) * 8) = hp;
    if (live) {
      *reinterpret_cast<u32x2*>(hseq + ((size_t)b * (T + 1) + t + 1) * H + u0) = hp;
      if (save) {
        bf16_t* gp = gates + ((size_t)b * T + t) * 4 * H + u0;
        *reinterpret_cast<u32x2*>(gp) = pack4(gi);
        *reinterpret_cast<u32x2*>(gp + H) = pack4(gf);
        *reinterpret_cast<u32x2*>(gp + 2 * H) = pack4(gg);
        *reinterpret_cast<u32x2*>(gp + 3 * H) = pack4(go);
        *reinterpret_cast<f32x4*>(cseq + ((size_t)b * (T + 1) + t + 1) * H + u0) = f32x4{c[0], c[1], c[2], c[3]};
      }
    }
#pragma unroll
    for (int ks = 0; ks < KI; ++ks) xc[ks] = xn[ks];
    __syncthreads();
    cur ^= 1;
  }
}

// dout   [B][T][H]    bf16 gradient of the output sequence (strides d_sb / d_st)
// gates  [B][T][4][H] bf16, cseq [B][T+1][H] fp32 from the forward pass
// w_hhT  [H][4H]      bf16 (W_hh transposed)
// dG     [B][T][4H]   bf16 pre-activation gate gradients (written)
// dh0/dc0 [B][H]      fp32 gradients of the initial state (written when non-null)
template <int H>
__global__ __launch_bounds__(H / 16 * 64) void lstm_bwd_kernel(
    const bf16_t* __restrict__ dout, long long d_sb, long long d_st, const bf16_t* __restrict__ gates,
    const float* __restrict__ cseq, const bf16_t* __restrict__ w_hhT, bf16_t* __restrict__ dG,
    float* __restrict__ dh0, float* __restrict__ dc0, int B, int T) {
  constexpr int G4 = 4 * H;
  constexpr int KS = G4 / 32;
  // gate-gradient tile: 16 rows x 4H bf16 (rows are multiples of 256 bytes), chunk ^ row swizzle as in the forward
  constexpr int ROWB = G4 * 2;
  __shared__ __attribute__((aligned(16))) char dgbuf[2 * 16 * ROWB];
  auto goff = [](int buf, int row, int chunk) { return (buf * 16 + row) * ROWB + ((chunk ^ row) << 4); };

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, q4 = lane >> 4;
  const int b = blockIdx.x * 16 + n;
  const bool live = b < B;
  const size_t bl = (size_t)(live ? b : B - 1);
  const int u0 = 16 * w + 4 * q4;

  bf16x8 a[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    a[ks] = *reinterpret_cast<const bf16x8*>(w_hhT + (size_t)(16 * w + n) * G4 + ks * 32 + 8 * q4);

  const bf16_t* gbase = gates + bl * (size_t)T * G4 + u0;
  const float* cbase = cseq + bl * (size_t)(T + 1) * H + u0;
  const bf16_t* dbase = dout + bl * d_sb + u0;

  struct Step { u32x2 g[4]; u32x2 d; f32x4 c, cp; };
  auto fetch = [&](int t) {
    Step s;
    const bf16_t* gp = gbase + (size_t)t * G4;
#pragma unroll
    for (int k = 0; k < 4; ++k) s.g[k] = *reinterpret_cast<const u32x2*>(gp + k * H);
    s.d = *reinterpret_cast<const u32x2*>(dbase + t * d_st);
    s.cp = *reinterpret_cast<const f32x4*>(cbase + (size_t)t * H);   // c_{t-1}; c_t is the previous step's cp
    return s;
  };

  float dh[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
  Step s = fetch(T - 1);
  s.c = *reinterpret_cast<const f32x4*>(cbase + (size_t)T * H);
  int cur = 0;
  for (int t = T - 1; t >= 0; --t) {
    float gi[4], gf[4], gg[4], go[4], dy[4];
    unpack4(s.g[0], gi); unpack4(s.g[1], gf); unpack4(s.g[2], gg); unpack4(s.g[3], go); unpack4(s.d, dy);
    float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dht = dy[i] + dh[i];
      const float tc = tanh_f(s.c[i]);
      const float dct = dc[i] + dht * go[i] * (1.f - tc * tc);
      dao[i] = dht * tc * go[i] * (1.f - go[i]);
      dai[i] = dct * gg[i] * gi[i] * (1.f - gi[i]);
      dag[i] = dct * gi[i] * (1.f - gg[i] * gg[i]);
      daf[i] = dct * s.cp[i] * gf[i] * (1.f - gf[i]);
      dc[i] = dct * gf[i];
    }
    const u32x2 pi = pack4(dai), pf = pack4(daf), pg = pack4(dag), po = pack4(dao);
    {
      constexpr int GC = H / 8;                       // chunks per gate
      const int c0 = 2 * w + (q4 >> 1), bo = (q4 & 1) * 8;
      *reinterpret_cast<u32x2*>(dgbuf + goff(cur, n, c0) + bo) = pi;
      *reinterpret_cast<u32x2*>(dgbuf + goff(cur, n, GC + c0) + bo) = pf;
      *reinterpret_cast<u32x2*>(dgbuf + goff(cur, n, 2 * GC + c0) + bo) = pg;
      *reinterpret_cast<u32x2*>(dgbuf + goff(cur, n, 3 * GC + c0) + bo) = po;
    }
    if (live) {
      bf16_t* gp = dG + ((size_t)b * T + t) * G4 + u0;
      *reinterpret_cast<u32x2*>(gp) = pi;
      *reinterpret_cast<u32x2*>(gp + H) = pf;
      *reinterpret_cast<u32x2*>(gp + 2 * H) = pg;
      *reinterpret_cast<u32x2*>(gp + 3 * H) = po;
    }
    if (t > 0) {                                    // every cell state is read once: this step's c_{t-1} is the next c_t
      const f32x4 c_next = s.cp;
      s = fetch(t - 1);
      s.c = c_next;
    }                    // next step's operands fly during the MFMA phase
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(dgbuf + goff(cur, n, ks * 4 + q4));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], bf, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dh[i] = acc[i];
    cur ^= 1;
  }
  if (live) {
    if (dh0 != nullptr) *reinterpret_cast<f32x4*>(dh0 + (size_t)b * H + u0) = f32x4{dh[0], dh[1], dh[2], dh[3]};
    if (dc0 != nullptr) *reinterpret_cast<f32x4*>(dc0 + (size_t)b * H + u0) = f32x4{dc[0], dc[1], dc[2], dc[3]};
  }
}

// ---------------------------------------------------------------- input gradient: dx[M][N] = dG[M][K] @ W_ih[K][N]
// A row GEMM with a huge M (= B*T), a short reduction (K = 4H <= 512) and a narrow output (N = I <= 128).  Operands
// in the MFMA's natural form: both with the reduction index contiguous - dG as it is, the weight as its [N][K]
// transpose (the arena's twin).  A wave owns 32 rows and ALL N columns: the row fragments come straight from global
// memory, once; the weight (<= 128 KiB) is re-read by every wave and stays in L2.  The weight is the MFMA's A operand
// and the rows its B operand, so a lane ends with 4 consecutive COLUMNS of one row: one 8-byte store.
template <int NT>
__global__ __launch_bounds__(256) void rows_gemm_nt_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ bt,
                                                           bf16_t* __restrict__ out, int M, int K) {
  constexpr int N = NT * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  if (m0 >= M) return;
  const int r16 = lane & 15, q4 = lane >> 4;
  const int row0 = min(m0 + r16, M - 1), row1 = min(m0 + 16 + r16, M - 1);
  const bf16_t* a0 = a + (size_t)row0 * K + q4 * 8;
  const bf16_t* a1 = a + (size_t)row1 * K + q4 * 8;
  const bf16_t* bp = bt + (size_t)r16 * K + q4 * 8;
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[0][i] = acc[1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int k = 0; k < K; k += 32) {
    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(a0 + k);
    const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(a1 + k);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(bp + (size_t)i * 16 * K + k);
      acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x0, acc[0][i], 0, 0, 0);
      acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x1, acc[1][i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = m0 + h * 16 + r16;
    if (row >= M) continue;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float f[4] = {acc[h][i][0], acc[h][i][1], acc[h][i][2], acc[h][i][3]};
      *reinterpret_cast<u32x2*>(out + (size_t)row * N + i * 16 + q4 * 4) = pack4(f);
    }
  }
}

template <int H, int I>
int launch_fwd(const void* x, long long x_sb, long long x_st, const void* w_ih, const void* w_hh, const void* bias,
               void* hseq, void* cseq, void* gates, int B, int T, int save, hipStream_t st) {
  const dim3 grid((B + 15) / 16), block(H / 16 * 64);
  hipLaunchKernelGGL((lstm_fwd_kernel<H, I>), grid, block, 0, st, (const bf16_t*)x, x_sb, x_st, (const bf16_t*)w_ih,
                     (const bf16_t*)w_hh, (const float*)bias, (bf16_t*)hseq, (float*)cseq, (bf16_t*)gates, B, T, save);
  return (int)hipGetLastError();
}

template <int H>
int launch_bwd(const void* dout, long long d_sb, long long d_st, const void* gates, const void* cseq,
               const void* w_hhT, void* dG, void* dh0, void* dc0, int B, int T, hipStream_t st) {
  const dim3 grid((B + 15) / 16), block(H / 16 * 64);
  hipLaunchKernelGGL((lstm_bwd_kernel<H>), grid, block, 0, st, (const bf16_t*)dout, d_sb, d_st, (const bf16_t*)gates,
                     (const float*)cseq, (const bf16_t*)w_hhT, (bf16_t*)dG, (float*)dh0, (float*)dc0, B, T);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// 1 when (hidden, input) has a compiled instance.
int ddl_lstm_supported(int H, int I) {
  return ((H == 64 || H == 128) && (I == 32 || I == 64 || I == 128)) ? 1 : 0;
}

int ddl_lstm_fwd(const void* x, long long x_sb, long long x_st, const void* w_ih, const void* w_hh, const void* bias,
                 void* hseq, void* cseq, void* gates, int B, int T, int I, int H, int save, void* stream) {
  if (B <= 0 || T <= 0 || !ddl_lstm_supported(H, I)) return -1;
  if (x == nullptr || w_ih == nullptr || w_hh == nullptr || bias == nullptr || hseq == nullptr || cseq == nullptr)
    return -2;
  if (save && gates == nullptr) return -2;
  if ((x_sb % 8) != 0 || (x_st % 8) != 0 || (((uintptr_t)x) & 15) != 0) return -3;   // 16-byte operand loads
  hipStream_t st = (hipStream_t)stream;
#define DDL_LSTM_FWD(HH, II) \
  if (H == HH && I == II) return launch_fwd<HH, II>(x, x_sb, x_st, w_ih, w_hh, bias, hseq, cseq, gates, B, T, save, st);
  DDL_LSTM_FWD(128, 32) DDL_LSTM_FWD(128, 64) DDL_LSTM_FWD(128, 128)
  DDL_LSTM_FWD(64, 32) DDL_LSTM_FWD(64, 64) DDL_LSTM_FWD(64, 128)
#undef DDL_LSTM_FWD
  return -1;
}

int ddl_lstm_bwd(const void* dout, long long d_sb, long long d_st, const void* gates, const void* cseq,
                 const void* w_hhT, void* dG, void* dh0, void* dc0, int B, int T, int H, void* stream) {
  if (B <= 0 || T <= 0 || !(H == 64 || H == 128)) return -1;
  if (dout == nullptr || gates == nullptr || cseq == nullptr || w_hhT == nullptr || dG == nullptr) return -2;
  if ((d_sb % 4) != 0 || (d_st % 4) != 0 || (((uintptr_t)dout) & 7) != 0) return -3;  // 8-byte gradient loads
  hipStream_t st = (hipStream_t)stream;
  if (H == 128) return launch_bwd<128>(dout, d_sb, d_st, gates, cseq, w_hhT, dG, dh0, dc0, B, T, st);
  return launch_bwd<64>(dout, d_sb, d_st, gates, cseq, w_hhT, dG, dh0, dc0, B, T, st);
}

// out[M][N] (bf16) = a[M][K] @ bt[N][K]^T; K a multiple of 32, N one of 32 / 64 / 128 (the LSTM input widths).
int ddl_rows_gemm_nt(const void* a, const void* bt, void* out, int M, int K, int N, void* stream) {
  if (M <= 0 || K <= 0 || (K % 32) != 0 || !(N == 32 || N == 64 || N == 128)) return -1;
  if (a == nullptr || bt == nullptr || out == nullptr) return -2;
  if (((((uintptr_t)a) | ((uintptr_t)bt)) & 15) != 0 || (((uintptr_t)out) & 7) != 0) return -3;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((M + 127) / 128), block(256);
  if (N == 32)
    hipLaunchKernelGGL((rows_gemm_nt_kernel<2>), grid, block, 0, st, (const bf16_t*)a, (const bf16_t*)bt, (bf16_t*)out, M, K);
  else if (N == 64)
    hipLaunchKernelGGL((rows_gemm_nt_kernel<4>), grid, block, 0, st, (const bf16_t*)a, (const bf16_t*)bt, (bf16_t*)out, M, K);
  else
    hipLaunchKernelGGL((rows_gemm_nt_kernel<8>), grid, block, 0, st, (const bf16_t*)a, (const bf16_t*)bt, (bf16_t*)out, M, K);
  return (int)hipGetLastError();
}

}  // extern "C"
