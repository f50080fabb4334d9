// Weight-gradient convolution for gfx950.
//
//   dw[k][r][s*C + c] = sum_{m=(n,p,q)} dy[m][k] * x[n, p*stride - pad + r, q*stride - pad + s, c]
//
// Per filter row r this is a GEMM  dY^T [K x M] * Xr [M x (S*C)]  whose reduction index m is the SLOW index of
// both operands in memory (NHWC).  Both tiles are therefore staged in their natural [pixel][channel] layout
// and the MFMA fragments are fetched with ds_read_b64_tr_b16 (the CDNA4 transposing LDS read): each 16-lane
// group pulls a 4-pixel x 16-channel block and receives it channel-major, i.e. 4 consecutive reduction
// elements of one channel per lane.  A and B use the identical read pattern, so the reduction order matches.
//
// Grid: (K tiles) x (S*C tiles) x R x split-M.  Each block walks its slice of M in 64-pixel steps
// (double-buffered LDS, register staging with zero fill for padding) and accumulates its fp32 tile into dw
// with global float atomics (dw must be zeroed by the caller); split-M is sized so the grid fills 256 CUs
// while the atomic traffic stays a small multiple of the weight size.
#include "common.h"
#include "host_util.h"
#include <stdlib.h>

struct WgradParams {
  const bf16_t* dy;   // [M][K]
  const bf16_t* x;    // [N][H][W][C]
  float* dw;          // [K][R][S*C]
  int N, H, W, C, K, P, Q, R, S, stride, pad;
  int M, SC;
  int tiles_k, tiles_n, splits, m_per_split;
  int plain;          // 1x1, stride 1, pad 0: x row index == m
  FastDiv div_pq, div_q, div_c;
};

// Byte offset of 16-byte chunk `ch` of row `row` inside a [rows][TW channels] bf16 tile.
// 256-byte rows (TW = 128) and 128-byte rows (TW = 64) need different XOR patterns to keep both the
// ds_write_b128 staging stores and the transposed 8-byte reads conflict-free.
template <int TW>
__device__ __forceinline__ int tr_off(int row, int ch) {
  if (TW == 128) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
  return row * 128 + ((ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2))) << 4);
}

template <int TK, int TN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int BKP = 64;                        // pixels per K-step
  constexpr int YCPR = TK / 8, XCPR = TN / 8;    // 16-byte chunks per tile row
  constexpr int YRPP = 256 / YCPR, XRPP = 256 / XCPR;
  constexpr int YI = BKP / YRPP, XI = BKP / XRPP;
  constexpr int YBYTES = BKP * TK * 2, XBYTES = BKP * TN * 2;
  constexpr int WK = TK / 2, WN = TN / 2;        // per-wave sub-tile
  constexpr int FK = WK / 16, FN = WN / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Ys = smem;
  char* const Xs = smem + 2 * YBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid & 1, wn = wid >> 1;

  // block -> (split, r, k tile, n tile), tiles fastest after the XCD remap: the tiles of one M-slice run
  // together on one XCD, so the dy / x rows they all stream are fetched from HBM once and re-read from that L2.
  unsigned b = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = b % p.tiles_n; b /= p.tiles_n;
  const int tk = b % p.tiles_k; b /= p.tiles_k;
  const int r = b % p.R; b /= p.R;
  const int split = (int)b;
  const int k0 = tk * TK, nn0 = tn * TN;

  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // ---- staging assignments ----
  const int ych = tid % YCPR, yrow0 = tid / YCPR;
  const int xch = tid % XCPR, xrow0 = tid / XCPR;
  const bool ycol_ok = (k0 + ych * 8) < p.K;
  const int nn = nn0 + xch * 8;                 // this thread's column inside the S*C axis
  const bool xcol_ok = nn < p.SC;
  const int xs = (int)fastdiv((unsigned)nn, p.div_c);   // filter column s
  const int xc = nn - xs * p.C;                           // channel

  u32x4 yr[YI], xr[XI];

  auto load_tile = [&](int mbase) {
#pragma unroll
    for (int i = 0; i < YI; ++i) {
      const int m = mbase + yrow0 + i * YRPP;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ycol_ok && m < m_end) v = *reinterpret_cast<const u32x4*>(p.dy + (size_t)m * (size_t)p.K + (size_t)(k0 + ych * 8));
      yr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int m = mbase + xrow0 + i * XRPP;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xcol_ok && m < m_end) {
        if (p.plain) {
          v = *reinterpret_cast<const u32x4*>(p.x + (size_t)m * (size_t)p.C + (size_t)xc);
        } else {
          const unsigned n = fastdiv((unsigned)m, p.div_pq);
          const unsigned rem = (unsigned)m - n * p.div_pq.d;
          const unsigned pp = fastdiv(rem, p.div_q);
          const unsigned qq = rem - pp * p.div_q.d;
          const int ih = (int)pp * p.stride - p.pad + r;
          const int iw = (int)qq * p.stride - p.pad + xs;
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
            const size_t off = ((size_t)((int)n * p.H * p.W + ih * p.W + iw)) * (size_t)p.C + (size_t)xc;
            v = *reinterpret_cast<const u32x4*>(p.x + off);
          }
        }
      }
      xr[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < YI; ++i)
      *reinterpret_cast<u32x4*>(Ys + buf * YBYTES + tr_off<TK>(yrow0 + i * YRPP, ych)) = yr[i];
#pragma unroll
    for (int i = 0; i < XI; ++i)
      *reinterpret_cast<u32x4*>(Xs + buf * XBYTES + tr_off<TN>(xrow0 + i * XRPP, xch)) = xr[i];
  };

  f32x4 acc[FK][FN];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane geometry: group g = lane>>4 owns reduction rows 8g..8g+7 of each 32-pixel sub-step;
  // inside the group lane 4q+pp addresses row q, 8 bytes at channel 4*pp of the 16-channel block.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp4 = lane & 3;

  const int nsteps = (m_end - m_begin + BKP - 1) / BKP;
  if (nsteps > 0) {
    load_tile(m_begin);
    store_tile(0);
  }
  __syncthreads();

  for (int t = 0; t < nsteps; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1 < nsteps);
    if (more) load_tile(m_begin + (t + 1) * BKP);
    const char* yb = Ys + buf * YBYTES;
    const char* xb = Xs + buf * XBYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[FK], bfr[FN];
      const int rbase = ks * 32 + 8 * g + q4;
#pragma unroll
      for (int i = 0; i < FK; ++i) {
        const int c0 = (wk * WK + i * 16) >> 3;   // first 16-byte chunk of the 16-channel block
        const int ch = c0 + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(yb + tr_off<TK>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        a[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c0 = (wn * WN + j * 16) >> 3;
        const int ch = c0 + (pp4 >> 1);
        const int sub = (pp4 & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xb + tr_off<TN>(rbase, ch) + sub));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(xb + tr_off<TN>(rbase + 4, ch) + sub));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < FK; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: D[row = k channel][col = (s,c)] -> fp32 tile in LDS -> row-contiguous float atomics ----
  // (each wave-instruction adds 256 contiguous bytes of one dw row: the shape the memory-side atomic units like)
  if (nsteps > 0) {
    float* const Ts = reinterpret_cast<float*>(smem);          // [TK][TN] fp32, reuses the staging buffers
#pragma unroll
    for (int i = 0; i < FK; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn * WN + j * 16 + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wk * WK + i * 16 + (lane >> 4) * 4 + e;
          Ts[row * TN + col] = acc[i][j][e];
        }
      }
    __syncthreads();
    constexpr int LPR = TN / 64;                                // 64-lane pieces per row
    for (int it = wid; it < TK * LPR; it += 4) {
      const int row = it / LPR, col = (it - row * LPR) * 64 + lane;
      const int k = k0 + row, cn = nn0 + col;
      if (k < p.K && cn < p.SC)
        atomicAdd(p.dw + ((size_t)k * (size_t)p.R + (size_t)r) * (size_t)p.SC + (size_t)cn, Ts[row * TN + col]);
    }
  }
}

template <int TK, int TN>
static int launch_wgrad(WgradParams& p, hipStream_t stream) {
  p.tiles_k = (p.K + TK - 1) / TK;
  p.tiles_n = (p.SC + TN - 1) / TN;
  const long long tiles = (long long)p.tiles_k * p.tiles_n * p.R;
  // One full round of resident blocks (2 per CU x 256 CUs), never more: a grid of 1.1 rounds runs as long as 2,
  // and every extra split adds a tile of fp32 atomic traffic (~1.3 TB/s chip-wide).  Splits are never shorter
  // than 256 pixels (4 K-steps).
  const int smem = 2 * 64 * (TK + TN) * 2;
  static const int cap_env = [] { const char* e = getenv("DDL_WGRAD_BLOCKS"); return e ? atoi(e) : 0; }();
  const int per_cu = (160 * 1024) / smem < 4 ? (160 * 1024) / smem : 4;      // LDS- and register-limited residency
  const long long capacity = cap_env > 0 ? cap_env : 256 * per_cu;
  long long splits = capacity / tiles;
  const long long max_splits = (p.M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  long long mps = (p.M + splits - 1) / splits;
  mps = ((mps + 63) / 64) * 64;
  splits = (p.M + mps - 1) / mps;
  p.splits = (int)splits;
  p.m_per_split = (int)mps;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_wgrad_kernel<TK, TN>), smem);
  if (rc != 0) return rc;
  const unsigned grid = (unsigned)(tiles * splits);
  hipLaunchKernelGGL((conv_wgrad_kernel<TK, TN>), dim3(grid), dim3(256), smem, stream, p);
  return (int)hipGetLastError();
}

// dw (fp32, [K][R][S][C]) must be zero on entry: the kernel accumulates with atomics.
extern "C" int ddl_conv_wgrad(const void* dy, const void* x, float* dw, int N, int H, int W, int C, int K, int P,
                              int Q, int R, int S, int stride, int pad, void* stream) {
  if (C % 8 != 0 || K % 8 != 0) return -1;
  const long long M = (long long)N * P * Q;
  if (M >= (1ll << 31) || (long long)N * H * W >= (1ll << 31) || M <= 0) return -3;
  WgradParams p;
  p.dy = (const bf16_t*)dy; p.x = (const bf16_t*)x; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.P = P; p.Q = Q; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.M = (int)M; p.SC = S * C;
  p.plain = (R == 1 && S == 1 && stride == 1 && pad == 0 && P == H && Q == W) ? 1 : 0;
  p.div_pq = make_fastdiv((unsigned)(P * Q));
  p.div_q = make_fastdiv((unsigned)Q);
  p.div_c = make_fastdiv((unsigned)C);
  hipStream_t st = (hipStream_t)stream;
  const bool k64 = K <= 64, n64 = p.SC <= 64 || (p.SC % 128 != 0 && p.SC % 64 == 0 && p.SC < 512);
  if (k64 && n64) return launch_wgrad<64, 64>(p, st);
  if (k64) return launch_wgrad<64, 128>(p, st);
  if (n64) return launch_wgrad<128, 64>(p, st);
  return launch_wgrad<128, 128>(p, st);
}
