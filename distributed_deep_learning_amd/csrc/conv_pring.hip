// Persistent variant of the LDS-DMA ring convolution (conv_ring.hip) for the layers whose time is the SUM of their
// K loop and their epilogue (short reductions, wide outputs: ResNet's 1x1 expansions and the conv1 dgrads with the
// residual addend).  Ablation of the ring kernel (profiles/r2_ring_kernel_ablation_b512.txt): 256->1024 @14x14 takes
// 0.107 ms, 0.061 without its epilogue and 0.076 with the epilogue alone - the epilogue runs at HBM write speed but a
// block cannot start its next tile before its stores have drained, and the chip alternates between "everybody
// multiplies" and "everybody stores".
//
// What changes:
//   * 512 blocks (two per CU) walk the tile list; the load cursor runs two K-steps ahead of the MFMA cursor ACROSS tile
//     boundaries, so the first two stages of tile T+1 are in flight while tile T is in its epilogue;
//   * the epilogue needs no block-wide transposition: every wave turns its own 64 x 64 accumulator block into full
//     128-byte row segments through a private 2 KiB window of the one ring stage that is free at that moment (the stage
//     of the last K-step; it is refilled only after the next K-step's barrier), 16 pixel rows per round;
//   * output stores are buffer stores whose out-of-range lanes are dropped by the descriptor's range check, so every
//     wave issues the SAME number of them (8) and the counted vmcnt of the next tile's first two K-steps can leave them
//     in flight: stores drain under the next tile's MFMA loop instead of in front of it;
//   * forward BatchNorm statistics and the masked residual addend of the dgrads are taken on the row segments, as in
//     conv_epilogue.h (pivot-shifted fp32 partials, fp64 slab atomics).
//
// Same math, operand roles, staging geometry and LDS swizzle as conv_ring.hip at BN = 128, BK = 32.
#include "common.h"
#include "host_util.h"
#include "conv_params.h"
#include <cstdlib>

namespace {

constexpr int PBM = 256, PBN = 128, PBK = 32, PNS = 3;
constexpr int PRP = PBK * 2;                    // LDS row pitch (bytes)
constexpr int PRW = 1024 / PRP;                 // rows per LDS-DMA wave-instruction (16)
constexpr int PXJ = PBM / (8 * PRW);            // X DMA instructions per thread per stage (2)
constexpr int PWJ = PBN / PRW / 8;              // W DMA instructions per thread per stage (1)
constexpr int PXBYTES = PBM * PRP, PSTAGE = (PBM + PBN) * PRP;
constexpr int PSTORES = 8;                      // output store instructions per wave and tile
constexpr unsigned POOB = 0xFFFFFFF0u;
constexpr int PWP = 144;                        // window row pitch: 64 channels (128 B) + 16 B pad, so that the accumulator
                                                // fragment i and the row group h enter every address as IMMEDIATE offsets
constexpr int PWIN = 16 * PWP;                  // per-wave epilogue window: 16 pixel rows x 64 channels (bf16)

__device__ __forceinline__ void pdma16(const __amdgpu_buffer_rsrc_t rsrc, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void pwait() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// EPI bits: 2 = forward BatchNorm partial sums, 4 = addend (+ optional 1-bit mask) - as in conv_ring.hip
template <int MODE, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_pring_kernel(const ConvParams p, const int tiles_total) {
  constexpr int WN = PBN / 2;                   // 64 channels per wave
  constexpr int TM = 4, TN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 3, wn = wid >> 2;

  // ---- this block's run of logical tiles: XCD x owns [base, base + cnt); its gridDim/8 blocks sweep it side by side
  const unsigned per = gridDim.x >> 3, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned q = (unsigned)tiles_total >> 3, r8 = (unsigned)tiles_total & 7u;
  const unsigned cnt = q + (xcd < r8 ? 1u : 0u);
  const unsigned base = xcd < r8 ? xcd * (q + 1u) : r8 * (q + 1u) + (xcd - r8) * q;
  if (slot >= cnt) return;                      // (never with tiles_total >= gridDim.x)
  const unsigned my_tiles = (cnt - slot + per - 1u) / per;

  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wmat, 0, (int)p.w_bytes, 0x00020000);
  const unsigned out_bytes = (unsigned)((size_t)p.M * (size_t)p.out_ld * 2u);
  const __amdgpu_buffer_rsrc_t osrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.addend, 0, (EPI & 4) ? (int)((size_t)p.M * (size_t)p.OC * 2u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.addend_mask, 0, ((EPI & 4) && p.addend_mask != nullptr) ? (int)((size_t)p.M * (size_t)(p.OC >> 3)) : 0,
      0x00020000);

  // ---- staging geometry (conv_ring.hip, BK = 32)
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ (3 * ((lane >> 5) & 1));
  // IC >= 32 (launcher): a thread's first 8-channel chunk always lies in tap (0, 0)
  const int ksteps = (p.Kred + PBK - 1) / PBK;

  // ---- load cursor: (tile, K-step) whose stage is issued next; runs two K-steps ahead of the MFMA cursor
  unsigned ld_iter = 0;                         // index into this block's tile run
  int ld_step = 0, ld_stage = 0;
  bool ld_valid = true;
  int nb[PXJ], bh[PXJ], bw[PXJ];
  unsigned wrow_off;
  bool wrow_ok;
  int kr, ks_, kc;

  auto decode = [&](unsigned iter) {
    const unsigned lid = base + iter * per + slot;
    const int tile_m = (int)(lid / (unsigned)p.tiles_n), tile_n = (int)(lid - (unsigned)tile_m * (unsigned)p.tiles_n);
    const int m0 = tile_m * PBM, n0 = tile_n * PBN;
#pragma unroll
    for (int j = 0; j < PXJ; ++j) {
      const int m = m0 + (j * 8 + wid) * PRW + lrow;
      if (m < p.M) {
        const int n = (int)fastdiv((unsigned)m, p.div_ohw);
        const int rem = m - n * (int)p.div_ohw.d;
        const int oh = (int)fastdiv((unsigned)rem, p.div_ow);
        const int ow = rem - oh * p.OW;
        nb[j] = n * p.IH * p.IW;
        if (MODE == 0) { bh[j] = oh * p.stride - p.pad; bw[j] = ow * p.stride - p.pad; }
        else           { bh[j] = oh + p.pad; bw[j] = ow + p.pad; }          // o_step == 1 only (checked by the launcher)
      } else {
        nb[j] = 0; bh[j] = -(1 << 28); bw[j] = -(1 << 28);
      }
    }
    const int n = n0 + wid * PRW + lrow;
    wrow_ok = n < p.OC;
    wrow_off = (unsigned)n * (unsigned)p.KredFull;
    kr = 0; ks_ = 0; kc = lchunk * 8;
  };

  auto issue_next = [&]() {
    const int r = p.r0 + kr * p.t_step, sx = p.s0 + ks_ * p.t_step;
    const bool kok = kr < p.nR;
    char* const sbase = smem + ld_stage * PSTAGE;
#pragma unroll
    for (int j = 0; j < PXJ; ++j) {
      int ih, iw; bool ok;
      if (MODE == 0) {
        ih = bh[j] + r; iw = bw[j] + sx;
        ok = ((unsigned)ih < (unsigned)p.IH) && ((unsigned)iw < (unsigned)p.IW);
      } else {
        ih = bh[j] - r; iw = bw[j] - sx;                                      // stride 1 (checked by the launcher)
        ok = (ih >= 0) && (iw >= 0) && (ih < p.IH) && (iw < p.IW);
      }
      const unsigned off = ((unsigned)(nb[j] + ih * p.IW + iw) * (unsigned)p.IC + (unsigned)kc) * 2u;
      pdma16(xsrc, sbase + (j * 8 + wid) * 1024, (ok && kok) ? off : POOB);
    }
    const unsigned woff = (unsigned)((r * p.S + sx) * p.IC + kc);
    pdma16(wsrc, sbase + PXBYTES + wid * 1024, (wrow_ok && kok) ? (wrow_off + woff) * 2u : POOB);
    kc += PBK;
    while (kc >= p.IC) {
      kc -= p.IC;
      if (++ks_ == p.nS) { ks_ = 0; ++kr; }
    }
    ld_stage = (ld_stage == PNS - 1) ? 0 : ld_stage + 1;
    if (++ld_step == ksteps) {
      ld_step = 0;
      if (++ld_iter < my_tiles) decode(ld_iter); else ld_valid = false;
    }
  };

  decode(0);
  issue_next();
  issue_next();                                 // ksteps >= 2 (launcher): both belong to the first tile

  int stage = 0;
  double run1 = 0.0, run2 = 0.0;                // EPI & 2: running sums of channel `lane` of this wave's 64 (fp64)
  for (unsigned it = 0; it < my_tiles; ++it) {
    const unsigned lid = base + it * per + slot;
    const int tile_m = (int)(lid / (unsigned)p.tiles_n), tile_n = (int)(lid - (unsigned)tile_m * (unsigned)p.tiles_n);
    const int m0 = tile_m * PBM, n0 = tile_n * PBN;
    const bool last_tile = it + 1 == my_tiles;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < ksteps; ++t) {
      // The loads of (tile, t) have landed once only YOUNGER operations are outstanding: the next K-step's 3 loads, and -
      // during the first two K-steps of a later tile - the previous tile's 8 output stores (issued between them).
      if (last_tile && t + 1 == ksteps) pwait<0>();
      else if (it > 0 && t < 2)         pwait<PXJ + PWJ + PSTORES>();
      else                              pwait<PXJ + PWJ>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* xb = smem + stage * PSTAGE;
      const char* wb = xb + PXBYTES;
      bf16x8 a[TN], b[TM];
      const int ch = lane >> 4;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * WN + i * 16 + (lane & 15);
        a[i] = *reinterpret_cast<const bf16x8*>(wb + row * PRP + ((ch ^ (3 * ((row >> 3) & 1))) << 4));
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm * 64 + j * 16 + (lane & 15);
        b[j] = *reinterpret_cast<const bf16x8*>(xb + row * PRP + ((ch ^ (3 * ((row >> 3) & 1))) << 4));
      }
      if (ld_valid) issue_next();               // into the stage the previous K-step vacated
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      stage = (stage == PNS - 1) ? 0 : stage + 1;
    }
    // ---- epilogue, wave-private: 4 rounds of 16 pixel rows x 64 channels
    const int rr = lane >> 3, cc = lane & 7;                  // read-back: row rr (+8), 16-byte chunk cc
    const int col = n0 + wn * WN + cc * 8;
    const bool col_ok = col < p.OC;
    const int wrow0 = m0 + wm * 64;                           // first pixel row of this wave
    // addend rows (and their mask bytes) are requested TWO rounds ahead of their use: the first two rounds' loads go out
    // before the barrier below, so their latency hides behind it and behind the first LDS round trip
    u32x4 av[TM][2];
    unsigned mk[TM][2];
    auto issue_addend = [&](int j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = wrow0 + j * 16 + h * 8 + rr;
        const bool ok = m < p.M && col_ok;
        const unsigned aoff = ok ? ((unsigned)m * (unsigned)p.OC + (unsigned)col) * 2u : POOB;
        av[j][h] = __builtin_amdgcn_raw_buffer_load_b128(asrc, (int)aoff, 0, 0);
        const unsigned moff = ok ? (unsigned)m * (unsigned)(p.OC >> 3) + (unsigned)(col >> 3) : POOB;
        mk[j][h] = p.addend_mask != nullptr ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(msrc, (int)moff, 0, 0) : 0xffu;
      }
    };
    if (EPI & 4) { issue_addend(0); issue_addend(1); }

    // every wave has read the last stage (its fragments are in registers): that stage - the one the next refill will
    // target - becomes the epilogue windows.  Raw barrier: a __syncthreads() would drain the prefetched stages of the
    // next tile.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int free_stage = (stage == 0) ? PNS - 1 : stage - 1;
    char* const win = smem + free_stage * PSTAGE + wid * PWIN;

    float s1[8], s2[8], kf[8];
    if (EPI & 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; kf[e] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      {
        char* const wp = win + (lane & 15) * PWP + (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
          pk[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
          *reinterpret_cast<u32x2*>(wp + i * 32) = pk;
        }
      }
      if ((EPI & 4) && j + 2 < TM) issue_addend(j + 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private window: no barrier, only my own LDS writes
      u32x4 v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[h] = *reinterpret_cast<const u32x4*>(win + (h * 8 + rr) * PWP + (cc << 4));
      }
      if ((EPI & 2) && j == 0) {
        // pivot of the statistics: this wave's first pixel row (valid whenever the wave has any valid row)
        const u32x4 pv = *reinterpret_cast<const u32x4*>(win + (cc << 4));
        unpack8(pv, kf);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the next round overwrites the window
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = wrow0 + j * 16 + h * 8 + rr;
        const bool ok = m < p.M && col_ok;
        u32x4 vo = v[h];
        if (EPI & 4) {
          float f[8], a8[8];
          unpack8(v[h], f);
          unpack8(av[j][h], a8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += ((mk[j][h] >> e) & 1u) ? a8[e] : 0.f;
          vo = pack8(f);
        }
        const unsigned ooff = ok ? ((unsigned)m * (unsigned)p.out_ld + (unsigned)col) * 2u : POOB;
        __builtin_amdgcn_raw_buffer_store_b128(vo, osrc, (int)ooff, 0, 0);
        if (EPI & 2) {
          float f[8];
          unpack8(v[h], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = ok ? f[e] - kf[e] : 0.f;
            s1[e] += d; s2[e] += d * d;
          }
        }
      }
    }
    if (EPI & 2) {
      // Fold the wave's 8 row groups through its window: [8][64] fp32 partials, then lane c sums column c - channel c of
      // this wave's 64 - so all 64 lanes rebuild plain sums in fp64 (no cross-lane shuffles, no divergent tail).
      float* const sm = reinterpret_cast<float*>(win);         // [8][64] floats + [64] pivots = 2304 B = PWIN
      float S[2], pivot = 0.f;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float* src = which == 0 ? s1 : s2;
        *reinterpret_cast<f32x4*>(sm + rr * 64 + cc * 8) = f32x4{src[0], src[1], src[2], src[3]};
        *reinterpret_cast<f32x4*>(sm + rr * 64 + cc * 8 + 4) = f32x4{src[4], src[5], src[6], src[7]};
        if (which == 0 && rr == 0) {
          *reinterpret_cast<f32x4*>(sm + 512 + cc * 8) = f32x4{kf[0], kf[1], kf[2], kf[3]};
          *reinterpret_cast<f32x4*>(sm + 512 + cc * 8 + 4) = f32x4{kf[4], kf[5], kf[6], kf[7]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += sm[r * 64 + lane];
        if (which == 0) pivot = sm[512 + lane];
        S[which] = t;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      const int nrows = min(64, max(0, p.M - wrow0));
      if (nrows > 0) {
        double sx, sxx;
        unshift_sums(S[0], S[1], pivot, (double)nrows, sx, sxx);
        run1 += sx; run2 += sxx;
      }
      // A block keeps its channel tile for as long as tiles_n divides the tile stride of its run (every power-of-two
      // OC / 128): the running fp64 sums go to the slab only when the channel tile changes, or at the end.
      bool flush = last_tile;
      if (!last_tile) {
        const unsigned nlid = lid + per;
        flush = (int)(nlid - (nlid / (unsigned)p.tiles_n) * (unsigned)p.tiles_n) != tile_n;
      }
      if (flush) {                                               // block-uniform
        double* const part = reinterpret_cast<double*>(win);     // [2][64] doubles = 1 KiB of the wave's window
        part[lane] = run1;
        part[64 + lane] = run2;
        run1 = 0.0; run2 = 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < 2 * PBN) {
          const int which = tid >> 7, c = tid & 127;             // c: channel inside the tile; its wave column is c >> 6
          const char* const st = smem + free_stage * PSTAGE;
          double s = 0.0;
#pragma unroll
          for (int w4 = 0; w4 < 4; ++w4)
            s += reinterpret_cast<const double*>(st + ((c >> 6) * 4 + w4) * PWIN)[which * 64 + (c & 63)];
          if (n0 + c < p.OC)
            slab_add(p.stats, ((size_t)(slot % DDL_BN_SLOTS) * 2 + (size_t)which) * (size_t)p.OC + (size_t)(n0 + c), s);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the windows are read before anyone's next K-step barrier
      }
    }
  }
}

template <int MODE, int EPI>
int launch_pring(ConvParams& p, hipStream_t stream) {
  constexpr int smem = PNS * PSTAGE;
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_pring_kernel<MODE, EPI>), smem);
  if (rc != 0) return rc;
  p.tiles_m = (p.M + PBM - 1) / PBM;
  p.tiles_n = (p.OC + PBN - 1) / PBN;
  const int tiles = p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((conv_pring_kernel<MODE, EPI>), dim3(512), dim3(512), smem, stream, p, tiles);
  return (int)hipGetLastError();
}

}  // namespace

// What the persistent kernel CAN run: plain forward / forward + statistics / plain dgrad / dgrad + (masked) addend, OC > 64,
// at least two K-steps of whole 32-channel chunks, unit output step, 32-bit byte offsets everywhere, and enough tiles
// for the 512 resident blocks to have a next tile to prefetch.
// What it runs BY DEFAULT (DDL_CONV_PRING=1) is what measures faster than the ring kernel on this part
// (bench/pring_probe.py, profiles/r2_persistent_ring_vs_ring_b512.txt): every dgrad with an addend (0.76-0.91x: the addend
// rows are requested two rounds ahead instead of inside the store loop) and forward + statistics with at most two
// K-steps (0.86x at 64 -> 256, 56x56); long reductions and narrow outputs are K-loop bound and stay on the ring kernel
// (1.02-1.15x there).  DDL_CONV_PRING=2: everything it can run (tests, probes); 0: never.
bool ddl_conv_pring_supported(const ConvParams& p) {
  static const int enabled = [] { const char* e = getenv("DDL_CONV_PRING"); return e ? atoi(e) : 1; }();
  if (!enabled) return false;
  if (p.OC <= 64 || p.bias != nullptr || p.relu || p.fin.counter != nullptr) return false;
  if (p.bnz != nullptr || p.bnz_mask != nullptr || p.trace != nullptr) return false;
  if (p.mode_hint == 0 && p.addend != nullptr) return false;
  if (p.mode_hint == 1 && (p.stats != nullptr || p.o_step != 1 || p.stride != 1)) return false;
  if (p.addend != nullptr && p.addend_step != 1) return false;
  if (p.addend == nullptr && p.addend_mask != nullptr) return false;
  if (p.out_ld != p.OC && p.mode_hint != 0) return false;
  const long long tiles = (long long)((p.M + PBM - 1) / PBM) * ((p.OC + PBN - 1) / PBN);
  constexpr int min_tiles = 1024;
  if (tiles < min_tiles || tiles < 512 || tiles >= (1ll << 30)) return false;
  const int ksteps = (p.Kred + PBK - 1) / PBK;
  if (ksteps < 2 || p.IC < PBK) return false;
  if (enabled == 1) {
    const bool addend_dgrad = p.mode_hint == 1 && p.addend != nullptr;
    const bool short_fwd = p.mode_hint == 0 && p.stats != nullptr && ksteps <= 2;
    if (!addend_dgrad && !short_fwd) return false;
  }
  const unsigned long long out_bytes = 2ull * (unsigned long long)p.M * (unsigned long long)p.out_ld;
  return p.in_bytes < 0xFFFF0000ull && p.w_bytes < 0xFFFF0000ull && out_bytes < 0xFFFF0000ull;
}

int ddl_launch_conv_pring(ConvParams& p, int mode, hipStream_t stream) {
  p.div_ohw = make_fastdiv((unsigned)(p.OH * p.OW));
  p.div_ow = make_fastdiv((unsigned)p.OW);
  if (mode == 0) return p.stats != nullptr ? launch_pring<0, 2>(p, stream) : launch_pring<0, 0>(p, stream);
  return p.addend != nullptr ? launch_pring<1, 4>(p, stream) : launch_pring<1, 0>(p, stream);
}
