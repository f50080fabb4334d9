// Implicit-GEMM convolution, BIG-TILE PING-PONG version (gfx950): forward and data-gradient of the MFMA-bound layers.
//
// What limits the ring kernels (conv_ring.hip / conv_pring.hip), measured this round (profiles/r3_big_tile_*.txt):
// NOT the matrix pipe and not the fabric (L2 hit rate 93 %), but the CU's vector-memory front end, which takes ONE
// request per ~4 clocks whatever its size.  A K-step of 32 channels makes every LDS row 64 bytes = one 64-byte request,
// so a CU pulls 16 B/clk from its L2 and a 256 x 128 x 32 block-step needs 24 LDS-DMA pieces x 64 clocks = 1536 clocks of
// request time for 512 clocks of MFMA: the 0.85-0.93 PFLOP/s of the 3x3 layers IS that ratio.  The first version of this
// kernel (256 x 256 tile, 64-byte rows, ping-pong) hit the same wall at 1.09 PFLOP/s: 32 pieces x 64 clocks per step.
//
// This version moves the same bytes in HALF the requests and needs fewer bytes per FLOP:
//
//   * K-step (stage) of 64 channels: 128-byte LDS rows, every LDS-DMA piece (8 rows x 128 B) is 8 full-line requests;
//   * block tile 256 pixels x 256 channels, ONE block per CU, 8 waves of 128 pixels x 64 channels: 64 KiB per stage for
//     8.4 MFLOP (128 FLOP/B against 87 for 256 x 128), 12 fragment reads for 32 MFMAs per 32-deep sub-step;
//   * the two wave groups (waves 0-3 / 4-7: one wave per SIMD each) run HALF A SUB-STEP APART: group 1 passes one extra
//     barrier before the loop, so in every barrier interval one group is in a LOAD phase (12 ds_read_b128, counted
//     vmcnt) while the other is in an MFMA phase (32 x v_mfma_f32_16x16x32_bf16 = 512 clocks of the SIMD's matrix pipe,
//     with the wave's LDS-DMA pieces issued between the MFMAs).  The matrix pipe always has a wave;
//   * persistent: 256 blocks walk the tile list (XCD-contiguous runs, as conv_pring.hip), the load cursors run ahead ACROSS
//     tile boundaries, the epilogue is wave-private (each wave transposes its 128 x 64 block through a window of its own
//     behind the ring, 16 pixel rows per round) and leaves its 16 output stores in flight under the next tile;
//   * uniform-tap addressing only (IC % 64 == 0, forward or stride-1 dgrad): scalar tap cursors, every pixel row carries
//     a byte offset and a bit mask of its in-bounds taps.
//
// Two stages of 64 KiB fit (+ 18 KiB of windows), so the ring is managed per UNIT: XT / XB = the pixel rows only group 0 /
// only group 1 reads (each group stages its own), W = the filter rows both read (each group stages half).  The pieces go
// out in the LOAD phases (the partner's MFMAs cover their issue time: ~85 clocks each; inside the MFMA phase every piece
// is a bubble in the wave's own matrix pipe - measured 1424 clocks for an MFMA phase with 6 pieces against 580 bare).
// With I(n) = n-th barrier interval, group 0 runs L(s,0) M(s,0) L(s,1) M(s,1) in I(4s) .. I(4s+3), group 1 one later:
//
//   unit (stage s+1)   slot free after (last reader)     issued in                 waited for at the end of    first read in
//   XT                 I(4s-2)  g0 L(s-1,1)               g0 L(s,0) = I(4s)         g0 M(s,1) = I(4s+3)         g0 L(s+1,0) = I(4s+4)
//   W, group-1 half    I(4s-1)  g1 L(s-1,1)               g1 L(s,0) = I(4s+1)       g1 L(s,1) = I(4s+3)         g0 L(s+1,0) = I(4s+4)
//   W, group-0 half    I(4s-1)  g1 L(s-1,1)               g0 L(s,1) = I(4s+2)       g0 M(s,1) = I(4s+3)         g0 L(s+1,0) = I(4s+4)
//   XB                 I(4s-1)  g1 L(s-1,1)               g1 L(s,1) = I(4s+3)       g1 M(s,1) = I(4s+4)         g1 L(s+1,0) = I(4s+5)
//
// Waits are literals: vmcnt(0) at the end of M(s,1) (everything of stage s+1 this wave staged), vmcnt(4) at the end of
// group 1's L(s,1) (its W pieces, leaving the four XB pieces it has just issued in flight).  Reads of a phase are drained
// (lgkmcnt 0) before the barrier that ends it.  Every wave stages 4 pieces per load phase: 16 pieces per interval.
#include "common.h"
#include "host_util.h"
#include "conv_params.h"
#include <cstdlib>

namespace {

constexpr int GBM = 256, GBN = 256;             // block tile: pixels x channels
constexpr int GBK = 64;                         // reduction depth per stage
constexpr int GRP = GBK * 2;                    // LDS row pitch (bytes): one full 128-byte line
constexpr int GXBYTES = GBM * GRP, GSTAGE = (GBM + GBN) * GRP;
constexpr unsigned GOOB = 0xFFFFFFF0u;
constexpr int GWP = 144;                        // window row pitch: 64 channels (128 B) + 16 B pad
constexpr int GWIN = 16 * GWP;                  // per-wave epilogue window: 16 pixel rows x 64 channels (bf16)
constexpr int GSTORES = 16;                     // output store instructions per wave and tile
constexpr int GTM = 8, GTN = 4;                 // fragments per wave: 128 pixels x 64 channels
constexpr int GSMEM = 2 * GSTAGE + 8 * GWIN;
static_assert(GSMEM <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void gdma16(const __amdgpu_buffer_rsrc_t rsrc, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the count needs a literal; the field holds 0..63)
// Tap cursor of one operand stream: stage index, tile, position inside the tile's reduction (all wave-uniform).
struct GCursor {
  unsigned stage;        // global stage index of this block (all tiles) the cursor points at
  unsigned iter;         // tile (index into the block's run) that stage belongs to
  int step;              // stage inside the tile
  int tr, ts, bit, kc;   // tap row / column inside the tap subset, its bit in the row masks, channel offset
};

// EPI bits: 2 = forward BatchNorm partial sums, 4 = addend (+ optional 1-bit mask) - as in conv_ring.hip / conv_pring.hip
// TRACE: diagnostic build (bench/big_trace.py) - every wave sums the shader cycles it spends per phase; stamps cost ~10 %.
template <int MODE, int EPI, bool TRACE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_big_kernel(const ConvParams p, const int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, w4 = wid & 3;       // ping-pong group (one wave per SIMD each); wave inside the group
  const int wm = grp, wn = w4;                  // this wave's 128 pixels x 64 channels

  // ---- this block's run of logical tiles: XCD x owns [base, base + cnt); its gridDim/8 blocks sweep it side by side
  const unsigned per = gridDim.x >> 3, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned q = (unsigned)tiles_total >> 3, r8 = (unsigned)tiles_total & 7u;
  const unsigned cnt = q + (xcd < r8 ? 1u : 0u);
  const unsigned base = xcd < r8 ? xcd * (q + 1u) : r8 * (q + 1u) + (xcd - r8) * q;
  if (slot >= cnt) return;                      // whole block: before any barrier
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

  // ---- staging geometry: a piece = 8 rows x 128 B; lane -> (row lane>>3, physical 16-byte chunk lane&7).  This group's
  // 128 rows of X and of W are pieces 0..15, wave w4 stages pieces w4, 4+w4, 8+w4, 12+w4.  The XOR swizzle of row r is
  // (r >> 1) & 7 (conv_ring.hip, BK = 64); with r = 128 g + 8 piece + (lane >> 3) it depends on the lane and w4 only.
  const int prow = lane >> 3;
  const int lchunk = (lane & 7) ^ ((4 * (w4 & 1) + (lane >> 4)) & 7);
  const int kstages = p.Kred / GBK;             // IC % 64 == 0 (launcher)
  const unsigned total_stages = my_tiles * (unsigned)kstages;

  unsigned xoff[4], vmask[4], woff[4];
  bool wok[4];
  GCursor c{0u, 0u, 0, 0, 0, 0, 0};             // the stage being STAGED (one ahead of the one being multiplied)

  auto decode = [&](unsigned iter) {
    const unsigned lid = base + iter * per + slot;
    const int tile_m = (int)(lid / (unsigned)p.tiles_n), tile_n = (int)(lid - (unsigned)tile_m * (unsigned)p.tiles_n);
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + grp * 128 + (j * 4 + w4) * 8 + prow;
      unsigned mk = 0u;
      int nb = 0, bh = 0, bw = 0;
      if (m < p.M) {
        const int n = (int)fastdiv((unsigned)m, p.div_ohw);
        const int rem = m - n * (int)p.div_ohw.d;
        const int oh = (int)fastdiv((unsigned)rem, p.div_ow);
        const int ow = rem - oh * p.OW;
        nb = n * p.IH * p.IW;
        if (MODE == 0) { bh = oh * p.stride - p.pad; bw = ow * p.stride - p.pad; }
        else           { bh = oh + p.pad; bw = ow + p.pad; }              // unit output step, stride 1 (launcher)
        for (int tr = 0, bit = 0; tr < p.nR; ++tr)
          for (int ts = 0; ts < p.nS; ++ts, ++bit) {
            const int r = p.r0 + tr * p.t_step, sx = p.s0 + ts * p.t_step;
            const int ih = MODE == 0 ? bh + r : bh - r, iw = MODE == 0 ? bw + sx : bw - sx;
            if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) mk |= 1u << bit;
          }
      }
      xoff[j] = ((unsigned)(nb + bh * p.IW + bw) * (unsigned)p.IC + (unsigned)(lchunk * 8)) * 2u;
      vmask[j] = mk;
      const int n = n0 + grp * 128 + (j * 4 + w4) * 8 + prow;
      wok[j] = n < p.OC;
      woff[j] = ((unsigned)n * (unsigned)p.KredFull + (unsigned)(lchunk * 8)) * 2u;
    }
  };
  auto advance = [&]() {                        // next stage; decodes the next tile's rows when the cursor enters it
    ++c.stage;
    c.kc += GBK;
    if (c.kc >= p.IC) {
      c.kc = 0; ++c.bit;
      if (++c.ts == p.nS) { c.ts = 0; ++c.tr; }
    }
    if (++c.step == kstages) {
      c.step = 0; c.tr = 0; c.ts = 0; c.bit = 0; c.kc = 0;
      if (++c.iter < my_tiles) decode(c.iter);
    }
  };
  // this wave's four X pieces / four W pieces of stage c.stage
  auto issue_x = [&]() {
    const int r = p.r0 + c.tr * p.t_step, sx = p.s0 + c.ts * p.t_step;
    const int tap_el = (r * p.IW + sx) * p.IC;
    const unsigned xdelta = (unsigned)((MODE == 0 ? tap_el : -tap_el) + c.kc) * 2u;
    char* const sbase = smem + (c.stage & 1u) * GSTAGE + grp * (128 * GRP);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gdma16(xsrc, sbase + (j * 4 + w4) * 1024, ((vmask[j] >> c.bit) & 1u) ? xoff[j] + xdelta : GOOB);
  };
  auto issue_w = [&]() {
    const int r = p.r0 + c.tr * p.t_step, sx = p.s0 + c.ts * p.t_step;
    const unsigned wdelta = (unsigned)((r * p.S + sx) * p.IC + c.kc) * 2u;
    char* const sbase = smem + (c.stage & 1u) * GSTAGE + GXBYTES + grp * (128 * GRP);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gdma16(wsrc, sbase + (j * 4 + w4) * 1024, wok[j] ? woff[j] + wdelta : GOOB);
  };

  // ---- prologue: stage 0 whole
  decode(0);
  issue_x();
  issue_w();
  advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // P: stage 0 is in LDS
  if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind
  __builtin_amdgcn_sched_barrier(0);

  // fragment read addresses: row * 128 + ((chunk ^ swizzle(row)) << 4); fragment rows step by 16, so swizzle(row) =
  // (lane >> 1) & 7 and every fragment of a wave is base + immediate; sub-step 1 flips bit 2 of the chunk
  const int frow = lane & 15, fch = (lane >> 4) ^ ((lane >> 1) & 7);
  const int a_off0 = GXBYTES + (wn * 64 + frow) * GRP + (fch << 4), a_off1 = GXBYTES + (wn * 64 + frow) * GRP + ((fch ^ 4) << 4);
  const int b_off0 = (wm * 128 + frow) * GRP + (fch << 4), b_off1 = (wm * 128 + frow) * GRP + ((fch ^ 4) << 4);
  char* const win = smem + 2 * GSTAGE + wid * GWIN;

  unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ts0 = 0, ts1 = 0;
#define STAMP(slot) if constexpr (TRACE) { ts1 = __builtin_amdgcn_s_memtime(); tr[slot] += ts1 - ts0; ts0 = ts1; }
  if constexpr (TRACE) { ts0 = __builtin_amdgcn_s_memtime(); tr[14] = ts0; }

  unsigned sg = 0;                              // global stage index of this block (all tiles)
  double run1 = 0.0, run2 = 0.0;                // EPI & 2: running sums of channel `lane` of this wave's 64 (fp64)
  for (unsigned it = 0; it < my_tiles; ++it) {
    const unsigned lid = base + it * per + slot;
    const int tile_m = (int)(lid / (unsigned)p.tiles_n), tile_n = (int)(lid - (unsigned)tile_m * (unsigned)p.tiles_n);
    const int m0 = tile_m * GBM, n0 = tile_n * GBN;

    f32x4 acc[GTN][GTM];
#pragma unroll
    for (int i = 0; i < GTN; ++i)
#pragma unroll
      for (int j = 0; j < GTM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < kstages; ++t, ++sg) {
      const char* sb = smem + (sg & 1u) * GSTAGE;
      const bool more = c.stage < total_stages;               // is there a stage sg+1 to stage
      bf16x8 a[GTN], b[GTM];
      // ================= sub-step 0: load phase (the other group multiplies meanwhile)
#pragma unroll
      for (int i = 0; i < GTN; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sb + a_off0 + i * 16 * GRP);
#pragma unroll
      for (int j = 0; j < GTM; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + b_off0 + j * 16 * GRP);
      if (more) { if (grp == 0) issue_x(); else issue_w(); }
      STAMP(0)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2)
      // ================= sub-step 0: MFMA phase
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < GTN; ++i)
#pragma unroll
        for (int j = 0; j < GTM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      STAMP(3)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4)
      // ================= sub-step 1: load phase
#pragma unroll
      for (int i = 0; i < GTN; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sb + a_off1 + i * 16 * GRP);
#pragma unroll
      for (int j = 0; j < GTM; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + b_off1 + j * 16 * GRP);
      if (more) { if (grp == 0) issue_w(); else issue_x(); }
      STAMP(6)
      if (grp == 1) {                                         // group 1's W pieces of L(sg, 0): group 0 reads them next
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      STAMP(7)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(8)
      // ================= sub-step 1: MFMA phase
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < GTN; ++i)
#pragma unroll
        for (int j = 0; j < GTM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (more) advance();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // everything this wave staged of stage sg+1 has landed
      STAMP(9)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(10)
    }

    // ---------------- epilogue, wave-private, no barrier: 8 rounds of 16 pixel rows x 64 channels through `win`
    const int rr = lane >> 3, cc = lane & 7;                  // read-back: row rr (+8), 16-byte chunk cc
    const int col = n0 + wn * 64 + cc * 8;
    const bool col_ok = col < p.OC;
    const int wrow0 = m0 + wm * 128;                          // first pixel row of this wave
    u32x4 av[2][2];                                           // addend rows, requested one round ahead
    unsigned mk[2][2];
    auto issue_addend = [&](int j, int s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = wrow0 + j * 16 + h * 8 + rr;
        const bool ok = m < p.M && col_ok;
        const unsigned aoff = ok ? ((unsigned)m * (unsigned)p.OC + (unsigned)col) * 2u : GOOB;
        av[s][h] = __builtin_amdgcn_raw_buffer_load_b128(asrc, (int)aoff, 0, 0);
        const unsigned moff = ok ? (unsigned)m * (unsigned)(p.OC >> 3) + (unsigned)(col >> 3) : GOOB;
        mk[s][h] = p.addend_mask != nullptr ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(msrc, (int)moff, 0, 0) : 0xffu;
      }
    };
    if (EPI & 4) issue_addend(0, 0);

    float s1[8], s2[8], kf[8];
    if (EPI & 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; kf[e] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < GTM; ++j) {
      {
        char* const wp = win + (lane & 15) * GWP + (lane >> 4) * 8;
#pragma unroll
        for (int i = 0; i < GTN; ++i) {
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
          pk[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
          *reinterpret_cast<u32x2*>(wp + i * 32) = pk;
        }
      }
      if ((EPI & 4) && j + 1 < GTM) issue_addend(j + 1, (j + 1) & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private window: only my own LDS writes
      u32x4 v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) v[h] = *reinterpret_cast<const u32x4*>(win + (h * 8 + rr) * GWP + (cc << 4));
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
          unpack8(av[j & 1][h], a8);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += ((mk[j & 1][h] >> e) & 1u) ? a8[e] : 0.f;
          vo = pack8(f);
        }
        const unsigned ooff = ok ? ((unsigned)m * (unsigned)p.out_ld + (unsigned)col) * 2u : GOOB;
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
      // Fold the wave's 8 row groups through its window ([8][64] fp32 partials + [64] pivots = 2304 B = GWIN), lane c sums
      // column c - channel c of this wave's 64 - and rebuilds plain sums in fp64 (conv_pring.hip).
      float* const sm = reinterpret_cast<float*>(win);
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
        float tsum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) tsum += sm[r * 64 + lane];
        if (which == 0) pivot = sm[512 + lane];
        S[which] = tsum;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      const int nrows = min(128, max(0, p.M - wrow0));
      if (nrows > 0) {
        double sx, sxx;
        unshift_sums(S[0], S[1], pivot, (double)nrows, sx, sxx);
        run1 += sx; run2 += sxx;
      }
      // A block keeps its channel tile while tiles_n divides the tile stride of its run: the running fp64 sums go to the
      // slab (one atomic per channel and WAVE) only when the channel tile changes, or at the end.
      bool flush = it + 1 == my_tiles;
      if (!flush) {
        const unsigned nlid = lid + per;
        flush = (int)(nlid - (nlid / (unsigned)p.tiles_n) * (unsigned)p.tiles_n) != tile_n;
      }
      if (flush) {
        const int c = n0 + wn * 64 + lane;
        if (c < p.OC) {
          const size_t sb2 = (size_t)((slot * 2u + (unsigned)wm) % DDL_BN_SLOTS) * 2;
          slab_add(p.stats, sb2 * (size_t)p.OC + (size_t)c, run1);
          slab_add(p.stats, (sb2 + 1) * (size_t)p.OC + (size_t)c, run2);
        }
        run1 = 0.0; run2 = 0.0;
      }
    }
    STAMP(5)
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();   // same barrier count in both groups
  if constexpr (TRACE) {
    if (lane == 0) {
      long long* const o = p.trace + ((size_t)blockIdx.x * 8 + (size_t)wid) * 16;
      for (int i = 0; i < 11; ++i) o[i] = (long long)tr[i];
      o[14] = (long long)(__builtin_amdgcn_s_memtime() - tr[14]);
      o[15] = (long long)total_stages;
    }
  }
#undef STAMP
}

template <int MODE, int EPI>
int launch_big_t(ConvParams& p, hipStream_t stream) {
  static SmemAttrOnce once;
  const int rc = once.ensure(reinterpret_cast<const void*>(&conv_big_kernel<MODE, EPI>), GSMEM);
  if (rc != 0) return rc;
  p.tiles_m = (p.M + GBM - 1) / GBM;
  p.tiles_n = (p.OC + GBN - 1) / GBN;
  const int tiles = p.tiles_m * p.tiles_n;
  if (p.trace != nullptr) {
    if constexpr (EPI == 2 || EPI == 0) {
      static SmemAttrOnce once_t;
      const int rt = once_t.ensure(reinterpret_cast<const void*>(&conv_big_kernel<MODE, EPI, true>), GSMEM);
      if (rt != 0) return rt;
      hipLaunchKernelGGL((conv_big_kernel<MODE, EPI, true>), dim3(256), dim3(512), GSMEM, stream, p, tiles);
      return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL((conv_big_kernel<MODE, EPI>), dim3(256), dim3(512), GSMEM, stream, p, tiles);
  return (int)hipGetLastError();
}

}  // namespace

// What the big-tile kernel CAN run: plain forward / forward + statistics / plain dgrad / dgrad + (masked) addend with
// uniform taps (IC % 64 == 0; forward at any stride, dgrad at stride 1 with unit output step), at most 32 taps, OC a
// multiple of 8 above 128, 32-bit byte offsets everywhere.
static bool big_can_run(const ConvParams& p) {
  if (p.OC <= 128 || p.bias != nullptr || p.relu || p.fin.counter != nullptr) return false;
  if (p.bnz != nullptr || p.bnz_mask != nullptr) return false;
  if (p.mode_hint == 0 && p.addend != nullptr) return false;
  if (p.mode_hint == 1 && (p.stats != nullptr || p.o_step != 1 || p.stride != 1)) return false;
  if (p.addend != nullptr && p.addend_step != 1) return false;
  if (p.addend == nullptr && p.addend_mask != nullptr) return false;
  if (p.out_ld != p.OC && p.mode_hint != 0) return false;
  if (p.IC % GBK != 0 || p.nR * p.nS > 32 || p.Kred != p.nR * p.nS * p.IC) return false;
  const unsigned long long out_bytes = 2ull * (unsigned long long)p.M * (unsigned long long)p.out_ld;
  return p.in_bytes < 0xFFFF0000ull && p.w_bytes < 0xFFFF0000ull && out_bytes < 0xFFFF0000ull;
}

// 0 = the layer goes elsewhere; 1 = 256 pixels x 256 channels.
// DDL_CONV_BIG: 0 never; 1 (default) the FORWARD convolutions it measures faster on (bench/big_probe.py,
// profiles/r3_big_tile_vs_ring_b*.txt: reductions of at least 1024 - the 3x3 layers from 128 channels on and the 1x1
// layers from 1024 input channels on; 0.80-0.90 of the ring kernel's time on whole tile rounds); 2 everything it can
// run; 3 = 1; 4 data gradients only; 5 forward and data gradients.  The data gradients are 0.80-0.87 in isolation too but
// LOSE inside the step (+0.5 % step time, bench/r3_ab_big.sh): a 146 KiB one-block-per-CU kernel leaves no room for the
// filter-gradient blocks that run beside the data gradient on the side stream, so the two serialise.
int ddl_conv_big_shape(const ConvParams& p) {
  static const int enabled = [] { const char* e = getenv("DDL_CONV_BIG"); return e ? atoi(e) : 1; }();
  if (!enabled || !big_can_run(p)) return 0;
  if (enabled == 2) return 1;
  if ((enabled == 1 || enabled == 3) && p.mode_hint != 0) return 0;
  if (enabled == 4 && p.mode_hint != 1) return 0;
  const long long tiles = (long long)((p.M + GBM - 1) / GBM) * ((p.OC + GBN - 1) / GBN);
  return (p.Kred >= 1024 && p.OC % GBN == 0 && p.addend == nullptr && tiles >= 180) ? 1 : 0;
}

// Tile rounds: 256 persistent blocks take whole tiles, so R + f rounds of tiles cost R + 1 rounds of time.  When the
// last round would be less than 60 % full the launcher cuts the batch at an image boundary: the images that make R
// whole rounds run here, the rest goes to the ring kernels (two 256 x 128 blocks per CU: finer tiles for the tail).
// Returns the number of leading images for this kernel (N: everything, no split).
int ddl_conv_big_split(const ConvParams& p) {
  static const int split_on = [] { const char* e = getenv("DDL_BIG_SPLIT"); return e ? atoi(e) : 1; }();
  const long long tiles_n = (p.OC + GBN - 1) / GBN, pix = (long long)p.OH * p.OW;
  const long long tiles = (long long)((p.M + GBM - 1) / GBM) * tiles_n;
  const long long R = tiles / 256, f = tiles - R * 256;
  if (!split_on || R == 0 || f == 0 || f >= 154 || p.out_ld != p.OC) return p.N;
  const long long n0 = (R * 256 / tiles_n) * GBM / pix;            // images whose tiles fit in R rounds
  if (n0 <= 0 || n0 >= p.N) return p.N;
  return (int)n0;
}

int ddl_launch_conv_big(ConvParams& p, int mode, hipStream_t stream) {
  p.div_ohw = make_fastdiv((unsigned)(p.OH * p.OW));
  p.div_ow = make_fastdiv((unsigned)p.OW);
  if (ddl_conv_big_shape(p) == 0) return -14;
  if (mode == 0) return p.stats != nullptr ? launch_big_t<0, 2>(p, stream) : launch_big_t<0, 0>(p, stream);
  return p.addend != nullptr ? launch_big_t<1, 4>(p, stream) : launch_big_t<1, 0>(p, stream);
}
