// Parameter block shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_ring.hip).
#pragma once
#include "common.h"

// Fused BatchNorm finalize of a forward conv's output statistics (conv_epilogue.h): the LAST block to finish a channel
// tile folds that tile's slab columns and writes mean / invstd / scale / shift (+ running statistics), so no separate
// finalize launch sits between the conv and the BN apply pass.
struct BnFin {
  unsigned* counter;       // [tiles_n] arrival counters: zero on entry, left zero (null: no fused finalize)
  float count, momentum, eps;
  const float* gamma;
  const float* beta;
  float* running_mean;     // may be null
  float* running_var;
  float* vec;              // [4][OC]: mean, invstd, scale, shift
};

struct ConvParams {
  const bf16_t* in;
  const bf16_t* wmat;
  bf16_t* out;
  const float* bias;   // optional [OC]
  const bf16_t* addend; // optional, same shape as out: out = conv + addend (gradient accumulation fused in dgrad)
  int addend_step, AH, AW;              // > 1: the addend is COMPACT [N][AH][AW][OC] and lands on pixels (h, w) % step == 0 only
  FastDiv div_fhw, div_fw;               // full-output pixel -> (n, h, w) for that case
  const unsigned char* addend_mask; // optional 1-bit-per-element mask on the addend ([pixels][OC/8] bytes, BN-apply layout)
  float* stats;        // optional [DDL_BN_SLOTS][2][OC] partial sums, accumulated with float atomics
  int N, OH, OW, IH, IW, IC, OC, R, S, stride, pad;
  int M, Kred;
  int tiles_m, tiles_n;
  int relu;
  // row grid -> output pixel: (oh*o_step + o_ph, ow*o_step + o_pw) inside a [N][FOH][FOW] tensor
  int o_step, o_ph, o_pw, FOH, FOW;
  // tap subset: r = r0 + tr*t_step (tr < nR), s = s0 + ts*t_step (ts < nS); Kred = nR*nS*IC; KredFull = R*S*IC
  int r0, s0, t_step, nR, nS, KredFull;
  unsigned long long in_bytes, w_bytes;
  FastDiv div_ohw, div_ow;
  long long* trace;                       // debug: [block][4] wall-clock stamps (start, loop begin, loop end, end) or null
  // dgrad only, with `stats`: BN-backward partial sums of the BatchNorm whose activation this dgrad differentiates.
  // bnz = that BN's input (raw conv output, geometry of `out`), bnz_mask = its ReLU bit mask (null: no ReLU).
  // The epilogue adds  sum(g), sum(g * z)  with  g = out * mask  into the stats slab.
  const bf16_t* bnz;
  const unsigned char* bnz_mask;
  // bn_flags bit 0: no mask tensor, the gate is (bnz > 0) (dx is the gradient of a ReLU OUTPUT bnz: conv+bias+ReLU
  // chains); bit 1: store the gated gradient (the consumer then needs no ReLU-backward pass; sum(g) = its bias grad)
  int bn_flags;
  // forward only: row pitch of `out` in elements (>= OC, multiple of 8).  The tile is written into a channel window of
  // a wider [pixels][out_ld] buffer (`out` already points at the window's first channel): a dense block's feature map
  // grows in place, no concatenation copy.
  int out_ld;
  BnFin fin;
  int fin_done;                           // set by a launcher whose kernel performs the fused finalize
  int mode_hint;                          // 0 fwd, 1 dgrad (set by the launcher before dispatch)                // row -> (n, oh, ow) decode without integer division   // operand sizes (for the buffer descriptors of the LDS-DMA kernel)
};

// conv_ring.hip
bool ddl_conv_ring_supported(const ConvParams& p);
int ddl_launch_conv_ring(ConvParams& p, int mode, hipStream_t stream);

// conv_pring.hip (persistent ring kernel with an asynchronous wave-private epilogue: wide outputs, many tiles)
bool ddl_conv_pring_supported(const ConvParams& p);
int ddl_launch_conv_pring(ConvParams& p, int mode, hipStream_t stream);

// conv_stem.hip (space-to-depth stem, forward: the input patch lives in LDS, the filter in registers)
bool ddl_conv_stem_supported(const ConvParams& p);
int ddl_launch_conv_stem(ConvParams& p, hipStream_t stream);

// conv_c64.hip (3x3 / stride 1 / pad 1 over 64 -> 64 channels, forward and data gradient: filter and input patch in LDS)
bool ddl_conv_c64_supported(const ConvParams& p, int mode);
int ddl_launch_conv_c64(ConvParams& p, int mode, hipStream_t stream);

// conv_big.hip (256 x 256 block tiles, two wave groups half a K-step apart, persistent: the MFMA-bound layers)
int ddl_conv_big_shape(const ConvParams& p);
int ddl_conv_big_split(const ConvParams& p);
int ddl_launch_conv_big(ConvParams& p, int mode, hipStream_t stream);
