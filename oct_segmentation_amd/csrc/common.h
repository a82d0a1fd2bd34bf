// Shared host/device declarations for the octseg gfx950 engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace octseg {

typedef unsigned short bf16_t;  // raw bfloat16 bits in HBM (same layout as torch.bfloat16)
struct f16_t { unsigned short bits; };  // raw IEEE half bits (torch.float16): the serving dtype of BASELINE config #5 (eval forward only)

enum DType { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
static inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

constexpr int MAX_SRC = 5;    // U-Net++ dense concat: up + 3 dense + encoder feature
constexpr int MAX_TAPS = 49;  // 7x7 (only used by the unit-test entry points)

// One source of a virtual (concatenated / upsampled / lazily normalised) conv input.
// The consumer applies  relu?(x * scale[c] + shift[c])  on load (scale == nullptr: identity).
struct SrcDesc {
  const void* ptr;     // NHWC, element type T
  const float* scale;  // per source-channel, or nullptr
  const float* shift;
  int C;               // channels of this source (== its channel stride)
  int c0;              // first channel inside the concatenation
  int H, W;            // stored spatial size
  int up;              // 1: nearest x2 upsample on read (coord >> 1)
  int relu;            // apply relu after the affine
};

// Destination slice for a conv output (forward: one; dgrad: one per forward source).
struct DstDesc {
  void* ptr;  // NHWC T, or NCHW f32 in head mode
  int C;      // channel stride of the destination tensor
  int c0;     // first output channel mapped to this destination
  int cn;     // number of channels
  int H, W;   // full spatial size of the destination tensor
  int accum;  // 1: add to what is there (a later gradient contribution), 0: plain store (first writer)
  int pool;   // 1: the destination is the half-resolution tensor behind a nearest-x2 upsample: every 2x2 output quad is
              //    summed in the epilogue (H, W are then the half-resolution extents; ostride must be 1)
};

enum OutMode { OUT_STORE = 0, OUT_ACCUM = 1, OUT_HEAD_NCHW = 2 };

struct ConvArgs {
  SrcDesc src[MAX_SRC];
  int nsrc;
  int Cin;         // total concatenated input channels
  int N, IH, IW;   // virtual input extent (after up)
  int OH, OW;      // output grid extent (tile iteration space)
  int ntaps;
  // int (not char) on purpose: a runtime-indexed char in the kernarg segment is fetched with a VMEM byte
  // load whose s_waitcnt vmcnt(0) drains the whole software pipeline; dwords go through the scalar cache.
  int tap_dy[MAX_TAPS], tap_dx[MAX_TAPS];  // input offset of each tap (pad folded in)
  int tap_w[MAX_TAPS];                     // weight slab index of each tap
  int min_dy, min_dx, span_y, span_x;              // bounding box of the tap offsets
  int istride;     // input coord = grid * istride + tap offset
  const void* W;   // [wtaps][Cout][Cin] T  (Cin contiguous)
  int Cout;
  DstDesc dst[MAX_SRC];
  int ndst;
  int ostride, ooy, oox;  // output pixel = grid * ostride + (ooy, oox)
  int out_mode;
  const float* bias;      // per-Cout, nullable
  int relu_out;           // eval with BatchNorm folded into the weight image: the epilogue stores relu(acc + bias)
  float* stat_slab;       // nullable: [slab rows][Cout][2] partial (sum, sumsq) per M tile
  int slab_row0;
  int src_uniform;        // set by launch_conv: every source starts on a K-chunk boundary (the per-chunk source choice is then scalar)
  int tile_order;         // set by launch_conv: 0 N-major, 1 M-major placement of the tiles on the XCDs (speed only)
  unsigned long long* stamp;  // diagnostic builds (-DOCTSEG_STAMP) only: per-phase cycle sums, else nullptr
  // fp32 master weights of the layer ([wtaps][wO][wI], the parameter arena's layout) for kernels that build their operands from them
  // instead of the packed image (thin.hip); wtrans = 1: data gradient (output rows run over wI, the contraction over wO);
  // wscale: eval forwards, the BatchNorm scale folded into output row `co` (nullptr: none).  Wmaster == nullptr: not available.
  const float* Wmaster;
  int wO, wI, wtrans;
  const float* wscale;
  // Per-source tap subsets (0: every source meets every tap).  taps_per_src = n > 0: the K chunks of source i only contract with the n taps
  // whose indices are packed four bits each in src_taps[i] -- ConvTranspose2d's data gradient as ONE stride-1 launch over dy's four parity
  // planes (each plane a source of its own, a view of dy with doubled strides), every plane with its own 2 x 2 of the 3 x 3 tap offsets.
  // Sources must start on K-chunk boundaries; routed to conv_mfma_kernel's masked loop.
  int taps_per_src;
  int src_taps[MAX_SRC];
};

struct WgradArgs {
  SrcDesc src[MAX_SRC];
  int nsrc;
  int Cin;
  int N, IH, IW;
  int OH, OW;       // grid extent of dy that is iterated
  int ntaps;
  int tap_dy[MAX_TAPS], tap_dx[MAX_TAPS];
  int tap_w[MAX_TAPS];
  int min_dy, min_dx, span_y, span_x;
  int istride;
  const void* dy;   // NHWC T, [N][DH][DW][dyC]
  int dyC, DH, DW;  // channel stride and full extent of dy
  int dstride, doy, dox;  // dy pixel = grid * dstride + (doy, dox)
  int Cout;         // real number of output channels (rows of dW)
  float* dW;        // [wtaps][Cout][Cin] fp32, accumulated with atomics
  int ksplit;       // number of pixel-tile groups (grid.z)
  int co_fast;      // set by the launcher: co tiles fastest in the XCD-local workgroup order (speed only)
  int wg_target;    // 0: the launcher's default split-K width (one workgroup per CU); > 0: at most this many workgroups -- the plan asks for a
                    // narrower launch when the weight gradient runs on the side stream BESIDE the chain's kernels (plan.cpp side_wgs)
  unsigned long long* stamp;  // diagnostic builds (-DOCTSEG_STAMP) only
};

}  // namespace octseg
