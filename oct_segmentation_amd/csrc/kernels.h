// Host-side launchers of the gfx950 kernels (all stream-ordered, never synchronise).
#pragma once
#include "common.h"

namespace octseg {

// conv_mfma.hip
int conv_num_mtiles(const ConvArgs& a, int dtype);       // M tiles of a launch as given
int conv_num_mtiles_flat(const ConvArgs& a, int dtype);  // ... as launch_conv runs it (= BN-stat slab rows)
// Layout of the packed weight image a launch expects in ConvArgs::W:
// [wtap][chunk][N tile][BN rows][RB bytes], zero padded, 16-byte chunks XOR-swizzled per row.
struct ConvPackInfo { int BN, RB, nchunks, ntiles; };
ConvPackInfo conv_pack_info(const ConvArgs& a, int dtype);
static inline size_t conv_image_bytes(const ConvPackInfo& p, int wtaps) {
  return (size_t)wtaps * p.nchunks * p.ntiles * p.BN * p.RB;
}
hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t st);
bool conv_masked_eligible(const ConvArgs& a, int dtype);   // launches with per-source tap subsets (ConvArgs::taps_per_src): can the masked loop take this one?

// gemm1x1.hip: stride-1, single-source 1x1 convs and their data gradients as a persistent GEMM (2-byte dtypes); consumes the same
// packed weight image as conv_mfma_kernel.  launch_conv routes eligible launches there; its BN-stat slab has one row per
// workgroup (gemm1x1_rows) instead of one per M tile.
bool gemm1x1_eligible(const ConvArgs& a, int dtype);
int gemm1x1_rows(const ConvArgs& a);
hipError_t launch_gemm1x1(int dtype, const ConvArgs& a, hipStream_t st);

// Deterministic-reduction mode (octseg_set_deterministic / OCTSEG_DETERMINISTIC=1): no floating-point atomic meets another
// workgroup's -- weight gradients without split-K, Dice sums and bias gradients by one workgroup per output.  Slower; bit-stable.
bool deterministic_mode();

// conv3x3p.hip: persistent 3x3 stride-1 conv / data gradient (2-byte dtypes, interior tiles, 128-channel N tile): one software pipeline
// over the flattened (tile, K chunk, tap) sequence of a workgroup; same weight image and BN-stat slab rows as conv_mfma_kernel's 16x16 tile
bool conv3x3p_eligible(const ConvArgs& a, int dtype);
int conv3x3p_rows(const ConvArgs& a);
hipError_t launch_conv3x3p(int dtype, const ConvArgs& a, hipStream_t st);

// thin.hip: 3x3 stride-1 layers with 16 / 32 input and <= 32 output channels on large maps (the last decoder block and the head) as
// HBM-bound persistent kernels: forward / data gradient (weights as the MFMA A operand, built from the fp32 master weights; BN-stat
// slab: one row per workgroup = thin_conv_rows) and weight gradient.  launch_conv / launch_wgrad route eligible launches there.
bool thin_conv_eligible(const ConvArgs& a, int dtype);
int thin_conv_rows(const ConvArgs& a);
hipError_t launch_thin_conv(int dtype, const ConvArgs& a, hipStream_t st);
bool thin_wgrad_eligible(const WgradArgs& a, int dtype);
hipError_t launch_thin_wgrad(int dtype, const WgradArgs& a, hipStream_t st);
// ... and the ResNet stem (7x7 stride 2, 3 -> 64) straight from the NCHW f32 frame: forward (+ BN partials, one slab row per workgroup =
// thin_stem_rows) and weight gradient; no im2col tensor
struct StemArgs {
  const float* img; int N, H, W;            // frame [N][3][H][W] f32
  int normalize; float mean[3], stdv[3];
  const float* w;                            // fp32 master [64][160], k = (r * 7 + s) * 3 + ci
  const float* wscale; const float* bias; int relu_out;   // eval: folded BatchNorm
  void* y;                                   // forward: raw output NHWC T [N][H/2][W/2][64]
  float* slab; int slab_row0;
  const void* dy; float* dW;                 // weight gradient: dy NHWC T [N][H/2][W/2][64], dW [64][160] accumulated
};
bool thin_stem_eligible(int dtype);
int thin_stem_rows(int N, int H, int W);
hipError_t launch_thin_stem_forward(int dtype, const StemArgs& s, hipStream_t st);
hipError_t launch_thin_stem_wgrad(int dtype, const StemArgs& s, hipStream_t st);

// pan.hip: smp's PAN decoder beside its convs -- MaxPool2d(2, 2), the feature-pyramid-attention block's one-channel pyramid (one workgroup, f32),
// its wide 7x7 -> 1-channel conv, the final u * mid + b1 mix, and a plain add
hipError_t launch_maxpool2(int dtype, const void* x, void* p, const void* dp, void* dx, int N, int H, int W, int C, int accum, hipStream_t st);   // dp == nullptr: forward
hipError_t launch_fpa_in_fwd(int dtype, const void* p, const float* w, const float* b, float* y, int N, int H, int W, int C, int K, hipStream_t st);
hipError_t launch_fpa_in_bwd(int dtype, const void* p, const float* dy, const float* w, void* dp, float* dw, float* db, int N, int H, int W, int C, int K,
                             hipStream_t st);
struct FpaPyrArgs {        // layers 0..5 = down1, down2, down3.1, down3.2, conv2, conv1 (each conv k x k + bias, BatchNorm2d(1), ReLU)
  int N, h, w, train;      // h x w: the block's input map; the pyramid lives on h/2, h/4, h/8
  float* scratch; float* gscratch; const float* duu;
  const float* w_[6]; const float* b_[6]; const float* g_[6]; const float* be_[6]; float* rm_[6]; float* rv_[6];
  float* dw_[6]; float* db_[6]; float* dg_[6]; float* dbe_[6];
};
size_t fpa_pyr_scratch_floats(int N, int h, int w);    // forward values kept for the backward (+ the upsampled map uu [N][h][w] and 12 statistics)
size_t fpa_pyr_gscratch_floats(int N, int h, int w);   // gradients; d(x1raw) ends at gscratch + N (h/2) (w/2)
hipError_t launch_fpa_pyr_fwd(const FpaPyrArgs& a, hipStream_t st);   // x1raw in scratch[0 .. n1) -> uu at scratch + fpa_pyr_uu_offset
hipError_t launch_fpa_pyr_bwd(const FpaPyrArgs& a, hipStream_t st);
static inline size_t fpa_pyr_uu_offset(int N, int h, int w) { return fpa_pyr_scratch_floats(N, h, w) - 16 - (size_t)N * h * w; }
hipError_t launch_fpa_mix(int dtype, const float* uu, const void* mid, const void* b1, void* out, const void* g, void* dmid, float* duu, int N, int HW, int C,
                          hipStream_t st);             // g == nullptr: forward out = uu mid + b1; else dmid = g uu, duu = sum_c g mid
hipError_t launch_add2(int dtype, const void* a, const void* b, void* out, size_t numel, hipStream_t st);

// pab.hip: the position attention block of smp's MAnet decoder between its convolutions (strided f32-accumulating GEMMs on activations,
// softmax over a whole position map, the un-transposed reshape of the attended map)
struct PabGemm {           // C[b][m][n] (+)= sum_k A[b][m][k] B[b][k][n]; strides in elements; *_f32: the operand is float whatever the plan's dtype
  const void* A; const void* B; void* C;
  size_t sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn;
  int M, N, K, batch, a_f32, b_f32, c_f32, accum;
};
hipError_t launch_pab_gemm(int dtype, const PabGemm& g, hipStream_t st);
hipError_t launch_pab_softmax(float* S, const float* P, int batch, size_t n, int backward, hipStream_t st);   // forward: in place; backward: dP -> dS in place
hipError_t launch_pab_mix(int dtype, const void* x, const float* M, void* y, float* dM, const void* dy, int N, int HW, int C, hipStream_t st);

// effnet.hip: the non-GEMM operators of efficientnet_pytorch's MBConvBlock (smp 'efficientnet-b0' / '-b5' / '-b7')
struct DwgArgs {           // depthwise K x K (3 | 5), stride 1 | 2, top/left padding `pad` (TF static "same"); w / dw: fp32 [K][K][C]
  const void* in; void* out; void* gin; const float* w; float* dw;   // backward: out = gout, gin = gradient of `in`
  int N, H, W, C, OH, OW, K, stride, pad, accum;
};
hipError_t launch_dwg_fwd(int dtype, const DwgArgs& a, hipStream_t st);
hipError_t launch_dwg_bwd_data(int dtype, const DwgArgs& a, hipStream_t st);
hipError_t launch_dwg_bwd_w(int dtype, const DwgArgs& a, hipStream_t st);
struct BnxArgs {           // forward: out = act(y * scale + shift) * dscale[n] + post;  backward: out = g(=post) * dscale[n] * act'(y * scale + shift)
  const void* y; const float* scale; const float* shift; const float* dscale; const void* post; void* out;
  size_t npix; int hw, C, act;    // act: 0 identity, 1 swish
};
hipError_t launch_bnx_fwd(int dtype, const BnxArgs& a, hipStream_t st);
hipError_t launch_bnx_bwd(int dtype, const BnxArgs& a, hipStream_t st);
struct SefcArgs {          // s = W2 swish(W1 m + b1) + b2 on pooled vectors m [N][C] (T); W1 [R][C], W2 [C][R] fp32; h / dh: float [N][R] scratch
  const void* m; void* s; const void* ds; void* dm;
  const float* w1; const float* b1; const float* w2; const float* b2; float* dw1; float* db1; float* dw2; float* db2;
  float* h; float* dh;
  int N, C, R;
  int act;                 // activation between the two layers: 1 swish (EfficientNet), 0 ReLU (MAnet's SE_ll / SE_hl)
};
hipError_t launch_sefc_fwd(int dtype, const SefcArgs& a, hipStream_t st);
hipError_t launch_sefc_bwd(int dtype, const SefcArgs& a, hipStream_t st);   // dm, dh, then the four parameter gradients (accumulated)

// se.hip: squeeze-excite gate of timm's SEModule (RegNetY): out (+)= in * sigmoid(s[n][c]) and the gate's own gradient
// ds[n][c] = sigmoid'(s) * sum_p g * x (float scratch `part`: N x se_dgate_shares(HW) x C floats)
// (s2 / ds2 != nullptr: MAnet's MFAB gate sigmoid(s) + sigmoid(s2) -- two excitations of one tensor)
hipError_t launch_se_gate(int dtype, const void* in, const void* s, void* out, int N, int HW, int C, int accum, hipStream_t st, const void* s2 = nullptr);
int se_dgate_shares(int HW);
hipError_t launch_se_dgate(int dtype, const void* g, const void* x, const void* s, void* ds, float* part, int N, int HW, int C, hipStream_t st,
                           const void* s2 = nullptr, void* ds2 = nullptr);

// wgrad_mfma.hip
hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t st);
// wgrad1x1.hip: stride-1 1x1 weight gradients (bf16, 64-divisible channel counts, flattened pixel list) as an LDS-DMA ring pipeline;
// wgrad_convt.hip: ConvTranspose2d(k4, s2, p1) weight gradient, all four output parities (16 taps) in one launch; `a` = any of the four
// per-parity launches of the transposed form (they share sources, extents, dy and dW)
bool wgrad_convt16_eligible(const WgradArgs& a, int dtype);
hipError_t launch_wgrad_convt16(int dtype, const WgradArgs& a, hipStream_t st);
// launch_wgrad routes eligible launches there after flatten_1x1
bool wgrad1x1_eligible(const WgradArgs& flattened, int dtype);
hipError_t launch_wgrad1x1(int dtype, const WgradArgs& flattened, hipStream_t st);

// elementwise.hip -----------------------------------------------------------------------------
// BatchNorm finalize (training): reduce the [rows][C][2] slab -> mean/var, scale/shift, running stats.
hipError_t launch_bn_finalize_train(const float* slab, int rows, int C, double count, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, float* scale, float* shift, float* mean,
                                    float* rstd, double* part, unsigned* counters, hipStream_t st);
// `part` (SLAB_PART_CAP x 2 doubles) and `counters` (64 zeroed uints) are the scratch of the two-level slab reduction.
constexpr int SLAB_PART_CAP = 16384;
// eval: scale/shift from the running statistics.
// statistics of a tensor with <= BN_SMALL_COUNT values per channel, two-pass in double from the conv output itself (no slab)
constexpr int BN_SMALL_COUNT = 1024;
hipError_t launch_bn_finalize_small(int dtype, const void* y, int count, int C, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                                    hipStream_t st);
hipError_t launch_bn_finalize_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift,
                                   hipStream_t st);

// eval: scale/shift of every BatchNorm of a plan in one launch (job table + channel prefix sums in the workspace)
struct BnEvalJob { size_t gamma_off, beta_off, rm_off, rv_off /*floats*/, ss_off /*bytes*/; int C;
                   size_t bias_off; /* floats: bias of the conv in front of this BN, folded into the shift (~0: none) */
                   float eps;       /* this BatchNorm's eps (> 0; 0: the launch's default -- 1e-5 torch, 1e-3 efficientnet_pytorch) */ };
hipError_t launch_bn_finalize_eval_all(const float* params, const float* buffers, void* ws, const BnEvalJob* tab, const unsigned* prefix,
                                       int njobs, unsigned total, float eps, hipStream_t st);

// out = relu?( y*scale+shift [+ res*rscale+rshift | + res] ) [+ post]   (NHWC, T)
struct BnActArgs {
  const void* y; const float* scale; const float* shift;
  const void* res; const float* rscale; const float* rshift;  // pre-activation residual (nullable)
  const void* post;                                           // post-activation addend (nullable)
  void* out; size_t npix; int C; int relu;
  unsigned char* maskbits;   // nullable (training forwards): one byte per 16-byte vector, bit i = (output element i > 0): the ReLU mask the
                             // BatchNorm backward would otherwise re-read the whole output tensor for, twice
};
hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t st);

// BatchNorm backward.  g = gradient w.r.t. the post value; mask: 0 none, 1 relu(y*scale+shift) > 0,
// 2 out > 0 (materialised tensor `out`).  Pass 1 writes [rows][C][2] partials (sum dz, sum dz*xhat).
struct BnBwdArgs {
  const void* g; const void* y; const void* out;  // out: mask source for mode 2
  const unsigned char* maskbits;                  // mode 2: the forward's saved mask bits (BnActArgs::maskbits) instead of `out` when not null
  const float* scale; const float* shift; const float* mean; const float* rstd; const float* gamma;
  size_t npix; int C; int mask;
  float* slab; int rows;                 // pass 1 output
  float* dgamma; float* dbeta;           // pass 2 output (fp32 grad arena, accumulated)
  float* coef;                           // [C][2]: (sum dz / M, sum dz*xhat / M)
  void* dy;                              // pass 3 output (may alias g)
  double* part; unsigned* counters;      // scratch of the two-level slab reduction (pass 2)
  void* res_grad; int res_store;         // pass 3, optional: the identity shortcut's gradient  res_grad (+)= g * mask  in the same sweep
  double* fpart; unsigned* fcnt;         // pass 1 finishing the reduction itself (no finalize launch): [8][32][4096] partials, [8][33][32] zeroed tickets (one 128-byte line each)
};                                       // (what masked_accum would re-read g and the mask source for); res_store = 1: first writer

hipError_t launch_bn_bwd_small(int dtype, const BnBwdArgs& a, hipStream_t st);   // reduce + finalize of a small tensor in one launch, double accumulation
bool bn_bwd_fused_finalize();            // (opt-in switch OCTSEG_FUSED_BNFIN=1: measured slower than the separate launch)
hipError_t launch_bn_bwd_reduce(int dtype, const BnBwdArgs& a, hipStream_t st);   // a.fpart && a.fcnt: also does what launch_bn_bwd_finalize does
hipError_t launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t st);
hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a, hipStream_t st);

// dst (+)= g * (mask ? out > 0 : 1)   (residual / skip gradient); store = 1: first writer, plain store
hipError_t launch_masked_accum(int dtype, void* dst, const void* g, const void* out_mask, size_t n, int store,
                               hipStream_t st);
// dst[n, y, x, c] += sum_{2x2} src[n, 2y+dy, 2x+dx, c]   (gradient of nearest x2 upsample)
hipError_t launch_pool2x2_accum(int dtype, void* dst, const void* src, int N, int H, int W, int C, int store,
                                hipStream_t st);
// per-channel sum over pixels of a NHWC T tensor -> out[c] += sum (bias gradients)
hipError_t launch_channel_sum(int dtype, const void* g, size_t npix, int Cstride, int C, float* out,
                              hipStream_t st);

// maxpool 3x3 s2 p1 (NHWC T)
hipError_t launch_maxpool_fwd(int dtype, const void* in, void* out, unsigned char* idx /*nullable: window position of every maximum*/,
                              int N, int H, int W, int C, hipStream_t st);
hipError_t launch_maxpool_bwd_idx(int dtype, const unsigned char* idx, const void* gout, void* gin, int N, int H, int W, int C, int store,
                                  hipStream_t st);
// stem: NCHW f32 image -> (normalise) -> im2col rows [N, H/2, W/2, KP] T for the k x k stride-2 pad-(k / 2) conv (k = 7 ResNet, 3 RegNet)
hipError_t launch_stem_im2col(int dtype, const float* img, void* col, int N, int H, int W, int KP,
                              const float* mean, const float* stdv, int normalize, hipStream_t st, int ksize = 7, int pad = -1 /* top / left; -1: ksize / 2 */);

// Dice loss (multilabel, from logits) and / or mean binary cross-entropy with logits + confusion counts, logits/target NCHW f32
constexpr int DICE_NS = 4;                                   // doubles per (image, class): I, S, T, BCE sum
enum { LOSS_DICE = 0, LOSS_BCE = 1, LOSS_DICE_BCE = 2 };     // = octseg_loss_kind
struct DiceArgs {
  const float* logits; const float* target; int B, C; size_t HW;
  int loss_kind;       // LOSS_*
  double* sums;        // [1 + B][C][DICE_NS]: I, S, T, BCE totals, then per-image replicas (zeroed by the launcher)
  long long* stats;    // [B][C][4]: tp, fp, fn, tn (zeroed by the launcher), nullable
  float* loss;         // scalar
};
hipError_t launch_dice_fwd(const DiceArgs& a, hipStream_t st);
// dL/dz written as NHWC T rows padded to CP channels (zeros beyond C)
hipError_t launch_dice_bwd(int dtype, const DiceArgs& a, float grad_scale, void* dlogits, int CP,
                           hipStream_t st);

// weight packing: master fp32 [taps][O][I] -> LDS-image slabs of a conv launch.  transpose = 0: rows are
// O, K runs over I (forward);  transpose = 1: rows are I, K runs over O (data gradient); K is padded
// to Kpad (>= the contraction length the launch uses).
hipError_t launch_pack_weight_image(int dtype, const float* w, void* img, int taps, int O, int I, int transpose,
                                    const ConvPackInfo& p, hipStream_t st);

// every weight image of a plan in one launch (job table + prefix sums live in the workspace)
struct PackJob { size_t src_off /*floats*/, dst_off /*bytes*/; int taps, O, I, transpose, BN, RB, nchunks, ntiles;
                 size_t scale_off; /* bytes in the workspace: eval scale[O] of the BatchNorm behind this conv, folded into the
                                      forward image when packing for eval (~0: none) */
                 int src_I;        /* row length of the SOURCE weight when the image covers a channel slice of it (0: I) */
                 int src_c0;       /* first source channel of that slice */
                 int tied;         /* 2: the masked data-gradient image of the same kernel (taps == 9 offsets, K = 4 O: dy's four parity planes);
                                      1: taps == 16, the image is the 4x4 stride-2 kernel that a nearest-x2 upsample followed by the 3x3
                                         source is equal to: K4[u][v] = sum of W3[r][s] over r in A(u), s in A(v), A(k) = [max(0, 2 - k), min(2, 3 - k)] */ };
// gradient of the tied image back into the 3x3 weight it was derived from: dW3[t][o][i] += (i < Ca ? sum of dK4 over the (u, v) whose
// A(u) x A(v) holds tap t : dW3s[t][o][i - Ca]);  dK4 [16][O][Ca], dW3s [9][O][Cs] (nullptr when Cs == 0), dW3 [9][O][Ca + Cs], all f32
hipError_t launch_tied_fold(const float* dK4, const float* dW3s, float* dW3, int O, int Ca, int Cs, hipStream_t st);
hipError_t launch_pack_all(int dtype, const float* params, void* ws, const PackJob* tab, const unsigned long long* prefix, int njobs,
                           unsigned long long total, int fold, hipStream_t st);

// fpn.hip: what the FPN decoder (smp decoders/fpn; reference configs/tune.yaml:9-18) adds -- GroupNorm(32) + ReLU (+ bilinear x2,
// align_corners=True) forward / backward, the resample's adjoint, nearest-x2 fill, merge-add + Dropout2d, x4 bilinear of the logits.
struct GnArgs {
  const void* y;            // raw conv output [N][HW][C] T
  const void* g; void* dy;  // backward: gradient w.r.t. relu(gn(y)) at y's resolution in, gradient w.r.t. y out (may alias)
  void* out;                // forward: relu(gn(y)), resampled
  const float* gamma; const float* beta; float* dgamma; float* dbeta;
  float* part;              // [N][S][C][2] partial sums (S = gn_num_slabs(HW))
  float* ss;                // [N][C][2]: scale = gamma * rstd, shift = beta - mean * scale
  float* stat;              // [N][G][2]: mean, rstd
  float* coef;              // [N][G][2]: backward group means
  size_t HW; int C, G, cpg; float eps;
};
int gn_num_slabs(size_t HW);
hipError_t launch_gn_forward(int dtype, const GnArgs& a, int N, int H, int W, int up, hipStream_t st);
hipError_t launch_gn_backward(int dtype, const GnArgs& a, int N, hipStream_t st);
hipError_t launch_bilinear_adjoint(int dtype, const void* gout, void* gin, int N, int H, int W, int C, int up, hipStream_t st);   // gin [N][H][W][C] <- gout [N][H*up][W*up][C]
hipError_t launch_up2_fill(int dtype, const void* in, void* out, int N, int H, int W, int C, hipStream_t st);                    // out [N][2H][2W][C] = nearest x2 of in
hipError_t launch_merge_drop(int dtype, const void* a0, const void* a1, const void* a2, const void* a3, const float* m, float mscale, void* out,
                             int N, size_t HW, int C, hipStream_t st);
hipError_t launch_drop_bwd(int dtype, const void* gout, const float* m, float mscale, void* gin, int N, size_t HW, int C, hipStream_t st);
hipError_t launch_bilinear_nchw(const float* z, float* out, int NC, int H, int W, int up, hipStream_t st);                          // NCHW f32, align_corners=True

// ---- DeepLabV3+ (deeplab.hip)
hipError_t launch_parity_permute(int dtype, const void* src, void* dst, int N, int H, int W, int C, int to_coarse, int accum, hipStream_t st);   // fine [N][H][W][C] <-> coarse [4N][H/2][W/2][C]
hipError_t launch_dw_conv(int dtype, const void* in, int inC, int ic0, void* out, int outC, int oc0, const float* w, int wC, int wc0, int N, int H,
                          int W, int C, int dil, int flip, int accum, hipStream_t st);                                                          // depthwise 3x3, dilation = padding = dil
hipError_t launch_dw_wgrad(int dtype, const void* in, int inC, int ic0, const void* gout, int goC, int oc0, float* dw, int wC, int wc0, int N, int H,
                           int W, int C, int dil, hipStream_t st);
hipError_t launch_image_sum(int dtype, const void* in, void* out, int N, int HW, int C, float div, hipStream_t st);                              // out [N][C] = sum over pixels / div
hipError_t launch_image_bcast(int dtype, const void* in, void* out, int N, int HW, int C, float scale, int accum, hipStream_t st);              // out [N][HW][C] (+)= scale * in [N][C]
hipError_t launch_drop_elem(int dtype, const void* in, const float* keep, float mscale, void* out, size_t numel, hipStream_t st);                // out = in * keep * mscale
hipError_t launch_bilinear_up(int dtype, const void* in, void* out, int N, int H, int W, int C, int up, hipStream_t st);                         // NHWC, align_corners=True
// ---- DeepLabV3 (deeplab.hip): dense dilated convs as plain 3x3 convs on a mosaic of the rate^2 sub-grids
hipError_t launch_mosaic(int dtype, const void* src, void* dst, int N, int H, int W, int C, int r, int to_mosaic, int accum, hipStream_t st);      // fine [N][H][W][C] <-> mosaic [N][r (hs + 1) + 1][r (ws + 1) + 1][C]
hipError_t launch_tensor_stats(int dtype, const void* y, size_t npix, int C, float* slab, int rows, hipStream_t st);   // (any C: channel chunks of <= 256 vectors)                            // slab [rows][C][2] = (sum, sum of squares)
// ---- PSPNet (deeplab.hip)
hipError_t launch_bin_mean(int dtype, const void* in, void* out, int N, int H, int W, int C, int k, hipStream_t st);                              // AdaptiveAvgPool2d((k, k)): out [N][k][k][C]
hipError_t launch_bin_mean_bwd(int dtype, const void* gout, void* gin, int N, int H, int W, int C, int k, int accum, hipStream_t st);
hipError_t launch_bilinear_resize(int dtype, const void* in, void* out, int N, int IH, int IW, int OH, int OW, int C, hipStream_t st);          // align_corners=True, any sizes
hipError_t launch_bilinear_resize_adjoint(int dtype, const void* gout, void* gin, int N, int IH, int IW, int OH, int OW, int C, hipStream_t st);
hipError_t launch_relu(int dtype, const void* in, const void* mask, void* out, size_t numel, hipStream_t st);                                   // mask == nullptr: max(in, 0); else in where mask > 0

// serving: out[n][y][x][out_ch] = (logits[n][ch] nearest-resized to OH x OW) > 0, out has OC channels per pixel
hipError_t launch_mask_assemble(const float* logits, int N, int C, int SH, int SW, int ch, float* out, int OH, int OW, int OC, int out_ch,
                                const int* rows, const int* cols, hipStream_t st);

// on-GPU training augmentation (augment.hip): AUG_NPARAM floats per frame, layout in the kernel's header comment
constexpr int AUG_NPARAM = 36;
hipError_t launch_augment(const float* img, const float* mask, float* img_out, float* mask_out, const float* params, int B, int C, int H,
                          int W, hipStream_t st);

// fused optimizers over the flat fp32 arenas
struct OptArgs {
  float* p; const float* g; float* m; float* v; size_t n;
  int kind;  // 0 SGD, 1 Adam, 2 RMSprop, 3 RAdam
  float lr, wd, beta1, beta2, eps, alpha, momentum; int step; float grad_scale;
};
hipError_t launch_optim_step(const OptArgs& a, hipStream_t st);

}  // namespace octseg
