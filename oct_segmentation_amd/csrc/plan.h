// plan.h -- static execution plan of one segmentation network on one GPU.
#pragma once
#include <string>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "../../include/octseg.h"

namespace octseg {

struct TensorInfo {
  int N, H, W, C;
  size_t off;    // byte offset of the activation in the workspace
  size_t goff;   // byte offset of its gradient buffer (valid when need_grad)
  bool need_grad;
  bool external; // logits / image: not in the workspace
  size_t mask_off = 0;  // bn_act outputs whose ReLU mask the backward needs: byte offset of their mask bits (1 byte per 16-byte vector), else 0
  int grad_alias = -1;  // >= 0: shares the gradient buffer of that tensor (the summands of a merge-add all receive the same gradient)
};

struct Value {   // what a consumer reads: tensor t, optionally through BN `bn` (scale/shift) lazily
  int t = -1;
  int bn = -1;
};

struct ParamInfo {
  std::string name;
  int kind, R, S, O, I, KP;
  size_t off, numel;
};

struct BNInfo {
  std::string name;
  int C;
  int gamma, beta;          // param indices
  size_t rm_off, rv_off;    // element offsets in the buffer arena
  size_t ss_off;            // byte offset in workspace: scale[C] shift[C] mean[C] rstd[C] coef[2C]
  double count;             // elements per channel of the tensor it normalises
  int rows;                 // slab rows produced by the conv epilogue
  bool lazy;                // consumed by conv sources (relu(bn(y))) -> BN backward runs in place
  int y;                    // tensor it normalises
  float eps = 1e-5f, momentum = 0.1f;   // torch defaults; efficientnet_pytorch: 1e-3 / 0.01
};

// GroupNorm(32) + ReLU of the FPN decoder's Conv3x3GNReLU (smp decoders/fpn): per-(image, group) statistics, no running buffers
struct GNInfo {
  std::string name;
  int C, G;
  int gamma, beta;          // param indices
  int y;                    // raw conv output it normalises
  size_t part_off, ss_off, stat_off, coef_off;   // workspace byte offsets: partial sums, scale/shift [N][C][2], mean/rstd and backward means [N][G][2]
};

struct ConvSrc { Value v; int up; int c0 = 0, cn = 0; };   // c0 / cn: channel slice [c0, c0 + cn) of the tensor (cn = 0: all of it) -- grouped convs

struct ConvLayer {
  std::string name;
  int R, S, stride, pad;
  bool transposed, head, stem;
  int Cin, Cout;
  int N, IH, IW, OH, OW;    // virtual input extent, output extent
  std::vector<ConvSrc> srcs;
  int out;                  // tensor id of the raw output (or -1 for the head: external logits)
  int bn;                   // BN that follows (stats emitted by the epilogue), or -1
  int w, b;                 // param indices (b = -1: no bias)
  size_t wimg_fwd_off, wimg_dgrad_off;   // byte offsets of the packed weight images in the workspace
  ConvPackInfo pk_fwd, pk_dgrad;         // their layouts (tile variant of the forward / dgrad launches)
  bool has_dgrad;
  int OP;                   // channel count of dy as the dgrad sees it (Cout rounded up to 16)
  bool accum_out = false;   // the forward ADDS into an existing tensor (FPNBlock: nearest-x2 fill, then the skip conv accumulates)
  int out_c0 = 0;           // first channel of `out` this layer writes, and (sliced) whether it owns only a channel slice of it: one group of a
  bool sliced = false;      // grouped conv (RegNet's conv2: groups = width / group width) -- G independent convs on channel slices
  int stem_k = 7;           // stem layers: kernel size of the im2col rows (7: torchvision ResNet, 3: timm RegNet)
  int fold_bn = -1;         // sliced layers: the BatchNorm of the whole tensor (not fed by this layer's epilogue in training); eval forwards
                            // fold its scale / shift for channels [out_c0, out_c0 + Cout) into this group's image and epilogue
  // Tied decomposition (training, 2-byte dtypes): srcs[0] is read through a nearest-x2 upsample, so over its Ca channels the 3x3 is equal to a
  // ConvTranspose2d(k4, s2, p1) of the LOW-resolution map with the 4x4 kernel K4 = sums of the 3x3 taps that land on one source pixel
  // (kernels.h PackJob::tied) -- 16 instead of 36 multiply-accumulates per source pixel, channel pair.  The skip sources (Cs channels)
  // keep their 3x3 as a launch of their own.  `tie` = which passes run that way: 1 forward, 2 data gradient, 4 weight gradient.
  int tie = 0, tie_Ca = 0, tie_Cs = 0;
  size_t tie_fu_off = 0, tie_fs_off = 0, tie_du_off = 0, tie_ds_off = 0;   // images: forward up / skip, data gradient up / skip
  ConvPackInfo tie_pk_fu, tie_pk_fs, tie_pk_du, tie_pk_ds;
  bool tie_du_masked = false;   // the data gradient over the upsampled source's channels is ONE masked launch over dy's parity planes (9-tap image, 4 O virtual channels)
};

enum OpKind { OP_STEM_COL, OP_CONV, OP_BN_FIN, OP_BN_ACT, OP_MAXPOOL,
              OP_UP2,        // out = nearest x2 of in (materialised; a skip conv accumulates into it)
              OP_GN,         // out = resample(relu(groupnorm(in))), up = 1 | 2 (bilinear, align_corners)
              OP_MERGE,      // out = (ins[0] + .. + ins[3]) * dropout keep mask
              OP_UPLOGITS,   // logits = bilinear x head_up of the head's low-resolution output
              // DeepLabV3+ (deeplab.hip)
              OP_PARITY,     // up = 1: out [4N][H/2][W/2] = parity sub-grids of in [N][H][W] (dilation 2 -> plain 3x3); up = 0: the inverse
              OP_DW,         // depthwise 3x3 (dilation = padding = up) of `in` into channels [oc0, oc0 + C) of `out`; weights param `dwp` at channel wc0
              OP_GAP,        // out [N][1][1][C] = mean over the pixels of in (AdaptiveAvgPool2d(1))
              OP_BCAST,      // out [N][H][W][C] = in [N][1][1][C] (bilinear resize of a 1x1 map)
              OP_DROPE,      // out = in * element-wise dropout keep mask / (1 - p)
              OP_UPB,        // out = bilinear x up of in (align_corners=True)
              // PSPNet (deeplab.hip)
              OP_BINPOOL,    // out [N][k][k][C] = AdaptiveAvgPool2d((k, k)) of in, k = up
              OP_RESIZE,     // out = bilinear resize of in to out's size (align_corners=True)
              OP_RELU,       // out = relu(in) of a plain tensor (biased conv without BatchNorm)
              OP_DROP2D,     // out = in * Dropout2d keep pattern [N][C] / (1 - p)
              // DeepLabV3 (deeplab.hip)
              OP_MOSAIC,     // out = mosaic of the rate^2 sub-grids of in (oc0 = 1), or in's mosaic gathered back (oc0 = 0); rate = up
              OP_STATS,      // BatchNorm partial sums of tensor `in` for BN `bn` (its producer is not a conv epilogue)
              OP_SEGATE,     // out = in * sigmoid(ins[0] [N][1][1][C]): the squeeze-excite gate (timm SEModule, RegNetY; se.hip)
              // EfficientNet (effnet.hip)
              OP_DWG,        // out = depthwise K x K (wc0) stride `up` conv of in, top/left padding oc0, weights param dwp [K][K][C]
              OP_BNX,        // out = act(bn(y)) * drop_connect[n] + post: act = up (0 identity, 1 swish), drop-connect block oc0 (-1: none)
              OP_SEFC,       // out [N][1][1][C] = W2 act(W1 in + b1) + b2: params ins[0..3] = w1, b1, w2, b2; up = reduction channels; oc0 = act (1 swish, 0 ReLU)
              OP_PAB,        // MAnet's position attention: out = in + reshape(softmax(center topT) bottom); ins[0..2] = top, center, bottom (pab.hip)
              // PAN (pan.hip)
              OP_FPA,        // feature pyramid attention: out = pyramid(in) * ins[0] (mid) + ins[1] (b1); parameters in octseg_plan::fpa
              OP_ADD };      // out = in + ins[0]
struct Op {
  OpKind kind;
  int conv = -1;   // OP_CONV
  int bn = -1;     // OP_BN_FIN
  // OP_BN_ACT: out = relu?(bn(y) + res) + post
  Value y, res;
  int post = -1;
  bool relu = true;
  int in = -1, out = -1;  // OP_MAXPOOL / OP_BN_ACT / OP_STEM_COL out
  int lane = 0;           // forward stream: 0 main, 1 side (decoder nodes that only need early encoder features)
  int gn = -1, up = 1;    // OP_GN
  int ins[4] = {-1, -1, -1, -1};   // OP_MERGE
  int dwp = -1, oc0 = 0, wc0 = 0;  // OP_DW: weight parameter, first output channel, first weight channel
  bool dw_first = true;            // OP_DW: first writer of the parameter's gradient slice? (all slices accumulate: atomics)
  size_t aux_off = 0;              // OP_SEFC: float scratch h [N][R], dh [N][R] in the workspace; OP_PAB: S / P [N][HW^2], dP [N][HW^2], M [N][HW][C]
  bool conv_bn = false;            // OP_BNX: the BatchNorm is fed (and, in eval, folded) by a conv epilogue
};

}  // namespace octseg

struct octseg_plan {
  std::string arch, encoder;
  int classes, B, H, W, dtype;
  std::vector<octseg::TensorInfo> tensors;
  std::vector<octseg::ParamInfo> params;
  std::vector<octseg::BNInfo> bns;
  std::vector<octseg::ConvLayer> convs;
  std::vector<octseg::GNInfo> gns;
  std::vector<octseg::Op> ops;
  // FPN: the head runs at stride 4 into z4 (NCHW f32) and is resampled x4 into the logits; its gradient comes back through dz4
  int head_up = 1;
  size_t z4_off = 0, dz4_off = 0;
  // stem through thin.hip: frame and normalisation of the last training forward (its weight gradient gathers the frame again)
  const float* stem_image = nullptr; int stem_normalize = 0; float stem_mean[3] = {0, 0, 0}, stem_std[3] = {1, 1, 1};
  const float* dropout_keep = nullptr;   // keep pattern of the next training forward (caller-owned device floats of {0, 1}): fpn [B][128]
                                         // (Dropout2d), deeplabv3plus [B][H/16][W/16][256] NHWC (element-wise Dropout of ASPP.project)
  float dropout_p = 0.2f;
  size_t param_numel = 0, buffer_numel = 0;
  size_t ws_bytes = 0;
  size_t act_begin = 0, act_end = 0, grad_begin = 0, grad_end = 0;
  size_t slab_off = 0, slab_bytes = 0;       // BN partial-sum slab (reused per layer; one per forward lane)
  bool has_lanes = false;
  size_t fin_part_off = 0, fin_cnt_off = 0;  // scratch of the two-level slab reduction (BN finalize)
  size_t bwd_part_off = 0, bwd_cnt_off = 0;  // scratch / tickets of the BN-backward reduce kernel that finishes its own reduction
  size_t pool_idx_off = 0;                   // maxpool: window position of every maximum (1 byte per output element)
  size_t tmp_off = 0, tmp_bytes = 0;         // dgrad temp for upsampled sources
  size_t se_part_off = 0;                    // float scratch of the squeeze-excite gate's own gradient (se.hip)
  int stem_pad = 3;                          // top / left padding of the stem conv (3: ResNet 7x7, 1: RegNet 3x3, 0: EfficientNet's static 'same')
  const float* drop_connect = nullptr;       // EfficientNet: id-skip factors of the next training forward, device float [blocks with id skip][B]
  std::vector<float> dc_rates;               // their drop_connect rates (0.2 * block index / blocks)
  std::string run_error;                     // non-empty: the plan exists (parameter table) but this frame size cannot run (PAN below 128 x 128)
  struct { int w[6], b[6], bn[6]; int pool = -1; size_t scratch_off = 0, gscratch_off = 0; } fpa;   // PAN's FPA pyramid: parameter / BatchNorm indices of its six
                                                                                                     // one-channel layers, the pooled-input tensor, f32 scratch
  size_t dlogits_off = 0;                    // NHWC padded dL/dlogits
  int loss_kind = 0;                         // LOSS_DICE | LOSS_BCE | LOSS_DICE_BCE (octseg_plan_set_loss)
  size_t dice_off = 0;                       // double sums[1 + B][C][DICE_NS]: totals, then one replica per image
  int col_tensor = -1;
  int stem_k = 7;                            // kernel size of the stem conv (7x7 s2 p3 ResNet, 3x3 s2 p1 RegNet): rows of OP_STEM_COL
  int dlogits_C = 16;
  double fwd_macs = 0;
  double exec_macs[3] = {0, 0, 0};           // multiply-accumulates a training step EXECUTES per pass (forward, data gradient, weight gradient):
                                             // below fwd_macs where the tied decomposition runs (ConvLayer::tie)
  size_t tie_scratch_off = 0;                // f32 [16][O][Ca] + [9][O][Cs]: gradients of the tied images of ONE layer (folded into the 3x3 gradient)
  // side stream of the backward: weight gradients only depend on dy and on saved activations, so they run
  // beside the dgrad / BN-backward chain (MFMA-bound next to HBM-bound work)
  // weight images currently in the workspace correspond to (packed_params, packed_ws) unless invalidated
  bool packed_valid = false;
  bool packed_fold = false;                // eval images (BatchNorm scale folded in) vs training images
  const void* packed_buffers = nullptr;    // running statistics the eval images were folded with
  const void* packed_ws = nullptr;
  const void* packed_params = nullptr;
  std::vector<octseg::PackJob> pack_jobs;          // host copy of the pack table
  std::vector<unsigned long long> pack_prefix;
  unsigned long long pack_total = 0;
  size_t pack_tab_off = 0, pack_prefix_off = 0;    // their place in the workspace
  const void* pack_tab_ws = nullptr;               // workspace the table was last uploaded to
  std::vector<octseg::BnEvalJob> bn_jobs;          // eval: all BN scale/shift in one launch
  std::vector<unsigned> bn_prefix;
  unsigned bn_total = 0;
  size_t bn_tab_off = 0, bn_prefix_off = 0;
  hipStream_t side = nullptr;        // second forward lane (default priority)
  hipStream_t side_bwd = nullptr;    // weight gradients of the backward (lowest priority: they yield to the chain's kernels)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_slice = nullptr;
  // eval forward as a hipGraph (octseg_plan_set_graph): captured on the second call with an unchanged argument set,
  // replayed while that set stays the same (a B=1 predict is ~400 launches of a few microseconds each)
  bool graph_enabled = false;
  int graph_seen = 0;                 // eager calls with the current key (the first sets the function attributes)
  hipGraphExec_t graph_exec = nullptr;
  struct GraphKey {
    const void* params; const void* buffers; const void* ws; const void* image; const void* logits; const void* stream;
    int normalize; float mean[3], stdv[3];
    bool operator==(const GraphKey& o) const {
      return params == o.params && buffers == o.buffers && ws == o.ws && image == o.image && logits == o.logits &&
             stream == o.stream && normalize == o.normalize && mean[0] == o.mean[0] && mean[1] == o.mean[1] &&
             mean[2] == o.mean[2] && stdv[0] == o.stdv[0] && stdv[1] == o.stdv[1] && stdv[2] == o.stdv[2];
    }
  } graph_key{};
  // the TRAINING step (forward + Dice + backward, octseg_net_train_step) as one hipGraph: ~800 launches, two streams and their event
  // edges become one launch; captured on the second call with an unchanged argument set, replayed while the set stays the same
  bool tgraph_enabled = false;
  int tgraph_seen = 0;
  hipGraphExec_t tgraph_exec = nullptr;
  struct TrainKey {
    const void *params, *grads, *buffers, *ws, *image, *target, *logits, *loss, *stats, *stream, *dropout, *drop_connect;
    int normalize; float mean[3], stdv[3], grad_scale;
    bool operator==(const TrainKey& o) const {
      return params == o.params && grads == o.grads && buffers == o.buffers && ws == o.ws && image == o.image && target == o.target &&
             logits == o.logits && loss == o.loss && stats == o.stats && stream == o.stream && dropout == o.dropout && drop_connect == o.drop_connect &&
             normalize == o.normalize && grad_scale == o.grad_scale && mean[0] == o.mean[0] && mean[1] == o.mean[1] &&
             mean[2] == o.mean[2] && stdv[0] == o.stdv[0] && stdv[1] == o.stdv[1] && stdv[2] == o.stdv[2];
    }
  } tgraph_key{};
};
