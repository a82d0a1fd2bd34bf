/* octseg.h -- C ABI of liboctseg_hip.so, the MI355X (gfx950) engine behind the OCT segmentation
 * hot path: encoder-decoder forward + backward + Dice loss + optimizer step.
 *
 * The reference (ViacheslavDanilov/oct_segmentation) has no FFI: its seam is Python.  Each entry
 * point below names the reference call it sits under (paths relative to the reference repo):
 *
 *   octseg_plan_create / _destroy      smp.create_model(arch, encoder_name, in_channels, classes)
 *                                      src/models/smp/model.py:38-44
 *   octseg_plan_set_dropout            the nn.Dropout2d inside smp's FPN decoder (arch "fpn") / the nn.Dropout of DeepLabV3+'s
 *                                      ASPP.project (arch "deeplabv3plus"): its keep pattern, injected
 *   octseg_plan_set_drop_connect       efficientnet_pytorch's drop_connect on the id skips of MBConv blocks (encoders "efficientnet-b*")
 *   octseg_plan_param_info / bn_info   the nn.Module parameter / buffer tree behind state_dict()
 *                                      (load_from_checkpoint, src/predict.py:39-48)
 *   octseg_net_forward                 OCTSegmentationModel.forward (normalize=1, model.py:65-71) and
 *                                      .predict's bare self.model(x) (normalize=0, model.py:192)
 *   octseg_dice_forward                smp.losses.DiceLoss(MULTILABEL_MODE, from_logits=True)
 *                                      (model.py:55,81,115) + smp.metrics.get_stats (utils.py:19-23)
 *   octseg_plan_set_loss               the choice of criterion at model.py:55 (the reference always builds DiceLoss; north_star also
 *                                      names BCE): Dice | torch.nn.functional.binary_cross_entropy_with_logits | their sum
 *   octseg_net_backward                loss.backward() that Lightning runs after training_step
 *                                      (model.py:73-95, train.py:130-133)
 *   octseg_net_train_step              training_step + loss.backward() as one call, optionally one replayed hipGraph (model.py:73-95)
 *   octseg_net_backward_sliced         the same under DDP: gradient buckets handed out while the backward still runs
 *                                      (train.py:122-133, devices > 1)
 *   octseg_optim_step                  configure_optimizers -> SGD|RMSprop|RAdam|Adam.step()
 *                                      (model.py:150-181)
 *   octseg_augment                     OCTDataset.get_img_augmentation applied in __getitem__ (dataset.py:119-123,160-207)
 *   octseg_mask_assemble               the per-frame epilogue of segment(): threshold, cv2 INTER_NEAREST resize to output_size,
 *                                      write into mask[:, :, CLASS_ID - 1] (src/predict.py:92-100, data/utils.py:16-33)
 *   octseg_plan_set_graph              (serving option, no reference counterpart) eval forwards of predict()
 *                                      (model.py:183-200) replayed as one hipGraph
 *   octseg_plan_params_changed         optimizer.step() / load_state_dict() side effect: weight images are stale
 *   octseg_conv2d_* / _convT_*         torch conv2d / conv_transpose2d primitives, exported so the
 *                                      parity tests can pin every kernel against torch CPU in isolation
 *   octseg_profile_* / octseg_debug_*  measurement aids of bench.py and tools/ (HIP-event brackets per launch, one-stream
 *                                      mode, s_memtime stamps in -DOCTSEG_STAMP builds); no reference counterpart
 *
 * Conventions: every function returns 0 on success or a negative octseg_status; the message of the
 * last failure on the calling thread is octseg_last_error().  Nothing throws across the ABI.  The
 * caller owns every buffer (device pointers, plain sizes); the library only enqueues kernels on the
 * caller's hipStream_t (passed as void*) and never synchronises or allocates in hot calls.  Plans
 * are thread-compatible (no concurrent calls on one plan).
 */
#ifndef OCTSEG_H
#define OCTSEG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  OCTSEG_OK = 0,
  OCTSEG_BAD_SHAPE = -1,        /* e.g. H or W not divisible by 32 (smp check_input_shape) */
  OCTSEG_BAD_DTYPE = -2,
  OCTSEG_UNSUPPORTED_ARCH = -3, /* unknown arch / encoder name */
  OCTSEG_HIP_ERROR = -4,
  OCTSEG_BAD_ARG = -5
} octseg_status;

/* OCTSEG_F16: IEEE half storage + v_mfma_f32_32x32x16_f16, the serving dtype of the ensemble path (reference src/predict.py; BASELINE
 * config #5).  Eval forwards only: octseg_net_forward(train = 1) and the backward entry points return OCTSEG_BAD_DTYPE for it. */
typedef enum { OCTSEG_F32 = 0, OCTSEG_BF16 = 1, OCTSEG_F16 = 2 } octseg_dtype;

typedef struct {
  const char* arch;     /* "unet" | "unetplusplus" | "linknet" | "fpn" | "deeplabv3plus" | "deeplabv3" | "pspnet" (case-insensitive) */
  const char* encoder;  /* "resnet18" | "resnet34" | "resnet50" | "resnet101" | "resnet152" | "timm-regnetx_002" | "timm-regnetx_064" |
                           "timm-regnety_120" | "efficientnet-b0" | "efficientnet-b5" | "efficientnet-b7" (smp encoder names) */
  int classes;          /* output channels */
  int batch, height, width;
  int dtype;            /* octseg_dtype: storage/MFMA input type of activations (accumulate is f32) */
} octseg_net_desc;

typedef struct octseg_plan octseg_plan;

/* kinds of parameter layout inside the flat fp32 parameter arena */
enum {
  OCTSEG_P_CONV = 0,   /* [R][S][O][I]  <- torch Conv2d weight [O][I][R][S]           */
  OCTSEG_P_CONVT = 1,  /* [R][S][O][I]  <- torch ConvTranspose2d weight [I][O][R][S]  */
  OCTSEG_P_STEM = 2,   /* [O][KP], k=(r*7+s)*3+ci zero padded to KP <- [O][3][7][7]   */
  OCTSEG_P_VEC = 3     /* bias / BN weight / BN bias, [O]                              */
};
typedef struct {
  char name[128];      /* state_dict key relative to the smp model, e.g. "encoder.layer1.0.conv1.weight" */
  int kind;
  int R, S, O, I, KP;
  size_t offset;       /* element offset into the parameter (and gradient) arena */
  size_t numel;
} octseg_param_info;
typedef struct {
  char name[128];      /* module path, e.g. "encoder.bn1"; buffers are <name>.running_mean / running_var */
  int C;
  size_t mean_offset, var_offset; /* element offsets into the buffer arena */
} octseg_bn_info;

int octseg_version(void);
const char* octseg_last_error(void);

int octseg_plan_create(const octseg_net_desc* desc, octseg_plan** out);
int octseg_plan_destroy(octseg_plan* plan);
size_t octseg_plan_workspace_bytes(const octseg_plan* plan);
size_t octseg_plan_param_numel(const octseg_plan* plan);   /* fp32 elements of the param / grad arenas */
size_t octseg_plan_buffer_numel(const octseg_plan* plan);  /* fp32 elements of the BN buffer arena */
int octseg_plan_num_params(const octseg_plan* plan);
int octseg_plan_param_info(const octseg_plan* plan, int index, octseg_param_info* out);
int octseg_plan_num_bn(const octseg_plan* plan);
int octseg_plan_bn_info(const octseg_plan* plan, int index, octseg_bn_info* out);
double octseg_plan_fwd_macs(const octseg_plan* plan);      /* conv multiply-accumulates of one forward */
/* multiply-accumulates a TRAINING step executes per pass: out3 = {forward, data gradient, weight gradient}.  Equal to fwd_macs unless the
 * plan runs the decoder's (nearest x2, concat, 3x3) layers as a 4x4 stride-2 transposed conv over the low-resolution map plus a 3x3 over the
 * skip channels (OCTSEG_TIED): the same function of the same weights in 16 instead of 36 products per source pixel. */
int octseg_plan_exec_macs(const octseg_plan* plan, double* out3);
/* test hook: workspace byte offsets of the raw output (NHWC, plan dtype) of conv layer `conv_name`
 * (module path, e.g. "decoder.blocks.0.conv1.0") and of its gradient; dims = {N,H,W,C}. */
int octseg_plan_find_tensor(const octseg_plan* plan, const char* conv_name, size_t* act_off,
                            size_t* grad_off, int* dims);

/* Measurement hooks (bench.py): between _start and _stop every MFMA conv launch and every BatchNorm sweep is bracketed
 * by HIP events on its launch stream.  _stop synchronises the device and fills out[12]:
 * out[3k+0..2] = {milliseconds, algorithmic work, launches} for k = 0 conv forward, 1 conv data-gradient, 2 weight
 * gradient (work = FLOPs) and k = 3 the HBM-bound BatchNorm sweeps bn_act / bn_bwd_reduce / bn_bwd_apply (work = bytes
 * every tensor they read or write once). */
int octseg_profile_start(void);
int octseg_profile_stop(double* out);

/* The forward packs the fp32 parameters into the kernels' weight images (bf16 / f32, LDS-slab order) and
 * reuses them on later calls with the same (params, workspace) pointers.  Call this after anything that
 * changes the parameter arena in place: optimizer.step(), load_state_dict(), an all-reduce of parameters. */
int octseg_plan_params_changed(octseg_plan* plan);

/* arch "fpn" (smp FPN, one of the reference's sweep architectures: configs/tune.yaml:9-18 through smp.create_model, model.py:38-44):
 * the Dropout2d(0.2) behind the merge needs a keep pattern in training -- device float [batch][128] of 0 / 1, caller-owned, read by the
 * next training forward AND its backward (kept channels are scaled by 1 / (1 - 0.2), torch's Dropout2d).  Eval forwards ignore it. */
/* arch "deeplabv3plus" (smp DeepLabV3Plus at its defaults: encoder_output_stride 16, decoder_channels 256, atrous rates (12, 24, 36);
 * same sweep, same call): the keep pattern is per ELEMENT of ASPP.project's output -- device float [batch][H/16][W/16][256] (NHWC) of
 * 0 / 1, kept elements scaled by 1 / (1 - 0.5).  A training forward with batch 1 fails like torch does ("Expected more than 1 value per
 * channel when training": the pooled ASPP branch's BatchNorm).
 * arch "deeplabv3" (smp DeepLabV3 at its defaults: output stride 8, dense ASPP): as "deeplabv3plus" with [batch][H/8][W/8][256].
 * arch "pspnet" (smp PSPNet at its defaults: encoder_depth 3, psp_out_channels 512, upsampling 8): Dropout2d(0.2) behind the fuse conv,
 * device float [batch][512] of 0 / 1.  Its parameter table still lists encoder.layer3 / layer4 (smp keeps them in state_dict): they
 * never run and their gradients are zero. */
int octseg_plan_set_dropout(octseg_plan* plan, const float* keep_dev);

/* encoder "efficientnet-b0" | "-b5" | "-b7" (efficientnet_pytorch through smp's EfficientNetEncoder; reference sweep configs/tune.yaml:25-28):
 * every MBConv block with an identity skip applies drop_connect in training -- x / (1 - rate) * floor(1 - rate + U[0, 1)) per sample, rate =
 * 0.2 * block index / blocks.  The caller draws the decisions and hands over the FACTORS: device float [octseg_plan_num_drop_connect()][batch]
 * of 0 or 1 / (1 - octseg_plan_drop_connect_rate(i)), caller-owned, read by the next training forward AND its backward.  Eval ignores it. */
int octseg_plan_set_drop_connect(octseg_plan* plan, const float* factors_dev);
int octseg_plan_num_drop_connect(const octseg_plan* plan);
float octseg_plan_drop_connect_rate(const octseg_plan* plan, int index);

/* Serving path (reference: src/models/smp/predict.py segment(), model.py:183-200 predict()): enable = 1 makes every
 * eval-mode octseg_net_forward of this plan run as a hipGraph -- the first call with a given argument set runs
 * eagerly, the second is captured, later ones replay it (one launch instead of ~400) for as long as the pointers,
 * the stream and the normalisation constants stay the same.  Keep the image / logits in persistent buffers. */
int octseg_plan_set_graph(octseg_plan* plan, int enable);

/* image: NCHW f32 [B,3,H,W]; logits: NCHW f32 [B,classes,H,W]; mean/std: 3 host floats (normalize=1).
 * train=1: batch statistics, running buffers updated, activations kept for backward -- and `image` itself must stay valid and
 * unchanged until that backward has been enqueued: the stem's weight gradient gathers the frame again instead of saving an im2col copy. */
int octseg_net_forward(octseg_plan* plan, const float* params, float* buffers, void* workspace,
                       const float* image, float* logits, int normalize, const float* mean,
                       const float* stdv, int train, void* stream);

/* Training augmentation on the GPU (reference src/models/smp/dataset.py:160-207: HorizontalFlip, ShiftScaleRotate, RandomCrop +
 * PadIfNeeded, GaussNoise, Perspective, RandomBrightnessContrast, HueSaturationValue).  The host draws the per-frame
 * decisions and parameters (oct_segmentation_amd/augment.py mirrors the reference's probabilities and ranges) and passes
 * OCTSEG_AUG_NPARAM floats per frame: [0..8] inverse homography (output pixel -> source pixel), [9] contrast alpha,
 * [10] brightness beta (x 255), [11] noise sigma, [12] seed bits, [13..15] hue / saturation / value shifts in OpenCV
 * 8-bit units, [16] flags (bit 0: HSV shift on), [20..28] inverse homography output pixel -> frame after crop + pad,
 * [29..32] crop window [x_lo, y_lo, x_hi, y_hi) in that frame (outside = padding = 0).  One bilinear gather of the image (constant-0 border), one nearest
 * gather per mask channel, photometric ops on the pixel, result clipped and rounded to the uint8 grid. */
#define OCTSEG_AUG_NPARAM 36
int octseg_augment(const float* img, const float* mask, float* img_out, float* mask_out, const float* params, int B,
                   int classes, int H, int W, void* stream);

/* Serving epilogue (reference src/predict.py:92-100): out[n][y][x][out_ch] = sigmoid(logits[n][ch]) > 0.5 after a nearest
 * resize from H x W to out_h x out_w.  logits: NCHW f32 [N,classes,H,W]; out: NHWC f32 [N,out_h,out_w,out_channels] (the
 * reference's 4-channel mask stack, channel = CLASS_ID - 1).  row_index[out_h] / col_index[out_w]: device int32 source
 * index of every output row / column -- the host mirror fills them with OpenCV's INTER_NEAREST rule (resizeNN:
 * min(floor(i * (1 / (out / in))), in - 1), what the reference's cv2.resize call computes); null = floor((i + 0.5) * H / out_h). */
int octseg_mask_assemble(const float* logits, int N, int classes, int H, int W, int ch, float* out, int out_h, int out_w,
                         int out_channels, int out_ch, const int* row_index, const int* col_index, void* stream);

/* Criterion evaluated by octseg_dice_forward / octseg_net_train_step and differentiated by the backward entry points.
 * OCTSEG_LOSS_DICE (default) = smp.losses.DiceLoss(MULTILABEL_MODE, from_logits=True), the reference's (model.py:55);
 * OCTSEG_LOSS_BCE = torch.nn.functional.binary_cross_entropy_with_logits(logits, target) (reduction 'mean' over every element);
 * OCTSEG_LOSS_DICE_BCE = their unweighted sum.  All three come out of the ONE pass over logits / target that also counts tp/fp/fn/tn. */
typedef enum { OCTSEG_LOSS_DICE = 0, OCTSEG_LOSS_BCE = 1, OCTSEG_LOSS_DICE_BCE = 2 } octseg_loss_kind;
int octseg_plan_set_loss(octseg_plan* plan, int kind);

/* loss: device f32 scalar; stats: device int64 [B][classes][4] = tp, fp, fn, tn (nullable). */
int octseg_dice_forward(octseg_plan* plan, void* workspace, const float* logits, const float* target,
                        float* loss, long long* stats, void* stream);

/* Must follow octseg_net_forward(train=1) + octseg_dice_forward on the same workspace.
 * grads (fp32 arena, same layout as params) is overwritten with d(grad_scale * loss)/dparams. */
int octseg_net_backward(octseg_plan* plan, const float* params, float* grads, void* workspace,
                        const float* logits, const float* target, float grad_scale, void* stream);

/* Forward (train) + Dice + backward in ONE call (reference: training_step + loss.backward(), src/models/smp/model.py:73-95 under Lightning).
 * Same launches as octseg_net_forward(train = 1) -> octseg_dice_forward -> octseg_net_backward.  octseg_plan_set_train_graph(plan, 1):
 * the call is captured into a hipGraph on its second use with an unchanged argument set (every pointer, the stream, the constants) and
 * replayed afterwards -- one launch per step instead of ~800 (keep image / target / logits / loss / stats in persistent buffers).
 * Not available together with the sliced (data-parallel) backward. */
int octseg_net_train_step(octseg_plan* plan, const float* params, float* grads, float* buffers, void* workspace, const float* image,
                          const float* target, float* logits, float* loss, long long* stats, int normalize, const float* mean,
                          const float* stdv, float grad_scale, void* stream);
int octseg_plan_set_train_graph(octseg_plan* plan, int enable);

/* Data-parallel variant (reference: torch DDP's bucketed gradient all-reduce overlapped with backward, which Lightning installs for
 * src/models/smp/train.py:122-133 when more than one GPU is visible).  Same launches; the gradient arena is cut into
 * `nslices` contiguous parameter-aligned ranges and cb(user, k, begin, end) -- element offsets into grads -- is called on the
 * calling host thread as soon as the last launch writing into slice k is enqueued; comm_stream (not the compute stream) has by
 * then been made to wait for those launches, so the collective the callback enqueues there overlaps the rest of the backward.
 * Slices are reported exactly once each, in completion order (decoder / head ranges first). */
typedef void (*octseg_slice_cb)(void* user, int slice, size_t begin, size_t end);
int octseg_net_backward_sliced(octseg_plan* plan, const float* params, float* grads, void* workspace, const float* logits,
                               const float* target, float grad_scale, void* stream, int nslices, void* comm_stream,
                               octseg_slice_cb cb, void* user);

/* kind: 0 SGD, 1 Adam, 2 RMSprop, 3 RAdam (torch defaults for everything not listed).
 * state_m / state_v: fp32 arenas of numel elements (may be NULL when unused by the kind). */
int octseg_optim_step(int kind, float* params, const float* grads, float* state_m, float* state_v,
                      size_t numel, float lr, float weight_decay, int step, float grad_scale,
                      void* stream);

/* Deterministic-reduction mode (also OCTSEG_DETERMINISTIC=1 in the environment): weight gradients without split-K atomics, Dice sums and
 * bias gradients by one workgroup per output -- two runs of the same step give bit-identical losses and gradients (the default mode
 * orders its floating-point atomics by arrival).  Slower; for tests and debugging (torch.use_deterministic_algorithms' counterpart). */
int octseg_set_deterministic(int on);

/* diagnostic hook: with a library built with -DOCTSEG_STAMP, octseg_conv2d_forward adds per-phase cycle
 * sums of the tap loop into dev_buf[6] (u64, device); a no-op in the shipped build. */
int octseg_debug_set_stamp(unsigned long long* dev_buf);
/* Measurement aid: on = 1 runs every launch of forward and backward on the caller's stream, one after the other
 * (no forward lanes, no weight-gradient side stream), so that per-kernel durations are those of the kernels alone. */
int octseg_debug_set_serial(int on);

/* ---- single-op entry points (NHWC device tensors of `dtype`; weights fp32 in arena layout) ---- */
/* y[N,OH,OW,Cout] = conv(x[N,H,W,Cin], w[R][S][Cout][Cin]) (+bias);  transposed=1: ConvTranspose2d
 * 4x4 s2 p1 with w[R][S][Cout][Cin].  scratch: device bytes >= octseg_conv2d_scratch_bytes(). */
size_t octseg_conv2d_scratch_bytes(int dtype, int N, int H, int W, int Cin, int Cout, int R, int S);
int octseg_conv2d_forward(int dtype, const void* x, const float* w, const float* bias, void* y, int N,
                          int H, int W, int Cin, int Cout, int R, int S, int stride, int pad,
                          int transposed, void* scratch, void* stream);
int octseg_conv2d_backward_data(int dtype, const void* dy, const float* w, void* dx, int N, int H, int W,
                                int Cin, int Cout, int R, int S, int stride, int pad, int transposed,
                                void* scratch, void* stream);
int octseg_conv2d_backward_weight(int dtype, const void* x, const void* dy, float* dw, int N, int H,
                                  int W, int Cin, int Cout, int R, int S, int stride, int pad,
                                  int transposed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCTSEG_H */
