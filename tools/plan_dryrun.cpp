// plan_dryrun.cpp -- host-only walk of the plan builder under AddressSanitizer + UBSan ("race detection / sanitizers").
//
// Built by `make -C oct_segmentation_amd/csrc asan` from the HOST halves of every source (hipcc --cuda-host-only
// -fsanitize=address,undefined): graph construction, workspace layout, tap tables, launch geometry, weight-image layouts, BN slab row
// counts and the one-launch pack / BN job tables of csrc/plan.cpp -- ~1400 lines of index arithmetic -- run for every architecture x
// encoder pair at the smallest legal frame, the benchmark frame and a non-square one, in all three dtypes, with no kernel launched
// (no GPU needed).  A subset of the plans is then driven through octseg_net_forward / _backward / _backward_sliced / octseg_optim_step and
// the graph-captured eval forward against tools/hip_host_stubs.cpp, a recording stand-in for the HIP runtime that checks every launch
// geometry and every memset / copy range.  Exit code 0 = the sanitizers and the stubs saw nothing.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/octseg.h"

// tools/hip_host_stubs.cpp: the recording HIP stand-in this binary links instead of libamdhip64
extern "C" void dry_register_range(const void* lo, size_t bytes);
extern "C" void dry_clear_ranges();
extern "C" unsigned long long dry_launches();
extern "C" unsigned long long dry_memops();
extern "C" unsigned long long dry_errors();

namespace {
int g_slices_seen = 0;
size_t g_slice_cover = 0;
void slice_cb(void*, int, size_t b, size_t e) { ++g_slices_seen; g_slice_cover += e - b; }

// One training step + eval forwards of a plan with every pointer inside fake, registered address ranges (never dereferenced: all device
// work is behind the stubs).  Walks run_forward / run_backward: source and destination descriptors, launch routing, split-K geometry,
// slab rows, stream forks and joins, slice bookkeeping.
int exercise(const octseg_net_desc& d, octseg_plan* p) {
  char* base = (char*)(uintptr_t)0x100000000000ull;
  auto carve = [&](size_t bytes) { char* r = base; base += (bytes + 4095) / 4096 * 4096 + 4096; dry_register_range(r, bytes); return r; };
  dry_clear_ranges();
  const size_t np = octseg_plan_param_numel(p), nb = octseg_plan_buffer_numel(p), ws = octseg_plan_workspace_bytes(p);
  const size_t px = (size_t)d.batch * d.height * d.width;
  float* params = (float*)carve(np * 4); float* grads = (float*)carve(np * 4); float* bufs = (float*)carve(nb * 4);
  float* m = (float*)carve(np * 4); float* v = (float*)carve(np * 4);
  void* wsp = carve(ws);
  float* image = (float*)carve(px * 3 * 4); float* logits = (float*)carve(px * d.classes * 4); float* target = (float*)carve(px * d.classes * 4);
  float* loss = (float*)carve(4); long long* stats = (long long*)carve((size_t)d.batch * d.classes * 4 * 8);
  const int dls = !strcmp(d.arch, "deeplabv3plus") ? 16 : !strcmp(d.arch, "deeplabv3") ? 8 : 0;   // element-wise [B][H/s][W/s][256]; fpn [B][128], pspnet [B][512]
  float* keep = (float*)carve(dls ? (size_t)d.batch * (d.height / dls) * (d.width / dls) * 256 * 4 : (size_t)d.batch * 512 * 4);
  octseg_plan_set_dropout(p, keep);                 // (ignored by the other architectures)
  float* dcf = (float*)carve((size_t)(octseg_plan_num_drop_connect(p) + 1) * d.batch * 4);
  octseg_plan_set_drop_connect(p, dcf);             // EfficientNet encoders: drop_connect factors of the id skips
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  void* st = (void*)(uintptr_t)0x4000; void* comm = (void*)(uintptr_t)0x4100;
  const unsigned long long l0 = dry_launches();
  if (d.dtype != OCTSEG_F16) {
    if (octseg_net_forward(p, params, bufs, wsp, image, logits, 1, mean, stdv, 1, st) != 0) { fprintf(stderr, "forward: %s\n", octseg_last_error()); return 10; }
    if (octseg_dice_forward(p, wsp, logits, target, loss, stats, st) != 0) return 11;
    if (octseg_net_backward(p, params, grads, wsp, logits, target, 1.0f, st) != 0) { fprintf(stderr, "backward: %s\n", octseg_last_error()); return 12; }
    for (int ns : {1, 3, 7}) {
      g_slices_seen = 0; g_slice_cover = 0;
      if (octseg_net_backward_sliced(p, params, grads, wsp, logits, target, 0.5f, st, ns, comm, slice_cb, nullptr) != 0) return 13;
      if (g_slice_cover != np || g_slices_seen < 1 || g_slices_seen > ns) { fprintf(stderr, "slices do not tile the arena (%d slices, %zu of %zu)\n", g_slices_seen, g_slice_cover, np); return 14; }
    }
    for (int kind = 0; kind < 4; ++kind)
      if (octseg_optim_step(kind, params, grads, m, v, np, 1e-3f, 1e-4f, 1, 1.0f, st) != 0) return 15;
    octseg_plan_params_changed(p);
    octseg_debug_set_serial(1);                      // the one-stream measurement mode takes other branches
    if (octseg_net_forward(p, params, bufs, wsp, image, logits, 1, mean, stdv, 1, st) != 0) return 16;
    if (octseg_net_backward(p, params, grads, wsp, logits, target, 1.0f, st) != 0) return 16;
    octseg_debug_set_serial(0);
    octseg_set_deterministic(1);                     // ... and so does the deterministic-reduction mode
    if (octseg_net_forward(p, params, bufs, wsp, image, logits, 1, mean, stdv, 1, st) != 0) return 17;
    if (octseg_dice_forward(p, wsp, logits, target, loss, nullptr, st) != 0) return 17;
    if (octseg_net_backward(p, params, grads, wsp, logits, target, 1.0f, st) != 0) return 17;
    octseg_set_deterministic(0);
  } else {
    if (octseg_net_forward(p, params, bufs, wsp, image, logits, 1, mean, stdv, 1, st) == 0) return 18;      // f16 trains nowhere
    if (octseg_net_backward(p, params, grads, wsp, logits, target, 1.0f, st) == 0) return 18;
  }
  if (octseg_net_forward(p, params, bufs, wsp, image, logits, 0, nullptr, nullptr, 0, st) != 0) { fprintf(stderr, "eval forward: %s\n", octseg_last_error()); return 19; }
  octseg_plan_set_graph(p, 1);                       // eager call, capture, replay
  for (int k = 0; k < 3; ++k)
    if (octseg_net_forward(p, params, bufs, wsp, image, logits, 0, nullptr, nullptr, 0, st) != 0) return 20;
  octseg_plan_set_graph(p, 0);
  if (dry_launches() - l0 < 50) { fprintf(stderr, "suspiciously few launches\n"); return 21; }
  return 0;
}
}  // namespace

int main() {
  const char* archs[] = {"unet", "unetplusplus", "linknet", "fpn", "deeplabv3plus", "pspnet", "deeplabv3", "manet", "pan"};
  const char* encs[] = {"resnet18", "resnet34", "resnet50", "resnet101", "resnet152", "timm-regnetx_002", "timm-regnetx_064", "timm-regnety_120",
                        "efficientnet-b0", "efficientnet-b5", "efficientnet-b7"};
  const int shapes[][3] = {{1, 32, 32}, {16, 704, 704}, {3, 96, 64}, {2, 64, 160}};      // (PAN needs >= 128 x 128 to run: its frames are doubled below)
  int plans = 0, executed = 0;
  unsigned long long checksum = 0;
  for (const char* arch : archs)
    for (const char* enc : encs)
      for (auto& sh : shapes)
        for (int dtype = 0; dtype < 3; ++dtype)
          for (int classes : {1, 4}) {
            if (classes == 4 && !(sh[1] == 96 || dtype == 1)) continue;   // (the class count only changes the head: a subset is enough)
            // Encoders outside the ResNets (timm RegNet: grouped convs as per-group layers on channel slices; EfficientNet: MBConv blocks):
            // built under U-Net, U-Net++ and FPN always; LinkNet / PSPNet only where a QUARTER of the feature widths is a multiple of the
            // 8-channel vector (RegNetY-120 both, EfficientNet-B5 PSPNet); the dilated DeepLab encoders never.  The builder refuses the rest
            // with OCTSEG_UNSUPPORTED_ARCH (checked here for every refused pair).
            const bool other_enc = !strncmp(enc, "timm-", 5) || !strncmp(enc, "efficientnet-", 13);
            if (other_enc && !strcmp(arch, "pan")) {        // PAN dilates its encoder: ResNets only
              octseg_net_desc dr{arch, enc, classes, sh[0], sh[1], sh[2], dtype};
              octseg_plan* pr = nullptr;
              if (octseg_plan_create(&dr, &pr) != OCTSEG_UNSUPPORTED_ARCH || pr) { fprintf(stderr, "pan over %s was not refused\n", enc); return 7; }
              continue;
            }
            if (other_enc && (!strcmp(arch, "linknet") || !strcmp(arch, "pspnet") || !strncmp(arch, "deeplab", 7))) {
              const bool ok_pair = (!strcmp(enc, "timm-regnety_120") && strncmp(arch, "deeplab", 7)) || (!strcmp(enc, "efficientnet-b5") && !strcmp(arch, "pspnet"));
              if (!ok_pair) {
                octseg_net_desc dr{arch, enc, classes, sh[0], sh[1], sh[2], dtype};
                octseg_plan* pr = nullptr;
                if (octseg_plan_create(&dr, &pr) != OCTSEG_UNSUPPORTED_ARCH || pr) { fprintf(stderr, "%s over %s was not refused\n", arch, enc); return 7; }
                continue;
              }
            }
            const int fs = (!strcmp(arch, "pan") && sh[1] < 128) ? 2 : 1;
            octseg_net_desc d{arch, enc, classes, sh[0], sh[1] * fs, sh[2] * fs, dtype};
            octseg_plan* p = nullptr;
            if (octseg_plan_create(&d, &p) != 0 || !p) {
              fprintf(stderr, "plan_create(%s, %s, B=%d %dx%d, dtype %d) failed: %s\n", arch, enc, sh[0], sh[1], sh[2], dtype, octseg_last_error());
              return 2;
            }
            ++plans;
            const size_t ws = octseg_plan_workspace_bytes(p), np = octseg_plan_param_numel(p), nb = octseg_plan_buffer_numel(p);
            checksum += ws % 1000003 + np + nb;
            size_t covered = 0;
            std::string first_conv;
            for (int i = 0; i < octseg_plan_num_params(p); ++i) {
              octseg_param_info pi;
              if (octseg_plan_param_info(p, i, &pi) != 0) return 3;
              if (pi.offset + pi.numel > np) { fprintf(stderr, "%s: parameter %s leaves the arena\n", arch, pi.name); return 4; }
              if (pi.offset < covered) { fprintf(stderr, "%s: parameter %s overlaps its predecessor\n", arch, pi.name); return 4; }
              covered = pi.offset + pi.numel;
              const size_t n = strlen(pi.name);
              if (first_conv.empty() && pi.kind == OCTSEG_P_CONV && n > 7 && strcmp(pi.name + n - 7, ".weight") == 0)
                first_conv.assign(pi.name, n - 7);
            }
            for (int i = 0; i < octseg_plan_num_bn(p); ++i) {
              octseg_bn_info bi;
              if (octseg_plan_bn_info(p, i, &bi) != 0) return 3;
              if (bi.mean_offset + bi.C > nb || bi.var_offset + bi.C > nb) { fprintf(stderr, "BN %s leaves the buffer arena\n", bi.name); return 4; }
            }
            size_t ao = 0, go = 0; int dims[4] = {0, 0, 0, 0};
            if (!first_conv.empty() && octseg_plan_find_tensor(p, first_conv.c_str(), &ao, &go, dims) == 0) {
              const size_t bytes = (size_t)dims[0] * dims[1] * dims[2] * dims[3] * (dtype == 0 ? 4 : 2);
              if (ao + bytes > ws || go + bytes > ws) { fprintf(stderr, "%s: tensor of %s leaves the workspace\n", arch, first_conv.c_str()); return 4; }
            }
            if (octseg_plan_fwd_macs(p) <= 0) return 5;
            // error paths: out-of-range queries must fail cleanly
            octseg_param_info bad;
            if (octseg_plan_param_info(p, -1, &bad) == 0 || octseg_plan_param_info(p, octseg_plan_num_params(p), &bad) == 0) return 6;
            if (octseg_plan_find_tensor(p, "no.such.layer", nullptr, nullptr, nullptr) == 0) return 6;
            octseg_plan_params_changed(p);
            // the executors: every pair on the small non-square frame (all dtypes), the three BASELINE training configs at full size
            const bool base_cfg = sh[1] == 704 && dtype == 1 && classes == 1 &&
                                  ((!strcmp(arch, "unetplusplus") && !strcmp(enc, "resnet101")) || (!strcmp(arch, "linknet") && !strcmp(enc, "resnet50")) ||
                                   (!strcmp(arch, "unet") && !strcmp(enc, "resnet50")));
            if (sh[1] == 64 || base_cfg || (sh[1] == 96 && classes == 4)) {
              const int rc = exercise(d, p);
              if (rc) { fprintf(stderr, "exercise(%s, %s, B=%d %dx%d, dtype %d) -> %d\n", arch, enc, sh[0], sh[1], sh[2], dtype, rc); return rc; }
              ++executed;
            }
            octseg_plan_destroy(p);
          }
  // shapes the builder must refuse (smp check_input_shape / argument checks) without touching memory it does not own
  octseg_plan* q = nullptr;
  octseg_net_desc badr1{"linknet", "timm-regnetx_002", 1, 1, 32, 32, 0}, badr2{"deeplabv3plus", "timm-regnetx_064", 1, 2, 64, 64, 1},
      badr3{"pspnet", "timm-regnetx_002", 1, 2, 64, 64, 0};
  if (octseg_plan_create(&badr1, &q) == 0 || octseg_plan_create(&badr2, &q) == 0 || octseg_plan_create(&badr3, &q) == 0) { fprintf(stderr, "an unsupported RegNet pair was accepted\n"); return 7; }
  octseg_net_desc bad1{"unet", "resnet18", 1, 1, 48, 64, 0}, bad2{"segformer", "resnet18", 1, 1, 32, 32, 0}, bad3{"unet", "vgg", 1, 1, 32, 32, 0},
      bad4{"unet", "resnet18", 0, 1, 32, 32, 0}, bad5{"unet", "resnet18", 1, 1, 32, 32, 7};
  for (octseg_net_desc* b : {&bad1, &bad2, &bad3, &bad4, &bad5})
    if (octseg_plan_create(b, &q) == 0) { fprintf(stderr, "a bad descriptor was accepted\n"); return 7; }
  for (int dt = 0; dt < 3; ++dt) checksum += octseg_conv2d_scratch_bytes(dt, 2, 64, 64, 24, 40, 3, 3) + octseg_conv2d_scratch_bytes(dt, 1, 8, 8, 2048, 512, 1, 1);
  if (dry_errors() != 0) { fprintf(stderr, "%llu launch / memory-range violations\n", dry_errors()); return 8; }
  printf("plan_dryrun: %d plans built, %d of them run through forward / backward / optimizer with recording HIP stubs (%llu launches, %llu memory "
         "operations checked) under ASan + UBSan, checksum %llu, 0 errors\n", plans, executed, dry_launches(), dry_memops(), checksum);
  return 0;
}
