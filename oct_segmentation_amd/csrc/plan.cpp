// plan.cpp -- graph builder (torchvision ResNet encoders + smp Unet / UnetPlusPlus / Linknet
// decoders), workspace planner and the forward / backward executors behind the C ABI.
//
// The graph is built once per (arch, encoder, classes, B, H, W, dtype).  Convolutions never see
// a materialised concat / upsample / BN-apply: a consumer reads `Value`s = (raw tensor, BN id)
// and applies relu(x*scale+shift) while staging (conv_mfma.hip).  Only residual-block outputs
// (and LinkNet skip sums) are materialised by bn_act.  Module / parameter names reproduce the
// smp 0.3.3 + torchvision module tree so that the Python facade can serve a reference
// state_dict (reference src/predict.py:39-48).
#include "plan.h"

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

using namespace octseg;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return fail(OCTSEG_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)

static size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- in-process kernel timing
// bench.py brackets every MFMA launch with HIP events on the launch stream (octseg_profile_start /
// _stop); classes: 0 conv forward, 1 conv data-gradient, 2 weight gradient.
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double flops; std::string name; };
bool g_prof_on = false;
bool g_capturing = false;   // a stream capture is in progress on this thread's call: no timing events inside it
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
static bool g_prof_hbm = false;   // set with octseg_debug_set_serial
struct ProfScope {
  hipStream_t st; ProfRec r; bool on;
  ProfScope(int kind, double flops, hipStream_t s, const std::string& name = std::string()) : st(s), on(g_prof_on && !g_capturing) {
    if (kind == 3 && !g_prof_hbm) on = false;   // the BatchNorm sweeps are only bracketed in the one-stream measurement pass
    if (!on) return;
    r.kind = kind; r.flops = flops; r.name = name; r.a = prof_event(); r.b = prof_event();
    if (!r.a || !r.b) { on = false; return; }
    (void)hipEventRecord(r.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, st);
    g_prof.push_back(r);
  }
};
}  // namespace

// ================================================================ tap tables / launch geometry
static void set_taps(int* tdy, int* tdx, int* tw, int n, int& ntaps, int& min_dy,
                     int& min_dx, int& span_y, int& span_x) {
  ntaps = n;
  if (n == 0) { min_dy = min_dx = 0; span_y = span_x = 1; return; }
  int mny = 127, mnx = 127, mxy = -127, mxx = -127;
  for (int i = 0; i < n; ++i) {
    mny = std::min(mny, (int)tdy[i]); mxy = std::max(mxy, (int)tdy[i]);
    mnx = std::min(mnx, (int)tdx[i]); mxx = std::max(mxx, (int)tdx[i]);
  }
  (void)tw;
  min_dy = mny; min_dx = mnx; span_y = mxy - mny + 1; span_x = mxx - mnx + 1;
}

struct Geom {
  int R, S, stride, pad;
  bool transposed;
  int N, IH, IW, Cin, OH, OW, Cout;
};

// forward launches: taps + grid + output mapping (sources / weights / destinations filled by caller)
static void fwd_launches(const Geom& g, std::vector<ConvArgs>& out) {
  if (!g.transposed) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    int n = 0;
    for (int r = 0; r < g.R; ++r)
      for (int s = 0; s < g.S; ++s) { a.tap_dy[n] = r - g.pad; a.tap_dx[n] = s - g.pad; a.tap_w[n] = r * g.S + s; ++n; }
    set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
    a.istride = g.stride; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.OH; a.OW = g.OW;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ostride = 1; a.ooy = a.oox = 0;
    out.push_back(a);
  } else {  // ConvTranspose2d k4 s2 p1: one launch per output parity, 2x2 taps each
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        int n = 0;
        for (int r = 0; r < g.R; ++r) {
          if (((py + g.pad - r) & 1) != 0) continue;
          for (int s = 0; s < g.S; ++s) {
            if (((px + g.pad - s) & 1) != 0) continue;
            a.tap_dy[n] = (py + g.pad - r) / 2; a.tap_dx[n] = (px + g.pad - s) / 2; a.tap_w[n] = r * g.S + s; ++n;
          }
        }
        set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
        a.istride = 1; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.IH; a.OW = g.IW;  // grid = input grid
        a.Cin = g.Cin; a.Cout = g.Cout; a.ostride = 2; a.ooy = py; a.oox = px;
        out.push_back(a);
      }
  }
}

// data-gradient launches: "input" is dy [N,OH,OW,Cout], "output" is dx [N,IH,IW,Cin]
static void dgrad_launches(const Geom& g, std::vector<ConvArgs>& out) {
  if (g.transposed) {  // gradient of ConvT = plain conv k4 s2 p1 over dy
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    int n = 0;
    for (int r = 0; r < g.R; ++r)
      for (int s = 0; s < g.S; ++s) { a.tap_dy[n] = r - g.pad; a.tap_dx[n] = s - g.pad; a.tap_w[n] = r * g.S + s; ++n; }
    set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
    a.istride = 2; a.N = g.N; a.IH = g.OH; a.IW = g.OW; a.OH = g.IH; a.OW = g.IW;
    a.Cin = g.Cout; a.Cout = g.Cin; a.ostride = 1;
    out.push_back(a);
  } else if (g.stride == 1) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    int n = 0;
    for (int r = 0; r < g.R; ++r)
      for (int s = 0; s < g.S; ++s) { a.tap_dy[n] = g.pad - r; a.tap_dx[n] = g.pad - s; a.tap_w[n] = r * g.S + s; ++n; }
    set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
    a.istride = 1; a.N = g.N; a.IH = g.OH; a.IW = g.OW; a.OH = g.IH; a.OW = g.IW;
    a.Cin = g.Cout; a.Cout = g.Cin; a.ostride = 1;
    out.push_back(a);
  } else {  // stride 2: one launch per input parity
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        int n = 0;
        for (int r = 0; r < g.R; ++r) {
          if (((py + g.pad - r) & 1) != 0) continue;
          for (int s = 0; s < g.S; ++s) {
            if (((px + g.pad - s) & 1) != 0) continue;
            a.tap_dy[n] = (py + g.pad - r) / 2; a.tap_dx[n] = (px + g.pad - s) / 2; a.tap_w[n] = r * g.S + s; ++n;
          }
        }
        set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
        a.istride = 1; a.N = g.N; a.IH = g.OH; a.IW = g.OW;
        a.OH = (g.IH - py + 1) / 2; a.OW = (g.IW - px + 1) / 2;
        a.Cin = g.Cout; a.Cout = g.Cin; a.ostride = 2; a.ooy = py; a.oox = px;
        out.push_back(a);
      }
  }
}

static void wgrad_launches(const Geom& g, std::vector<WgradArgs>& out) {
  if (!g.transposed) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    int n = 0;
    for (int r = 0; r < g.R; ++r)
      for (int s = 0; s < g.S; ++s) { a.tap_dy[n] = r - g.pad; a.tap_dx[n] = s - g.pad; a.tap_w[n] = r * g.S + s; ++n; }
    set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
    a.istride = g.stride; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.OH; a.OW = g.OW;
    a.Cin = g.Cin; a.Cout = g.Cout; a.DH = g.OH; a.DW = g.OW; a.dstride = 1;
    out.push_back(a);
  } else {
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        WgradArgs a;
        memset(&a, 0, sizeof(a));
        int n = 0;
        for (int r = 0; r < g.R; ++r) {
          if (((py + g.pad - r) & 1) != 0) continue;
          for (int s = 0; s < g.S; ++s) {
            if (((px + g.pad - s) & 1) != 0) continue;
            a.tap_dy[n] = (py + g.pad - r) / 2; a.tap_dx[n] = (px + g.pad - s) / 2; a.tap_w[n] = r * g.S + s; ++n;
          }
        }
        set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
        a.istride = 1; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.IH; a.OW = g.IW;
        a.Cin = g.Cin; a.Cout = g.Cout; a.DH = g.OH; a.DW = g.OW; a.dstride = 2; a.doy = py; a.dox = px;
        out.push_back(a);
      }
  }
}

// Data gradient of the tied ConvTranspose2d (Geom of the ConvT: IH x IW = the low-resolution map): the adjoint of each forward parity launch is a
// stride-1 2x2-tap conv over ONE parity plane of dy (the caller presents the plane as a tensor of its own: pointer offset, doubled pixel and row
// strides), all four accumulating into the low-resolution gradient.  Four launches of 4 taps over 176^2 planes instead of one 16-tap stride-2
// launch whose 34 x 34 window does not fit a double-buffered LDS tile.  out[k] belongs to parity (k >> 1, k & 1).
static void tied_dgrad_launches(const Geom& g, std::vector<ConvArgs>& out) {
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      int n = 0;
      for (int r = 0; r < g.R; ++r) {
        if (((py + g.pad - r) & 1) != 0) continue;
        for (int s = 0; s < g.S; ++s) {
          if (((px + g.pad - s) & 1) != 0) continue;
          a.tap_dy[n] = -((py + g.pad - r) / 2); a.tap_dx[n] = -((px + g.pad - s) / 2); a.tap_w[n] = r * g.S + s; ++n;
        }
      }
      set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
      a.istride = 1; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.IH; a.OW = g.IW;
      a.Cin = g.Cout; a.Cout = g.Cin; a.ostride = 1;
      out.push_back(a);
    }
}

static double layer_macs(const ConvLayer& L);
// A/B switch: the tied data gradient as four 2x2-tap launches over dy's parity planes instead of one 16-tap stride-2 launch over dy
static bool tie_dgrad_planes() { return getenv("OCTSEG_TIED_DGRAD_PLANES") != nullptr; }

// ... and as ONE stride-1 launch whose four sources are dy's parity planes (virtual channels [p O, (p + 1) O) = plane p), every plane contracting
// with its own 2 x 2 of the 3 x 3 tap offsets (ConvArgs::taps_per_src, conv_mfma.hip's masked loop): plane (py, px) at offset (dy, dx) carries
// kernel tap r = py + 1 + 2 dy, s = px + 1 + 2 dx where that lies in [0, 3].  Sources (pointers, strides) are filled by the caller.
static void tied_dgrad_masked(const Geom& g, ConvArgs& a) {
  memset(&a, 0, sizeof(a));
  int n = 0;
  for (int t = 0; t < 9; ++t) { a.tap_dy[n] = t / 3 - 1; a.tap_dx[n] = t % 3 - 1; a.tap_w[n] = t; ++n; }
  set_taps(a.tap_dy, a.tap_dx, a.tap_w, n, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x);
  a.istride = 1; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.IH; a.OW = g.IW;
  a.Cin = 4 * g.Cout; a.Cout = g.Cin; a.ostride = 1;
  a.taps_per_src = 4; a.nsrc = 4;
  for (int p = 0; p < 4; ++p) {
    const int py = p >> 1, px = p & 1;
    int list = 0, k = 0;
    for (int t = 0; t < 9; ++t) {
      const int r = py + 1 + 2 * (t / 3 - 1), s = px + 1 + 2 * (t % 3 - 1);
      if (r >= 0 && r <= 3 && s >= 0 && s <= 3) list |= t << (4 * k++);
    }
    a.src_taps[p] = list;
    SrcDesc d{};
    d.C = 2 * g.Cout; d.c0 = p * g.Cout; d.H = g.IH; d.W = 2 * g.IW;
    a.src[p] = d;
  }
}

// OCTSEG_TIED=[f][d][w]: which passes of the decoder's (nearest x2, concat, 3x3) layers run the tied decomposition (ConvLayer::tie).  Default `dw`.
// The weight gradient is the same fp32 sum of the same bf16 products in another order, 16 instead of 36 of them per source pixel; the data
// gradient contracts dy with the 4x4 image (sums of taps rounded to bf16 once) at the low resolution instead of with nine taps at the high one
// followed by a 2x2 pool.  U-Net++/resnet101 16 x 704^2: weight-gradient class 21.4 -> 18.4 ms, data-gradient class 21.6 -> 20.1 ms
// (profiles/r4_tied_ab.txt).  `f` is exact up to bf16 rounding too but not faster: four parity launches that add into the output plus a sweep
// for the BatchNorm statistics -- opt-in.  OCTSEG_TIED=0 (or any string without f / d / w): none.
static int tie_mask() {   // (read when a plan is built, so that one process can hold plans of both kinds)
  const char* e = getenv("OCTSEG_TIED");
  if (e == nullptr) return 2 | 4;
  int v = 0;
  for (const char* c = e; *c; ++c) v |= *c == 'f' ? 1 : *c == 'd' ? 2 : *c == 'w' ? 4 : 0;
  return v;
}

// ================================================================ graph builder
// Side streams carry work that is off the critical chain: the second forward lane (default priority -- at low priority the fp16 ensemble's
// B = 1 replay, whose lanes ARE its critical path, fell from 148 to 121 frames/s) and the backward's weight gradients.  The latter's stream is
// created with the LOWEST priority: the chain's kernels win the dispatch whenever compute units free up, the weight gradients fill what is
// left (round 4, ABAB on one box, eager steps: 67.75-68.0 ms per step against 68.2-68.3 at the default priority, 69.1 at the highest; under
// graph replay round 3 saw no difference).  OCTSEG_SIDE_PRIORITY=normal|high|low sets both (A/B switch).
static hipError_t create_side_stream(hipStream_t* st, bool backward = false) {
  static const char* pr = getenv("OCTSEG_SIDE_PRIORITY");
  const char mode = pr != nullptr ? pr[0] : (backward ? 'l' : 'n');
  if (mode == 'l' || mode == 'h') {
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, mode == 'l' ? least : greatest);
  }
  // (a side stream confined to a share of the compute units -- hipExtStreamCreateWithCUMask, 6 / 4 / 7 of every 8 CUs -- so that the caller's
  //  stream always finds free CUs for its sweeps: 75.2 -> 91.7 ms per step whatever the share; measured once, not kept)
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

static bool serial_mode();

namespace {

struct Builder {
  octseg_plan* P;
  size_t esz;

  int tensor(int N, int H, int W, int C, bool need_grad = true, bool external = false) {
    TensorInfo t{N, H, W, C, 0, 0, need_grad, external};
    P->tensors.push_back(t);
    return (int)P->tensors.size() - 1;
  }
  int param(const std::string& name, int kind, int R, int S, int O, int I, int KP) {
    ParamInfo p;
    p.name = name; p.kind = kind; p.R = R; p.S = S; p.O = O; p.I = I; p.KP = KP;
    p.numel = kind == OCTSEG_P_VEC ? (size_t)O : kind == OCTSEG_P_STEM ? (size_t)O * KP : (size_t)R * S * O * I;
    p.off = P->param_numel;
    P->param_numel += (p.numel + 3) / 4 * 4;  // keep 16-byte alignment of every parameter
    P->params.push_back(p);
    return (int)P->params.size() - 1;
  }
  int bn(const std::string& name, int C, int y, bool lazy) {
    BNInfo b;
    b.name = name; b.C = C;
    b.gamma = param(name + ".weight", OCTSEG_P_VEC, 1, 1, C, 1, 0);
    b.beta = param(name + ".bias", OCTSEG_P_VEC, 1, 1, C, 1, 0);
    b.rm_off = P->buffer_numel; P->buffer_numel += C;
    b.rv_off = P->buffer_numel; P->buffer_numel += C;
    b.ss_off = 0; b.rows = 0; b.lazy = lazy; b.y = y;
    const TensorInfo& t = P->tensors[y];
    b.count = (double)t.N * t.H * t.W;
    P->bns.push_back(b);
    return (int)P->bns.size() - 1;
  }

  // conv (+ optional BN whose statistics the epilogue emits).  Returns Value{raw output, bn}.
  Value conv(const std::string& name, const std::vector<ConvSrc>& srcs, int Cout, int R, int stride, int pad,
             const std::string& bn_name, bool bias, bool transposed = false, bool head = false,
             bool stem = false, bool bn_lazy = true, int accum_into = -1, bool defer_fin = false) {
    ConvLayer L;
    L.name = name; L.R = R; L.S = R; L.stride = stride; L.pad = pad;
    L.transposed = transposed; L.head = head; L.stem = stem; L.srcs = srcs; L.Cout = Cout;
    int Cin = 0;
    const TensorInfo& t0 = P->tensors[srcs[0].v.t];
    L.N = t0.N; L.IH = t0.H << srcs[0].up; L.IW = t0.W << srcs[0].up;
    for (auto& s : srcs) Cin += s.cn ? s.cn : P->tensors[s.v.t].C;
    L.Cin = Cin;
    L.stem_k = P->stem_k;
    if (transposed) { L.OH = L.IH * 2; L.OW = L.IW * 2; }
    else { L.OH = (L.IH + 2 * pad - R) / stride + 1; L.OW = (L.IW + 2 * pad - R) / stride + 1; }
    if (stem) L.w = param(name + ".weight", OCTSEG_P_STEM, P->stem_k, P->stem_k, Cout, 3, Cin);
    else L.w = param(name + ".weight", transposed ? OCTSEG_P_CONVT : OCTSEG_P_CONV, R, R, Cout, Cin, 0);
    L.b = bias ? param(name + ".bias", OCTSEG_P_VEC, 1, 1, Cout, 1, 0) : -1;
    L.OP = (Cout + 15) / 16 * 16;
    L.out = head ? -1 : (accum_into >= 0 ? accum_into : tensor(L.N, L.OH, L.OW, Cout));
    L.accum_out = accum_into >= 0;
    L.bn = -1; L.wimg_fwd_off = L.wimg_dgrad_off = 0; L.has_dgrad = false;
    P->convs.push_back(L);
    const int ci = (int)P->convs.size() - 1;
    Op op; op.kind = OP_CONV; op.conv = ci;
    P->ops.push_back(op);
    const double taps = (double)R * R;
    // MACs: every output pixel of a plain conv sees R*S*Cin; ConvT k4 s2 sees 4 taps per output pixel
    P->fwd_macs += (double)L.N * L.OH * L.OW * Cout * (stem ? 3.0 * P->stem_k * P->stem_k : (double)Cin * (transposed ? 4.0 : taps));
    Value v; v.t = L.out; v.bn = -1;
    if (!bn_name.empty()) {
      const int b = bn(bn_name, Cout, L.out, bn_lazy);
      P->convs[ci].bn = b;
      if (!defer_fin) {            // (deferred: the statistics come from another tensor, stats_fin() finishes the BatchNorm)
        Op f; f.kind = OP_BN_FIN; f.bn = b;
        P->ops.push_back(f);
      }
      v.bn = b;
    }
    return v;
  }
  // Grouped k x k conv (timm RegNet's conv2: groups = width / group width) + BatchNorm: G independent convs, each reading a channel
  // slice of `in` and writing a channel slice of ONE output tensor through the ordinary conv kernels (slices are pointer offsets: the
  // descriptors carry the channel stride separately).  The parameter of group g is named <name>.weight#g<g>: the host mirror joins the
  // groups along dim 0 into torch's [Cout][gw][k][k] tensor.  The BatchNorm's statistics come from the whole tensor (OP_STATS).
  Value gconv(const std::string& name, Value in, int C, int R, int stride, int pad, int gw, const std::string& bn_name) {
    const TensorInfo ti = P->tensors[in.t];
    const int OH = (ti.H + 2 * pad - R) / stride + 1, OW = (ti.W + 2 * pad - R) / stride + 1;
    const int out = tensor(ti.N, OH, OW, C);
    for (int g = 0; g < C / gw; ++g) {
      ConvSrc sc; sc.v = in; sc.up = 0; sc.c0 = g * gw; sc.cn = gw;
      conv(name, {sc}, gw, R, stride, pad, "", false, false, false, false, true, out);
      ConvLayer& L = P->convs.back();
      L.accum_out = false; L.sliced = true; L.out_c0 = g * gw;
      P->params[L.w].name = name + ".weight#g" + std::to_string(g);
    }
    const int bi = bn(bn_name, C, out, true);
    for (auto& L : P->convs) if (L.sliced && L.out == out) L.fold_bn = bi;
    stats_fin(bi, out);
    Value v; v.t = out; v.bn = bi;
    return v;
  }
  // out = relu?(bn(y) + res) + post, materialised
  int bn_act(Value y, Value res, int post, bool relu) {
    const TensorInfo& t = P->tensors[y.t];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_BN_ACT; op.y = y; op.res = res; op.post = post; op.relu = relu; op.out = o;
    P->ops.push_back(op);
    P->bns[y.bn].lazy = false;
    if (res.t >= 0 && res.bn >= 0) P->bns[res.bn].lazy = false;
    return o;
  }
  int maxpool(int in) {
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, t.H / 2, t.W / 2, t.C);
    Op op; op.kind = OP_MAXPOOL; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  // ---- FPN decoder pieces (smp decoders/fpn/decoder.py, restated in oracle/nets.py)
  int up2(int in) {   // F.interpolate(x, scale_factor=2, mode='nearest'), materialised: the FPNBlock's skip conv accumulates into it
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, t.H * 2, t.W * 2, t.C);
    Op op; op.kind = OP_UP2; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int gn_act(const std::string& name, int y, int up) {   // GroupNorm(32) + ReLU (+ bilinear x2, align_corners=True)
    const TensorInfo& t = P->tensors[y];
    GNInfo g;
    g.name = name; g.C = t.C; g.G = 32; g.y = y;
    g.gamma = param(name + ".weight", OCTSEG_P_VEC, 1, 1, t.C, 1, 0);
    g.beta = param(name + ".bias", OCTSEG_P_VEC, 1, 1, t.C, 1, 0);
    g.part_off = g.ss_off = g.stat_off = g.coef_off = 0;
    P->gns.push_back(g);
    const int o = tensor(t.N, t.H * up, t.W * up, t.C);
    Op op; op.kind = OP_GN; op.in = y; op.out = o; op.gn = (int)P->gns.size() - 1; op.up = up;
    P->ops.push_back(op);
    return o;
  }
  // ---- DeepLabV3+ pieces (smp decoders/deeplabv3, restated in oracle/nets.py; kernels in deeplab.hip)
  int parity(int in, bool to_coarse) {   // [N][H][W] -> [4N][H/2][W/2] parity sub-grids (a dilation-2 3x3 is a plain 3x3 on them), or back
    const TensorInfo& t = P->tensors[in];
    const int o = to_coarse ? tensor(t.N * 4, t.H / 2, t.W / 2, t.C) : tensor(t.N / 4, t.H * 2, t.W * 2, t.C);
    Op op; op.kind = OP_PARITY; op.in = in; op.out = o; op.up = to_coarse ? 1 : 0;
    P->ops.push_back(op);
    return o;
  }
  int dw_param(const std::string& name, int C) { return param(name + ".weight", OCTSEG_P_CONV, 3, 3, C, 1, 0); }   // torch [C][1][3][3]
  void dw(int in, int out, int oc0, int dwp, int wc0, int dil) {   // depthwise 3x3 of `in` into channels [oc0, oc0 + C_in) of `out`
    const TensorInfo& t = P->tensors[in];
    Op op; op.kind = OP_DW; op.in = in; op.out = out; op.oc0 = oc0; op.dwp = dwp; op.wc0 = wc0; op.up = dil;
    P->ops.push_back(op);
    P->fwd_macs += (double)t.N * t.H * t.W * t.C * 9.0;
  }
  int gap(int in) {                       // AdaptiveAvgPool2d(1)
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, 1, 1, t.C);
    Op op; op.kind = OP_GAP; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int bcast(int in, int H, int W) {       // F.interpolate of a 1x1 map to H x W
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, H, W, t.C);
    Op op; op.kind = OP_BCAST; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int drope(int in) {                     // nn.Dropout (element-wise), keep mask injected
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_DROPE; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int upb(int in, int up) {               // nn.UpsamplingBilinear2d(scale_factor=up)
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, t.H * up, t.W * up, t.C);
    Op op; op.kind = OP_UPB; op.in = in; op.out = o; op.up = up;
    P->ops.push_back(op);
    return o;
  }
  // ---- DeepLabV3 pieces: a dense dilated 3x3 conv as a plain conv on the mosaic of its rate^2 sub-grids (deeplab.hip)
  int mosaic(int in, int r, bool to_mosaic, int H = 0, int W = 0) {
    const TensorInfo& t = P->tensors[in];
    int o;
    if (to_mosaic) { const int hs = (t.H + r - 1) / r, ws = (t.W + r - 1) / r; o = tensor(t.N, r * (hs + 1) + 1, r * (ws + 1) + 1, t.C); }
    else o = tensor(t.N, H, W, t.C);
    Op op; op.kind = OP_MOSAIC; op.in = in; op.out = o; op.up = r; op.oc0 = to_mosaic ? 1 : 0;
    P->ops.push_back(op);
    return o;
  }
  // the BatchNorm `bn` (created by a conv with defer_fin) normalises tensor y, not the conv's own output: statistics from y, then finalize
  void stats_fin(int bn, int y) {
    const TensorInfo& t = P->tensors[y];
    P->bns[bn].y = y;
    P->bns[bn].count = (double)t.N * t.H * t.W;
    Op s; s.kind = OP_STATS; s.bn = bn; s.in = y;
    P->ops.push_back(s);
    Op f; f.kind = OP_BN_FIN; f.bn = bn;
    P->ops.push_back(f);
  }
  // ---- PSPNet pieces (smp decoders/pspnet, restated in oracle/nets.py)
  int binpool(int in, int k) {            // nn.AdaptiveAvgPool2d((k, k))
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, k, k, t.C);
    Op op; op.kind = OP_BINPOOL; op.in = in; op.out = o; op.up = k;
    P->ops.push_back(op);
    return o;
  }
  int resize(int in, int H, int W) {      // F.interpolate(size=(H, W), mode='bilinear', align_corners=True)
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, H, W, t.C);
    Op op; op.kind = OP_RESIZE; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int relu(int in) {
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_RELU; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int drop2d(int in) {                    // nn.Dropout2d, keep pattern [N][C] injected
    const TensorInfo& t = P->tensors[in];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_DROP2D; op.in = in; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  // timm SEModule(channels, rd_channels): x * sigmoid(fc2(relu(fc1(mean_hw(x))))), x = the materialised activation `in`
  int se(const std::string& name, int in, int rd) {
    const TensorInfo t = P->tensors[in];
    const int g = gap(in);
    const Value f1 = conv(name + ".fc1", {{mat_(g), 0}}, rd, 1, 1, 0, "", true);
    const int r1 = relu(f1.t);
    const Value f2 = conv(name + ".fc2", {{mat_(r1), 0}}, t.C, 1, 1, 0, "", true);
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_SEGATE; op.in = in; op.ins[0] = f2.t; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  static Value mat_(int t) { Value v; v.t = t; v.bn = -1; return v; }
  // ---- PAN pieces (smp decoders/pan, restated in oracle/nets.py; kernels in pan.hip)
  int add2(int a, int c) {
    const TensorInfo t = P->tensors[a];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_ADD; op.in = a; op.ins[0] = c; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  // GAUBlock(x = low-level feature, y = high-level map): bilinear(y) + relu(bn(conv3x3(x))) * sigmoid(bn(conv1x1(mean(y))))
  int gau(const std::string& pre, int x, int y) {
    const TensorInfo tx = P->tensors[x];
    const Value v2 = conv(pre + ".conv2.conv", {{mat_(x), 0}}, 32, 3, 1, 1, pre + ".conv2.bn", true);
    const int xg = bn_act(v2, Value(), -1, true);
    const Value v1 = conv(pre + ".conv1.1.conv", {{mat_(gap(y)), 0}}, 32, 1, 1, 0, pre + ".conv1.1.bn", true);
    const int s = bn_act(v1, Value(), -1, false);
    const int z = tensor(tx.N, tx.H, tx.W, 32);
    { Op op; op.kind = OP_SEGATE; op.in = xg; op.ins[0] = s; op.out = z; P->ops.push_back(op); }
    return add2(resize(y, tx.H, tx.W), z);
  }
  // FPABlock: global branch b1, 1x1 branch mid, the one-channel pyramid (pan.hip), out = pyramid * mid + b1
  int fpa(const std::string& pre, int x) {
    const TensorInfo t = P->tensors[x];
    const Value vb = conv(pre + ".branch1.1.conv", {{mat_(gap(x)), 0}}, 32, 1, 1, 0, pre + ".branch1.1.bn", true);
    const int b1 = bn_act(vb, Value(), -1, true);
    const Value vm = conv(pre + ".mid.0.conv", {{mat_(x), 0}}, 32, 1, 1, 0, pre + ".mid.0.bn", true);
    const int mid = bn_act(vm, Value(), -1, true);
    const char* names[6] = {".down1.1", ".down2.1", ".down3.1", ".down3.2", ".conv2", ".conv1"};
    const int ks[6] = {7, 5, 3, 3, 5, 7};
    for (int l = 0; l < 6; ++l) {
      const std::string n = pre + names[l];
      P->fpa.w[l] = param(n + ".conv.weight", OCTSEG_P_CONV, ks[l], ks[l], 1, l == 0 ? t.C : 1, 0);
      P->fpa.b[l] = param(n + ".conv.bias", OCTSEG_P_VEC, 1, 1, 1, 1, 0);
      P->fpa.bn[l] = bn(n + ".bn", 1, x, false);
    }
    P->fpa.pool = tensor(t.N, t.H / 2, t.W / 2, t.C);                 // MaxPool2d(2, 2) of the feature (the 7x7 conv's input)
    const int o = tensor(t.N, t.H, t.W, 32);
    Op op; op.kind = OP_FPA; op.in = x; op.ins[0] = mid; op.ins[1] = b1; op.out = o;
    P->ops.push_back(op);
    P->fwd_macs += (double)t.N * (t.H / 2) * (t.W / 2) * 49.0 * t.C;
    return o;
  }
  // ---- MAnet pieces (smp decoders/manet, restated in oracle/nets.py; kernels in pab.hip / se.hip / effnet.hip's sefc)
  // nn.Sequential(AdaptiveAvgPool2d(1), Conv2d(C, rd, 1), ReLU, Conv2d(rd, C, 1), Sigmoid) up to the sigmoid: the excitation s [N][1][1][C]
  int se_relu(const std::string& name, int in, int rd) {
    const TensorInfo t = P->tensors[in];
    const int g = gap(in);
    const int s = tensor(t.N, 1, 1, t.C);
    Op f; f.kind = OP_SEFC; f.in = g; f.out = s; f.up = rd; f.oc0 = 0;
    f.ins[0] = param(name + ".1.weight", OCTSEG_P_CONV, 1, 1, rd, t.C, 0);
    f.ins[1] = param(name + ".1.bias", OCTSEG_P_VEC, 1, 1, rd, 1, 0);
    f.ins[2] = param(name + ".3.weight", OCTSEG_P_CONV, 1, 1, t.C, rd, 0);
    f.ins[3] = param(name + ".3.bias", OCTSEG_P_VEC, 1, 1, t.C, 1, 0);
    P->ops.push_back(f);
    return s;
  }
  int gate2(int in, int s1, int s2) {    // in * (sigmoid(s1) + sigmoid(s2))
    const TensorInfo t = P->tensors[in];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_SEGATE; op.in = in; op.ins[0] = s1; op.ins[1] = s2; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  int pab(const std::string& name, int x) {
    const TensorInfo t = P->tensors[x];
    const Value top = conv(name + ".top_conv", {{mat_(x), 0}}, 64, 1, 1, 0, "", true);
    const Value center = conv(name + ".center_conv", {{mat_(x), 0}}, 64, 1, 1, 0, "", true);
    const Value bottom = conv(name + ".bottom_conv", {{mat_(x), 0}}, t.C, 3, 1, 1, "", true);
    const int y = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_PAB; op.in = x; op.ins[0] = top.t; op.ins[1] = center.t; op.ins[2] = bottom.t; op.out = y;
    P->ops.push_back(op);
    P->fwd_macs += (double)t.N * t.H * t.W * t.H * t.W * (64.0 + t.C);
    return conv(name + ".out_conv", {{mat_(y), 0}}, t.C, 3, 1, 1, "", true).t;
  }
  // ---- EfficientNet pieces (efficientnet_pytorch MBConvBlock, restated in oracle/nets.py; kernels in effnet.hip)
  void set_bn_effnet(int bi) { P->bns[bi].eps = 1e-3f; P->bns[bi].momentum = 0.01f; }
  // out = act(bn(y)) * drop_connect + post, materialised (swish has no lazy form in the conv kernels' staging)
  int bnx(Value y, int act, int post, int dc_block, bool conv_bn) {
    const TensorInfo t = P->tensors[y.t];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_BNX; op.y = y; op.up = act; op.post = post; op.oc0 = dc_block; op.out = o; op.conv_bn = conv_bn;
    P->ops.push_back(op);
    P->bns[y.bn].lazy = false;
    return o;
  }
  // depthwise K x K conv (torch weight [C][1][K][K] -> arena [K][K][C]) + BatchNorm (statistics from the output tensor)
  Value dwg(const std::string& name, int in, int K, int stride, int pad, const std::string& bn_name) {
    const TensorInfo t = P->tensors[in];
    const int OH = (t.H + stride - 1) / stride, OW = (t.W + stride - 1) / stride;      // TF "same"
    const int o = tensor(t.N, OH, OW, t.C);
    Op op; op.kind = OP_DWG; op.in = in; op.out = o; op.dwp = param(name + ".weight", OCTSEG_P_CONV, K, K, t.C, 1, 0); op.up = stride; op.oc0 = pad; op.wc0 = K;
    P->ops.push_back(op);
    P->fwd_macs += (double)t.N * OH * OW * t.C * K * K;
    const int bi = bn(bn_name, t.C, o, false);
    set_bn_effnet(bi);
    stats_fin(bi, o);
    Value v; v.t = o; v.bn = bi;
    return v;
  }
  // squeeze-excite of an MBConv block: mean -> W1, b1 -> swish -> W2, b2 -> sigmoid gate (reduction widths 4 .. 160: a kernel of its own)
  int se_effnet(const std::string& pre, int in, int rd) {
    const TensorInfo t = P->tensors[in];
    const int g = gap(in);
    const int s = tensor(t.N, 1, 1, t.C);
    Op f; f.kind = OP_SEFC; f.in = g; f.out = s; f.up = rd; f.oc0 = 1;
    f.ins[0] = param(pre + "._se_reduce.weight", OCTSEG_P_CONV, 1, 1, rd, t.C, 0);
    f.ins[1] = param(pre + "._se_reduce.bias", OCTSEG_P_VEC, 1, 1, rd, 1, 0);
    f.ins[2] = param(pre + "._se_expand.weight", OCTSEG_P_CONV, 1, 1, t.C, rd, 0);
    f.ins[3] = param(pre + "._se_expand.bias", OCTSEG_P_VEC, 1, 1, t.C, 1, 0);
    P->ops.push_back(f);
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_SEGATE; op.in = in; op.ins[0] = s; op.out = o;
    P->ops.push_back(op);
    return o;
  }
  // parameters and buffers of a layer the graph never runs (smp's get_encoder(depth=3) keeps layer3 / layer4 in the module and in state_dict)
  void dead_conv(const std::string& name, int Cout, int Cin, int R) { param(name + ".weight", OCTSEG_P_CONV, R, R, Cout, Cin, 0); }
  void dead_bn(const std::string& name, int C, int any_tensor) { bn(name, C, any_tensor, false); }
  int merge4(const int (&ins)[4]) {   // MergeBlock('add') + Dropout2d: every summand receives the same gradient -> one shared buffer
    const TensorInfo& t = P->tensors[ins[0]];
    const int o = tensor(t.N, t.H, t.W, t.C);
    Op op; op.kind = OP_MERGE; op.out = o;
    for (int i = 0; i < 4; ++i) { op.ins[i] = ins[i]; if (i > 0) P->tensors[ins[i]].grad_alias = ins[0]; }
    P->ops.push_back(op);
    return o;
  }
};

Value mat(int t) { Value v; v.t = t; v.bn = -1; return v; }

// torchvision ResNet; returns materialised features f1..f5
// dilate4: smp's make_dilated(output_stride=16) -- every conv of layer4 at stride 1 / dilation 2.  Built as the ordinary layer4 (stride 1)
// on the parity re-arrangement of layer3's output (deeplab.hip header): no dilated conv kernel exists or is needed.
// depth: smp encoder_depth (5, or 3 for PSPNet: layer3 / layer4 keep their parameters and buffers but no op).
// dilate3: smp's make_dilated(8) on top -- layer3 at dilation 2 (one parity re-arrangement), layer4 at dilation 4 (= dilation 2 on layer3's
// sub-grids: a second, nested re-arrangement); both are undone behind layer4.
std::vector<int> build_resnet(Builder& b, const std::string& enc, bool dilate4 = false, int depth = 5, bool dilate3 = false) {
  octseg_plan* P = b.P;
  const bool bottleneck = enc == "resnet50" || enc == "resnet101" || enc == "resnet152";
  int nblocks[4];
  if (enc == "resnet18") { int v[4] = {2, 2, 2, 2}; memcpy(nblocks, v, sizeof v); }
  else if (enc == "resnet34" || enc == "resnet50") { int v[4] = {3, 4, 6, 3}; memcpy(nblocks, v, sizeof v); }
  else if (enc == "resnet152") { int v[4] = {3, 8, 36, 3}; memcpy(nblocks, v, sizeof v); }
  else { int v[4] = {3, 4, 23, 3}; memcpy(nblocks, v, sizeof v); }
  const int KP = 160;  // 7*7*3 = 147 padded to a multiple of 32
  P->col_tensor = b.tensor(P->B, P->H / 2, P->W / 2, KP, false);
  { Op op; op.kind = OP_STEM_COL; op.out = P->col_tensor; P->ops.push_back(op); }
  Value ystem = b.conv("encoder.conv1", {{mat(P->col_tensor), 0}}, 64, 1, 1, 0, "encoder.bn1", false, false, false, true);
  std::vector<int> feats;
  int f1 = b.bn_act(ystem, Value(), -1, true);
  feats.push_back(f1);
  int x = b.maxpool(f1);
  int inplanes = 64;
  const int planes_l[4] = {64, 128, 256, 512};
  for (int li = 0; li < 4; ++li) {
    const int planes = planes_l[li];
    const int exp = bottleneck ? 4 : 1;
    const bool dil = (dilate4 && li == 3) || (dilate3 && li >= 2);
    if (li + 2 > depth) {               // a stage behind the last feature the decoder reads: parameters only
      for (int bi = 0; bi < nblocks[li]; ++bi) {
        const int stride = (bi == 0 && li > 0) ? 2 : 1;
        const std::string pre = "encoder.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
        const bool ds = (stride != 1) || (inplanes != planes * exp);
        if (!bottleneck) {
          b.dead_conv(pre + ".conv1", planes, inplanes, 3); b.dead_bn(pre + ".bn1", planes, x);
          b.dead_conv(pre + ".conv2", planes, planes, 3); b.dead_bn(pre + ".bn2", planes, x);
        } else {
          b.dead_conv(pre + ".conv1", planes, inplanes, 1); b.dead_bn(pre + ".bn1", planes, x);
          b.dead_conv(pre + ".conv2", planes, planes, 3); b.dead_bn(pre + ".bn2", planes, x);
          b.dead_conv(pre + ".conv3", planes * 4, planes, 1); b.dead_bn(pre + ".bn3", planes * 4, x);
        }
        if (ds) { b.dead_conv(pre + ".downsample.0", planes * exp, inplanes, 1); b.dead_bn(pre + ".downsample.1", planes * exp, x); }
        inplanes = planes * exp;
      }
      continue;
    }
    if (dil) x = b.parity(x, true);
    for (int bi = 0; bi < nblocks[li]; ++bi) {
      const int stride = (bi == 0 && li > 0 && !dil) ? 2 : 1;
      const std::string pre = "encoder.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      const bool ds = (stride != 1) || (inplanes != planes * exp);
      Value last;
      if (!bottleneck) {
        Value v1 = b.conv(pre + ".conv1", {{mat(x), 0}}, planes, 3, stride, 1, pre + ".bn1", false);
        last = b.conv(pre + ".conv2", {{v1, 0}}, planes, 3, 1, 1, pre + ".bn2", false);
      } else {
        Value v1 = b.conv(pre + ".conv1", {{mat(x), 0}}, planes, 1, 1, 0, pre + ".bn1", false);
        Value v2 = b.conv(pre + ".conv2", {{v1, 0}}, planes, 3, stride, 1, pre + ".bn2", false);
        last = b.conv(pre + ".conv3", {{v2, 0}}, planes * 4, 1, 1, 0, pre + ".bn3", false);
      }
      Value res = mat(x);
      if (ds) res = b.conv(pre + ".downsample.0", {{mat(x), 0}}, planes * exp, 1, stride, 0, pre + ".downsample.1", false);
      x = b.bn_act(last, res, -1, true);
      inplanes = planes * exp;
    }
    if (dil && li == 3) { x = b.parity(x, false); if (dilate3) x = b.parity(x, false); }
    feats.push_back(x);
  }
  return feats;  // f1 (S/2) .. f5 (S/32)
}

// timm RegNet as smp's RegNetEncoder wraps it (oracle/nets.py RegNetEncoder; reference configs/tune.yaml:19-24: timm-regnetx_002 /
// timm-regnetx_064): stem 3x3 s2 -> 32 + BN + ReLU (im2col rows of 27 values padded to 32, then the GEMM the ResNet stem uses in f32),
// four stages of bottleneck blocks -- conv1 1x1, conv2 GROUPED 3x3 (stride 2 in a stage's first block), conv3 1x1 without activation,
// 1x1 stride-s conv shortcut where the shape changes, ReLU behind the sum.  Widths / depths / group width: timm generate_regnet.
struct RegNetCfg { int w[4], d[4], gw; bool se; };   // se: RegNetY -- SEModule behind conv2 with round(0.25 * block input channels) reduction channels
static bool regnet_cfg(const std::string& enc, RegNetCfg& c) {
  if (enc == "timm-regnetx_002") { c = RegNetCfg{{24, 56, 152, 368}, {1, 1, 4, 7}, 8, false}; return true; }
  if (enc == "timm-regnetx_064") { c = RegNetCfg{{168, 392, 784, 1624}, {2, 4, 10, 1}, 56, false}; return true; }
  if (enc == "timm-regnety_120") { c = RegNetCfg{{224, 448, 896, 2240}, {2, 5, 11, 1}, 112, true}; return true; }
  return false;
}
std::vector<int> build_regnet(Builder& b, const RegNetCfg& cfg, int depth) {
  octseg_plan* P = b.P;
  P->stem_k = 3; P->stem_pad = 1;
  const int KP = 32;   // 3 * 3 * 3 = 27 padded
  P->col_tensor = b.tensor(P->B, P->H / 2, P->W / 2, KP, false);
  { Op op; op.kind = OP_STEM_COL; op.out = P->col_tensor; P->ops.push_back(op); }
  Value ystem = b.conv("encoder.stem.conv", {{mat(P->col_tensor), 0}}, 32, 1, 1, 0, "encoder.stem.bn", false, false, false, true);
  std::vector<int> feats;
  int x = b.bn_act(ystem, Value(), -1, true);
  feats.push_back(x);
  int prev = 32;
  for (int si = 0; si < 4; ++si) {
    const int w = cfg.w[si];
    for (int bi = 0; bi < cfg.d[si]; ++bi) {
      const int stride = bi == 0 ? 2 : 1;
      const std::string pre = "encoder.s" + std::to_string(si + 1) + ".b" + std::to_string(bi + 1);
      if (si + 2 > depth) {   // (smp encoder_depth 3: the stage keeps its parameters and buffers, no op)
        b.dead_conv(pre + ".conv1.conv", w, prev, 1); b.dead_bn(pre + ".conv1.bn", w, x);
        for (int g = 0; g < w / cfg.gw; ++g) { b.dead_conv(pre + ".conv2.conv", cfg.gw, cfg.gw, 3); P->params.back().name = pre + ".conv2.conv.weight#g" + std::to_string(g); }
        b.dead_bn(pre + ".conv2.bn", w, x);
        if (cfg.se) {
          const int rd = (int)lround(prev * 0.25);
          b.dead_conv(pre + ".se.fc1", rd, w, 1); b.param(pre + ".se.fc1.bias", OCTSEG_P_VEC, 1, 1, rd, 1, 0);
          b.dead_conv(pre + ".se.fc2", w, rd, 1); b.param(pre + ".se.fc2.bias", OCTSEG_P_VEC, 1, 1, w, 1, 0);
        }
        b.dead_conv(pre + ".conv3.conv", w, w, 1); b.dead_bn(pre + ".conv3.bn", w, x);
        if (prev != w || stride != 1) { b.dead_conv(pre + ".downsample.conv", w, prev, 1); b.dead_bn(pre + ".downsample.bn", w, x); }
        prev = w;
        continue;
      }
      Value v1 = b.conv(pre + ".conv1.conv", {{mat(x), 0}}, w, 1, 1, 0, pre + ".conv1.bn", false);
      Value v2 = b.gconv(pre + ".conv2.conv", v1, w, 3, stride, 1, cfg.gw, pre + ".conv2.bn");
      if (cfg.se) {   // RegNetY: the gate acts on relu(bn(conv2)), materialised for it
        const int xa = b.bn_act(v2, Value(), -1, true);
        v2 = mat(b.se(pre + ".se", xa, (int)lround(prev * 0.25)));
      }
      Value v3 = b.conv(pre + ".conv3.conv", {{v2, 0}}, w, 1, 1, 0, pre + ".conv3.bn", false);
      Value res = mat(x);
      if (prev != w || stride != 1) res = b.conv(pre + ".downsample.conv", {{mat(x), 0}}, w, 1, stride, 0, pre + ".downsample.bn", false);
      x = b.bn_act(v3, res, -1, true);
      prev = w;
    }
    if (si + 2 <= depth) feats.push_back(x);
  }
  return feats;
}

// efficientnet_pytorch's EfficientNet as smp's EfficientNetEncoder runs it (oracle/nets.py EfficientNetEncoder; reference configs/tune.yaml:
// 25-28: efficientnet-b0 / -b5 / -b7): stem 3x3 s2 (static "same" padding: top / left 0) + BN + swish, MBConv blocks -- expand 1x1 + BN + swish
// (expand ratio 6), depthwise k3 / k5 stride 1 / 2 + BN + swish, squeeze-excite with swish, project 1x1 + BN, id skip with drop_connect
// where stride 1 and equal widths --, features behind smp's stage indices; _conv_head / _bn1 stay as never-run parameters.
struct EffBlock { int k, stride, expand, cin, cout, se, pad; };
struct EffCfg { int stem, head, stage_idx[3]; std::vector<EffBlock> blocks; };
static bool effnet_cfg(const std::string& enc, EffCfg& c) {
  double w, d; int size;
  if (enc == "efficientnet-b0") { w = 1.0; d = 1.0; size = 224; int si[3] = {3, 5, 9}; memcpy(c.stage_idx, si, sizeof si); }
  else if (enc == "efficientnet-b5") { w = 1.6; d = 2.2; size = 456; int si[3] = {8, 13, 27}; memcpy(c.stage_idx, si, sizeof si); }
  else if (enc == "efficientnet-b7") { w = 2.0; d = 3.1; size = 600; int si[3] = {11, 18, 38}; memcpy(c.stage_idx, si, sizeof si); }
  else return false;
  auto rf = [&](int f) { const double x = f * w; int n = std::max(8, (int)(x + 4) / 8 * 8); if (n < 0.9 * x) n += 8; return n; };
  auto same_pad_top = [](int ih, int k, int s) { const int oh = (ih + s - 1) / s; const int pad = std::max((oh - 1) * s + k - ih, 0); return pad / 2; };
  static const int B[7][6] = {{1, 3, 1, 1, 32, 16}, {2, 3, 2, 6, 16, 24}, {2, 5, 2, 6, 24, 40}, {3, 3, 2, 6, 40, 80}, {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192},
                              {1, 3, 1, 6, 192, 320}};
  c.stem = rf(32); c.head = rf(1280); c.blocks.clear();
  size = (size + 1) / 2;       // behind the stem (its own static padding: top 0 for the even nominal sizes 224 / 456 / 600)
  for (auto& r : B) {
    const int rep = (int)ceil(d * r[0]), cin = rf(r[4]), cout = rf(r[5]);
    for (int j = 0; j < rep; ++j) {
      const int st = j == 0 ? r[2] : 1, ci = j == 0 ? cin : cout;
      c.blocks.push_back(EffBlock{r[1], st, r[3], ci, cout, std::max(1, (int)(ci * 0.25)), same_pad_top(size, r[1], st)});
      size = (size + st - 1) / st;
    }
  }
  return true;
}
std::vector<int> build_effnet(Builder& b, const EffCfg& cfg, int depth) {
  octseg_plan* P = b.P;
  P->stem_k = 3; P->stem_pad = 0;
  const int KP = 32;
  P->col_tensor = b.tensor(P->B, P->H / 2, P->W / 2, KP, false);
  { Op op; op.kind = OP_STEM_COL; op.out = P->col_tensor; P->ops.push_back(op); }
  Value ystem = b.conv("encoder._conv_stem", {{mat(P->col_tensor), 0}}, cfg.stem, 1, 1, 0, "encoder._bn0", false, false, false, true);
  b.set_bn_effnet(ystem.bn);
  std::vector<int> feats;
  int x = b.bnx(ystem, 1, -1, -1, true);
  feats.push_back(x);
  const int nb = (int)cfg.blocks.size();
  int stage = 0;      // blocks [0, stage_idx[0]) -> feature 2, ...
  bool live = true;
  for (int bi = 0; bi < nb; ++bi) {
    const EffBlock& e = cfg.blocks[bi];
    const std::string pre = "encoder._blocks." + std::to_string(bi);
    const int mid = e.cin * e.expand;
    if (stage < 3 && bi == cfg.stage_idx[stage]) { feats.push_back(x); ++stage; if ((int)feats.size() >= depth) live = false; }
    if (!live) {        // (smp encoder_depth 3: the later blocks keep parameters and buffers, no op)
      if (e.expand != 1) { b.dead_conv(pre + "._expand_conv", mid, e.cin, 1); b.dead_bn(pre + "._bn0", mid, x); }
      b.param(pre + "._depthwise_conv.weight", OCTSEG_P_CONV, e.k, e.k, mid, 1, 0); b.dead_bn(pre + "._bn1", mid, x);
      b.dead_conv(pre + "._se_reduce", e.se, mid, 1); b.param(pre + "._se_reduce.bias", OCTSEG_P_VEC, 1, 1, e.se, 1, 0);
      b.dead_conv(pre + "._se_expand", mid, e.se, 1); b.param(pre + "._se_expand.bias", OCTSEG_P_VEC, 1, 1, mid, 1, 0);
      b.dead_conv(pre + "._project_conv", e.cout, mid, 1); b.dead_bn(pre + "._bn2", e.cout, x);
      continue;
    }
    int t = x;
    if (e.expand != 1) {
      Value v = b.conv(pre + "._expand_conv", {{mat(x), 0}}, mid, 1, 1, 0, pre + "._bn0", false);
      b.set_bn_effnet(v.bn);
      t = b.bnx(v, 1, -1, -1, true);
    }
    Value vd = b.dwg(pre + "._depthwise_conv", t, e.k, e.stride, e.pad, pre + "._bn1");
    const int td = b.bnx(vd, 1, -1, -1, false);
    const int ts = b.se_effnet(pre, td, e.se);
    Value vp = b.conv(pre + "._project_conv", {{mat(ts), 0}}, e.cout, 1, 1, 0, pre + "._bn2", false);
    b.set_bn_effnet(vp.bn);
    const bool id_skip = e.stride == 1 && e.cin == e.cout;
    int dc = -1;
    if (id_skip && bi > 0) { dc = (int)P->dc_rates.size(); P->dc_rates.push_back(0.2f * (float)bi / (float)nb); }   // (block 0: rate 0 -> no drop)
    x = b.bnx(vp, 0, id_skip ? x : -1, dc, true);
  }
  if (live) feats.push_back(x);
  // smp deletes only _fc: the classifier's 1x1 conv and BatchNorm stay in the module and in state_dict
  b.dead_conv("encoder._conv_head", cfg.head, cfg.blocks.back().cout, 1);
  b.dead_bn("encoder._bn1", cfg.head, x);
  return feats;
}

Value unet_block(Builder& b, const std::string& pre, Value x, const std::vector<Value>& skips, int cout) {
  std::vector<ConvSrc> srcs;
  srcs.push_back({x, 1});
  for (auto& s : skips) srcs.push_back({s, 0});
  Value v1 = b.conv(pre + ".conv1.0", srcs, cout, 3, 1, 1, pre + ".conv1.1", false);
  return b.conv(pre + ".conv2.0", {{v1, 0}}, cout, 3, 1, 1, pre + ".conv2.1", false);
}

}  // namespace

static std::string lower(const char* s) {
  std::string r(s ? s : "");
  for (auto& c : r) c = (char)tolower((unsigned char)c);
  return r;
}

// Forward lanes.  A dense decoder (U-Net++) has nodes that do not need the deepest encoder feature: they are moved right
// behind the encoder op that completes their inputs and run on a side stream beside the deep encoder stages and the
// deepest decoder nodes, whose small grids (44^2 / 22^2 maps) leave most of the chip idle (measured: +1.7 % frames/s
// with everything that does not depend on layer4 on the side lane, +0.8 % with only what does not depend on layer3).  Any topological order is a valid plan; the backward
// walks the same list in reverse.  Nothing moves for U-Net / LinkNet (their decoders start at the deepest feature).
static void assign_lanes(octseg_plan* P) {
  const int n = (int)P->ops.size();
  auto op_name = [&](const Op& o) -> std::string {
    if (o.kind == OP_CONV) return P->convs[o.conv].name;
    if (o.kind == OP_BN_FIN) return P->bns[o.bn].name;
    return std::string();
  };
  int enc_end = n, l3_begin = -1;
  for (int i = 0; i < n; ++i) {
    const std::string nm = op_name(P->ops[i]);
    if (nm.rfind("decoder.", 0) == 0 || nm.rfind("segmentation_head", 0) == 0) { enc_end = i; break; }
  }
  const char* lane_stage = getenv("OCTSEG_LANE_STAGE");   // experiments: encoder stage the side lane may not depend on
  const std::string stage = std::string("encoder.") + (lane_stage ? lane_stage : "layer4") + ".";
  for (int i = 0; i < enc_end; ++i)
    if (op_name(P->ops[i]).rfind(stage, 0) == 0) { l3_begin = i; break; }
  if (enc_end >= n || l3_begin < 0) return;
  std::vector<int> prod_t(P->tensors.size(), -1), fin_bn(P->bns.size(), -1), depmax(n, -1);
  auto dep = [&](int i, int j) {   // op i reads what op j wrote
    if (j < 0) return;
    depmax[i] = std::max(depmax[i], j < enc_end ? j : depmax[j]);
  };
  auto dep_val = [&](int i, const Value& v) {
    if (v.t >= 0) dep(i, prod_t[v.t]);
    if (v.bn >= 0) dep(i, fin_bn[v.bn]);
  };
  for (int i = 0; i < n; ++i) {
    const Op& o = P->ops[i];
    switch (o.kind) {
      case OP_STEM_COL: prod_t[o.out] = i; break;
      case OP_CONV: {
        const ConvLayer& L = P->convs[o.conv];
        for (auto& sct : L.srcs) dep_val(i, sct.v);
        if (L.out >= 0) prod_t[L.out] = i;
        break;
      }
      case OP_BN_FIN: dep(i, prod_t[P->bns[o.bn].y]); fin_bn[o.bn] = i; break;
      case OP_BN_ACT: dep_val(i, o.y); dep_val(i, o.res); if (o.post >= 0) dep(i, prod_t[o.post]); prod_t[o.out] = i; break;
      case OP_MAXPOOL: dep(i, prod_t[o.in]); prod_t[o.out] = i; break;
    }
  }
  std::vector<std::vector<int>> after(enc_end);   // side ops to insert behind encoder op e
  std::vector<char> side(n, 0);
  bool any = false;
  for (int i = enc_end; i < n; ++i)
    if (depmax[i] >= 0 && depmax[i] < l3_begin && op_name(P->ops[i]).rfind("decoder.", 0) == 0) {
      // the lane starts no earlier than the last op in front of layer3 that it can follow
      side[i] = 1; any = true;
      after[depmax[i]].push_back(i);
    }
  if (!any) return;
  std::vector<Op> order;
  order.reserve(n);
  for (int e = 0; e < enc_end; ++e) {
    order.push_back(P->ops[e]);
    for (int i : after[e]) { Op o = P->ops[i]; o.lane = 1; order.push_back(o); }
  }
  for (int i = enc_end; i < n; ++i)
    if (!side[i]) order.push_back(P->ops[i]);
  P->ops.swap(order);
  P->has_lanes = true;
}

static int build_plan(octseg_plan* P) {
  Builder b{P, dtype_size(P->dtype)};
  const bool dlv3 = P->arch == "deeplabv3";
  std::vector<int> f;
  RegNetCfg rcfg;
  const bool regnet = regnet_cfg(P->encoder, rcfg);
  EffCfg ecfg;
  const bool effnet = effnet_cfg(P->encoder, ecfg);
  if ((effnet || regnet) && P->arch == "pan") return fail(OCTSEG_UNSUPPORTED_ARCH, "PAN dilates its encoder (output stride 16): built over the ResNets");
  if (effnet) {
    if (P->arch == "deeplabv3plus" || dlv3) return fail(OCTSEG_UNSUPPORTED_ARCH, "EfficientNet encoders cannot be dilated (smp raises for DeepLabV3 / DeepLabV3+ over them too)");
    f = build_effnet(b, ecfg, P->arch == "pspnet" ? 3 : 5);
  } else if (regnet) {
    if (P->arch == "deeplabv3plus" || dlv3) return fail(OCTSEG_UNSUPPORTED_ARCH, "the dilated RegNet encoders (smp make_dilated) are not built");
    if ((P->arch == "linknet" && (rcfg.w[3] / 4) % 8 != 0) || (P->arch == "pspnet" && (rcfg.w[1] / 4) % 8 != 0))   // LinkNet's decoder blocks and PSPNet's pyramid branches run on a QUARTER of a feature's channels
      return fail(OCTSEG_UNSUPPORTED_ARCH, P->arch + " over " + P->encoder + ": its decoder narrows a feature to a quarter of its channels (" +
                  std::to_string(rcfg.w[P->arch == "pspnet" ? 1 : 3]) + " / 4 is not a multiple of the 8-channel vector the NHWC kernels move)");
    f = build_regnet(b, rcfg, P->arch == "pspnet" ? 3 : 5);
  } else {
    f = build_resnet(b, P->encoder, P->arch == "deeplabv3plus" || P->arch == "pan" || dlv3, P->arch == "pspnet" ? 3 : 5, dlv3);  // f[0]=f1 .. f[4]=f5
  }
  while (f.size() < 5) f.push_back(f.back());       // (PSPNet: three features; the slots of the others are never read)
  std::vector<int> fr(f.rbegin(), f.rend());          // features[1:][::-1]: f5, f4, f3, f2, f1
  std::vector<int> ench;
  for (int t : fr) ench.push_back(P->tensors[t].C);
  const int dec[5] = {256, 128, 64, 32, 16};
  Value x;
  int head_k = 3;
  if (P->arch == "unet") {
    x = mat(fr[0]);
    for (int i = 0; i < 5; ++i) {
      std::vector<Value> skips;
      if (i < 4) skips.push_back(mat(fr[i + 1]));
      x = unet_block(b, "decoder.blocks." + std::to_string(i), x, skips, dec[i]);
    }
  } else if (P->arch == "unetplusplus") {
    // smp UnetPlusPlusDecoder
    std::vector<int> skip_ch(ench.begin() + 1, ench.end());
    skip_ch.push_back(0);
    std::map<std::string, Value> dense;
    auto key = [](int d, int l) { return "x_" + std::to_string(d) + "_" + std::to_string(l); };
    const int depth = 4;
    for (int layer = 0; layer < 4; ++layer) {
      for (int d = 0; d < depth - layer; ++d) {
        if (layer == 0) {
          const int cout = d == 0 ? dec[0] : skip_ch[d];
          dense[key(d, d)] = unet_block(b, "decoder.blocks." + key(d, d), mat(fr[d]), {mat(fr[d + 1])}, cout);
        } else {
          const int li = d + layer;
          std::vector<Value> cat;
          for (int idx = d + 1; idx <= li; ++idx) cat.push_back(dense[key(idx, li)]);
          cat.push_back(mat(fr[li + 1]));
          const int cout = d == 0 ? dec[li] : skip_ch[li];
          dense[key(d, li)] = unet_block(b, "decoder.blocks." + key(d, li), dense[key(d, li - 1)], cat, cout);
        }
      }
    }
    x = unet_block(b, "decoder.blocks." + key(0, depth), dense[key(0, depth - 1)], {}, dec[4]);
  } else if (P->arch == "pan") {
    // smp PAN (reference sweep, configs/tune.yaml:18) at its defaults: encoder_output_stride 16 (layer4 dilated, as DeepLabV3+), decoder_channels
    // 32, FPA on the last feature, three GAU blocks down to stride 4, 3x3 head + UpsamplingBilinear2d(4)
    head_k = 3;
    P->head_up = 4;
    const TensorInfo t5 = P->tensors[f[4]];
    // (frames below 128 x 128 leave nothing for the pyramid's third max-pool -- torch fails there too; the plan is still built, for its
    //  parameter table, and refuses to run: run_forward)
    if (t5.H < 8 || t5.W < 8) P->run_error = "PAN needs frames of at least 128 x 128 (its pyramid pools the stride-16 feature three times)";
    const int x5 = b.fpa("decoder.fpa", f[4]);
    const int x4 = b.gau("decoder.gau3", f[3], x5);
    const int x3 = b.gau("decoder.gau2", f[2], x4);
    x = mat(b.gau("decoder.gau1", f[1], x3));
  } else if (P->arch == "manet") {
    // smp MAnet (reference sweep, configs/tune.yaml:17) at its defaults: PAB on the deepest feature, MFAB blocks (SE gates on the upsampled
    // high-level path and on the skip, summed) where there is a skip, a plain U-Net block for the last one; 3x3 head on 16 channels
    x = mat(b.pab("decoder.center", fr[0]));
    for (int i = 0; i < 5; ++i) {
      const std::string pre = "decoder.blocks." + std::to_string(i);
      const int in_ch = i == 0 ? ench[0] : dec[i - 1];
      if (i < 4) {
        const int skip = fr[i + 1], skip_ch = ench[i + 1];
        Value v1 = b.conv(pre + ".hl_conv.0.0", {{x, 0}}, in_ch, 3, 1, 1, pre + ".hl_conv.0.1", false);
        Value v2 = b.conv(pre + ".hl_conv.1.0", {{v1, 0}}, skip_ch, 1, 1, 0, pre + ".hl_conv.1.1", false);
        const int hl = b.bn_act(v2, Value(), -1, true);
        const int rd = std::max(1, skip_ch / 16);
        const int s_ll = b.se_relu(pre + ".SE_ll", skip, rd);          // (parameter order of the module: SE_ll before SE_hl)
        const int s_hl = b.se_relu(pre + ".SE_hl", hl, rd);            // mean of the nearest-x2 upsampled map = mean of the map
        const int gated = b.gate2(hl, s_hl, s_ll);                      // the gate is per (image, channel): applied BEFORE the upsample
        x = unet_block(b, pre, mat(gated), {mat(skip)}, dec[i]);
      } else {
        x = unet_block(b, pre, x, {}, dec[i]);
      }
    }
  } else if (P->arch == "linknet") {
    head_k = 1;
    std::vector<int> ch = ench;
    ch.push_back(32);
    x = mat(fr[0]);
    for (int i = 0; i < 5; ++i) {
      const std::string pre = "decoder.blocks." + std::to_string(i) + ".block";
      const int cin = ch[i], mid = cin / 4, cout = ch[i + 1];
      if (mid % 8 != 0)
        return fail(OCTSEG_UNSUPPORTED_ARCH, "linknet over " + P->encoder + ": its decoder narrows a feature to a quarter of its channels (" +
                    std::to_string(cin) + " / 4 is not a multiple of the 8-channel vector the NHWC kernels move)");
      Value v1 = b.conv(pre + ".0.0", {{x, 0}}, mid, 1, 1, 0, pre + ".0.1", false);
      Value v2 = b.conv(pre + ".1.0", {{v1, 0}}, mid, 4, 2, 1, pre + ".1.1", true, true);
      Value v3 = b.conv(pre + ".2.0", {{v2, 0}}, cout, 1, 1, 0, pre + ".2.1", false);
      if (i < 4) x = mat(b.bn_act(v3, Value(), fr[i + 1], true));
      else x = v3;
    }
  } else if (P->arch == "fpn") {
    // smp FPN (reference sweep, configs/tune.yaml:9-18): pyramid_channels 256, segmentation_channels 128, merge 'add', Dropout2d(0.2),
    // head = 1x1 conv at stride 4 + UpsamplingBilinear2d(4).  f[1..4] = c2..c5 (strides 4..32).
    head_k = 1;
    P->head_up = 4;
    int pyr[4];
    pyr[0] = b.conv("decoder.p5", {{mat(f[4]), 0}}, 256, 1, 1, 0, "", true).t;
    const char* lvl[3] = {"decoder.p4", "decoder.p3", "decoder.p2"};
    for (int i = 0; i < 3; ++i) {
      const int up = b.up2(pyr[i]);
      b.conv(std::string(lvl[i]) + ".skip_conv", {{mat(f[3 - i]), 0}}, 256, 1, 1, 0, "", true, false, false, false, true, up);
      pyr[i + 1] = up;
    }
    int seg[4];
    for (int i = 0; i < 4; ++i) {
      const int nup = 3 - i, nblk = nup > 1 ? nup : 1;
      int t = pyr[i];
      for (int j = 0; j < nblk; ++j) {
        const std::string pre = "decoder.seg_blocks." + std::to_string(i) + ".block." + std::to_string(j) + ".block";
        const Value y = b.conv(pre + ".0", {{mat(t), 0}}, 128, 3, 1, 1, "", false);
        t = b.gn_act(pre + ".1", y.t, nup > 0 ? 2 : 1);
      }
      seg[i] = t;
    }
    x = mat(b.merge4(seg));
  } else if (P->arch == "deeplabv3") {
    // smp DeepLabV3 (reference sweep, configs/tune.yaml:9-18) with its defaults: output stride 8, dense ASPP (12, 24, 36), decoder_channels
    // 256, 3x3 conv + BN + ReLU, head = 1x1 conv + UpsamplingBilinear2d(8).  Only the last feature is read.
    head_k = 1;
    P->head_up = 8;
    P->dropout_p = 0.5f;
    const int X = f[4];                       // stride 8
    const TensorInfo tx = P->tensors[X];
    const std::string A = "decoder.0";
    std::vector<ConvSrc> cat;
    cat.push_back({b.conv(A + ".convs.0.0", {{mat(X), 0}}, 256, 1, 1, 0, A + ".convs.0.1", false), 0});
    const int rates[3] = {12, 24, 36};
    for (int i = 0; i < 3; ++i) {             // ASPPConv: dense dilated 3x3 as a plain 3x3 on the mosaic of its sub-grids, BN, ReLU
      const std::string pre = A + ".convs." + std::to_string(i + 1);
      const int m = b.mosaic(X, rates[i], true);
      const Value v = b.conv(pre + ".0", {{mat(m), 0}}, 256, 3, 1, 1, pre + ".1", false, false, false, false, true, -1, true);
      const int yf = b.mosaic(v.t, rates[i], false, tx.H, tx.W);
      b.stats_fin(v.bn, yf);
      Value vf; vf.t = yf; vf.bn = v.bn;
      cat.push_back({vf, 0});
    }
    {
      const std::string pre = A + ".convs.4";
      const int g = b.gap(X);
      const Value v = b.conv(pre + ".1", {{mat(g), 0}}, 256, 1, 1, 0, pre + ".2", false);
      cat.push_back({mat(b.bcast(b.bn_act(v, Value(), -1, true), tx.H, tx.W)), 0});
    }
    const Value pr = b.conv(A + ".project.0", cat, 256, 1, 1, 0, A + ".project.1", false);
    const int pd = b.drope(b.bn_act(pr, Value(), -1, true));
    x = b.conv("decoder.1", {{mat(pd), 0}}, 256, 3, 1, 1, "decoder.2", false);
  } else if (P->arch == "pspnet") {
    // smp PSPNet (reference sweep, configs/tune.yaml:9-18) with its defaults: encoder_depth 3 (the stride-8 feature), pyramid pooling to
    // 1 / 2 / 3 / 6 bins, 1x1 conv to 512 + BN + ReLU, Dropout2d(0.2), 3x3 head + UpsamplingBilinear2d(8)
    head_k = 3;
    P->head_up = 8;
    P->dropout_p = 0.2f;
    const int X = f[2];
    const TensorInfo tx = P->tensors[X];
    const int sizes[4] = {1, 2, 3, 6};
    if ((tx.C / 4) % 8 != 0)
      return fail(OCTSEG_UNSUPPORTED_ARCH, "pspnet over " + P->encoder + ": its pyramid branches run on a quarter of the feature's channels (" +
                  std::to_string(tx.C) + " / 4 is not a multiple of the 8-channel vector the NHWC kernels move)");
    std::vector<ConvSrc> cat;
    for (int i = 0; i < 4; ++i) {
      const std::string pre = "decoder.psp.blocks." + std::to_string(i) + ".pool.1";
      const int g = b.binpool(X, sizes[i]);
      int a;
      if (sizes[i] == 1) a = b.relu(b.conv(pre + ".0", {{mat(g), 0}}, tx.C / 4, 1, 1, 0, "", true).t);      // no BatchNorm on a 1x1 map: biased conv
      else a = b.bn_act(b.conv(pre + ".0", {{mat(g), 0}}, tx.C / 4, 1, 1, 0, pre + ".1", false), Value(), -1, true);
      cat.push_back({mat(b.resize(a, tx.H, tx.W)), 0});
    }
    cat.push_back({mat(X), 0});
    const Value v = b.conv("decoder.conv.0", cat, 512, 1, 1, 0, "decoder.conv.1", false);
    x = mat(b.drop2d(b.bn_act(v, Value(), -1, true)));
  } else if (P->arch == "deeplabv3plus") {
    // smp DeepLabV3Plus (reference sweep, configs/tune.yaml:9-18) with its defaults: encoder_output_stride 16, decoder_channels 256,
    // atrous rates (12, 24, 36), 48-channel high-resolution branch from the stride-4 feature, head = 1x1 conv + UpsamplingBilinear2d(4).
    head_k = 1;
    P->head_up = 4;
    P->dropout_p = 0.5f;
    const int X = f[4];                       // stride 16 (layer4 dilated)
    const TensorInfo tx = P->tensors[X];
    const std::string A = "decoder.aspp.0";
    std::vector<ConvSrc> cat;
    cat.push_back({b.conv(A + ".convs.0.0", {{mat(X), 0}}, 256, 1, 1, 0, A + ".convs.0.1", false), 0});
    const int rates[3] = {12, 24, 36};
    for (int i = 0; i < 3; ++i) {             // ASPPSeparableConv: depthwise dilated 3x3, pointwise 1x1, BN, ReLU
      const std::string pre = A + ".convs." + std::to_string(i + 1);
      const int wp = b.dw_param(pre + ".0.0", tx.C);
      const int t = b.tensor(tx.N, tx.H, tx.W, tx.C);
      b.dw(X, t, 0, wp, 0, rates[i]);
      cat.push_back({b.conv(pre + ".0.1", {{mat(t), 0}}, 256, 1, 1, 0, pre + ".1", false), 0});
    }
    {                                         // ASPPPooling: mean, 1x1 conv, BN (over the batch only), ReLU, resize = broadcast
      const std::string pre = A + ".convs.4";
      const int g = b.gap(X);
      const Value v = b.conv(pre + ".1", {{mat(g), 0}}, 256, 1, 1, 0, pre + ".2", false);
      cat.push_back({mat(b.bcast(b.bn_act(v, Value(), -1, true), tx.H, tx.W)), 0});
    }
    const Value pr = b.conv(A + ".project.0", cat, 256, 1, 1, 0, A + ".project.1", false);
    const int pd = b.drope(b.bn_act(pr, Value(), -1, true));
    const int wp1 = b.dw_param("decoder.aspp.1.0", 256);
    const int t1 = b.tensor(tx.N, tx.H, tx.W, 256);
    b.dw(pd, t1, 0, wp1, 0, 1);
    const Value a2 = b.conv("decoder.aspp.1.1", {{mat(t1), 0}}, 256, 1, 1, 0, "decoder.aspp.2", false);
    const int au = b.upb(b.bn_act(a2, Value(), -1, true), 4);
    const Value h1 = b.conv("decoder.block1.0", {{mat(f[1]), 0}}, 48, 1, 1, 0, "decoder.block1.1", false);
    const int hm = b.bn_act(h1, Value(), -1, true);
    const int wp2 = b.dw_param("decoder.block2.0.0", 256 + 48);
    const TensorInfo tu = P->tensors[au];
    const int t2 = b.tensor(tu.N, tu.H, tu.W, 256 + 48);     // torch.cat([aspp, high_res]) exists only as the depthwise conv's output
    b.dw(au, t2, 0, wp2, 0, 1);
    b.dw(hm, t2, 256, wp2, 256, 1);
    x = b.conv("decoder.block2.0.1", {{mat(t2), 0}}, 256, 1, 1, 0, "decoder.block2.1", false);
  } else {
    return fail(OCTSEG_UNSUPPORTED_ARCH, "unknown arch '" + P->arch + "' (unet | unetplusplus | linknet | fpn | deeplabv3plus | deeplabv3 | pspnet | manet | pan)");
  }
  b.conv("segmentation_head.0", {{x, 0}}, P->classes, head_k, 1, head_k / 2, "", true, false, true);
  if (P->head_up > 1) { Op op; op.kind = OP_UPLOGITS; P->ops.push_back(op); }

  if (!regnet && !effnet && P->arch != "manet" && P->arch != "pan" && P->arch != "fpn" && P->arch != "deeplabv3plus" && P->arch != "pspnet" && P->arch != "deeplabv3") assign_lanes(P);

  // ---------------- workspace layout ----------------
  P->dlogits_C = 16;
  const size_t esz = dtype_size(P->dtype);
  size_t off = 0;
  P->act_begin = off;
  for (auto& t : P->tensors) { t.off = off; off += align_up((size_t)t.N * t.H * t.W * t.C * esz); }
  P->act_end = off;
  P->grad_begin = off;
  for (auto& t : P->tensors)
    if (t.need_grad && t.grad_alias < 0) { t.goff = off; off += align_up((size_t)t.N * t.H * t.W * t.C * esz); }
  for (auto& t : P->tensors)
    if (t.need_grad && t.grad_alias >= 0) t.goff = P->tensors[t.grad_alias].goff;
  P->grad_end = off;
  {
    static const bool no_bits = getenv("OCTSEG_NO_MASKBITS") != nullptr;   // A/B switch: the backward re-reads the output tensor for the mask
    for (auto& op : P->ops)
      if (!no_bits && op.kind == OP_BN_ACT && op.relu && op.post < 0) {
        TensorInfo& t = P->tensors[op.out];
        const size_t nvec = (size_t)t.N * t.H * t.W * t.C * esz / 16;
        t.mask_off = off; off += align_up(nvec);
      }
  }
  for (auto& g : P->gns) {
    const TensorInfo& t = P->tensors[g.y];
    const size_t S = (size_t)gn_num_slabs((size_t)t.H * t.W);
    g.part_off = off; off += align_up((size_t)t.N * S * g.C * 2 * sizeof(float));
    g.ss_off = off; off += align_up((size_t)t.N * g.C * 2 * sizeof(float));
    g.stat_off = off; off += align_up((size_t)t.N * g.G * 2 * sizeof(float));
    g.coef_off = off; off += align_up((size_t)t.N * g.G * 2 * sizeof(float));
  }
  if (P->head_up > 1) {
    const size_t h4 = P->H / P->head_up, w4 = P->W / P->head_up;
    P->z4_off = off; off += align_up((size_t)P->B * P->classes * h4 * w4 * sizeof(float));
    P->dz4_off = off; off += align_up((size_t)P->B * h4 * w4 * 16 * esz);
  }
  for (auto& bn : P->bns) { bn.ss_off = off; off += align_up((size_t)bn.C * 6 * sizeof(float)); }
  size_t slab = 0, tmp = 0, tie_scratch = 0;
  for (auto& L : P->convs) {
    Geom g{L.R, L.S, L.stride, L.pad, L.transposed, L.N, L.IH, L.IW, L.Cin, L.OH, L.OW, L.Cout};
    if (L.stem) { g.R = g.S = 1; g.pad = 0; }
    const int wtaps = g.R * g.S;
    std::vector<ConvArgs> la;
    fwd_launches(g, la);
    L.pk_fwd = conv_pack_info(la[0], P->dtype);
    L.wimg_fwd_off = off; off += align_up(conv_image_bytes(L.pk_fwd, wtaps));
    L.has_dgrad = false;
    for (auto& s : L.srcs) L.has_dgrad = L.has_dgrad || P->tensors[s.v.t].need_grad;
    if (L.has_dgrad) {
      std::vector<ConvArgs> ld;
      dgrad_launches(g, ld);
      ConvArgs d0 = ld[0];
      for (auto& d : ld) if (d.ntaps > 0) { d0 = d; break; }
      d0.Cin = L.head ? P->dlogits_C : L.Cout;
      L.pk_dgrad = conv_pack_info(d0, P->dtype);
      L.wimg_dgrad_off = off; off += align_up(conv_image_bytes(L.pk_dgrad, wtaps));
    }
    L.tie = 0;
    if (tie_mask() != 0 && esz == 2 && !L.transposed && !L.stem && !L.head && !L.sliced && !L.accum_out && L.R == 3 && L.S == 3 && L.stride == 1 &&
        L.pad == 1 && L.bn >= 0 && L.b < 0 && !L.srcs.empty() && L.srcs[0].up && L.srcs[0].cn == 0 && P->tensors[L.srcs[0].v.t].need_grad) {
      bool ok = true;
      for (size_t i = 1; i < L.srcs.size(); ++i) ok = ok && !L.srcs[i].up && L.srcs[i].cn == 0 && P->tensors[L.srcs[i].v.t].need_grad;
      const int Ca = P->tensors[L.srcs[0].v.t].C, Cs = L.Cin - Ca;
      // (narrow layers stay whole: below 64 channels a launch is a single K chunk and the thin kernels own the 16 / 32-channel decoder tail)
      if (ok && Ca >= 64 && Ca % 8 == 0 && Cs % 8 == 0 && L.Cout >= 32 && (Cs == 0 || Cs >= 32)) {
        L.tie = tie_mask(); L.tie_Ca = Ca; L.tie_Cs = Cs;
        const Geom gu{4, 4, 2, 1, true, L.N, L.IH / 2, L.IW / 2, Ca, L.OH, L.OW, L.Cout};
        const Geom gs{3, 3, 1, 1, false, L.N, L.IH, L.IW, Cs, L.OH, L.OW, L.Cout};
        std::vector<ConvArgs> v;
        fwd_launches(gu, v);
        L.tie_pk_fu = conv_pack_info(v[0], P->dtype);
        L.tie_fu_off = off; off += align_up(conv_image_bytes(L.tie_pk_fu, 16));
        v.clear();
        L.tie_du_masked = false;
        if (!tie_dgrad_planes() && L.Cout % 64 == 0) {   // one masked launch over the four parity planes: the planes are whole K chunks
          ConvArgs am;
          tied_dgrad_masked(gu, am);
          DstDesc dd{}; dd.H = gu.IH; dd.W = gu.IW; dd.C = Ca; dd.cn = Ca; am.dst[0] = dd; am.ndst = 1;
          if (conv_masked_eligible(am, P->dtype)) { L.tie_du_masked = true; v.push_back(am); }
        }
        if (!L.tie_du_masked) { if (tie_dgrad_planes()) tied_dgrad_launches(gu, v); else dgrad_launches(gu, v); }
        L.tie_pk_du = conv_pack_info(v[0], P->dtype);
        L.tie_du_off = off; off += align_up(conv_image_bytes(L.tie_pk_du, L.tie_du_masked ? 9 : 16));
        if (Cs > 0) {
          v.clear(); fwd_launches(gs, v);
          L.tie_pk_fs = conv_pack_info(v[0], P->dtype);
          L.tie_fs_off = off; off += align_up(conv_image_bytes(L.tie_pk_fs, 9));
          v.clear(); dgrad_launches(gs, v);
          L.tie_pk_ds = conv_pack_info(v[0], P->dtype);
          L.tie_ds_off = off; off += align_up(conv_image_bytes(L.tie_pk_ds, 9));
        }
        tie_scratch = std::max(tie_scratch, ((size_t)16 * Ca + (size_t)9 * Cs) * L.Cout * sizeof(float));
      }
    }
    if ((L.tie & 1) && L.bn >= 0) {
      P->bns[L.bn].rows = 512;   // the launches accumulate into the output: its statistics come from a sweep over the finished tensor
      slab = std::max(slab, (size_t)512 * L.Cout * 2 * sizeof(float));
    } else if (L.bn >= 0 && L.stem && (P->stem_k == 7 && thin_stem_eligible(P->dtype))) {
      P->bns[L.bn].rows = thin_stem_rows(L.N, P->H, P->W);   // the stem runs in thin.hip straight from the frame: one slab row per workgroup
      slab = std::max(slab, (size_t)P->bns[L.bn].rows * L.Cout * 2 * sizeof(float));
    } else if (L.bn >= 0) {
      int rows = 0;
      for (auto& a : la) {
        // geometry-only descriptors, so that the tile count equals what run_forward will launch
        // (everything the kernel choice looks at: launch_conv routes stride-1 single-source 1x1 layers to gemm1x1.hip, whose
        //  slab has one row per workgroup; run_forward checks that it lands on the same row count)
        a.nsrc = 0;
        int c0 = 0;
        for (auto& s : L.srcs) {
          SrcDesc d{}; d.H = P->tensors[s.v.t].H; d.W = P->tensors[s.v.t].W; d.up = s.up; d.C = P->tensors[s.v.t].C; d.c0 = c0;
          c0 += s.cn ? s.cn : d.C; a.src[a.nsrc++] = d;
        }
        DstDesc dd{}; dd.H = L.OH; dd.W = L.OW; dd.C = L.Cout; dd.cn = L.Cout; a.dst[0] = dd; a.ndst = 1;
        a.bias = L.b >= 0 ? (const float*)(uintptr_t)16 : nullptr;   // presence only
        a.stat_slab = (float*)(uintptr_t)16;                          // (the rows are those of a TRAINING forward)
        a.Wmaster = (const float*)(uintptr_t)16; a.wO = L.Cout; a.wI = L.Cin; a.wtrans = 0;
        a.out_mode = L.head ? OUT_HEAD_NCHW : OUT_STORE;
        rows += conv_num_mtiles_flat(a, P->dtype);
      }
      P->bns[L.bn].rows = rows;
      slab = std::max(slab, (size_t)rows * L.Cout * 2 * sizeof(float));
    }
    for (auto& s : L.srcs)
      if (s.up) tmp = std::max(tmp, (size_t)L.N * L.IH * L.IW * P->tensors[s.v.t].C * esz);
  }
  for (auto& bn : P->bns)   // a BatchNorm behind a grouped conv: no conv epilogue feeds it, OP_STATS writes `rows` partial sums of the whole tensor
    if (bn.rows == 0) bn.rows = 512;
  // one-launch weight packing: job table + prefix sums (uploaded into the workspace on first use)
  P->pack_jobs.clear(); P->pack_prefix.clear(); P->pack_total = 0;
  for (auto& L : P->convs) {
    const int taps = L.stem ? 1 : L.R * L.S;
    for (int tr = 0; tr < 2; ++tr) {
      if (tr == 1 && !L.has_dgrad) continue;
      const ConvPackInfo& pk = tr ? L.pk_dgrad : L.pk_fwd;
      PackJob j{P->params[L.w].off, tr ? L.wimg_dgrad_off : L.wimg_fwd_off, taps, L.Cout, L.Cin, tr, pk.BN, pk.RB, pk.nchunks, pk.ntiles,
                (!tr && L.bn >= 0) ? P->bns[L.bn].ss_off : (!tr && L.fold_bn >= 0) ? P->bns[L.fold_bn].ss_off + (size_t)L.out_c0 * sizeof(float) : ~(size_t)0};
      P->pack_prefix.push_back(P->pack_total);
      P->pack_jobs.push_back(j);
      P->pack_total += (unsigned long long)taps * pk.nchunks * pk.ntiles * pk.BN * (pk.RB / 16);
    }
    if (L.tie) {   // the tied images: 4x4 kernel over the upsampled source's channels, the plain 3x3 over the skip channels (training only: no fold)
      auto add = [&](size_t dst, int ntaps, int I, int c0, int tr, int tied, const ConvPackInfo& pk) {
        PackJob j{P->params[L.w].off, dst, ntaps, L.Cout, I, tr, pk.BN, pk.RB, pk.nchunks, pk.ntiles, ~(size_t)0, L.Cin, c0, tied};
        P->pack_prefix.push_back(P->pack_total);
        P->pack_jobs.push_back(j);
        P->pack_total += (unsigned long long)ntaps * pk.nchunks * pk.ntiles * pk.BN * (pk.RB / 16);
      };
      add(L.tie_fu_off, 16, L.tie_Ca, 0, 0, 1, L.tie_pk_fu);
      add(L.tie_du_off, L.tie_du_masked ? 9 : 16, L.tie_Ca, 0, 1, L.tie_du_masked ? 2 : 1, L.tie_pk_du);
      if (L.tie_Cs > 0) {
        add(L.tie_fs_off, 9, L.tie_Cs, L.tie_Ca, 0, 0, L.tie_pk_fs);
        add(L.tie_ds_off, 9, L.tie_Cs, L.tie_Ca, 1, 0, L.tie_pk_ds);
      }
    }
  }
  P->bn_jobs.clear(); P->bn_prefix.clear(); P->bn_total = 0;
  for (auto& b : P->bns) {
    P->bn_prefix.push_back(P->bn_total);
    P->bn_jobs.push_back(BnEvalJob{P->params[b.gamma].off, P->params[b.beta].off, b.rm_off, b.rv_off, b.ss_off, b.C, ~(size_t)0, b.eps});
    P->bn_total += (unsigned)b.C;
  }
  for (auto& L : P->convs)   // a biased conv in front of a BatchNorm (LinkNet's ConvTranspose2d): its bias folds into the eval shift
    if (L.bn >= 0 && L.b >= 0) P->bn_jobs[L.bn].bias_off = P->params[L.b].off;
  P->bn_tab_off = off; off += align_up(P->bn_jobs.size() * sizeof(BnEvalJob));
  P->bn_prefix_off = off; off += align_up(P->bn_prefix.size() * sizeof(unsigned));
  P->pack_tab_off = off; off += align_up(P->pack_jobs.size() * sizeof(PackJob));
  P->pack_prefix_off = off; off += align_up(P->pack_prefix.size() * sizeof(unsigned long long));
  // the BN backward reduce uses up to 1024 slab rows
  for (auto& bn : P->bns) slab = std::max(slab, (size_t)1024 * bn.C * 2 * sizeof(float));
  P->slab_off = off; P->slab_bytes = align_up(slab); off += 2 * align_up(slab);          // one slab per forward lane
  P->fin_part_off = off; off += 2 * align_up((size_t)SLAB_PART_CAP * 2 * sizeof(double));   // two-level slab reduction scratch (per lane)
  P->fin_cnt_off = off; off += align_up(2 * 64 * sizeof(unsigned));
  P->bwd_part_off = off; off += align_up((size_t)8 * 32 * 4096 * sizeof(double));   // [column][group][256 vectors x 8 channels x 2]
  P->bwd_cnt_off = off; off += align_up((size_t)8 * 33 * 32 * sizeof(unsigned));   // one 128-byte line per ticket
  {
    size_t pool_elems = 0;
    for (auto& op : P->ops)
      if (op.kind == OP_MAXPOOL) { const TensorInfo& t = P->tensors[op.out]; pool_elems = std::max(pool_elems, (size_t)t.N * t.H * t.W * t.C); }
    P->pool_idx_off = off; off += align_up(pool_elems);   // one buffer: the ResNet stems have exactly one max-pool
  }
  P->tmp_off = off; P->tmp_bytes = tmp; off += align_up(tmp);
  P->tie_scratch_off = off; off += align_up(tie_scratch);
  for (int k = 0; k < 3; ++k) {
    P->exec_macs[k] = P->fwd_macs;
    for (auto& L : P->convs)   // a tied pass runs 16 of the 36 multiply-accumulates per low-resolution pixel over the upsampled source's channels
      if (L.tie & (1 << k)) P->exec_macs[k] -= layer_macs(L) * L.tie_Ca / L.Cin * (5.0 / 9.0);
  }
  {
    size_t se_part = 0;
    for (auto& op : P->ops)
      if (op.kind == OP_SEGATE) { const TensorInfo& t = P->tensors[op.in]; se_part = std::max(se_part, (size_t)t.N * se_dgate_shares(t.H * t.W) * t.C * sizeof(float)); }
    P->se_part_off = off; off += align_up(se_part);
    for (auto& op : P->ops)
      if (op.kind == OP_SEFC) { const TensorInfo& t = P->tensors[op.in]; op.aux_off = off; off += align_up((size_t)2 * t.N * op.up * sizeof(float)); }
      else if (op.kind == OP_FPA) {
        const TensorInfo& t = P->tensors[op.in];
        P->fpa.scratch_off = off; off += align_up(fpa_pyr_scratch_floats(t.N, t.H, t.W) * sizeof(float));
        P->fpa.gscratch_off = off; off += align_up((fpa_pyr_gscratch_floats(t.N, t.H, t.W) + (size_t)t.N * t.H * t.W) * sizeof(float));   // + d uu
      }
      else if (op.kind == OP_PAB) {
        const TensorInfo& t = P->tensors[op.in];
        const size_t hw = (size_t)t.H * t.W;
        op.aux_off = off; off += align_up((size_t)t.N * hw * hw * sizeof(float)) * 2 + align_up((size_t)t.N * hw * t.C * sizeof(float));
      }
  }
  P->dlogits_off = off; off += align_up((size_t)P->B * P->H * P->W * P->dlogits_C * esz);
  P->dice_off = off; off += align_up((size_t)(1 + P->B) * P->classes * DICE_NS * sizeof(double));   // totals + per-image replicas
  P->ws_bytes = off;
  return OCTSEG_OK;
}

// ================================================================ execution helpers
namespace {

struct Exec {
  octseg_plan* P;
  const float* params;
  float* grads;
  float* buffers;
  char* ws;
  hipStream_t st;
  int train;
  hipStream_t wst = nullptr;  // stream of the weight-gradient launches (side stream or st)
  std::vector<char> ginit;   // backward: has the gradient buffer of tensor t been written yet?
  // first contribution stores, later ones accumulate
  int claim(int t) { const int acc = ginit[t] ? 1 : 0; ginit[t] = 1; return acc; }

  void* act(int t) const { return ws + P->tensors[t].off; }
  void* grad(int t) const { return ws + P->tensors[t].goff; }
  float* ss(int bn) const { return (float*)(ws + P->bns[bn].ss_off); }
  float* bn_scale(int bn) const { return ss(bn); }
  float* bn_shift(int bn) const { return ss(bn) + P->bns[bn].C; }
  float* bn_mean(int bn) const { return ss(bn) + 2 * P->bns[bn].C; }
  float* bn_rstd(int bn) const { return ss(bn) + 3 * P->bns[bn].C; }
  float* bn_coef(int bn) const { return ss(bn) + 4 * P->bns[bn].C; }

  GnArgs gn_args(int gi) const {
    const GNInfo& g = P->gns[gi];
    const TensorInfo& t = P->tensors[g.y];
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.y = act(g.y);
    a.gamma = params + P->params[g.gamma].off; a.beta = params + P->params[g.beta].off;
    if (grads) { a.dgamma = grads + P->params[g.gamma].off; a.dbeta = grads + P->params[g.beta].off; }
    a.part = (float*)(ws + g.part_off); a.ss = (float*)(ws + g.ss_off); a.stat = (float*)(ws + g.stat_off); a.coef = (float*)(ws + g.coef_off);
    a.HW = (size_t)t.H * t.W; a.C = g.C; a.G = g.G; a.cpg = g.C / g.G; a.eps = 1e-5f;
    return a;
  }
  Geom geom(const ConvLayer& L) const {
    Geom g{L.R, L.S, L.stride, L.pad, L.transposed, L.N, L.IH, L.IW, L.Cin, L.OH, L.OW, L.Cout};
    if (L.stem) { g.R = g.S = 1; g.pad = 0; }
    return g;
  }
  int fill_srcs(const ConvLayer& L, SrcDesc* src) const {
    int c0 = 0, n = 0;
    for (auto& s : L.srcs) {
      const TensorInfo& t = P->tensors[s.v.t];
      SrcDesc d;
      d.ptr = (char*)act(s.v.t) + (size_t)s.c0 * dtype_size(P->dtype);          // (channel slice of a grouped conv: pointer offset,
      d.scale = s.v.bn >= 0 ? bn_scale(s.v.bn) + s.c0 : nullptr;                 //  the channel stride stays the tensor's)
      d.shift = s.v.bn >= 0 ? bn_shift(s.v.bn) + s.c0 : nullptr;
      d.C = t.C; d.c0 = c0; d.H = t.H; d.W = t.W; d.up = s.up; d.relu = s.v.bn >= 0 ? 1 : 0;
      src[n++] = d;
      c0 += s.cn ? s.cn : t.C;
    }
    return n;
  }
};

}  // namespace

// algorithmic multiply-accumulates of one pass (forward = dgrad = wgrad) over a conv layer
static double layer_macs(const ConvLayer& L) {
  return (double)L.N * L.OH * L.OW * L.Cout * (L.stem ? 3.0 * L.stem_k * L.stem_k : (double)L.Cin * (L.transposed ? 4.0 : (double)L.R * L.S));
}

// fold = 1 (eval forwards): scale / shift of every BatchNorm from the running statistics first (one launch), then every forward
// image with its BatchNorm's scale folded in (reference: what conv + BN in eval mode computes, src/predict.py / model.py:183-200);
// fold = 0 (training): plain images.  Cached until the parameters / buffers change or the mode flips.
static int pack_all_weights(Exec& E, bool fold) {
  octseg_plan* P = E.P;
  if (P->packed_valid && P->packed_fold == fold && P->packed_ws == (const void*)E.ws && P->packed_params == (const void*)E.params &&
      (!fold || P->packed_buffers == (const void*)E.buffers))
    return OCTSEG_OK;
  if (P->pack_tab_ws != (const void*)E.ws) {   // first use of this workspace: upload the job table
    HIPCHK(hipMemcpyAsync(E.ws + P->pack_tab_off, P->pack_jobs.data(), P->pack_jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, E.st));
    HIPCHK(hipMemcpyAsync(E.ws + P->pack_prefix_off, P->pack_prefix.data(), P->pack_prefix.size() * sizeof(unsigned long long),
                          hipMemcpyHostToDevice, E.st));
    if (!P->bn_jobs.empty()) {
      HIPCHK(hipMemcpyAsync(E.ws + P->bn_tab_off, P->bn_jobs.data(), P->bn_jobs.size() * sizeof(BnEvalJob), hipMemcpyHostToDevice, E.st));
      HIPCHK(hipMemcpyAsync(E.ws + P->bn_prefix_off, P->bn_prefix.data(), P->bn_prefix.size() * sizeof(unsigned), hipMemcpyHostToDevice, E.st));
    }
    P->pack_tab_ws = E.ws;
  }
  if (fold && !P->bn_jobs.empty())
    HIPCHK(launch_bn_finalize_eval_all(E.params, E.buffers, E.ws, (const BnEvalJob*)(E.ws + P->bn_tab_off),
                                       (const unsigned*)(E.ws + P->bn_prefix_off), (int)P->bn_jobs.size(), P->bn_total, 1e-5f, E.st));
  HIPCHK(launch_pack_all(P->dtype, E.params, E.ws, (const PackJob*)(E.ws + P->pack_tab_off),
                         (const unsigned long long*)(E.ws + P->pack_prefix_off), (int)P->pack_jobs.size(), P->pack_total, fold ? 1 : 0, E.st));
  P->packed_valid = true; P->packed_fold = fold; P->packed_ws = E.ws; P->packed_params = E.params; P->packed_buffers = E.buffers;
  return OCTSEG_OK;
}

static const void* fwd_weight(const Exec& E, const ConvLayer& L) { return E.ws + L.wimg_fwd_off; }

// Split-K width of the multi-tap weight gradients when they run on the side stream beside the chain's kernels.  A weight gradient that fills
// all 256 CUs with one long-running workgroup each (110-150 KB of LDS, 380 registers) leaves the data gradient and the BatchNorm sweeps of the
// chain nothing to start on until its workgroups retire; on 144-160 CUs it takes 30 % longer by itself (24.5 against 18.9 ms per step at 160, alone)
// and the step gets shorter: 65.1 -> 63.0 ms at 144 workgroups (ABAB on one box; 112: 67.8, 128: 63.5, 136: 63.0, 152: 63.3, 160: 63.4-63.7, 176: 63.7).  On the caller's own stream (one-stream
// mode, the kernels-alone pass of bench.py) nothing runs beside it and it keeps the full width.  OCTSEG_WGRAD_SIDE_WGS=n (A/B; 256 = full width).
static int side_wgs() {
  static const int v = getenv("OCTSEG_WGRAD_SIDE_WGS") ? atoi(getenv("OCTSEG_WGRAD_SIDE_WGS")) : 144;
  return v;
}

static Geom tie_geom_up(const ConvLayer& L) { return Geom{4, 4, 2, 1, true, L.N, L.IH / 2, L.IW / 2, L.tie_Ca, L.OH, L.OW, L.Cout}; }
static Geom tie_geom_skip(const ConvLayer& L) { return Geom{3, 3, 1, 1, false, L.N, L.IH, L.IW, L.tie_Cs, L.OH, L.OW, L.Cout}; }

// Training forward of a tied layer (ConvLayer::tie & 1): the skip channels' 3x3 stores the output, the four parity launches of the 4x4
// stride-2 kernel over the low-resolution source add to it, one sweep takes the BatchNorm statistics of the finished tensor.
static int tied_forward(Exec& E, const ConvLayer& L, hipStream_t st, float* slab) {
  octseg_plan* P = E.P;
  SrcDesc src[MAX_SRC];
  const int ns = E.fill_srcs(L, src);
  DstDesc d;
  d.ptr = E.act(L.out); d.C = L.Cout; d.c0 = 0; d.cn = L.Cout; d.H = L.OH; d.W = L.OW; d.accum = 0; d.pool = 0;
  const double macs = layer_macs(L);   // (profile classes keep the reference graph's count: the nine taps over every upsampled pixel)
  if (L.tie_Cs > 0) {
    std::vector<ConvArgs> v;
    fwd_launches(tie_geom_skip(L), v);
    ConvArgs& a = v[0];
    a.nsrc = ns - 1;
    for (int i = 1; i < ns; ++i) { a.src[i - 1] = src[i]; a.src[i - 1].c0 -= L.tie_Ca; }
    a.W = E.ws + L.tie_fs_off;
    a.dst[0] = d; a.ndst = 1; a.out_mode = OUT_STORE;
    ProfScope ps(0, 2.0 * macs * L.tie_Cs / L.Cin, st, L.name);
    HIPCHK(launch_conv(P->dtype, a, st));
  }
  std::vector<ConvArgs> v;
  fwd_launches(tie_geom_up(L), v);
  for (auto& a : v) {
    a.nsrc = 1; a.src[0] = src[0]; a.src[0].up = 0; a.src[0].c0 = 0;
    a.W = E.ws + L.tie_fu_off;
    d.accum = L.tie_Cs > 0 ? 1 : 0;   // (the four parities are disjoint: without a skip launch each one stores its own pixels)
    a.dst[0] = d; a.ndst = 1; a.out_mode = OUT_STORE;
    ProfScope ps(0, 2.0 * macs * L.tie_Ca / L.Cin / 4.0, st, L.name);
    HIPCHK(launch_conv(P->dtype, a, st));
  }
  const BNInfo& b = P->bns[L.bn];
  HIPCHK(launch_tensor_stats(P->dtype, E.act(L.out), (size_t)L.N * L.OH * L.OW, b.C, slab, b.rows, st));
  return OCTSEG_OK;
}

static int run_forward(Exec& E, const float* image, float* logits, int normalize, const float* mean, const float* stdv) {
  octseg_plan* P = E.P;
  if (!P->run_error.empty()) return fail(OCTSEG_BAD_SHAPE, P->run_error);
  if (E.train)
    for (auto& b : P->bns)
      if (b.count <= 1.0) {   // torch.nn.functional.batch_norm raises the same way (reference runs it in training)
        const TensorInfo& t = P->tensors[b.y];
        char buf[192];
        snprintf(buf, sizeof buf, "Expected more than 1 value per channel when training, got input size torch.Size([%d, %d, %d, %d])",
                 t.N, t.C, t.H, t.W);
        return fail(OCTSEG_BAD_SHAPE, buf);
      }
  int rc = pack_all_weights(E, !E.train);
  if (rc) return rc;
  if (E.train) HIPCHK(hipMemsetAsync(E.ws + P->fin_cnt_off, 0, 2 * 64 * sizeof(unsigned), E.st));
  // eval: BatchNorm is folded -- scale into the weight images (pack_all_weights), shift into the conv epilogue's bias, ReLU into
  // the epilogue of every conv whose BatchNorm is only read through relu(bn(y)): consumers stage plain activations
  const bool folded = !E.train;
  // ---- forward lanes (assign_lanes): lane-1 ops go to the side stream; a lane waits for the other one only when it
  // reads something the other lane produced and has not synchronised with since
  static const bool no_lanes = getenv("OCTSEG_NO_FWD_LANES") != nullptr;
  const bool lanes = P->has_lanes && !no_lanes && !serial_mode();
  hipStream_t lst[2] = {E.st, E.st};
  if (lanes) {
    // training forwards share the backward's lowest-priority stream (the second lane yields to the encoder chain: 69.1 / 68.6 against 69.7 /
    // 69.0 ms per step, ABAB); eval forwards -- the ensemble's B = 1 replay, whose lanes are its critical path -- and captured steps keep the
    // default priority
    const bool low = E.train && !P->tgraph_enabled;
    hipStream_t* lsp = low ? &P->side_bwd : &P->side;
    if (!*lsp) HIPCHK(create_side_stream(lsp, low));
    if (!P->ev_fork) {
      HIPCHK(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming));
    }
    lst[1] = *lsp;
  }
  std::vector<int> tseq(P->tensors.size(), 0), bseq(P->bns.size(), 0);   // producer: lane * 2^24 + sequence number on it
  int enq[2] = {1, 0}, seen[2] = {0, 0};   // ops enqueued per lane (main starts at 1: everything in front of the loop);
                                           // seen[l]: how much of the OTHER lane lane l has waited for
  auto need = [&](int lane, int stamp) -> int {   // make `lane` wait for the producer stamped `stamp`
    if (!lanes) return OCTSEG_OK;
    const int pl = stamp >> 24, ps = stamp & 0xffffff;
    if (pl == lane || ps <= seen[lane]) return OCTSEG_OK;
    hipEvent_t ev = lane == 1 ? P->ev_fork : P->ev_join;
    HIPCHK(hipEventRecord(ev, lst[pl]));
    HIPCHK(hipStreamWaitEvent(lst[lane], ev, 0));
    seen[lane] = enq[pl];
    return OCTSEG_OK;
  };
  auto need_val = [&](int lane, const Value& v) -> int {
    int rc2 = OCTSEG_OK;
    if (v.t >= 0) rc2 = need(lane, tseq[v.t]);
    if (!rc2 && v.bn >= 0) rc2 = need(lane, bseq[v.bn]);
    return rc2;
  };
  for (auto& op : P->ops) {
    const int lane = lanes ? op.lane : 0;
    hipStream_t st = lst[lane];
    const int stamp = (lane << 24) | (enq[lane] + 1);
    float* slab_l = (float*)(E.ws + P->slab_off + (size_t)lane * P->slab_bytes);
    double* part_l = (double*)(E.ws + P->fin_part_off) + (size_t)lane * SLAB_PART_CAP * 2;
    unsigned* cnt_l = (unsigned*)(E.ws + P->fin_cnt_off) + lane * 64;
    if (lane == 1 && enq[1] == 0) { rc = need(1, 1); if (rc) return rc; }   // the side lane starts behind the setup work
    switch (op.kind) {
      case OP_STEM_COL: {
        const TensorInfo& t = P->tensors[op.out];
        // (2-byte dtypes: the stem conv gathers its im2col rows straight from the frame in LDS, thin.hip KSTEM -- no 634 MB tensor)
        if (!(P->stem_k == 7 && thin_stem_eligible(P->dtype)))
          HIPCHK(launch_stem_im2col(P->dtype, image, E.act(op.out), P->B, P->H, P->W, t.C, mean, stdv, normalize, st, P->stem_k, P->stem_pad));
        tseq[op.out] = stamp;
        break;
      }
      case OP_CONV: {
        const ConvLayer& L = P->convs[op.conv];
        for (auto& sct : L.srcs) { rc = need_val(lane, sct.v); if (rc) return rc; }
        if (L.stem && (P->stem_k == 7 && thin_stem_eligible(P->dtype))) {
          StemArgs sa;
          memset(&sa, 0, sizeof(sa));
          sa.img = image; sa.N = P->B; sa.H = P->H; sa.W = P->W; sa.normalize = normalize;
          for (int i = 0; i < 3; ++i) { sa.mean[i] = normalize ? mean[i] : 0.f; sa.stdv[i] = normalize ? stdv[i] : 1.f; }
          sa.w = E.params + P->params[L.w].off;
          if (folded && L.bn >= 0) { sa.wscale = E.bn_scale(L.bn); sa.bias = E.bn_shift(L.bn); sa.relu_out = P->bns[L.bn].lazy ? 1 : 0; }
          sa.y = E.act(L.out);
          sa.slab = (L.bn >= 0 && E.train) ? slab_l : nullptr; sa.slab_row0 = 0;
          {
            ProfScope ps(0, 2.0 * layer_macs(L), st, L.name);
            HIPCHK(launch_thin_stem_forward(P->dtype, sa, st));
          }
          if (E.train) {   // the weight gradient re-gathers the frame: remember where it is (the caller keeps it alive until the backward)
            P->stem_image = image; P->stem_normalize = normalize;
            for (int i = 0; i < 3; ++i) { P->stem_mean[i] = sa.mean[i]; P->stem_std[i] = sa.stdv[i]; }
          }
          tseq[L.out] = stamp;
          break;
        }
        if ((L.tie & 1) && E.train && !folded) {
          rc = tied_forward(E, L, st, slab_l);
          if (rc) return rc;
          tseq[L.out] = stamp;
          break;
        }
        std::vector<ConvArgs> la;
        fwd_launches(E.geom(L), la);
        int row0 = 0;
        for (auto& a : la) {
          a.nsrc = E.fill_srcs(L, a.src);
          a.W = fwd_weight(E, L);
          a.bias = L.b >= 0 ? E.params + P->params[L.b].off : nullptr;
          // (ConvT: [4][4][O][I], same indexing; the stem's im2col GEMM: [O][KP] = one tap of KP input channels)
          a.Wmaster = E.params + P->params[L.w].off; a.wO = L.Cout; a.wI = L.Cin; a.wtrans = 0;
          if (folded) {
            for (int i = 0; i < a.nsrc; ++i) { a.src[i].scale = nullptr; a.src[i].shift = nullptr; a.src[i].relu = 0; }
            if (L.bn >= 0) { a.bias = E.bn_shift(L.bn); a.relu_out = P->bns[L.bn].lazy ? 1 : 0; a.wscale = E.bn_scale(L.bn); }
            else if (L.fold_bn >= 0) {   // one group of a grouped conv: its slice of the tensor's BatchNorm
              a.bias = E.bn_shift(L.fold_bn) + L.out_c0; a.relu_out = P->bns[L.fold_bn].lazy ? 1 : 0; a.wscale = E.bn_scale(L.fold_bn) + L.out_c0;
            }
          }
          a.ndst = 1;
          DstDesc d;
          d.accum = L.accum_out ? 1 : 0; d.pool = 0;
          if (L.head) {   // NCHW f32: the caller's logits, or (FPN) the stride-4 map that OP_UPLOGITS resamples
            d.ptr = P->head_up > 1 ? (void*)(E.ws + P->z4_off) : (void*)logits;
            d.C = L.Cout; d.c0 = 0; d.cn = L.Cout; d.H = L.OH; d.W = L.OW; a.out_mode = OUT_HEAD_NCHW;
          }
          else {
            d.ptr = (char*)E.act(L.out) + (size_t)L.out_c0 * dtype_size(P->dtype);
            d.C = L.sliced ? P->tensors[L.out].C : L.Cout; d.c0 = 0; d.cn = L.Cout; d.H = L.OH; d.W = L.OW; a.out_mode = OUT_STORE;
          }
          a.dst[0] = d;
          a.stat_slab = (L.bn >= 0 && E.train) ? slab_l : nullptr;
          a.slab_row0 = row0;
          row0 += conv_num_mtiles_flat(a, P->dtype);
          ProfScope ps(0, 2.0 * layer_macs(L) / (double)la.size(), st, L.name);
          HIPCHK(launch_conv(P->dtype, a, st));
        }
        if (L.bn >= 0 && E.train && row0 != P->bns[L.bn].rows)
          return fail(OCTSEG_BAD_ARG, "internal: BN-statistics slab rows of " + L.name + " differ between plan and launch");
        if (L.out >= 0) tseq[L.out] = stamp;
        break;
      }
      case OP_BN_FIN: {
        const BNInfo& b = P->bns[op.bn];
        const float* gamma = E.params + P->params[b.gamma].off;
        const float* beta = E.params + P->params[b.beta].off;
        rc = need(lane, tseq[b.y]);   // same lane as its conv by construction; kept for safety
        if (rc) return rc;
        if (E.train && b.count <= (double)BN_SMALL_COUNT)   // small tensors (pooled ASPP branch, 2x2 .. 16x16 maps): exact two-pass statistics
          HIPCHK(launch_bn_finalize_small(P->dtype, E.act(b.y), (int)b.count, b.C, gamma, beta, E.buffers + b.rm_off, E.buffers + b.rv_off, b.momentum,
                                          b.eps, E.bn_scale(op.bn), E.bn_shift(op.bn), E.bn_mean(op.bn), E.bn_rstd(op.bn), st));
        else if (E.train)
          HIPCHK(launch_bn_finalize_train(slab_l, b.rows, b.C, b.count, gamma, beta,
                                          E.buffers + b.rm_off, E.buffers + b.rv_off, b.momentum, b.eps, E.bn_scale(op.bn),
                                          E.bn_shift(op.bn), E.bn_mean(op.bn), E.bn_rstd(op.bn), part_l, cnt_l, st));
        // (eval: done for every BatchNorm at once in front of the loop)
        bseq[op.bn] = E.train ? stamp : 1;
        break;
      }
      case OP_BN_ACT: {
        const TensorInfo& t = P->tensors[op.out];
        rc = need_val(lane, op.y); if (rc) return rc;
        rc = need_val(lane, op.res); if (rc) return rc;
        if (op.post >= 0) { rc = need(lane, tseq[op.post]); if (rc) return rc; }
        BnActArgs a;
        memset(&a, 0, sizeof(a));
        a.y = E.act(op.y.t);
        if (!folded) { a.scale = E.bn_scale(op.y.bn); a.shift = E.bn_shift(op.y.bn); }
        if (op.res.t >= 0) {
          a.res = E.act(op.res.t);
          if (op.res.bn >= 0 && !folded) { a.rscale = E.bn_scale(op.res.bn); a.rshift = E.bn_shift(op.res.bn); }
        }
        if (op.post >= 0) a.post = E.act(op.post);
        a.out = E.act(op.out); a.npix = (size_t)t.N * t.H * t.W; a.C = t.C; a.relu = op.relu;
        if (E.train && t.mask_off) a.maskbits = (unsigned char*)(E.ws + t.mask_off);
        {
          const double tb = (double)a.npix * a.C * dtype_size(P->dtype);
          ProfScope ps(3, tb * (2 + (a.res ? 1 : 0) + (a.post ? 1 : 0)), st, "bn_act");
          HIPCHK(launch_bn_act(P->dtype, a, st));
        }
        tseq[op.out] = stamp;
        break;
      }
      case OP_UP2: {
        const TensorInfo& t = P->tensors[op.in];
        HIPCHK(launch_up2_fill(P->dtype, E.act(op.in), E.act(op.out), t.N, t.H, t.W, t.C, st));
        break;
      }
      case OP_GN: {
        const TensorInfo& t = P->tensors[op.in];
        const GnArgs ga = E.gn_args(op.gn);
        GnArgs a2 = ga;
        a2.out = E.act(op.out);
        HIPCHK(launch_gn_forward(P->dtype, a2, t.N, t.H, t.W, op.up, st));
        break;
      }
      case OP_MERGE: {
        const TensorInfo& t = P->tensors[op.out];
        if (E.train && P->dropout_keep == nullptr)
          return fail(OCTSEG_BAD_ARG, "FPN training forward: no Dropout2d keep mask set (octseg_plan_set_dropout: device float [B][128] of 0 / 1)");
        HIPCHK(launch_merge_drop(P->dtype, E.act(op.ins[0]), E.act(op.ins[1]), E.act(op.ins[2]), E.act(op.ins[3]),
                                 E.train ? P->dropout_keep : nullptr, 1.0f / (1.0f - P->dropout_p), E.act(op.out), t.N, (size_t)t.H * t.W, t.C, st));
        break;
      }
      case OP_PARITY: {
        const TensorInfo& tf = P->tensors[op.up ? op.in : op.out];   // the fine tensor
        HIPCHK(launch_parity_permute(P->dtype, E.act(op.in), E.act(op.out), tf.N, tf.H, tf.W, tf.C, op.up, 0, st));
        break;
      }
      case OP_DW: {
        const TensorInfo& ti = P->tensors[op.in];
        const TensorInfo& to = P->tensors[op.out];
        const ParamInfo& w = P->params[op.dwp];
        HIPCHK(launch_dw_conv(P->dtype, E.act(op.in), ti.C, 0, E.act(op.out), to.C, op.oc0, E.params + w.off, w.O, op.wc0, ti.N, ti.H, ti.W, ti.C,
                              op.up, 0, 0, st));
        break;
      }
      case OP_GAP: {
        const TensorInfo& t = P->tensors[op.in];
        HIPCHK(launch_image_sum(P->dtype, E.act(op.in), E.act(op.out), t.N, t.H * t.W, t.C, (float)(t.H * t.W), st));
        break;
      }
      case OP_BCAST: {
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_image_bcast(P->dtype, E.act(op.in), E.act(op.out), t.N, t.H * t.W, t.C, 1.f, 0, st));
        break;
      }
      case OP_DROPE: {
        const TensorInfo& t = P->tensors[op.out];
        if (E.train && P->dropout_keep == nullptr)
          return fail(OCTSEG_BAD_ARG, "DeepLabV3(+) training forward: no dropout keep mask set (octseg_plan_set_dropout: device float "
                                      "[B][H/s][W/s][256] of 0 / 1, NHWC; s = 16, DeepLabV3: 8)");
        HIPCHK(launch_drop_elem(P->dtype, E.act(op.in), E.train ? P->dropout_keep : nullptr, 1.0f / (1.0f - P->dropout_p), E.act(op.out),
                                (size_t)t.N * t.H * t.W * t.C, st));
        break;
      }
      case OP_UPB: {
        const TensorInfo& t = P->tensors[op.in];
        HIPCHK(launch_bilinear_up(P->dtype, E.act(op.in), E.act(op.out), t.N, t.H, t.W, t.C, op.up, st));
        break;
      }
      case OP_BINPOOL: {
        const TensorInfo& t = P->tensors[op.in];
        HIPCHK(launch_bin_mean(P->dtype, E.act(op.in), E.act(op.out), t.N, t.H, t.W, t.C, op.up, st));
        break;
      }
      case OP_MOSAIC: {
        const TensorInfo& tf = P->tensors[op.oc0 ? op.in : op.out];   // the fine tensor
        HIPCHK(launch_mosaic(P->dtype, E.act(op.in), E.act(op.out), tf.N, tf.H, tf.W, tf.C, op.up, op.oc0, 0, st));
        break;
      }
      case OP_STATS: {
        if (E.train) {
          const BNInfo& b = P->bns[op.bn];
          const TensorInfo& t = P->tensors[op.in];
          HIPCHK(launch_tensor_stats(P->dtype, E.act(op.in), (size_t)t.N * t.H * t.W, b.C, slab_l, b.rows, st));
        }
        break;
      }
      case OP_SEGATE: {
        const TensorInfo& t = P->tensors[op.in];
        HIPCHK(launch_se_gate(P->dtype, E.act(op.in), E.act(op.ins[0]), E.act(op.out), t.N, t.H * t.W, t.C, 0, st,
                              op.ins[1] >= 0 ? E.act(op.ins[1]) : nullptr));
        break;
      }
      case OP_ADD: {
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_add2(P->dtype, E.act(op.in), E.act(op.ins[0]), E.act(op.out), (size_t)t.N * t.H * t.W * t.C, st));
        break;
      }
      case OP_FPA: {
        const TensorInfo& t = P->tensors[op.in];
        float* scr = (float*)(E.ws + P->fpa.scratch_off);
        HIPCHK(launch_maxpool2(P->dtype, E.act(op.in), E.act(P->fpa.pool), nullptr, nullptr, t.N, t.H, t.W, t.C, 0, st));
        HIPCHK(launch_fpa_in_fwd(P->dtype, E.act(P->fpa.pool), E.params + P->params[P->fpa.w[0]].off, E.params + P->params[P->fpa.b[0]].off, scr, t.N, t.H / 2,
                                 t.W / 2, t.C, 7, st));
        FpaPyrArgs a;
        memset(&a, 0, sizeof(a));
        a.N = t.N; a.h = t.H; a.w = t.W; a.train = E.train; a.scratch = scr;
        for (int l = 0; l < 6; ++l) {
          const BNInfo& bn = P->bns[P->fpa.bn[l]];
          a.w_[l] = E.params + P->params[P->fpa.w[l]].off; a.b_[l] = E.params + P->params[P->fpa.b[l]].off;
          a.g_[l] = E.params + P->params[bn.gamma].off; a.be_[l] = E.params + P->params[bn.beta].off;
          a.rm_[l] = E.buffers + bn.rm_off; a.rv_[l] = E.buffers + bn.rv_off;
        }
        HIPCHK(launch_fpa_pyr_fwd(a, st));
        HIPCHK(launch_fpa_mix(P->dtype, scr + fpa_pyr_uu_offset(t.N, t.H, t.W), E.act(op.ins[0]), E.act(op.ins[1]), E.act(op.out), nullptr, nullptr, nullptr, t.N,
                              t.H * t.W, 32, st));
        break;
      }
      case OP_PAB: {
        const TensorInfo& t = P->tensors[op.in];
        const int hw = t.H * t.W;
        float* S = (float*)(E.ws + op.aux_off);
        float* M = (float*)(E.ws + op.aux_off + 2 * align_up((size_t)t.N * hw * hw * sizeof(float)));
        PabGemm g;
        memset(&g, 0, sizeof(g));
        g.batch = t.N;
        // S[i][j] = sum_k center[i][k] top[j][k]
        g.A = E.act(op.ins[1]); g.sAb = (size_t)hw * 64; g.sAm = 64; g.sAk = 1;
        g.B = E.act(op.ins[0]); g.sBb = (size_t)hw * 64; g.sBk = 1; g.sBn = 64;
        g.C = S; g.sCb = (size_t)hw * hw; g.sCm = hw; g.sCn = 1; g.c_f32 = 1; g.M = hw; g.N = hw; g.K = 64;
        HIPCHK(launch_pab_gemm(P->dtype, g, st));
        HIPCHK(launch_pab_softmax(S, nullptr, t.N, (size_t)hw * hw, 0, st));
        // M[i][c] = sum_j P[i][j] bottom[j][c]
        g.A = S; g.a_f32 = 1; g.sAb = (size_t)hw * hw; g.sAm = hw; g.sAk = 1;
        g.B = E.act(op.ins[2]); g.b_f32 = 0; g.sBb = (size_t)hw * t.C; g.sBk = t.C; g.sBn = 1;
        g.C = M; g.sCb = (size_t)hw * t.C; g.sCm = t.C; g.sCn = 1; g.M = hw; g.N = t.C; g.K = hw;
        HIPCHK(launch_pab_gemm(P->dtype, g, st));
        HIPCHK(launch_pab_mix(P->dtype, E.act(op.in), M, E.act(op.out), nullptr, nullptr, t.N, hw, t.C, st));
        break;
      }
      case OP_DWG: {
        const TensorInfo& ti = P->tensors[op.in];
        const TensorInfo& to = P->tensors[op.out];
        DwgArgs a;
        memset(&a, 0, sizeof(a));
        a.in = E.act(op.in); a.out = E.act(op.out); a.w = E.params + P->params[op.dwp].off;
        a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C; a.OH = to.H; a.OW = to.W; a.K = op.wc0; a.stride = op.up; a.pad = op.oc0;
        HIPCHK(launch_dwg_fwd(P->dtype, a, st));
        break;
      }
      case OP_BNX: {
        const TensorInfo& t = P->tensors[op.out];
        rc = need_val(lane, op.y); if (rc) return rc;
        BnxArgs a;
        memset(&a, 0, sizeof(a));
        a.y = E.act(op.y.t);
        if (!(folded && op.conv_bn)) { a.scale = E.bn_scale(op.y.bn); a.shift = E.bn_shift(op.y.bn); }   // (eval: a conv's BatchNorm is in its epilogue already)
        if (E.train && op.oc0 >= 0) {
          if (P->drop_connect == nullptr)
            return fail(OCTSEG_BAD_ARG, "EfficientNet training forward: no drop_connect factors set (octseg_plan_set_drop_connect: device float [" +
                                        std::to_string(P->dc_rates.size()) + "][B] of 0 or 1 / (1 - rate))");
          a.dscale = P->drop_connect + (size_t)op.oc0 * t.N;
        }
        a.post = op.post >= 0 ? E.act(op.post) : nullptr;
        a.out = E.act(op.out); a.npix = (size_t)t.N * t.H * t.W; a.hw = t.H * t.W; a.C = t.C; a.act = op.up;
        HIPCHK(launch_bnx_fwd(P->dtype, a, st));
        tseq[op.out] = stamp;
        break;
      }
      case OP_SEFC: {
        const TensorInfo& t = P->tensors[op.in];
        SefcArgs a;
        memset(&a, 0, sizeof(a));
        a.m = E.act(op.in); a.s = E.act(op.out);
        a.w1 = E.params + P->params[op.ins[0]].off; a.b1 = E.params + P->params[op.ins[1]].off;
        a.w2 = E.params + P->params[op.ins[2]].off; a.b2 = E.params + P->params[op.ins[3]].off;
        a.h = (float*)(E.ws + op.aux_off); a.dh = a.h + (size_t)t.N * op.up;
        a.N = t.N; a.C = t.C; a.R = op.up; a.act = op.oc0;
        HIPCHK(launch_sefc_fwd(P->dtype, a, st));
        break;
      }
      case OP_RESIZE: {
        const TensorInfo& ti = P->tensors[op.in];
        const TensorInfo& to = P->tensors[op.out];
        HIPCHK(launch_bilinear_resize(P->dtype, E.act(op.in), E.act(op.out), ti.N, ti.H, ti.W, to.H, to.W, ti.C, st));
        break;
      }
      case OP_RELU: {
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_relu(P->dtype, E.act(op.in), nullptr, E.act(op.out), (size_t)t.N * t.H * t.W * t.C, st));
        break;
      }
      case OP_DROP2D: {
        const TensorInfo& t = P->tensors[op.out];
        if (E.train && P->dropout_keep == nullptr)
          return fail(OCTSEG_BAD_ARG, "PSPNet training forward: no Dropout2d keep mask set (octseg_plan_set_dropout: device float [B][512] of 0 / 1)");
        if (E.train) HIPCHK(launch_drop_bwd(P->dtype, E.act(op.in), P->dropout_keep, 1.0f / (1.0f - P->dropout_p), E.act(op.out), t.N, (size_t)t.H * t.W, t.C, st));
        else HIPCHK(launch_drop_elem(P->dtype, E.act(op.in), nullptr, 1.f, E.act(op.out), (size_t)t.N * t.H * t.W * t.C, st));   // eval: identity (copy)
        break;
      }
      case OP_UPLOGITS: {
        const int h4 = P->H / P->head_up, w4 = P->W / P->head_up;
        HIPCHK(launch_bilinear_nchw((const float*)(E.ws + P->z4_off), logits, P->B * P->classes, h4, w4, P->head_up, st));
        break;
      }
      case OP_MAXPOOL: {
        const TensorInfo& t = P->tensors[op.in];
        rc = need(lane, tseq[op.in]); if (rc) return rc;
        HIPCHK(launch_maxpool_fwd(P->dtype, E.act(op.in), E.act(op.out), E.train ? (unsigned char*)(E.ws + P->pool_idx_off) : nullptr, t.N,
                                  t.H, t.W, t.C, st));
        tseq[op.out] = stamp;
        break;
      }
    }
    ++enq[lane];
  }
  if (lanes && enq[1] > seen[0]) {   // join: the caller's stream owns everything again
    HIPCHK(hipEventRecord(P->ev_join, lst[1]));
    HIPCHK(hipStreamWaitEvent(lst[0], P->ev_join, 0));
  }
  return OCTSEG_OK;
}

// BN backward of BN `bn` over raw tensor y: g -> dy (written to grad(y))
static int bn_backward(Exec& E, int bn, const void* g, int mask, const void* out_mask, void* res_grad = nullptr, int res_store = 0,
                       const unsigned char* maskbits = nullptr) {
  octseg_plan* P = E.P;
  const BNInfo& b = P->bns[bn];
  const TensorInfo& t = P->tensors[b.y];
  BnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.y = E.act(b.y); a.out = out_mask; a.maskbits = maskbits;
  a.scale = E.bn_scale(bn); a.shift = E.bn_shift(bn); a.mean = E.bn_mean(bn); a.rstd = E.bn_rstd(bn);
  a.gamma = E.params + P->params[b.gamma].off;
  a.npix = (size_t)t.N * t.H * t.W; a.C = b.C; a.mask = mask;
  a.slab = (float*)(E.ws + P->slab_off);
  a.part = (double*)(E.ws + P->fin_part_off); a.counters = (unsigned*)(E.ws + P->fin_cnt_off);
  const int VEC = P->dtype == DT_F32 ? 4 : 8;
  const int vpc = b.C / VEC;
  const int tpv = vpc >= 256 ? 1 : 256 / vpc;
  size_t rows = (a.npix + tpv - 1) / tpv;
  // slab rows = workgroups of the reduce pass (<= 1024: the slab's size).  Measured on U-Net++/resnet101: 256 rows everywhere +1.7 % at
  // --batch 2, +-0 at 4, -2 % at 16; 64 rows -9 % at 2; a size rule (256 rows up to 4 M elements) moved nothing: 1024 stays.
  static const size_t rows_env = getenv("OCTSEG_BN_ROWS") ? (size_t)atoi(getenv("OCTSEG_BN_ROWS")) : 0;   // experiments
  const size_t rows_cap = rows_env ? rows_env : 1024;
  if (rows > rows_cap) rows = rows_cap;
  a.rows = (int)rows;
  a.dgamma = E.grads + P->params[b.gamma].off;
  a.dbeta = E.grads + P->params[b.beta].off;
  a.coef = E.bn_coef(bn);
  a.dy = E.grad(b.y);
  a.res_grad = res_grad; a.res_store = res_store;
  const bool fused_fin = bn_bwd_fused_finalize();   // the reduce kernel finishes the reduction itself: no finalize launch
  if (fused_fin) { a.fpart = (double*)(E.ws + P->bwd_part_off); a.fcnt = (unsigned*)(E.ws + P->bwd_cnt_off); }
  E.ginit[b.y] = 1;   // written (stored) by the apply pass below
  const double tbytes = (double)a.npix * b.C * dtype_size(P->dtype);   // class 3 = HBM-bound sweeps: "flops" carries algorithmic bytes
  if (a.npix <= (size_t)BN_SMALL_COUNT) {   // small tensors: reduce, finalize and apply in one launch, in double (elementwise.hip)
    ProfScope ps(3, tbytes * ((mask == 2 && !maskbits) ? 5 : 4), E.st, b.name + ".bwd_small");
    HIPCHK(launch_bn_bwd_small(P->dtype, a, E.st));
    return OCTSEG_OK;
  } else {
    {
      ProfScope ps(3, tbytes * ((mask == 2 && !maskbits) ? 3 : 2), E.st, b.name + ".bwd_reduce");
      HIPCHK(launch_bn_bwd_reduce(P->dtype, a, E.st));
    }
    if (!fused_fin) HIPCHK(launch_bn_bwd_finalize(a, E.st));
  }
  {
    ProfScope ps(3, tbytes * (((mask == 2 && !maskbits) ? 4 : 3) + (res_grad ? (res_store ? 1 : 2) : 0)), E.st, b.name + ".bwd_apply");
    HIPCHK(launch_bn_bwd_apply(P->dtype, a, E.st));
  }
  return OCTSEG_OK;
}

static int conv_backward(Exec& E, const ConvLayer& L, const void* dy, int dyC) {
  octseg_plan* P = E.P;
  const size_t esz = dtype_size(P->dtype);
  const Geom g = E.geom(L);
  if (L.stem && (P->stem_k == 7 && thin_stem_eligible(P->dtype))) {
    // the forward built no im2col tensor.  Weight gradient straight from the frame (thin.hip); the deterministic-reduction mode keeps the
    // atomics-free kernel and rebuilds the im2col rows for it here.  The frame needs no gradient.
    if (P->stem_image == nullptr) return fail(OCTSEG_BAD_ARG, "backward without a training forward of this plan (stem frame unknown)");
    hipStream_t ws_ = E.wst ? E.wst : E.st;
    if (ws_ != E.st) {
      HIPCHK(hipEventRecord(P->ev_fork, E.st));
      HIPCHK(hipStreamWaitEvent(ws_, P->ev_fork, 0));
    }
    if (P->dtype == DT_BF16 && !deterministic_mode()) {
      StemArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.img = P->stem_image; sa.N = P->B; sa.H = P->H; sa.W = P->W; sa.normalize = P->stem_normalize;
      for (int i = 0; i < 3; ++i) { sa.mean[i] = P->stem_mean[i]; sa.stdv[i] = P->stem_std[i]; }
      sa.dy = dy; sa.dW = E.grads + P->params[L.w].off;
      ProfScope ps(2, 2.0 * layer_macs(L), ws_, L.name);
      HIPCHK(launch_thin_stem_wgrad(P->dtype, sa, ws_));
      return OCTSEG_OK;
    }
    const TensorInfo& tc = P->tensors[P->col_tensor];
    HIPCHK(launch_stem_im2col(P->dtype, P->stem_image, E.act(P->col_tensor), P->B, P->H, P->W, tc.C, P->stem_mean, P->stem_std,
                              P->stem_normalize, ws_));
  }
  // The weight gradient depends on dy and on saved activations only: forked in front of the layer's data gradient.  OCTSEG_WGRAD_BEHIND (A/B):
  // forked behind it, so that it would start together with the BatchNorm sweeps of the layer below (HBM-bound; at <= 128 registers they fit
  // beside its one-wave-per-SIMD workgroups) -- measured 70.2 against 68.5 ms per step (round 4, ABAB on one box; round 2: 78.8 against 78.4):
  // the side stream then idles through every data gradient's first half and the step's tail grows.
  static const bool wgrad_first = getenv("OCTSEG_WGRAD_BEHIND") == nullptr;
  auto wgrad_part = [&]() -> int {
    // weight gradient (+ bias gradient) on the side stream: fork after everything that produced dy
    hipStream_t ws_ = E.wst ? E.wst : E.st;
    if (ws_ != E.st) {
      HIPCHK(hipEventRecord(P->ev_fork, E.st));
      HIPCHK(hipStreamWaitEvent(ws_, P->ev_fork, 0));
    }
    // bias gradient
    if (L.b >= 0)
      HIPCHK(launch_channel_sum(P->dtype, dy, (size_t)L.N * L.OH * L.OW, dyC, L.Cout, E.grads + P->params[L.b].off, ws_));
    if (L.tie & 4) {
      // tied: gradient of the 4x4 image (low-resolution source x dy's parity planes) and of the skip slice's 3x3 into scratch, folded into the
      // 3x3 gradient by one sweep (the side stream runs the layers one after the other: one scratch serves them all)
      SrcDesc src[MAX_SRC];
      const int ns = E.fill_srcs(L, src);
      const int Ca = L.tie_Ca, Cs = L.tie_Cs;
      float* dK4 = (float*)(E.ws + P->tie_scratch_off);
      float* dW3s = dK4 + (size_t)16 * L.Cout * Ca;
      HIPCHK(hipMemsetAsync(dK4, 0, ((size_t)16 * Ca + (size_t)9 * Cs) * L.Cout * sizeof(float), ws_));
      const double macs = layer_macs(L);
      std::vector<WgradArgs> lw;
      wgrad_launches(tie_geom_up(L), lw);
      for (auto& a : lw) {
        a.nsrc = 1; a.src[0] = src[0]; a.src[0].up = 0; a.src[0].c0 = 0;
        a.dy = dy; a.dyC = dyC; a.dW = dK4; a.stamp = nullptr;
        a.wg_target = ws_ != E.st ? side_wgs() : 0;
      }
      if (wgrad_convt16_eligible(lw[0], P->dtype)) {   // all four parities from one staged window (wgrad_convt.hip)
        ProfScope ps(2, 2.0 * macs * Ca / L.Cin, ws_, L.name);
        HIPCHK(launch_wgrad_convt16(P->dtype, lw[0], ws_));
      } else {
        for (auto& a : lw) {
          ProfScope ps(2, 2.0 * macs * Ca / L.Cin / 4.0, ws_, L.name);
          HIPCHK(launch_wgrad(P->dtype, a, ws_));
        }
      }
      if (Cs > 0) {
        lw.clear();
        wgrad_launches(tie_geom_skip(L), lw);
        WgradArgs& a = lw[0];
        a.nsrc = ns - 1;
        for (int i = 1; i < ns; ++i) { a.src[i - 1] = src[i]; a.src[i - 1].c0 -= Ca; }
        a.dy = dy; a.dyC = dyC; a.dW = dW3s; a.stamp = nullptr;
        a.wg_target = ws_ != E.st ? side_wgs() : 0;
        ProfScope ps(2, 2.0 * macs * Cs / L.Cin, ws_, L.name);
        HIPCHK(launch_wgrad(P->dtype, a, ws_));
      }
      HIPCHK(launch_tied_fold(dK4, Cs > 0 ? dW3s : nullptr, E.grads + P->params[L.w].off, L.Cout, Ca, Cs, ws_));
    } else {
      std::vector<WgradArgs> lw;
      wgrad_launches(g, lw);
      for (auto& a : lw) {
        a.nsrc = E.fill_srcs(L, a.src);
        a.dy = dy; a.dyC = dyC;
        a.dW = E.grads + P->params[L.w].off;
        a.stamp = nullptr;
        a.wg_target = ws_ != E.st ? side_wgs() : 0;
      }
      if (L.transposed && lw.size() == 4 && wgrad_convt16_eligible(lw[0], P->dtype)) {   // ConvTranspose2d: the four parities in one launch
        ProfScope ps(2, 2.0 * layer_macs(L), ws_, L.name);
        HIPCHK(launch_wgrad_convt16(P->dtype, lw[0], ws_));
        lw.clear();
      }
      const double nl = (double)lw.size();
      for (auto& a : lw) {
        ProfScope ps(2, 2.0 * layer_macs(L) / nl, ws_, L.name);
        HIPCHK(launch_wgrad(P->dtype, a, ws_));
      }
    }
    return OCTSEG_OK;
  };
  auto dgrad_part = [&]() -> int {
    // data gradient
    bool any = false;
    for (auto& s : L.srcs) any = any || P->tensors[s.v.t].need_grad;
    if (!any) return OCTSEG_OK;
    if (L.tie & 2) {
      // tied: the low-resolution source's gradient from the four parity planes of dy (2x2 taps each, all of them cover the whole map: the first
      // stores unless somebody wrote before, the others add); the skip sources' from a 3x3 data gradient over their own channels
      const int ti0 = L.srcs[0].v.t;
      const TensorInfo& t0 = P->tensors[ti0];
      const double macs = layer_macs(L);
      std::vector<ConvArgs> lu;
      const bool planes = tie_dgrad_planes() && !L.tie_du_masked;
      if (L.tie_du_masked) { lu.resize(1); tied_dgrad_masked(tie_geom_up(L), lu[0]); }
      else if (planes) tied_dgrad_launches(tie_geom_up(L), lu);
      else dgrad_launches(tie_geom_up(L), lu);   // (one launch: 16 taps at stride 2 over dy)
      const int acc0 = E.claim(ti0);
      for (int k = 0; k < (int)lu.size(); ++k) {
        ConvArgs& a = lu[k];
        const int py = k >> 1, px = k & 1;
        if (L.tie_du_masked) {
          for (int p = 0; p < 4; ++p) {   // plane p of dy as a tensor of its own: first pixel (p >> 1, p & 1), doubled pixel and row strides
            SrcDesc& s = a.src[p];
            s.ptr = (const char*)dy + ((size_t)(p >> 1) * L.OW + (p & 1)) * dyC * esz;
            s.scale = nullptr; s.shift = nullptr; s.C = 2 * dyC; s.c0 = p * L.Cout; s.H = L.OH / 2; s.W = L.OW; s.up = 0; s.relu = 0;
          }
        } else {
          SrcDesc s;
          s.ptr = (const char*)dy + (planes ? ((size_t)py * L.OW + px) * dyC * esz : 0);
          s.scale = nullptr; s.shift = nullptr; s.C = planes ? 2 * dyC : dyC; s.c0 = 0; s.H = planes ? L.OH / 2 : L.OH; s.W = L.OW; s.up = 0; s.relu = 0;
          a.src[0] = s; a.nsrc = 1;
          a.Cin = L.Cout;
        }
        a.W = E.ws + L.tie_du_off;
        DstDesc d;
        d.ptr = E.grad(ti0); d.C = t0.C; d.c0 = 0; d.cn = L.tie_Ca; d.H = t0.H; d.W = t0.W; d.accum = k == 0 ? acc0 : 1; d.pool = 0;
        a.dst[0] = d; a.ndst = 1; a.out_mode = OUT_STORE; a.bias = nullptr; a.stat_slab = nullptr;
        ProfScope ps(1, 2.0 * macs * L.tie_Ca / L.Cin / (double)lu.size(), E.st, L.name);
        HIPCHK(launch_conv(P->dtype, a, E.st));
      }
      if (L.tie_Cs > 0) {
        std::vector<ConvArgs> ls;
        dgrad_launches(tie_geom_skip(L), ls);
        ConvArgs& a = ls[0];
        SrcDesc s;
        s.ptr = dy; s.scale = nullptr; s.shift = nullptr; s.C = dyC; s.c0 = 0; s.H = L.OH; s.W = L.OW; s.up = 0; s.relu = 0;
        a.src[0] = s; a.nsrc = 1;
        a.Cin = L.Cout;
        a.W = E.ws + L.tie_ds_off;
        int nd = 0, c0 = 0;
        for (size_t i = 1; i < L.srcs.size(); ++i) {
          const int ti = L.srcs[i].v.t;
          const TensorInfo& t = P->tensors[ti];
          DstDesc d;
          d.ptr = E.grad(ti); d.C = t.C; d.c0 = c0; d.cn = t.C; d.H = L.IH; d.W = L.IW; d.accum = E.claim(ti); d.pool = 0;
          a.dst[nd++] = d;
          c0 += t.C;
        }
        a.ndst = nd; a.out_mode = OUT_STORE; a.bias = nullptr; a.stat_slab = nullptr;
        ProfScope ps(1, 2.0 * macs * L.tie_Cs / L.Cin, E.st, L.name);
        HIPCHK(launch_conv(P->dtype, a, E.st));
      }
      return OCTSEG_OK;
    }
    std::vector<ConvArgs> ld;
    dgrad_launches(g, ld);
    // destinations: the forward sources' gradient buffers; upsampled sources go through a temp.  A destination
    // that nobody has written yet in this backward is stored to (no memset + read-modify-write), unless this
    // conv does not cover it completely (1x1 stride 2: only one pixel parity) -- then it is zeroed first.
    bool full_cover = true;
    for (auto& a : ld) if (a.ntaps == 0) full_cover = false;
    DstDesc dst[MAX_SRC];
    int nd = 0, c0 = 0;
    int up_src = -1;
    for (size_t i = 0; i < L.srcs.size(); ++i) {
      const int ti = L.srcs[i].v.t;
      const TensorInfo& t = P->tensors[ti];
      const int scn = L.srcs[i].cn ? L.srcs[i].cn : t.C;      // channels of this source inside the layer's input (a slice for grouped convs)
      const size_t soff = (size_t)L.srcs[i].c0 * esz;
      DstDesc d;
      d.C = t.C; d.c0 = c0; d.cn = scn; d.H = L.IH; d.W = L.IW; d.accum = 0; d.pool = 0;
      static const bool no_fuse_pool = getenv("OCTSEG_NO_FUSED_POOL") != nullptr;
      if (L.srcs[i].up && t.need_grad && !no_fuse_pool && ld.size() == 1 && ld[0].ostride == 1 && (L.IH % 2) == 0 && (L.IW % 2) == 0) {
        // gradient of the nearest-x2 upsample: the dgrad epilogue sums the 2x2 quads straight into the source's gradient
        d.ptr = E.grad(ti); d.H = t.H; d.W = t.W; d.pool = 1;
        d.accum = E.claim(ti);
      } else if (L.srcs[i].up) {
        d.ptr = E.ws + P->tmp_off;     // fully covered by this dgrad, pooled into the source afterwards
        up_src = (int)i;
      } else if (!t.need_grad) {
        d.ptr = E.ws + P->tmp_off; d.accum = 0;   // never happens for multi-source convs; keeps the descriptor valid
      } else if (L.srcs[i].cn) {
        // one group of a grouped conv: it owns a channel slice of the source's gradient.  The first group to arrive zeroes the whole
        // tensor, every group then accumulates into its slice (first-writer stores are per tensor, not per slice)
        if (!E.ginit[ti]) { HIPCHK(hipMemsetAsync(E.grad(ti), 0, (size_t)t.N * t.H * t.W * t.C * esz, E.st)); E.ginit[ti] = 1; }
        d.ptr = (char*)E.grad(ti) + soff;
        d.accum = 1;
      } else {
        d.ptr = E.grad(ti);
        d.accum = E.claim(ti);
        if (!d.accum && !full_cover) {
          HIPCHK(hipMemsetAsync(d.ptr, 0, (size_t)t.N * t.H * t.W * t.C * esz, E.st));
          d.accum = 1;
        }
      }
      dst[nd++] = d;
      c0 += scn;
    }
    for (auto& a : ld) {
      SrcDesc s;
      s.ptr = dy; s.scale = nullptr; s.shift = nullptr; s.C = dyC; s.c0 = 0; s.H = L.OH; s.W = L.OW; s.up = 0; s.relu = 0;
      a.src[0] = s; a.nsrc = 1;
      a.Cin = L.sliced ? L.Cout : dyC;  // contraction runs over the (padded) output channels; the pad columns of the image are zero (a group: its own channels, dyC is the stride)
      a.W = E.ws + L.wimg_dgrad_off;
      if (!L.stem && !L.transposed) { a.Wmaster = E.params + P->params[L.w].off; a.wO = L.Cout; a.wI = L.Cin; a.wtrans = 1; }
      for (int i = 0; i < nd; ++i) a.dst[i] = dst[i];
      a.ndst = nd;
      a.out_mode = OUT_STORE;   // per-destination accumulate flags decide
      a.bias = nullptr; a.stat_slab = nullptr;
      ProfScope ps(1, 2.0 * layer_macs(L) / (double)ld.size(), E.st, L.name);
      HIPCHK(launch_conv(P->dtype, a, E.st));
    }
    if (up_src >= 0) {
      const TensorInfo& t = P->tensors[L.srcs[up_src].v.t];
      const int ti = L.srcs[up_src].v.t;
      const int acc = E.claim(ti);
      HIPCHK(launch_pool2x2_accum(P->dtype, E.grad(ti), E.ws + P->tmp_off, t.N, t.H, t.W, t.C, acc ? 0 : 1, E.st));
    }
    return OCTSEG_OK;
  };
  if (wgrad_first) { const int rc = wgrad_part(); if (rc) return rc; }
  { const int rc = dgrad_part(); if (rc) return rc; }
  if (!wgrad_first) { const int rc = wgrad_part(); if (rc) return rc; }
  return OCTSEG_OK;
}

// Gradient-arena slices handed to the caller as soon as their last writer is enqueued (data-parallel overlap of the
// all-reduce with the rest of the backward, octseg_net_backward_sliced).
struct SliceCtx {
  int n = 0;
  hipStream_t comm = nullptr;
  octseg_slice_cb cb = nullptr;
  void* user = nullptr;
  std::vector<size_t> bounds;   // n + 1 element offsets into the arena, parameter-aligned, ascending
  std::vector<int> last_op;     // per slice: index of the op whose backward writes into it last (-1: nobody)
};

static void slice_plan(const octseg_plan* P, SliceCtx& S) {
  const int n = S.n;
  S.bounds.assign(n + 1, 0);
  S.bounds[n] = P->param_numel;
  for (int k = 1; k < n; ++k) {   // boundary k = start of the first parameter at or behind k/n of the arena
    const size_t want = P->param_numel * (size_t)k / n;
    size_t b = P->param_numel;
    for (auto& q : P->params) if (q.off >= want && q.off < b) b = q.off;
    S.bounds[k] = b;
  }
  for (int k = 1; k <= n; ++k) S.bounds[k] = std::max(S.bounds[k], S.bounds[k - 1]);
  S.last_op.assign(n, -1);
  auto touch = [&](int oi, int param) {
    if (param < 0) return;
    const size_t off = P->params[param].off;
    for (int k = 0; k < n; ++k)
      if (off >= S.bounds[k] && off < S.bounds[k + 1]) { if (S.last_op[k] < 0 || oi < S.last_op[k]) S.last_op[k] = oi; }
  };
  for (int oi = 0; oi < (int)P->ops.size(); ++oi) {   // the backward walks the ops downwards: the last writer has the SMALLEST index
    const Op& op = P->ops[oi];
    if (op.kind == OP_CONV) { touch(oi, P->convs[op.conv].w); touch(oi, P->convs[op.conv].b); }
    else if (op.kind == OP_BN_FIN) { if (P->bns[op.bn].lazy) { touch(oi, P->bns[op.bn].gamma); touch(oi, P->bns[op.bn].beta); } }
    else if (op.kind == OP_GN) { touch(oi, P->gns[op.gn].gamma); touch(oi, P->gns[op.gn].beta); }
    else if (op.kind == OP_DW || op.kind == OP_DWG) touch(oi, op.dwp);
    else if (op.kind == OP_BNX) { touch(oi, P->bns[op.y.bn].gamma); touch(oi, P->bns[op.y.bn].beta); }
    else if (op.kind == OP_SEFC) { for (int i = 0; i < 4; ++i) touch(oi, op.ins[i]); }
    else if (op.kind == OP_FPA) {
      for (int l = 0; l < 6; ++l) { touch(oi, P->fpa.w[l]); touch(oi, P->fpa.b[l]); touch(oi, P->bns[P->fpa.bn[l]].gamma); touch(oi, P->bns[P->fpa.bn[l]].beta); }
    }
    else if (op.kind == OP_BN_ACT) {
      touch(oi, P->bns[op.y.bn].gamma); touch(oi, P->bns[op.y.bn].beta);
      if (op.res.t >= 0 && op.res.bn >= 0) { touch(oi, P->bns[op.res.bn].gamma); touch(oi, P->bns[op.res.bn].beta); }
    }
  }
}

static int run_backward(Exec& E, const float* logits, const float* target, float grad_scale, SliceCtx* S = nullptr) {
  octseg_plan* P = E.P;
  if (S) slice_plan(P, *S);
  // slice k is complete once everything enqueued so far on the dgrad stream and on the weight-gradient stream has run:
  // the communication stream is made to wait for both, then the caller enqueues its collective there
  auto fire = [&](int k) -> int {
    if (S->bounds[k + 1] == S->bounds[k]) return OCTSEG_OK;
    if (!P->ev_slice) HIPCHK(hipEventCreateWithFlags(&P->ev_slice, hipEventDisableTiming));
    HIPCHK(hipEventRecord(P->ev_slice, E.st));
    HIPCHK(hipStreamWaitEvent(S->comm, P->ev_slice, 0));
    if (E.wst && E.wst != E.st) {
      HIPCHK(hipEventRecord(P->ev_slice, E.wst));
      HIPCHK(hipStreamWaitEvent(S->comm, P->ev_slice, 0));
    }
    S->cb(S->user, k, S->bounds[k], S->bounds[k + 1]);
    return OCTSEG_OK;
  };
  HIPCHK(hipMemsetAsync(E.grads, 0, P->param_numel * sizeof(float), E.st));
  HIPCHK(hipMemsetAsync(E.ws + P->fin_cnt_off, 0, 2 * 64 * sizeof(unsigned), E.st));
  HIPCHK(hipMemsetAsync(E.ws + P->bwd_cnt_off, 0, 8 * 33 * 32 * sizeof(unsigned), E.st));
  E.ginit.assign(P->tensors.size(), 0);
  static const bool no_side = getenv("OCTSEG_NO_SIDE_STREAM") != nullptr;   // A/B switch
  if (!no_side && !serial_mode()) {
    // (a step that is being captured / replayed as one hipGraph keeps the default priority: replaying a graph whose side branch was captured
    //  from a lowest-priority stream took 34.9 instead of 20.7 ms per step at 2 frames, profiles/r4_graph_ab.txt)
    hipStream_t* wsp = P->tgraph_enabled ? &P->side : &P->side_bwd;
    if (!*wsp) HIPCHK(create_side_stream(wsp, !P->tgraph_enabled));
    if (!P->ev_fork) {
      HIPCHK(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming));
    }
    E.wst = *wsp;
    // the side stream must see the zeroed parameter-gradient arena
    HIPCHK(hipEventRecord(P->ev_fork, E.st));
    HIPCHK(hipStreamWaitEvent(E.wst, P->ev_fork, 0));
  }
  // dL/dlogits (NHWC, padded channels)
  DiceArgs da;
  memset(&da, 0, sizeof(da));
  da.logits = logits; da.target = target; da.B = P->B; da.C = P->classes; da.HW = (size_t)P->H * P->W;
  da.sums = (double*)(E.ws + P->dice_off); da.loss_kind = P->loss_kind;
  HIPCHK(launch_dice_bwd(P->dtype, da, grad_scale, E.ws + P->dlogits_off, P->dlogits_C, E.st));
  int rc;
  for (int oi = (int)P->ops.size() - 1; oi >= 0; --oi) {
    const Op& op = P->ops[oi];
    switch (op.kind) {
      case OP_STEM_COL: break;
      case OP_CONV: {
        const ConvLayer& L = P->convs[op.conv];
        if (L.head) rc = conv_backward(E, L, E.ws + (P->head_up > 1 ? P->dz4_off : P->dlogits_off), P->dlogits_C);
        else if (L.sliced) rc = conv_backward(E, L, (char*)E.grad(L.out) + (size_t)L.out_c0 * dtype_size(P->dtype), P->tensors[L.out].C);
        else rc = conv_backward(E, L, E.grad(L.out), L.Cout);
        if (rc) return rc;
        break;
      }
      case OP_BN_FIN: {
        const BNInfo& b = P->bns[op.bn];
        if (b.lazy) {  // consumers accumulated d/d relu(bn(y)) into grad(y): turn it into dy in place
          rc = bn_backward(E, op.bn, E.grad(b.y), 1, nullptr);
          if (rc) return rc;
        }
        break;
      }
      case OP_BN_ACT: {
        const TensorInfo& t = P->tensors[op.out];
        const void* G = E.grad(op.out);
        const size_t n = (size_t)t.N * t.H * t.W * t.C;
        // main branch: with a post-add the relu mask must come from bn(y) itself
        const int mask = !op.relu ? 0 : (op.post >= 0 ? 1 : 2);
        // identity shortcut of a residual block (no BatchNorm on it, ReLU mask from the block's output): its gradient G * mask is
        // written by the main branch's apply sweep, which holds G and the mask already (was a masked_accum pass of its own)
        static const bool no_fuse_res = getenv("OCTSEG_NO_FUSED_RESGRAD") != nullptr;   // A/B switch
        const bool fuse_res = !no_fuse_res && mask == 2 && op.res.t >= 0 && op.res.bn < 0 && P->tensors[op.res.t].need_grad &&
                              E.grad(op.res.t) != G && E.grad(op.res.t) != E.grad(P->bns[op.y.bn].y);
        const unsigned char* mbits = (mask == 2 && t.mask_off) ? (const unsigned char*)(E.ws + t.mask_off) : nullptr;
        if (fuse_res) {
          const int acc = E.claim(op.res.t);
          rc = bn_backward(E, op.y.bn, G, mask, E.act(op.out), E.grad(op.res.t), acc ? 0 : 1, mbits);
        } else {
          rc = bn_backward(E, op.y.bn, G, mask, E.act(op.out), nullptr, 0, mbits);
        }
        if (rc) return rc;
        if (op.res.t >= 0 && !fuse_res) {
          if (op.res.bn >= 0) {
            rc = bn_backward(E, op.res.bn, G, op.relu ? 2 : 0, E.act(op.out), nullptr, 0, (op.relu && t.mask_off) ? (const unsigned char*)(E.ws + t.mask_off) : nullptr);
            if (rc) return rc;
          } else if (P->tensors[op.res.t].need_grad) {
            const int acc = E.claim(op.res.t);
            HIPCHK(launch_masked_accum(P->dtype, E.grad(op.res.t), G, op.relu ? E.act(op.out) : nullptr, n, acc ? 0 : 1, E.st));
          }
        }
        if (op.post >= 0 && P->tensors[op.post].need_grad) {
          const int acc = E.claim(op.post);
          HIPCHK(launch_masked_accum(P->dtype, E.grad(op.post), G, nullptr, n, acc ? 0 : 1, E.st));
        }
        break;
      }
      case OP_DROP2D: {
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_drop_bwd(P->dtype, E.grad(op.out), P->dropout_keep, 1.0f / (1.0f - P->dropout_p), E.grad(op.in), t.N, (size_t)t.H * t.W, t.C, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_RELU: {
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_relu(P->dtype, E.grad(op.out), E.act(op.out), E.grad(op.in), (size_t)t.N * t.H * t.W * t.C, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_RESIZE: {
        const TensorInfo& ti = P->tensors[op.in];
        const TensorInfo& to = P->tensors[op.out];
        HIPCHK(launch_bilinear_resize_adjoint(P->dtype, E.grad(op.out), E.grad(op.in), ti.N, ti.H, ti.W, to.H, to.W, ti.C, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_STATS: break;
      case OP_DWG: {
        const TensorInfo& ti = P->tensors[op.in];
        const TensorInfo& to = P->tensors[op.out];
        DwgArgs a;
        memset(&a, 0, sizeof(a));
        a.in = E.act(op.in); a.out = E.grad(op.out); a.w = E.params + P->params[op.dwp].off; a.dw = E.grads + P->params[op.dwp].off;
        a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C; a.OH = to.H; a.OW = to.W; a.K = op.wc0; a.stride = op.up; a.pad = op.oc0;
        HIPCHK(launch_dwg_bwd_w(P->dtype, a, E.st));
        if (ti.need_grad) {
          a.gin = E.grad(op.in); a.accum = E.claim(op.in);
          HIPCHK(launch_dwg_bwd_data(P->dtype, a, E.st));
        }
        break;
      }
      case OP_BNX: {        // gradient wrt bn(y) into grad(y) (act', drop_connect factor), the ordinary BatchNorm backward on it in place; post: + g
        const TensorInfo& t = P->tensors[op.out];
        BnxArgs a;
        memset(&a, 0, sizeof(a));
        a.y = E.act(op.y.t); a.scale = E.bn_scale(op.y.bn); a.shift = E.bn_shift(op.y.bn);
        a.dscale = op.oc0 >= 0 ? P->drop_connect + (size_t)op.oc0 * t.N : nullptr;
        a.post = E.grad(op.out); a.out = E.grad(op.y.t);
        a.npix = (size_t)t.N * t.H * t.W; a.hw = t.H * t.W; a.C = t.C; a.act = op.up;
        HIPCHK(launch_bnx_bwd(P->dtype, a, E.st));
        rc = bn_backward(E, op.y.bn, E.grad(op.y.t), 0, nullptr);
        if (rc) return rc;
        if (op.post >= 0 && P->tensors[op.post].need_grad) {
          const int acc = E.claim(op.post);
          HIPCHK(launch_masked_accum(P->dtype, E.grad(op.post), E.grad(op.out), nullptr, (size_t)t.N * t.H * t.W * t.C, acc ? 0 : 1, E.st));
        }
        break;
      }
      case OP_SEFC: {
        const TensorInfo& t = P->tensors[op.in];
        SefcArgs a;
        memset(&a, 0, sizeof(a));
        a.m = E.act(op.in); a.ds = E.grad(op.out); a.dm = E.grad(op.in);
        a.w1 = E.params + P->params[op.ins[0]].off; a.w2 = E.params + P->params[op.ins[2]].off;
        a.dw1 = E.grads + P->params[op.ins[0]].off; a.db1 = E.grads + P->params[op.ins[1]].off;
        a.dw2 = E.grads + P->params[op.ins[2]].off; a.db2 = E.grads + P->params[op.ins[3]].off;
        a.h = (float*)(E.ws + op.aux_off); a.dh = a.h + (size_t)t.N * op.up;
        a.N = t.N; a.C = t.C; a.R = op.up; a.act = op.oc0;
        HIPCHK(launch_sefc_bwd(P->dtype, a, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_SEGATE: {     // d x (+)= g * sigmoid(s);  d s = sigmoid'(s) * sum_p g * x
        const TensorInfo& t = P->tensors[op.in];
        const bool two = op.ins[1] >= 0;
        HIPCHK(launch_se_dgate(P->dtype, E.grad(op.out), E.act(op.in), E.act(op.ins[0]), E.grad(op.ins[0]), (float*)(E.ws + P->se_part_off), t.N, t.H * t.W,
                               t.C, E.st, two ? E.act(op.ins[1]) : nullptr, two ? E.grad(op.ins[1]) : nullptr));
        E.ginit[op.ins[0]] = 1;
        if (two) E.ginit[op.ins[1]] = 1;
        const int acc = E.claim(op.in);
        HIPCHK(launch_se_gate(P->dtype, E.grad(op.out), E.act(op.ins[0]), E.grad(op.in), t.N, t.H * t.W, t.C, acc, E.st, two ? E.act(op.ins[1]) : nullptr));
        break;
      }
      case OP_ADD: {
        const TensorInfo& t = P->tensors[op.out];
        const size_t n = (size_t)t.N * t.H * t.W * t.C;
        for (int src : {op.in, op.ins[0]})
          if (P->tensors[src].need_grad) {
            const int acc = E.claim(src);
            HIPCHK(launch_masked_accum(P->dtype, E.grad(src), E.grad(op.out), nullptr, n, acc ? 0 : 1, E.st));
          }
        break;
      }
      case OP_FPA: {
        const TensorInfo& t = P->tensors[op.in];
        const int n1 = t.N * (t.H / 2) * (t.W / 2);
        float* scr = (float*)(E.ws + P->fpa.scratch_off);
        float* gs = (float*)(E.ws + P->fpa.gscratch_off);
        float* duu = gs + fpa_pyr_gscratch_floats(t.N, t.H, t.W);
        // out = uu * mid + b1
        HIPCHK(launch_fpa_mix(P->dtype, scr + fpa_pyr_uu_offset(t.N, t.H, t.W), E.act(op.ins[0]), nullptr, nullptr, E.grad(op.out), E.grad(op.ins[0]), duu, t.N,
                              t.H * t.W, 32, E.st));
        E.ginit[op.ins[0]] = 1;
        HIPCHK(launch_image_sum(P->dtype, E.grad(op.out), E.grad(op.ins[1]), t.N, t.H * t.W, 32, 1.f, E.st));
        E.ginit[op.ins[1]] = 1;
        FpaPyrArgs a;
        memset(&a, 0, sizeof(a));
        a.N = t.N; a.h = t.H; a.w = t.W; a.train = 1; a.scratch = scr; a.gscratch = gs; a.duu = duu;
        for (int l = 0; l < 6; ++l) {
          const BNInfo& bn = P->bns[P->fpa.bn[l]];
          a.w_[l] = E.params + P->params[P->fpa.w[l]].off; a.b_[l] = E.params + P->params[P->fpa.b[l]].off;
          a.g_[l] = E.params + P->params[bn.gamma].off; a.be_[l] = E.params + P->params[bn.beta].off;
          a.rm_[l] = E.buffers; a.rv_[l] = E.buffers;
          a.dw_[l] = E.grads + P->params[P->fpa.w[l]].off; a.db_[l] = E.grads + P->params[P->fpa.b[l]].off;
          a.dg_[l] = E.grads + P->params[bn.gamma].off; a.dbe_[l] = E.grads + P->params[bn.beta].off;
        }
        HIPCHK(launch_fpa_pyr_bwd(a, E.st));
        // the wide 7x7 conv: d x1raw sits behind the first n1 floats of the gradient scratch; its input's gradient goes back through the max-pool
        HIPCHK(launch_fpa_in_bwd(P->dtype, E.act(P->fpa.pool), gs + n1, E.params + P->params[P->fpa.w[0]].off, E.grad(P->fpa.pool),
                                 E.grads + P->params[P->fpa.w[0]].off, E.grads + P->params[P->fpa.b[0]].off, t.N, t.H / 2, t.W / 2, t.C, 7, E.st));
        {
          const int acc = E.claim(op.in);
          HIPCHK(launch_maxpool2(P->dtype, E.act(op.in), nullptr, E.grad(P->fpa.pool), E.grad(op.in), t.N, t.H, t.W, t.C, acc, E.st));
        }
        break;
      }
      case OP_PAB: {
        const TensorInfo& t = P->tensors[op.in];
        const int hw = t.H * t.W;
        const size_t sbytes = align_up((size_t)t.N * hw * hw * sizeof(float));
        float* Pm = (float*)(E.ws + op.aux_off);
        float* dP = (float*)(E.ws + op.aux_off + sbytes);
        float* dM = (float*)(E.ws + op.aux_off + 2 * sbytes);
        // y = x + reshape(M): the gradient of y flows into x as it is, and into M through the same index map
        {
          const int acc = E.claim(op.in);
          HIPCHK(launch_masked_accum(P->dtype, E.grad(op.in), E.grad(op.out), nullptr, (size_t)t.N * hw * t.C, acc ? 0 : 1, E.st));
        }
        HIPCHK(launch_pab_mix(P->dtype, nullptr, nullptr, nullptr, dM, E.grad(op.out), t.N, hw, t.C, E.st));
        PabGemm g;
        memset(&g, 0, sizeof(g));
        g.batch = t.N;
        // dP[i][j] = sum_c dM[i][c] bottom[j][c]
        g.A = dM; g.a_f32 = 1; g.sAb = (size_t)hw * t.C; g.sAm = t.C; g.sAk = 1;
        g.B = E.act(op.ins[2]); g.sBb = (size_t)hw * t.C; g.sBk = 1; g.sBn = t.C;
        g.C = dP; g.c_f32 = 1; g.sCb = (size_t)hw * hw; g.sCm = hw; g.sCn = 1; g.M = hw; g.N = hw; g.K = t.C;
        HIPCHK(launch_pab_gemm(P->dtype, g, E.st));
        // d bottom[j][c] = sum_i P[i][j] dM[i][c]
        g.A = Pm; g.a_f32 = 1; g.sAb = (size_t)hw * hw; g.sAm = 1; g.sAk = hw;
        g.B = dM; g.b_f32 = 1; g.sBb = (size_t)hw * t.C; g.sBk = t.C; g.sBn = 1;
        g.C = E.grad(op.ins[2]); g.c_f32 = 0; g.sCb = (size_t)hw * t.C; g.sCm = t.C; g.sCn = 1; g.M = hw; g.N = t.C; g.K = hw;
        HIPCHK(launch_pab_gemm(P->dtype, g, E.st));
        E.ginit[op.ins[2]] = 1;
        HIPCHK(launch_pab_softmax(dP, Pm, t.N, (size_t)hw * hw, 1, E.st));     // dP -> dS in place
        // d center[i][k] = sum_j dS[i][j] top[j][k];   d top[j][k] = sum_i dS[i][j] center[i][k]
        g.A = dP; g.a_f32 = 1; g.sAb = (size_t)hw * hw; g.sAm = hw; g.sAk = 1;
        g.B = E.act(op.ins[0]); g.b_f32 = 0; g.sBb = (size_t)hw * 64; g.sBk = 64; g.sBn = 1;
        g.C = E.grad(op.ins[1]); g.sCb = (size_t)hw * 64; g.sCm = 64; g.sCn = 1; g.M = hw; g.N = 64; g.K = hw;
        HIPCHK(launch_pab_gemm(P->dtype, g, E.st));
        E.ginit[op.ins[1]] = 1;
        g.sAm = 1; g.sAk = hw;
        g.B = E.act(op.ins[1]);
        g.C = E.grad(op.ins[0]);
        HIPCHK(launch_pab_gemm(P->dtype, g, E.st));
        E.ginit[op.ins[0]] = 1;
        break;
      }
      case OP_MOSAIC: {     // the inverse re-arrangement of the gradient (gutters of a mosaic gradient are zero)
        const TensorInfo& tf = P->tensors[op.oc0 ? op.in : op.out];
        const int acc = E.claim(op.in);
        HIPCHK(launch_mosaic(P->dtype, E.grad(op.out), E.grad(op.in), tf.N, tf.H, tf.W, tf.C, op.up, op.oc0 ? 0 : 1, op.oc0 ? acc : 0, E.st));
        break;
      }
      case OP_BINPOOL: {
        const TensorInfo& t = P->tensors[op.in];
        const int acc = E.claim(op.in);
        HIPCHK(launch_bin_mean_bwd(P->dtype, E.grad(op.out), E.grad(op.in), t.N, t.H, t.W, t.C, op.up, acc, E.st));
        break;
      }
      case OP_UPB: {
        const TensorInfo& t = P->tensors[op.in];
        HIPCHK(launch_bilinear_adjoint(P->dtype, E.grad(op.out), E.grad(op.in), t.N, t.H, t.W, t.C, op.up, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_DROPE: {
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_drop_elem(P->dtype, E.grad(op.out), P->dropout_keep, 1.0f / (1.0f - P->dropout_p), E.grad(op.in),
                                (size_t)t.N * t.H * t.W * t.C, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_BCAST: {      // gradient of a broadcast: per-image sums
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_image_sum(P->dtype, E.grad(op.out), E.grad(op.in), t.N, t.H * t.W, t.C, 1.f, E.st));
        E.ginit[op.in] = 1;
        break;
      }
      case OP_GAP: {        // gradient of the mean: broadcast / HW, next to the other consumers of the ASPP input
        const TensorInfo& t = P->tensors[op.in];
        const int acc = E.claim(op.in);
        HIPCHK(launch_image_bcast(P->dtype, E.grad(op.out), E.grad(op.in), t.N, t.H * t.W, t.C, 1.f / (float)(t.H * t.W), acc, E.st));
        break;
      }
      case OP_DW: {
        const TensorInfo& ti = P->tensors[op.in];
        const TensorInfo& to = P->tensors[op.out];
        const ParamInfo& w = P->params[op.dwp];
        HIPCHK(launch_dw_wgrad(P->dtype, E.act(op.in), ti.C, 0, E.grad(op.out), to.C, op.oc0, E.grads + w.off, w.O, op.wc0, ti.N, ti.H, ti.W, ti.C,
                               op.up, E.st));
        if (ti.need_grad) {
          const int acc = E.claim(op.in);
          HIPCHK(launch_dw_conv(P->dtype, E.grad(op.out), to.C, op.oc0, E.grad(op.in), ti.C, 0, E.params + w.off, w.O, op.wc0, ti.N, ti.H, ti.W,
                                ti.C, op.up, 1, acc, E.st));
        }
        break;
      }
      case OP_PARITY: {     // the inverse permutation of the gradient
        const TensorInfo& tf = P->tensors[op.up ? op.in : op.out];
        const int acc = E.claim(op.in);
        HIPCHK(launch_parity_permute(P->dtype, E.grad(op.out), E.grad(op.in), tf.N, tf.H, tf.W, tf.C, op.up ? 0 : 1, acc, E.st));
        break;
      }
      case OP_UPLOGITS: {   // adjoint of the x4 bilinear resample: dL/dlogits (NHWC, padded channels) -> gradient of the stride-4 map
        const int h4 = P->H / P->head_up, w4 = P->W / P->head_up;
        HIPCHK(launch_bilinear_adjoint(P->dtype, E.ws + P->dlogits_off, E.ws + P->dz4_off, P->B, h4, w4, P->dlogits_C, P->head_up, E.st));
        break;
      }
      case OP_MERGE: {      // every summand's gradient = dropout mask * gradient of the sum: written once into the shared buffer
        const TensorInfo& t = P->tensors[op.out];
        HIPCHK(launch_drop_bwd(P->dtype, E.grad(op.out), P->dropout_keep, 1.0f / (1.0f - P->dropout_p), E.grad(op.ins[0]), t.N, (size_t)t.H * t.W,
                               t.C, E.st));
        for (int i = 0; i < 4; ++i) E.ginit[op.ins[i]] = 1;
        break;
      }
      case OP_GN: {
        const TensorInfo& t = P->tensors[op.in];
        GnArgs ga = E.gn_args(op.gn);
        if (op.up > 1) {      // gradient w.r.t. relu(gn(y)) at y's resolution: adjoint of the bilinear x2, parked in y's gradient buffer
          HIPCHK(launch_bilinear_adjoint(P->dtype, E.grad(op.out), E.grad(op.in), t.N, t.H, t.W, t.C, op.up, E.st));
          ga.g = E.grad(op.in);
        } else {
          ga.g = E.grad(op.out);
        }
        ga.dy = E.grad(op.in);
        E.ginit[op.in] = 1;
        HIPCHK(launch_gn_backward(P->dtype, ga, t.N, E.st));
        break;
      }
      case OP_UP2: {        // gradient of the nearest x2: 2x2 sums into the coarser level
        const TensorInfo& t = P->tensors[op.in];
        const int acc = E.claim(op.in);
        HIPCHK(launch_pool2x2_accum(P->dtype, E.grad(op.in), E.grad(op.out), t.N, t.H, t.W, t.C, acc ? 0 : 1, E.st));
        break;
      }
      case OP_MAXPOOL: {
        const TensorInfo& t = P->tensors[op.in];
        const int acc = E.claim(op.in);
        HIPCHK(launch_maxpool_bwd_idx(P->dtype, (const unsigned char*)(E.ws + P->pool_idx_off), E.grad(op.out), E.grad(op.in), t.N, t.H, t.W,
                                      t.C, acc ? 0 : 1, E.st));
        break;
      }
    }
    if (S)
      for (int k = 0; k < S->n; ++k)
        if (S->last_op[k] == oi) { rc = fire(k); if (rc) return rc; }
  }
  if (S)
    for (int k = 0; k < S->n; ++k)
      if (S->last_op[k] < 0) { rc = fire(k); if (rc) return rc; }   // (a slice nobody writes: zeros, still part of the exchange)
  if (E.wst && E.wst != E.st) {   // join: the caller's stream owns the complete gradient arena again
    HIPCHK(hipEventRecord(P->ev_join, E.wst));
    HIPCHK(hipStreamWaitEvent(E.st, P->ev_join, 0));
  }
  return OCTSEG_OK;
}

// ================================================================ C ABI
extern "C" {

int octseg_version(void) { return 100; }
const char* octseg_last_error(void) { return g_err.c_str(); }

int octseg_plan_create(const octseg_net_desc* d, octseg_plan** out) {
  if (!d || !out) return fail(OCTSEG_BAD_ARG, "null argument");
  *out = nullptr;
  if (d->dtype != OCTSEG_F32 && d->dtype != OCTSEG_BF16 && d->dtype != OCTSEG_F16) return fail(OCTSEG_BAD_DTYPE, "dtype must be f32, bf16 or f16");
  if (d->height <= 0 || d->width <= 0 || d->height % 32 != 0 || d->width % 32 != 0) {
    char buf[256];
    snprintf(buf, sizeof buf, "Wrong input shape height=%d, width=%d. Expected image height and width divisible by 32.",
             d->height, d->width);
    return fail(OCTSEG_BAD_SHAPE, buf);
  }
  if (d->batch <= 0 || d->classes <= 0 || d->classes > 16) return fail(OCTSEG_BAD_SHAPE, "batch > 0 and 1 <= classes <= 16 required");
  const std::string enc = lower(d->encoder);
  if (enc != "resnet18" && enc != "resnet34" && enc != "resnet50" && enc != "resnet101" && enc != "resnet152" && enc != "timm-regnetx_002" &&
      enc != "timm-regnetx_064" && enc != "timm-regnety_120" && enc != "efficientnet-b0" && enc != "efficientnet-b5" && enc != "efficientnet-b7")
    return fail(OCTSEG_UNSUPPORTED_ARCH, "unknown encoder '" + enc + "' (resnet18 | resnet34 | resnet50 | resnet101 | resnet152 | timm-regnetx_002 | timm-regnetx_064 | timm-regnety_120 | efficientnet-b0 | efficientnet-b5 | efficientnet-b7)");
  octseg_plan* P = new octseg_plan();
  P->arch = lower(d->arch); P->encoder = enc; P->classes = d->classes;
  P->B = d->batch; P->H = d->height; P->W = d->width; P->dtype = d->dtype;
  const int rc = build_plan(P);
  if (rc) { delete P; return rc; }
  *out = P;
  return OCTSEG_OK;
}
int octseg_plan_destroy(octseg_plan* p) {
  if (p) {
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->side_bwd) (void)hipStreamDestroy(p->side_bwd);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->ev_slice) (void)hipEventDestroy(p->ev_slice);
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->tgraph_exec) (void)hipGraphExecDestroy(p->tgraph_exec);
  }
  delete p;
  return OCTSEG_OK;
}
size_t octseg_plan_workspace_bytes(const octseg_plan* p) { return p ? p->ws_bytes : 0; }
size_t octseg_plan_param_numel(const octseg_plan* p) { return p ? p->param_numel : 0; }
size_t octseg_plan_buffer_numel(const octseg_plan* p) { return p ? p->buffer_numel : 0; }
int octseg_plan_num_params(const octseg_plan* p) { return p ? (int)p->params.size() : 0; }
int octseg_plan_num_bn(const octseg_plan* p) { return p ? (int)p->bns.size() : 0; }
double octseg_plan_fwd_macs(const octseg_plan* p) { return p ? p->fwd_macs : 0.0; }
int octseg_plan_exec_macs(const octseg_plan* p, double* out3) {
  if (p == nullptr || out3 == nullptr) return OCTSEG_BAD_ARG;
  for (int k = 0; k < 3; ++k) out3[k] = p->exec_macs[k];
  return OCTSEG_OK;
}

int octseg_plan_param_info(const octseg_plan* p, int i, octseg_param_info* o) {
  if (!p || !o || i < 0 || i >= (int)p->params.size()) return fail(OCTSEG_BAD_ARG, "param index out of range");
  const ParamInfo& q = p->params[i];
  memset(o, 0, sizeof(*o));
  snprintf(o->name, sizeof o->name, "%s", q.name.c_str());
  o->kind = q.kind; o->R = q.R; o->S = q.S; o->O = q.O; o->I = q.I; o->KP = q.KP; o->offset = q.off; o->numel = q.numel;
  return OCTSEG_OK;
}
int octseg_plan_bn_info(const octseg_plan* p, int i, octseg_bn_info* o) {
  if (!p || !o || i < 0 || i >= (int)p->bns.size()) return fail(OCTSEG_BAD_ARG, "bn index out of range");
  const BNInfo& b = p->bns[i];
  memset(o, 0, sizeof(*o));
  snprintf(o->name, sizeof o->name, "%s", b.name.c_str());
  o->C = b.C; o->mean_offset = b.rm_off; o->var_offset = b.rv_off;
  return OCTSEG_OK;
}

int octseg_profile_start(void) {
  for (auto& r : g_prof) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof.clear();
  g_prof_on = true;
  return OCTSEG_OK;
}
// out[3*k + {0,1,2}] = {milliseconds, algorithmic FLOPs, launches} of class k = 0 fwd, 1 dgrad, 2 wgrad.
// Synchronises the device (bench / test use only).
int octseg_profile_stop(double* out) {
  g_prof_on = false;
  if (!out) return fail(OCTSEG_BAD_ARG, "null argument");
  HIPCHK(hipDeviceSynchronize());
  for (int i = 0; i < 12; ++i) out[i] = 0.0;
  FILE* dump = nullptr;
  if (const char* path = getenv("OCTSEG_PROFILE_DUMP")) dump = fopen(path, "w");
  if (dump) fprintf(dump, "layer,class,ms,gflop,tflops\n");
  for (auto& r : g_prof) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    out[3 * r.kind] += ms; out[3 * r.kind + 1] += r.flops; out[3 * r.kind + 2] += 1.0;
    if (dump) fprintf(dump, "%s,%s,%.4f,%.3f,%.1f\n", r.name.c_str(), r.kind == 0 ? "fwd" : r.kind == 1 ? "dgrad" : r.kind == 2 ? "wgrad" : "hbm", ms,
                      r.flops / 1e9, ms > 0 ? r.flops / (ms * 1e-3) / 1e12 : 0.0);
  }
  if (dump) fclose(dump);
  for (auto& r : g_prof) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof.clear();
  return OCTSEG_OK;
}

// debug / test hook: byte offsets (inside the workspace) of a conv layer's raw output and its gradient
int octseg_plan_find_tensor(const octseg_plan* p, const char* conv_name, size_t* act_off, size_t* grad_off, int* dims) {
  if (!p || !conv_name) return fail(OCTSEG_BAD_ARG, "null argument");
  for (auto& L : p->convs)
    if (L.name == conv_name && L.out >= 0) {
      const TensorInfo& t = p->tensors[L.out];
      if (act_off) *act_off = t.off;
      if (grad_off) *grad_off = t.goff;
      if (dims) { dims[0] = t.N; dims[1] = t.H; dims[2] = t.W; dims[3] = t.C; }
      return OCTSEG_OK;
    }
  return fail(OCTSEG_BAD_ARG, std::string("no conv layer named ") + conv_name);
}

// The packed weight images in the workspace are reused until the caller says the parameters changed
// (optimizer step, load_state_dict); a fresh plan / another workspace or arena repacks by itself.
int octseg_plan_params_changed(octseg_plan* p) {
  if (!p) return fail(OCTSEG_BAD_ARG, "null argument");
  p->packed_valid = false;
  return OCTSEG_OK;
}

int octseg_net_forward(octseg_plan* p, const float* params, float* buffers, void* workspace, const float* image,
                       float* logits, int normalize, const float* mean, const float* stdv, int train, void* stream) {
  if (!p || !params || !buffers || !workspace || !image || !logits) return fail(OCTSEG_BAD_ARG, "null argument");
  if (normalize && (!mean || !stdv)) return fail(OCTSEG_BAD_ARG, "normalize=1 needs mean/std");
  if (train && p->dtype == OCTSEG_F16)
    return fail(OCTSEG_BAD_DTYPE, "f16 is a serving dtype (eval forwards, reference predict.py); train in bf16 or f32");
  Exec E{p, params, nullptr, buffers, (char*)workspace, (hipStream_t)stream, train};
  if (train || !p->graph_enabled) return run_forward(E, image, logits, normalize, mean, stdv);
  // ---- eval forward through a hipGraph
  octseg_plan::GraphKey key{params, buffers, workspace, image, logits, stream, normalize, {0, 0, 0}, {1, 1, 1}};
  if (normalize) for (int i = 0; i < 3; ++i) { key.mean[i] = mean[i]; key.stdv[i] = stdv[i]; }
  if (!(key == p->graph_key)) {   // new argument set: drop the old graph, start over with eager calls
    if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }
    p->graph_key = key; p->graph_seen = 0;
  }
  // weight images are packed outside the graph (a replay never repacks; octseg_plan_params_changed brings us here)
  int rc = pack_all_weights(E, true);
  if (rc) return rc;
  if (p->graph_exec) { HIPCHK(hipGraphLaunch(p->graph_exec, E.st)); return OCTSEG_OK; }
  if (p->graph_seen++ == 0) return run_forward(E, image, logits, normalize, mean, stdv);   // eager warm-up call
  hipGraph_t g = nullptr;
  HIPCHK(hipStreamBeginCapture(E.st, hipStreamCaptureModeThreadLocal));
  rc = run_forward(E, image, logits, normalize, mean, stdv);
  const hipError_t ce = hipStreamEndCapture(E.st, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (ce != hipSuccess) { if (g) (void)hipGraphDestroy(g); return fail(OCTSEG_HIP_ERROR, hipGetErrorString(ce)); }
  const hipError_t ie = hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess) { p->graph_exec = nullptr; return fail(OCTSEG_HIP_ERROR, hipGetErrorString(ie)); }
  HIPCHK(hipGraphLaunch(p->graph_exec, E.st));
  return OCTSEG_OK;
}

// Eval-mode forwards of this plan are captured into a hipGraph and replayed while the argument set (pointers,
// stream, normalisation constants) stays the same.  Training calls are never captured.
int octseg_plan_set_graph(octseg_plan* p, int enable) {
  if (!p) return fail(OCTSEG_BAD_ARG, "null argument");
  p->graph_enabled = enable != 0;
  if (!enable && p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; p->graph_seen = 0; }
  return OCTSEG_OK;
}

// FPN's Dropout2d(0.2) (smp decoders/fpn: self.dropout after the merge): the keep pattern of the NEXT training forward(s), device float
// [B][128] of 0 / 1 (kept channels are scaled by 1 / (1 - p) as torch does).  The caller draws it (the reference's pattern comes from
// torch's global RNG and is not reproducible across implementations anyway); the backward reuses the same pointer.
int octseg_plan_set_dropout(octseg_plan* p, const float* keep_dev) {
  if (!p) return fail(OCTSEG_BAD_ARG, "null argument");
  p->dropout_keep = keep_dev;
  return OCTSEG_OK;
}

// Training-input augmentation on the GPU (dataset.py:160-207): see augment.hip.  img [B,3,H,W] f32 BGR 0..255, mask
// [B,classes,H,W] f32 0/1, params device f32 [B][OCTSEG_AUG_NPARAM]; outputs must not alias the inputs.
int octseg_augment(const float* img, const float* mask, float* img_out, float* mask_out, const float* params, int B, int classes, int H,
                   int W, void* stream) {
  if (!img || !mask || !img_out || !mask_out || !params) return fail(OCTSEG_BAD_ARG, "null argument");
  if (B <= 0 || classes <= 0 || H <= 0 || W <= 0) return fail(OCTSEG_BAD_SHAPE, "augment: empty batch or frame");
  if (img == img_out || mask == mask_out) return fail(OCTSEG_BAD_ARG, "augment: outputs must not alias the inputs (gather)");
  HIPCHK(launch_augment(img, mask, img_out, mask_out, params, B, classes, H, W, (hipStream_t)stream));
  return OCTSEG_OK;
}

// Serving epilogue of predict.py:92-100: sigmoid(logits[:, ch]) > 0.5, nearest resize (index tables: cv2 INTER_NEAREST) to out_h x out_w,
// written to channel out_ch of the NHWC mask stack out[N][out_h][out_w][out_channels] (f32 0/1).
int octseg_mask_assemble(const float* logits, int N, int classes, int H, int W, int ch, float* out, int out_h, int out_w,
                         int out_channels, int out_ch, const int* row_index, const int* col_index, void* stream) {
  if (!logits || !out) return fail(OCTSEG_BAD_ARG, "null argument");
  if (N <= 0 || classes <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || ch < 0 || ch >= classes || out_ch < 0 ||
      out_ch >= out_channels)
    return fail(OCTSEG_BAD_SHAPE, "mask_assemble: channel / extent out of range");
  HIPCHK(launch_mask_assemble(logits, N, classes, H, W, ch, out, out_h, out_w, out_channels, out_ch, row_index, col_index,
                              (hipStream_t)stream));
  return OCTSEG_OK;
}

int octseg_dice_forward(octseg_plan* p, void* workspace, const float* logits, const float* target, float* loss,
                        long long* stats, void* stream) {
  if (!p || !workspace || !logits || !target || !loss) return fail(OCTSEG_BAD_ARG, "null argument");
  DiceArgs a;
  memset(&a, 0, sizeof(a));
  a.logits = logits; a.target = target; a.B = p->B; a.C = p->classes; a.HW = (size_t)p->H * p->W;
  a.sums = (double*)((char*)workspace + p->dice_off); a.stats = stats; a.loss = loss; a.loss_kind = p->loss_kind;
  HIPCHK(launch_dice_fwd(a, (hipStream_t)stream));
  return OCTSEG_OK;
}

int octseg_net_backward(octseg_plan* p, const float* params, float* grads, void* workspace, const float* logits,
                        const float* target, float grad_scale, void* stream) {
  if (!p || !params || !grads || !workspace || !logits || !target) return fail(OCTSEG_BAD_ARG, "null argument");
  if (p->dtype == OCTSEG_F16) return fail(OCTSEG_BAD_DTYPE, "f16 is a serving dtype: no backward");
  Exec E{p, params, grads, nullptr, (char*)workspace, (hipStream_t)stream, 1};
  return run_backward(E, logits, target, grad_scale);
}

// One training step's device work in one call: forward (batch statistics) + Dice (+ confusion counts) + backward -- what training_step
// and loss.backward() enqueue in the reference (src/models/smp/model.py:73-95, Lightning's automatic optimisation).  With
// octseg_plan_set_train_graph(plan, 1) the call is captured into a hipGraph (second call with an unchanged argument set) and replayed:
// the ~800 launches, the weight-gradient side stream, the forward lane and their event edges become ONE launch -- the host cost of a
// step drops from tens of milliseconds of enqueueing to microseconds, which is what small per-GPU batches (strong scaling) need.
// The replay runs the very launches of the eager call (weight images are repacked inside: the parameters change every step).
int octseg_net_train_step(octseg_plan* p, const float* params, float* grads, float* buffers, void* workspace, const float* image,
                          const float* target, float* logits, float* loss, long long* stats, int normalize, const float* mean,
                          const float* stdv, float grad_scale, void* stream) {
  if (!p || !params || !grads || !buffers || !workspace || !image || !target || !logits || !loss)
    return fail(OCTSEG_BAD_ARG, "null argument");
  if (normalize && (!mean || !stdv)) return fail(OCTSEG_BAD_ARG, "normalize=1 needs mean/std");
  if (p->dtype == OCTSEG_F16) return fail(OCTSEG_BAD_DTYPE, "f16 is a serving dtype (eval forwards, reference predict.py); train in bf16 or f32");
  hipStream_t st = (hipStream_t)stream;
  auto body = [&]() -> int {
    Exec Ef{p, params, nullptr, buffers, (char*)workspace, st, 1};
    int rc = run_forward(Ef, image, logits, normalize, mean, stdv);
    if (rc) return rc;
    DiceArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = logits; a.target = target; a.B = p->B; a.C = p->classes; a.HW = (size_t)p->H * p->W;
    a.sums = (double*)((char*)workspace + p->dice_off); a.stats = stats; a.loss = loss; a.loss_kind = p->loss_kind;
    HIPCHK(launch_dice_fwd(a, st));
    Exec Eb{p, params, grads, nullptr, (char*)workspace, st, 1};
    return run_backward(Eb, logits, target, grad_scale);
  };
  if (!p->tgraph_enabled || serial_mode()) return body();
  octseg_plan::TrainKey key{params, grads, buffers, workspace, image, target, logits, loss, stats, stream, p->dropout_keep, p->drop_connect,
                            normalize, {0, 0, 0}, {1, 1, 1}, grad_scale};
  if (normalize) for (int i = 0; i < 3; ++i) { key.mean[i] = mean[i]; key.stdv[i] = stdv[i]; }
  if (!(key == p->tgraph_key)) {
    if (p->tgraph_exec) { (void)hipGraphExecDestroy(p->tgraph_exec); p->tgraph_exec = nullptr; }
    p->tgraph_key = key; p->tgraph_seen = 0;
  }
  // a replay repacks the UNFOLDED training weight images and rewrites the BatchNorm scale / shift in the workspace behind the host
  // cache's back: drop the cache, or an eval forward of this plan would take the hit and run its folded epilogue on unfolded images
  if (p->tgraph_exec) { HIPCHK(hipGraphLaunch(p->tgraph_exec, st)); p->packed_valid = false; return OCTSEG_OK; }
  if (p->tgraph_seen++ == 0) return body();   // eager warm-up: function attributes, job tables, side stream and events exist afterwards
  p->packed_valid = false;                    // the captured step must contain the weight packing (a replay meets new parameters)
  hipGraph_t g = nullptr;
  HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  g_capturing = true;
  const int rc = body();
  g_capturing = false;
  const hipError_t ce = hipStreamEndCapture(st, &g);
  p->packed_valid = false;                    // (nothing was executed: the images in the workspace are whatever the last real step left)
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (ce != hipSuccess) { if (g) (void)hipGraphDestroy(g); return fail(OCTSEG_HIP_ERROR, std::string("training-step capture: ") + hipGetErrorString(ce)); }
  const hipError_t ie = hipGraphInstantiate(&p->tgraph_exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess) { p->tgraph_exec = nullptr; return fail(OCTSEG_HIP_ERROR, hipGetErrorString(ie)); }
  HIPCHK(hipGraphLaunch(p->tgraph_exec, st));
  p->packed_valid = false;
  return OCTSEG_OK;
}
// Loss behind octseg_dice_forward / the backward's dL/dlogits: smp DiceLoss (the reference, model.py:55), mean BCE-with-logits, or
// their sum.  A captured training step holds the old kind's kernels' arguments: drop it.
int octseg_plan_set_loss(octseg_plan* p, int kind) {
  if (!p) return fail(OCTSEG_BAD_ARG, "null argument");
  if (kind < LOSS_DICE || kind > LOSS_DICE_BCE) return fail(OCTSEG_BAD_ARG, "loss kind must be 0 (dice), 1 (bce) or 2 (dice + bce)");
  if (kind != p->loss_kind && p->tgraph_exec) { (void)hipGraphExecDestroy(p->tgraph_exec); p->tgraph_exec = nullptr; p->tgraph_seen = 0; }
  p->loss_kind = kind;
  return OCTSEG_OK;
}
// EfficientNet's drop_connect (efficientnet_pytorch.utils.drop_connect inside MBConvBlock.forward): the caller draws the per-sample keep
// decisions -- randomness stays with the caller, as with Dropout -- and hands over the FACTORS keep / (1 - rate).
int octseg_plan_set_drop_connect(octseg_plan* p, const float* factors_dev) {
  if (!p) return fail(OCTSEG_BAD_ARG, "null argument");
  p->drop_connect = factors_dev;
  return OCTSEG_OK;
}
int octseg_plan_num_drop_connect(const octseg_plan* p) { return p ? (int)p->dc_rates.size() : 0; }
float octseg_plan_drop_connect_rate(const octseg_plan* p, int i) { return (p && i >= 0 && i < (int)p->dc_rates.size()) ? p->dc_rates[i] : -1.f; }
int octseg_plan_set_train_graph(octseg_plan* p, int enable) {
  if (!p) return fail(OCTSEG_BAD_ARG, "null argument");
  p->tgraph_enabled = enable != 0;
  if (!enable && p->tgraph_exec) { (void)hipGraphExecDestroy(p->tgraph_exec); p->tgraph_exec = nullptr; p->tgraph_seen = 0; }
  return OCTSEG_OK;
}

// Data-parallel backward: the same launches as octseg_net_backward; the gradient arena is cut into `nslices` contiguous,
// parameter-aligned ranges and `cb(user, k, begin, end)` (element offsets) is called on the calling host thread as soon as the
// last launch that writes into slice k has been enqueued -- `comm_stream` has by then been made to wait for it, so a collective
// the callback enqueues on comm_stream runs beside the rest of the backward (reference: the bucketed, overlapped gradient
// all-reduce of torch DDP that Lightning sets up, src/models/smp/train.py:122-133).  Slices complete in backward order
// (head / decoder parameters first); every slice is reported exactly once.
int octseg_net_backward_sliced(octseg_plan* p, const float* params, float* grads, void* workspace, const float* logits,
                               const float* target, float grad_scale, void* stream, int nslices, void* comm_stream,
                               octseg_slice_cb cb, void* user) {
  if (!p || !params || !grads || !workspace || !logits || !target || !cb) return fail(OCTSEG_BAD_ARG, "null argument");
  if (nslices < 1 || nslices > 64) return fail(OCTSEG_BAD_ARG, "1 <= nslices <= 64 required");
  if (p->dtype == OCTSEG_F16) return fail(OCTSEG_BAD_DTYPE, "f16 is a serving dtype: no backward");
  if (!comm_stream || comm_stream == stream) return fail(OCTSEG_BAD_ARG, "comm_stream must be a stream of its own");
  Exec E{p, params, grads, nullptr, (char*)workspace, (hipStream_t)stream, 1};
  SliceCtx S;
  S.n = nslices; S.comm = (hipStream_t)comm_stream; S.cb = cb; S.user = user;
  return run_backward(E, logits, target, grad_scale, &S);
}

int octseg_optim_step(int kind, float* params, const float* grads, float* m, float* v, size_t numel, float lr,
                      float wd, int step, float grad_scale, void* stream) {
  if (!params || !grads) return fail(OCTSEG_BAD_ARG, "null argument");
  if (kind < 0 || kind > 3) return fail(OCTSEG_BAD_ARG, "optimizer kind must be 0..3");
  if ((kind == 1 || kind == 3) && (!m || !v)) return fail(OCTSEG_BAD_ARG, "Adam/RAdam need both state arenas");
  if (kind == 2 && !v) return fail(OCTSEG_BAD_ARG, "RMSprop needs the second-moment arena");
  OptArgs a;
  memset(&a, 0, sizeof(a));
  a.p = params; a.g = grads; a.m = m; a.v = v; a.n = numel; a.kind = kind; a.lr = lr; a.wd = wd;
  a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f; a.alpha = 0.99f; a.momentum = 0.f; a.step = step; a.grad_scale = grad_scale;
  HIPCHK(launch_optim_step(a, (hipStream_t)stream));
  return OCTSEG_OK;
}

// ---------------------------------------------------------------- single-op entry points
static bool geom_ok(int dtype, int Cin, int Cout, int R, int S, int stride, int transposed) {
  const int v = dtype == OCTSEG_F32 ? 4 : 8;
  (void)Cout;
  if (Cin % v != 0) return false;
  if (R != S || R < 1 || R > 7) return false;
  if (stride != 1 && stride != 2) return false;
  if (transposed && !(R == 4 && stride == 2)) return false;
  return true;
}
static unsigned long long* g_stamp = nullptr;
static bool g_serial = false;   // octseg_debug_set_serial: one stream, no lanes (isolated kernel durations)
// diagnostic builds only (-DOCTSEG_STAMP): device buffer of 6 u64 receiving the per-phase cycle sums
int octseg_debug_set_stamp(unsigned long long* dev_buf) { g_stamp = dev_buf; return OCTSEG_OK; }
int octseg_debug_set_serial(int on) { g_serial = on != 0; g_prof_hbm = g_serial; return OCTSEG_OK; }

static Geom op_geom(int N, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int transposed) {
  Geom g{R, S, stride, pad, transposed != 0, N, H, W, Cin, 0, 0, Cout};
  if (transposed) { g.OH = H * 2; g.OW = W * 2; } else { g.OH = (H + 2 * pad - R) / stride + 1; g.OW = (W + 2 * pad - S) / stride + 1; }
  return g;
}
size_t octseg_conv2d_scratch_bytes(int dtype, int N, int H, int W, int Cin, int Cout, int R, int S) {
  // upper bound over stride / transposed variants: both images use rows padded to <= 128 and K to <= 64 elements
  (void)N; (void)H; (void)W;
  const size_t esz = dtype == OCTSEG_F32 ? 4 : 2;
  const size_t rows_f = ((size_t)Cout + 127) / 128 * 128, k_f = ((size_t)Cin + 63) / 64 * 64;
  const size_t rows_d = ((size_t)Cin + 127) / 128 * 128, k_d = ((size_t)Cout + 63) / 64 * 64;
  return align_up((size_t)R * S * rows_f * k_f * esz) + align_up((size_t)R * S * rows_d * k_d * esz);
}

int octseg_conv2d_forward(int dtype, const void* x, const float* w, const float* bias, void* y, int N, int H, int W,
                          int Cin, int Cout, int R, int S, int stride, int pad, int transposed, void* scratch, void* stream) {
  if (!geom_ok(dtype, Cin, Cout, R, S, stride, transposed)) return fail(OCTSEG_BAD_SHAPE, "unsupported conv geometry");
  hipStream_t st = (hipStream_t)stream;
  const Geom g = op_geom(N, H, W, Cin, Cout, R, S, stride, pad, transposed);
  std::vector<ConvArgs> la;
  fwd_launches(g, la);
  const ConvPackInfo pk = conv_pack_info(la[0], dtype);
  HIPCHK(launch_pack_weight_image(dtype, w, scratch, R * S, Cout, Cin, 0, pk, st));
  for (auto& a : la) {
    SrcDesc s; s.ptr = x; s.scale = nullptr; s.shift = nullptr; s.C = Cin; s.c0 = 0; s.H = H; s.W = W; s.up = 0; s.relu = 0;
    a.src[0] = s; a.nsrc = 1; a.W = scratch; a.bias = bias;
    if (!transposed) { a.Wmaster = w; a.wO = Cout; a.wI = Cin; a.wtrans = 0; }
    DstDesc d; d.ptr = y; d.C = Cout; d.c0 = 0; d.cn = Cout; d.H = g.OH; d.W = g.OW; d.accum = 0; d.pool = 0;
    a.dst[0] = d; a.ndst = 1; a.out_mode = OUT_STORE; a.stat_slab = nullptr; a.stamp = g_stamp;
    HIPCHK(launch_conv(dtype, a, st));
  }
  return OCTSEG_OK;
}

int octseg_conv2d_backward_data(int dtype, const void* dy, const float* w, void* dx, int N, int H, int W, int Cin,
                                int Cout, int R, int S, int stride, int pad, int transposed, void* scratch, void* stream) {
  const int v = dtype == OCTSEG_F32 ? 4 : 8;
  if (!geom_ok(dtype, Cin, Cout, R, S, stride, transposed) || Cout % v != 0) return fail(OCTSEG_BAD_SHAPE, "unsupported conv geometry");
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == OCTSEG_F32 ? 4 : 2;
  const Geom g = op_geom(N, H, W, Cin, Cout, R, S, stride, pad, transposed);
  HIPCHK(hipMemsetAsync(dx, 0, (size_t)N * H * W * Cin * esz, st));
  std::vector<ConvArgs> ld;
  dgrad_launches(g, ld);
  ConvArgs d0 = ld[0];
  for (auto& d : ld) if (d.ntaps > 0) { d0 = d; break; }
  d0.Cin = Cout;
  const ConvPackInfo pk = conv_pack_info(d0, dtype);
  HIPCHK(launch_pack_weight_image(dtype, w, scratch, R * S, Cout, Cin, 1, pk, st));
  for (auto& a : ld) {
    SrcDesc s; s.ptr = dy; s.scale = nullptr; s.shift = nullptr; s.C = Cout; s.c0 = 0; s.H = g.OH; s.W = g.OW; s.up = 0; s.relu = 0;
    a.src[0] = s; a.nsrc = 1; a.Cin = Cout; a.W = scratch; a.bias = nullptr;
    if (!transposed) { a.Wmaster = w; a.wO = Cout; a.wI = Cin; a.wtrans = 1; }
    DstDesc d; d.ptr = dx; d.C = Cin; d.c0 = 0; d.cn = Cin; d.H = H; d.W = W; d.accum = 1; d.pool = 0;
    a.dst[0] = d; a.ndst = 1; a.out_mode = OUT_ACCUM; a.stat_slab = nullptr;
    HIPCHK(launch_conv(dtype, a, st));
  }
  return OCTSEG_OK;
}

int octseg_conv2d_backward_weight(int dtype, const void* x, const void* dy, float* dw, int N, int H, int W, int Cin,
                                  int Cout, int R, int S, int stride, int pad, int transposed, void* stream) {
  const int v = dtype == OCTSEG_F32 ? 4 : 8;
  if (!geom_ok(dtype, Cin, Cout, R, S, stride, transposed) || Cout % v != 0) return fail(OCTSEG_BAD_SHAPE, "unsupported conv geometry");
  hipStream_t st = (hipStream_t)stream;
  Geom g{R, S, stride, pad, transposed != 0, N, H, W, Cin, 0, 0, Cout};
  if (transposed) { g.OH = H * 2; g.OW = W * 2; } else { g.OH = (H + 2 * pad - R) / stride + 1; g.OW = (W + 2 * pad - S) / stride + 1; }
  HIPCHK(hipMemsetAsync(dw, 0, (size_t)R * S * Cout * Cin * sizeof(float), st));
  std::vector<WgradArgs> lw;
  wgrad_launches(g, lw);
  for (auto& a : lw) {
    SrcDesc s; s.ptr = x; s.scale = nullptr; s.shift = nullptr; s.C = Cin; s.c0 = 0; s.H = H; s.W = W; s.up = 0; s.relu = 0;
    a.src[0] = s; a.nsrc = 1; a.dy = dy; a.dyC = Cout; a.dW = dw; a.stamp = g_stamp;
  }
  if (transposed && lw.size() == 4 && wgrad_convt16_eligible(lw[0], dtype)) {   // the plan's route for a ConvTranspose2d (conv_backward)
    HIPCHK(launch_wgrad_convt16(dtype, lw[0], st));
    return OCTSEG_OK;
  }
  for (auto& a : lw) HIPCHK(launch_wgrad(dtype, a, st));
  return OCTSEG_OK;
}

}  // extern "C"

static bool serial_mode() { return g_serial; }

static int g_deterministic = -1;   // -1: not decided yet (environment)
bool octseg::deterministic_mode() {
  if (g_deterministic < 0) g_deterministic = getenv("OCTSEG_DETERMINISTIC") != nullptr ? 1 : 0;
  return g_deterministic != 0;
}
extern "C" int octseg_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return OCTSEG_OK; }
