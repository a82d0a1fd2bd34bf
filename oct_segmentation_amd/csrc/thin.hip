// thin.hip -- the thin full-resolution 3x3 layers (<= 32 channels in, <= 32 out) as HBM-bound streaming kernels on gfx950.
//
// Which layers: the last decoder block and the head of U-Net / U-Net++ (reference: smp DecoderBlock conv1 / conv2 at full resolution and
// SegmentationHead, src/models/smp/model.py:38-44 -> smp.create_model; SURVEY Appendix B: x_0_4 32 -> 16 and 16 -> 16 @704^2, head 16 -> C),
// forward, data gradient and weight gradient.  Their arithmetic intensity is 8..72 FLOP/B: the roof is HBM (8 TB/s), not MFMA.
// Through conv_mfma_kernel / wgrad_mfma_kernel (128-byte K chunks, 32..64-channel N tiles, one 16x16-pixel tile per workgroup, a
// 64 x 64 x taps weight-gradient tile) they ran at 68..126 TFLOP/s = 1.2 TB/s of algorithmic bytes, 15 % of the HBM roof (round 2).
//
// Here: a persistent 256-thread workgroup walks 32-pixel-wide tiles; the (tile + halo) window is staged ONCE into LDS with the lazy
// BatchNorm + ReLU (and the nearest-x2 read) applied, and
//   * conv / dgrad (thin_conv_kernel): the WEIGHTS are the MFMA A operand, built once per workgroup from the fp32 master weights and
//     kept in registers for every tile (row = output channel, k = (tap, input channel) pairs packed into 32-wide k-steps: 5 steps for
//     16 channels, 9 for 32); the B operand is 16 pixels of a window row read with one ds_read_b128 per k-step (conflict-free: 32-byte
//     pixels, or 96-byte pitch for 64-byte pixels).  v_mfma_f32_16x16x32 leaves D[row = channel 4g + j][col = pixel] : every lane
//     owns 4 consecutive channels of one pixel = one 8-byte NHWC store, no LDS transpose.  BatchNorm partial sums stay in registers
//     over all tiles of a workgroup (one slab row per workgroup).  The gradient of a nearest-x2 upsample is summed in registers
//     (two rows into one accumulator, lane pairs by DPP) in f32 and rounded once.
//   * wgrad (thin_wgrad_kernel): contraction over pixels, both operands fetched with ds_read_b64_tr_b16 from the pixel-major images
//     (k order permuted so that a 32-lane half reads 8 consecutive pixels = 256 contiguous bytes: conflict-free), 9 x (CIN / 16)
//     accumulators of 16 x 16 per wave kept across all tiles of the workgroup, one LDS reduction + one round of fp32 atomics at the end.
// No packed weight image is read: both kernels take the fp32 master weights ([R][S][O][I]) and round them exactly as pack_all does.
//
// The same two kernels, by template KIND, also run the other layers that are bound by HBM rather than MFMA (round 3):
//   K11   1x1 convs with 16 / 32 channels in and <= 32 out on full-resolution maps (LinkNet's last decoder block and its head,
//         smp LinknetDecoder / SegmentationHead): one tap, no halo;
//   KT22  the four output-parity launches of a ConvTranspose2d(k4, s2, p1) forward with 16 / 32 channels (LinkNet's TransposeX2 at
//         full resolution): 2x2 taps, outputs written at stride 2;
//   KSTEM the ResNet stem, conv 7x7 stride 2 over the NCHW f32 frame (torchvision ResNet.conv1; the reference's forward normalises in
//         front of it, src/models/smp/model.py:65-71): the (pixel, 160-wide k) im2col rows of a 4 x 32-pixel tile are gathered and
//         normalised straight into LDS -- the 634 MB im2col tensor of rounds 1-2 (written, read by the GEMM, read again by its weight
//         gradient) no longer exists in training steps.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <cstdlib>
#include <cstring>

namespace octseg {

namespace {

constexpr int THIN_TW = 32;          // tile width in pixels (two 16-pixel MFMA column groups)
constexpr int THIN_NTHR = 256;
constexpr int THIN_MAXWG = 768;      // persistent workgroups (3 per CU: ~150 registers per lane)

enum { K33 = 0, K11 = 1, KT22 = 2, KSTEM = 3 };

struct ThinArgs {
  const char* x; const float* scale; const float* shift; int relu, up, sH, sW;   // source: NHWC T [N][sH][sW][CIN] (stored extent), lazy BN
  int N, OH, OW;                     // tile grid = virtual input extent (stride 1); KSTEM: the stem's output extent
  int ntaps, tdy[9], tdx[9], tw[9];  // window offset (0 .. 2 HALO) and master-weight tap of each tap
  const float* w; int wO, wI, wtrans; const float* wscale;   // fp32 master [taps][wO][wI] (KSTEM: [wO][160]); wtrans: rows over I, contraction over O
  int Cout;                          // output channels (rows of the A operand that are real)
  const float* bias; int relu_out;
  float* slab; int slab_row0;        // BN partial sums: one row per workgroup, [rows][Cout][2]
  char* y; int yC, yH, yW, accum, pool, head;   // destination (NHWC T; head: NCHW f32 [N][Cout][yH][yW])
  int os, ooy, oox;                  // output pixel = grid * os + (ooy, oox)
  int tiles_x, tiles_y, ntiles;
  const char* dy; int dyC; float* dW;   // wgrad only
  const float* img; int IH, IW, normalize; float mean[3], inv[3];   // KSTEM: NCHW f32 frame, (x - mean) * inv on the fly
};

template <int KIND, int CIN> struct ThinK {
  static constexpr bool STEM = KIND == KSTEM;
  static constexpr int NTAPS = KIND == K33 ? 9 : (KIND == KT22 ? 4 : 1);
  static constexpr int HALO = (KIND == K33 || KIND == KT22) ? 1 : 0;
  static constexpr int RW = THIN_TW + 2 * HALO;
  // LDS bytes per window pixel: 32-byte pixels as they are; 64-byte pixels at pitch 96 and the stem's 320-byte im2col rows at pitch 416:
  // then both the ds_read_b128 fragment reads (16 pixels x one 16-byte slice) and the transposed reads (8 pixels x one 32-byte block)
  // of a 32-lane group fall on distinct banks
  static constexpr int PIXB = STEM ? 416 : (CIN == 16 ? 32 : 96);
  static constexpr int KTOT = STEM ? 160 : NTAPS * CIN;
  static constexpr int STEPS = (KTOT + 31) / 32;      // 32-wide k-steps over the (tap, channel) pairs
  static constexpr int VPP = STEM ? 20 : CIN / 8;     // 16-byte vectors per pixel
  static constexpr int NCB = STEM ? 10 : CIN / 16;    // 16-wide column blocks of the weight gradient
};

// Stage the (TH + 2 HALO) x RW window of tile (n, y0, x0) into LDS: lazy BN + ReLU, nearest-x2 read, zero outside the image.
template <typename T, int KIND, int CIN, int TH>
__device__ __forceinline__ void thin_stage(const ThinArgs& a, char* lds, int n, int y0, int x0, int tid, const float (&sc)[8],
                                           const float (&sh)[8], bool has_aff) {
  typedef ThinK<KIND, CIN> K;
  constexpr int PIXB = K::PIXB, VPP = K::VPP, RW = K::RW, HALO = K::HALO;
  constexpr int NPX = (TH + 2 * HALO) * RW, NVEC = NPX * VPP, MAXV = (NVEC + THIN_NTHR - 1) / THIN_NTHR;
  const int cv = tid % VPP;
  const char* img = a.x + ((size_t)n * a.sH * a.sW * CIN + cv * 8) * sizeof(T);
  uint4 v[MAXV];
  bool ok[MAXV];
  int dst[MAXV];
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int vi = min(tid + u * THIN_NTHR, NVEC - 1);      // (the clamped tail re-stages the last vector: same value, harmless)
    const int px = vi / VPP;
    const int hy = px / RW;
    const int hx = px - hy * RW;
    const int iy = y0 - HALO + hy, ix = x0 - HALO + hx;
    ok[u] = (unsigned)iy < (unsigned)a.OH && (unsigned)ix < (unsigned)a.OW;
    const int iyc = min(max(iy, 0), a.OH - 1) >> a.up, ixc = min(max(ix, 0), a.OW - 1) >> a.up;
    v[u] = *(const uint4*)(img + ((size_t)iyc * a.sW + ixc) * (CIN * sizeof(T)));
    dst[u] = px * PIXB + cv * 16;
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    uint4 w = v[u];
    if (has_aff) w = Tr<T>::affine(w, sc, sh, a.relu);
    if (!ok[u]) w = make_uint4(0, 0, 0, 0);
    *(uint4*)(lds + dst[u]) = w;
  }
}

// Stem: the im2col rows of a TH x 32 tile of the 7x7 stride-2 conv, k = (r * 7 + s) * 3 + ci padded to 160, as [pixel][160] T at pitch
// PIXB.  Two phases: (1) the frame window the tile needs ((2 TH + 5) rows x 69 columns x 3 channels) comes in with coalesced row loads
// -- one round trip per tile -- is normalised ((x - mean) * inv; zero outside the frame: conv2d pads the NORMALISED input) and parked in
// LDS as T; (2) after a barrier every thread assembles its im2col vectors from that window with LDS reads (a 160-entry offset table, one
// 16-byte table read per vector).  (Gathering the 147 taps of every pixel from global memory took 8 scalar loads per vector and five
// dependent round trips per tile: 1.1 ms for the stem forward, measured.)
// Stem forward: k-steps of a tile unrolled by 2, two waves per SIMD.  Fully unrolled (six k-steps x four weight blocks of LDS reads hoisted
// in front of the MFMAs) the kernel sat at 256 registers with 63 spilled: 0.47 ms; unrolled by 2 it needs 216 and spills nothing: 0.32 ms
// (U-Net/resnet50 704^2 step 644 -> 650 frames/s).  Three waves per SIMD with 768 workgroups spill 49 registers: 0.44 ms.  Fewer, larger
// tiles (TH = 8) or 256 workgroups are slower (0.54 / 0.62 ms): the kernel lives on workgroups in flight, not on work per workgroup.
#ifndef STEM_WPS
#define STEM_WPS 2
#endif
#ifndef STEM_UNROLL
#define STEM_UNROLL 2
#endif
constexpr int STEM_WC = 72;                                     // window row pitch in elements (69 used)
template <int TH> struct StemWin {
  static constexpr int ROWS = 2 * TH + 5, ELEMS = 3 * ROWS * STEM_WC;
  static constexpr int BYTES = (ELEMS * 2 + 320 + 15) / 16 * 16;   // window (T = 2 bytes) + offset table (160 x u16)
};
// The three pieces of the stem staging; the tile loop issues the frame loads of tile t + 1 (stem_load) BEFORE it works on tile t, so the
// one global round trip per tile hides under the im2col assembly and the MFMAs of the tile before (two workgroups per CU cannot hide it).
template <int TH> struct StemRegs {
  static constexpr int ROWS = StemWin<TH>::ROWS, NEL = 3 * ROWS * 69, MAXE = (NEL + THIN_NTHR - 1) / THIN_NTHR;
  float v[MAXE];
};
template <int TH>
__device__ __forceinline__ void stem_load(const ThinArgs& a, StemRegs<TH>& rg, int tile, int tid) {
  constexpr int ROWS = StemRegs<TH>::ROWS, NEL = StemRegs<TH>::NEL, MAXE = StemRegs<TH>::MAXE;
  const int n = tile / (a.tiles_x * a.tiles_y);
  const int rem = tile - n * a.tiles_x * a.tiles_y;
  const int tyi = rem / a.tiles_x;
  const int gy0 = 2 * (tyi * TH) - 3, gx0 = 2 * ((rem - tyi * a.tiles_x) * THIN_TW) - 3;
  const size_t plane = (size_t)a.IH * a.IW;
  const float* img = a.img + (size_t)n * 3 * plane;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    const int e = min(tid + u * THIN_NTHR, NEL - 1);
    const int wc = e % 69, rr = e / 69, wr = rr % ROWS, ci = rr / ROWS;
    const int iy = gy0 + wr, ix = gx0 + wc;
    const bool ok = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
    const float raw = img[(size_t)ci * plane + (size_t)min(max(iy, 0), a.IH - 1) * a.IW + min(max(ix, 0), a.IW - 1)];
    rg.v[u] = ok ? (raw - a.mean[ci]) * a.inv[ci] : 0.f;     // (normalize = 0: mean 0, inv 1)
  }
}
// ILV = false: planar [ci][row][STEM_WC] (the weight gradient's im2col assembly);  ILV = true: [row][column * 3 + ci] at STEM_ROWB bytes
// per row -- the 21 taps (s, ci) of one kernel row of one output pixel are then 42 CONTIGUOUS bytes, which the forward reads as fragments
constexpr int STEM_ROWB = 416;                                  // 69 x 3 x 2 = 414 bytes used
template <typename T, int TH, bool ILV>
__device__ __forceinline__ void stem_store(const StemRegs<TH>& rg, char* aux, int tid) {
  constexpr int ROWS = StemRegs<TH>::ROWS, NEL = StemRegs<TH>::NEL, MAXE = StemRegs<TH>::MAXE;
  unsigned short* wimg = (unsigned short*)aux;
#pragma unroll
  for (int u = 0; u < MAXE; ++u) {
    const int e = min(tid + u * THIN_NTHR, NEL - 1);
    const int wc = e % 69, rr = e / 69, wr = rr % ROWS, ci = rr / ROWS;
    if constexpr (ILV) wimg[wr * (STEM_ROWB / 2) + wc * 3 + ci] = Tr<T>::bits16(rg.v[u]);
    else wimg[(ci * ROWS + wr) * STEM_WC + wc] = Tr<T>::bits16(rg.v[u]);
  }
}
template <int TH>
__device__ __forceinline__ void stem_table(char* aux, int tid) {
  unsigned short* otab = (unsigned short*)(aux + StemWin<TH>::ELEMS * 2);
  if (tid < 160) {
    const int tap = tid / 3, ci = tid - tap * 3, r = tap / 7, s = tap - r * 7;
    otab[tid] = tid < 147 ? (unsigned short)((ci * StemWin<TH>::ROWS + r) * STEM_WC + s) : (unsigned short)0xffff;
  }
}
template <int TH>
__device__ __forceinline__ void stem_build(char* lds, const char* aux, int tid) {
  constexpr int PIXB = ThinK<KSTEM, 16>::PIXB, NVEC = TH * THIN_TW * 20, PER = NVEC / THIN_NTHR;
  static_assert(NVEC % THIN_NTHR == 0, "whole passes");
  const unsigned short* wimg = (const unsigned short*)aux;
  const unsigned short* otab = (const unsigned short*)(aux + StemWin<TH>::ELEMS * 2);
#pragma unroll 2
  for (int u = 0; u < PER; ++u) {
    const int vi = tid + u * THIN_NTHR;
    const int px = vi / 20, kv = vi - px * 20;
    const int pbase = (2 * (px >> 5)) * STEM_WC + 2 * (px & 31);
    const uint4 o4 = *(const uint4*)(otab + kv * 8);
    const unsigned ow[4] = {o4.x, o4.y, o4.z, o4.w};
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned o0 = ow[i] & 0xffffu, o1 = ow[i] >> 16;
      const unsigned e0 = o0 == 0xffffu ? 0u : (unsigned)wimg[pbase + o0], e1 = o1 == 0xffffu ? 0u : (unsigned)wimg[pbase + o1];
      w[i] = e0 | (e1 << 16);
    }
    *(uint4*)(lds + px * PIXB + kv * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace

template <typename T, int KIND, int CIN, int NB, int TH>
__global__ __launch_bounds__(THIN_NTHR, NB == 1 ? 3 : (KIND == KSTEM ? STEM_WPS : 2)) void thin_conv_kernel(const ThinArgs a) {
  typedef ThinK<KIND, CIN> K;
  constexpr int PIXB = K::PIXB, RW = K::RW, HALO = K::HALO;
  constexpr bool STEM = K::STEM;
  // stem: k is re-laid as 7 kernel rows x 24 (21 taps (s, ci) + 3 of padding) = 6 k-steps; k-group q = 4 s + g is row q / 3, part q % 3:
  // eight CONSECUTIVE elements of the interleaved frame window -- the B fragment is four ds_read_b32 straight from the window, no im2col tile
  constexpr int STEPS = STEM ? 6 : K::STEPS;
  constexpr int RPW = TH / 4;                              // tile rows per wave
  constexpr int WINB = STEM ? 0 : (TH + 2 * HALO) * RW * PIXB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* win = smem;
  float* red = (float*)(smem + WINB);                      // [4 waves][NB * 16][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- A operand: the weights, rounded to T, in registers for the whole kernel.  Lane (row = r16, k-slice g) of k-step s holds
  // k = 32 s + 8 g + j, j = 0..7  <->  tap k / CIN, contraction channel k % CIN  (stem: k is the im2col column)
  uint4 wf[STEM ? 1 : NB][STEM ? 1 : STEPS];
  char* wlds = smem + WINB + 4 * NB * 16 * 2 * sizeof(float) + (STEM ? StemWin<STEM ? TH : 4>::BYTES + 64 : 0);   // (stem only)
  int toff[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {   // (stem: only the fragment offsets here; its weights go to LDS in a rolled loop below)
    const int sq = 4 * s + g, sr = sq / 3, spart = sq - 3 * sr;
    if constexpr (STEM) toff[s] = min(sr, 6) * STEM_ROWB + 16 * spart;
  }
#pragma unroll(STEM ? 1 : STEPS)
  for (int s = 0; s < STEPS; ++s) {
    const int k0 = 32 * s + 8 * g;
    const int sq = 4 * s + g, sr = sq / 3, spart = sq - 3 * sr;          // stem: kernel row, 8-element part of its 24
    const int ti = STEM ? 0 : k0 / CIN, c0 = STEM ? 8 * spart : k0 % CIN;
    const bool tv = STEM ? sr < 7 : ti < K::NTAPS;
    const int tc = STEM ? 0 : (tv ? ti : K::NTAPS - 1);
    if constexpr (!STEM) toff[s] = (a.tdy[tc] * RW + a.tdx[tc]) * PIXB + c0 * (int)sizeof(T);
    const int tw = a.tw[tc];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int row = b * 16 + r16;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        bool ok = tv && row < a.Cout;
        size_t idx;
        if (STEM) { ok = ok && c < 21; idx = (size_t)row * 160 + sr * 21 + c; }
        else if (a.wtrans) { ok = ok && c < a.wO && row < a.wI; idx = ((size_t)tw * a.wO + c) * a.wI + row; }
        else { ok = ok && row < a.wO && c < a.wI; idx = ((size_t)tw * a.wO + row) * a.wI + c; }
        float x = ok ? a.w[idx] : 0.f;
        if (ok && a.wscale != nullptr) x *= a.wscale[row];     // eval: BatchNorm scale folded into the weights (as pack_all does)
        v[j] = x;
      }
      const uint4 wv = make_uint4(Tr<T>::pk(v[0], v[1]), Tr<T>::pk(v[2], v[3]), Tr<T>::pk(v[4], v[5]), Tr<T>::pk(v[6], v[7]));
      // stem: 4 row blocks x 6 k-steps = 96 registers of weights would spill: they live in LDS, lane-linear (conflict-free b128 reads)
      if constexpr (STEM) { if (wave == 0) *(uint4*)(wlds + ((s * NB + b) * 64 + lane) * 16) = wv; }
      else wf[b][s] = wv;
    }
  }
  // lazy BN parameters of this thread's channel vector (staging)
  float sc[8], sh[8];
  const bool has_aff = !STEM && a.scale != nullptr;
  {
    const int cv = STEM ? 0 : tid % K::VPP;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = has_aff ? a.scale[cv * 8 + i] : 1.f; sh[i] = has_aff ? a.shift[cv * 8 + i] : 0.f; }
  }
  float bias[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = b * 16 + 4 * g + j;
      bias[b][j] = (a.bias != nullptr && ch < a.Cout) ? a.bias[ch] : 0.f;
    }
  float s1[NB][4], s2[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[b][j] = 0.f; s2[b][j] = 0.f; }
  const bool want_stats = a.slab != nullptr;
  const int ES = (int)sizeof(T);

  StemRegs<STEM ? TH : 1> srg;
  char* saux = (char*)red + 4 * NB * 16 * 2 * sizeof(float);
  if constexpr (STEM) {
    // the fragment of the last 8-element part of a kernel row reads 3 elements past its 21 taps (zero WEIGHTS there): what it reads must be
    // finite -- inside a row that is the next pixel's data, at the end of a row / of the window it is padding, zeroed once here (0 x NaN = NaN)
    for (int i = tid; i < (StemWin<TH>::BYTES + 64) / 4; i += THIN_NTHR) ((unsigned*)saux)[i] = 0u;
    stem_load<TH>(a, srg, blockIdx.x, tid);
  }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int rem = tile - n * a.tiles_x * a.tiles_y;
    const int tyi = rem / a.tiles_x;
    const int y0 = tyi * TH, x0 = (rem - tyi * a.tiles_x) * THIN_TW;
    __syncthreads();                       // every wave is done reading the previous window
    if constexpr (STEM) {
      stem_store<T, TH, true>(srg, saux, tid);
      stem_load<TH>(a, srg, min(tile + (int)gridDim.x, a.ntiles - 1), tid);   // the next tile's frame window: in flight from here on
    } else {
      thin_stage<T, KIND, CIN, TH>(a, win, n, y0, x0, tid, sc, sh, has_aff);
    }
    __syncthreads();
    if (!a.pool) {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int yy = wave * RPW + rr, y = y0 + yy;
#pragma unroll(NB >= 4 ? 1 : 2)
        for (int seg = 0; seg < 2; ++seg) {
          const char* base = STEM ? saux + (2 * yy) * STEM_ROWB + 12 * (seg * 16 + r16) : win + (yy * RW + seg * 16 + r16) * PIXB;
          f32x4_t acc[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll(STEM ? STEM_UNROLL : STEPS)
          for (int s = 0; s < STEPS; ++s) {
            uint4 bf;
            if constexpr (STEM) {   // 4-byte aligned (12 bytes per output pixel): four dwords
              const unsigned* q4 = (const unsigned*)(base + toff[s]);
              bf = make_uint4(q4[0], q4[1], q4[2], q4[3]);
            } else {
              bf = *(const uint4*)(base + toff[s]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              if constexpr (STEM) Tr<T>::mma16(*(const uint4*)(wlds + ((s * NB + b) * 64 + lane) * 16), bf, acc[b]);
              else Tr<T>::mma16(wf[b][s], bf, acc[b]);
            }
          }
          const int x = x0 + seg * 16 + r16;
          const bool pok = y < a.OH && x < a.OW;
          const int yo = y * a.os + a.ooy, xo = x * a.os + a.oox;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              val[j] = acc[b][j] + bias[b][j];
              if (a.relu_out) val[j] = clamp_lo(val[j], 0.f);
              if (want_stats && pok) { s1[b][j] += val[j]; s2[b][j] += val[j] * val[j]; }
            }
            const int ch0 = b * 16 + 4 * g;
            if (!pok || ch0 >= a.Cout) continue;
            if (a.head) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (ch0 + j < a.Cout) ((float*)a.y)[(((size_t)n * a.Cout + ch0 + j) * a.yH + yo) * a.yW + xo] = val[j];
            } else {
              uint2* gp = (uint2*)(a.y + ((((size_t)n * a.yH + yo) * a.yW + xo) * a.yC + ch0) * ES);
              if (a.accum) {
                const uint2 old = *gp;
                val[0] += Tr<T>::lo(old.x); val[1] += Tr<T>::hi(old.x); val[2] += Tr<T>::lo(old.y); val[3] += Tr<T>::hi(old.y);
              }
              *gp = make_uint2(Tr<T>::pk(val[0], val[1]), Tr<T>::pk(val[2], val[3]));
            }
          }
        }
      }
    } else if constexpr (KIND == K33 && RPW >= 2) {
      // gradient of a nearest-x2 upsample: the 2x2 quad is summed in f32 -- rows 2q and 2q + 1 into ONE accumulator, the two
      // pixels of a lane pair by a cross-lane add -- and rounded once, at half resolution
#pragma unroll
      for (int rp = 0; rp < RPW / 2; ++rp) {
        const int yy = wave * RPW + 2 * rp, y = y0 + yy;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
          f32x4_t acc[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dyr = 0; dyr < 2; ++dyr) {
            const char* base = win + ((yy + dyr) * RW + seg * 16 + r16) * PIXB;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
              const uint4 bf = *(const uint4*)(base + toff[s]);
#pragma unroll
              for (int b = 0; b < NB; ++b) Tr<T>::mma16(wf[b][s], bf, acc[b]);
            }
          }
          const int x = x0 + seg * 16 + r16;
          const bool pok = y < a.OH && x < a.OW && (r16 & 1) == 0;      // (OH, OW even: a quad is inside the image as a whole)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = acc[b][j] + __shfl_xor(acc[b][j], 1);
            const int ch0 = b * 16 + 4 * g;
            if (!pok || ch0 >= a.Cout) continue;
            uint2* gp = (uint2*)(a.y + ((((size_t)n * a.yH + (y >> 1)) * a.yW + (x >> 1)) * a.yC + ch0) * ES);
            if (a.accum) {
              const uint2 old = *gp;
              val[0] += Tr<T>::lo(old.x); val[1] += Tr<T>::hi(old.x); val[2] += Tr<T>::lo(old.y); val[3] += Tr<T>::hi(old.y);
            }
            *gp = make_uint2(Tr<T>::pk(val[0], val[1]), Tr<T>::pk(val[2], val[3]));
          }
        }
      }
    }
  }
  if (want_stats) {
    // fixed order: pixels of a lane, then the 16 lanes of a k-slice group, then the four waves -> deterministic
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = s1[b][j], v = s2[b][j];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
        if (r16 == 0) { red[((wave * NB * 16) + b * 16 + 4 * g + j) * 2] = u; red[((wave * NB * 16) + b * 16 + 4 * g + j) * 2 + 1] = v; }
      }
    __syncthreads();
    if (tid < NB * 16 && tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * NB * 16 + tid) * 2]; t2 += red[(w * NB * 16 + tid) * 2 + 1]; }
      float* o = a.slab + ((size_t)(a.slab_row0 + blockIdx.x) * a.Cout + tid) * 2;
      o[0] = t1; o[1] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient
typedef __attribute__((ext_vector_type(4))) short thin_s16x4_t;
typedef __attribute__((address_space(3))) thin_s16x4_t thin_lds_s16x4_t;

// DYB = 16-channel row blocks of dy (1, 2, or 4 for the stem's 64 output channels); dy pixels sit at pitch 32 / 96 / 160 bytes so that
// the transposed reads of a 32-lane half (8 consecutive pixels x one 32-byte block) fall on distinct banks
template <int DYB> struct ThinDy { static constexpr int PITCH = DYB == 1 ? 32 : (DYB == 2 ? 96 : 160); };

template <typename T, int KIND, int CIN, int DYB, int TH>
__global__ __launch_bounds__(THIN_NTHR, 2) void thin_wgrad_kernel(const ThinArgs a) {
  typedef ThinK<KIND, CIN> K;
  constexpr int PIXB = K::PIXB, NCB = K::NCB, NT = K::NTAPS, RW = K::RW, HALO = K::HALO;
  constexpr bool STEM = K::STEM;
  constexpr int RPW = TH / 4;
  constexpr int WINB = STEM ? TH * THIN_TW * PIXB : (TH + 2 * HALO) * RW * PIXB;
  constexpr int DYP = ThinDy<DYB>::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* win = smem;
  char* dyt = smem + WINB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;

  f32x4_t acc[NT][DYB][NCB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int d = 0; d < DYB; ++d)
#pragma unroll
      for (int b = 0; b < NCB; ++b) acc[t][d][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int toff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) toff[t] = STEM ? 0 : (a.tdy[t] * RW + a.tdx[t]) * PIXB;
  // transposed-read addressing (ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns; lane 4q + p supplies row q,
  // columns 4p..4p+3 and receives column lane & 15 of the 4 rows).  k order of a 32-pixel row: group g reads pixels 4g..4g+3 (first
  // read) and 16+4g..16+4g+3 (second), the same for both operands; a 32-lane half then touches 8 consecutive pixels per read.
  const int ya0 = (4 * g + q) * DYP + p * 8;
  const int xa0 = (4 * g + q) * PIXB + p * 8;

  float sc[8], sh[8];
  const bool has_aff = !STEM && a.scale != nullptr;
  {
    const int cv = STEM ? 0 : tid % K::VPP;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = has_aff ? a.scale[cv * 8 + i] : 1.f; sh[i] = has_aff ? a.shift[cv * 8 + i] : 0.f; }
  }
  const int ES = (int)sizeof(T);
  StemRegs<STEM ? TH : 1> srg;
  char* saux = dyt + TH * THIN_TW * DYP;
  if constexpr (STEM) { stem_table<TH>(saux, tid); stem_load<TH>(a, srg, blockIdx.x, tid); }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int rem = tile - n * a.tiles_x * a.tiles_y;
    const int tyi = rem / a.tiles_x;
    const int y0 = tyi * TH, x0 = (rem - tyi * a.tiles_x) * THIN_TW;
    __syncthreads();
    if constexpr (STEM) {
      stem_store<T, TH, false>(srg, saux, tid);
      __syncthreads();
      stem_load<TH>(a, srg, min(tile + (int)gridDim.x, a.ntiles - 1), tid);
      stem_build<TH>(win, saux, tid);
    } else {
      thin_stage<T, KIND, CIN, TH>(a, win, n, y0, x0, tid, sc, sh, has_aff);
    }
    {   // dy tile: TH x 32 pixels x DYB * 16 channels (two 16-byte vectors per block), zero outside the image
      constexpr int VPPY = 2 * DYB, NV = TH * THIN_TW * VPPY, PER = NV / THIN_NTHR;
      static_assert(NV % THIN_NTHR == 0, "whole passes");
      uint4 v[PER];
      bool ok[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int vi = tid + u * THIN_NTHR, px = vi / VPPY, cv = vi - px * VPPY;
        const int ty = px >> 5, tx = px & 31;
        const int y = y0 + ty, x = x0 + tx;
        ok[u] = y < a.OH && x < a.OW;
        const int yc = min(y, a.OH - 1), xc = min(x, a.OW - 1);
        v[u] = *(const uint4*)(a.dy + ((((size_t)n * a.OH + yc) * a.OW + xc) * a.dyC + cv * 8) * ES);
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int vi = tid + u * THIN_NTHR, px = vi / VPPY, cv = vi - px * VPPY;
        *(uint4*)(dyt + px * DYP + cv * 16) = ok[u] ? v[u] : make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int yy = wave * RPW + rr;
      const char* yrow = dyt + yy * THIN_TW * DYP + ya0;
      struct Pair { thin_s16x4_t lo, hi; };
      uint4 yf[DYB];
#pragma unroll
      for (int d = 0; d < DYB; ++d) {
        Pair ya;
        ya.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(yrow + d * 32));
        ya.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(yrow + d * 32 + 16 * DYP));
        yf[d] = __builtin_bit_cast(uint4, ya);
      }
      const char* xrow = win + yy * (STEM ? THIN_TW : RW) * PIXB + xa0;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NCB; ++b) {
          Pair xb;
          xb.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(xrow + toff[t] + b * 32));
          xb.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(xrow + toff[t] + b * 32 + 16 * PIXB));
          const uint4 xf = __builtin_bit_cast(uint4, xb);
#pragma unroll
          for (int d = 0; d < DYB; ++d) Tr<T>::mma16(yf[d], xf, acc[t][d][b]);
        }
    }
  }
  // ---- combine the four waves in LDS, then one round of fp32 atomics: D[row = co (4g + j)][col = ci (lane & 15)]
  __syncthreads();
  float* sum = (float*)smem;                                // [NT][DYB][NCB][16 co][16 ci]
  constexpr int NSUM = NT * DYB * NCB * 256;
  for (int i = tid; i < NSUM; i += THIN_NTHR) sum[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int d = 0; d < DYB; ++d)
#pragma unroll
      for (int b = 0; b < NCB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&sum[(((t * DYB + d) * NCB + b) * 16 + 4 * g + j) * 16 + i16], acc[t][d][b][j]);
  __syncthreads();
  for (int i = tid; i < NSUM; i += THIN_NTHR) {
    const int ci = i & 15, col = (i >> 4) & 15, tb = i >> 8, b = tb % NCB, td = tb / NCB, d = td % DYB, t = td / DYB;
    const int co = d * 16 + col, cin = b * 16 + ci;
    if (STEM) { if (co < a.Cout && cin < 160) atomicAdd(a.dW + (size_t)co * 160 + cin, sum[i]); }
    else if (co < a.Cout && cin < a.wI) atomicAdd(a.dW + ((size_t)a.tw[t] * a.wO + co) * a.wI + cin, sum[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- host side
namespace {

static bool thin_on() {
  static const bool on = getenv("OCTSEG_NO_THIN") == nullptr;   // A/B switch
  return on;
}
static bool thin_ext_on() {
  static const bool on = getenv("OCTSEG_NO_THIN_EXT") == nullptr;   // A/B switch: 1x1 / ConvT-parity / stem kinds (round 3)
  return on && thin_on();
}
// tap offsets relative to a window whose origin sits `halo` pixels above / left of the tile
static int thin_kind(const int* tdy, const int* tdx, int ntaps, int istride, int ostride) {
  if (istride != 1) return -1;
  if (ntaps == 9 && ostride == 1) {
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
      const int oy = tdy[t] + 1, ox = tdx[t] + 1;
      if (oy < 0 || oy > 2 || ox < 0 || ox > 2) return -1;
      seen |= 1u << (oy * 3 + ox);
    }
    return seen == 0x1ffu ? K33 : -1;
  }
  if (!thin_ext_on()) return -1;
  if (ntaps == 1 && ostride == 1 && tdy[0] == 0 && tdx[0] == 0) return K11;
  if (ntaps == 4 && ostride == 2) {
    for (int t = 0; t < 4; ++t) if (tdy[t] < -1 || tdy[t] > 1 || tdx[t] < -1 || tdx[t] > 1) return -1;
    return KT22;
  }
  return -1;
}
struct ThinGeom { int TH, tiles_x, tiles_y, ntiles, G; };
static ThinGeom thin_geom(int N, int OH, int OW, int TH) {
  ThinGeom g;
  g.TH = TH;
  g.tiles_x = (OW + THIN_TW - 1) / THIN_TW; g.tiles_y = (OH + g.TH - 1) / g.TH;
  g.ntiles = N * g.tiles_x * g.tiles_y;
  g.G = g.ntiles < THIN_MAXWG ? g.ntiles : THIN_MAXWG;
  return g;
}
static int thin_th(int Cin) { return Cin == 16 ? 16 : 8; }

template <typename T, int KIND, int CIN, int NB, int TH>
static hipError_t thin_launch_k(const ThinArgs& ta, int G, hipStream_t st) {
  typedef ThinK<KIND, CIN> K;
  const size_t win = K::STEM ? 0 : (size_t)(TH + 2 * K::HALO) * K::RW * K::PIXB;
  const size_t lds = win + (size_t)4 * NB * 16 * 2 * sizeof(float) + (K::STEM ? (size_t)StemWin<TH>::BYTES + 64 + (size_t)6 * NB * 64 * 16 : 0);
  hipLaunchKernelGGL((thin_conv_kernel<T, KIND, CIN, NB, TH>), dim3(G), dim3(THIN_NTHR), lds, st, ta);
  return hipGetLastError();
}
template <typename T, int KIND>
static hipError_t thin_launch(const ThinArgs& ta, int Cin, int NB, int G, hipStream_t st) {
  if (Cin == 16) return NB == 1 ? thin_launch_k<T, KIND, 16, 1, 16>(ta, G, st) : thin_launch_k<T, KIND, 16, 2, 16>(ta, G, st);
  return NB == 1 ? thin_launch_k<T, KIND, 32, 1, 8>(ta, G, st) : thin_launch_k<T, KIND, 32, 2, 8>(ta, G, st);
}
static void thin_fill_taps(ThinArgs& ta, const int* tdy, const int* tdx, const int* tw, int ntaps, int halo) {
  ta.ntaps = ntaps;
  for (int t = 0; t < 9; ++t) {
    const int s = t < ntaps ? t : ntaps - 1;
    ta.tdy[t] = tdy[s] + halo; ta.tdx[t] = tdx[s] + halo; ta.tw[t] = tw[s];
  }
}

}  // namespace

bool thin_conv_eligible(const ConvArgs& a, int dtype) {
  if (!thin_on() || dtype == DT_F32 || a.Wmaster == nullptr) return false;
  if (a.nsrc != 1 || a.ndst != 1) return false;
  const int kind = thin_kind(a.tap_dy, a.tap_dx, a.ntaps, a.istride, a.ostride);
  if (kind < 0) return false;
  if (kind != KT22 && (a.ooy != 0 || a.oox != 0)) return false;
  if (a.Cin != 16 && a.Cin != 32) return false;
  const SrcDesc& s = a.src[0];
  const DstDesc& d = a.dst[0];
  if (s.C != a.Cin || s.c0 != 0 || a.IH != a.OH || a.IW != a.OW || (s.H << s.up) != a.IH || (s.W << s.up) != a.IW) return false;
  if (a.out_mode == OUT_HEAD_NCHW) {
    if (a.Cout > 16 || d.pool || kind == KT22) return false;
  } else {
    if ((a.Cout != 16 && a.Cout != 32) || d.c0 != 0 || d.cn != a.Cout || d.C != a.Cout) return false;
    if (d.pool) { if (kind != K33 || d.H * 2 != a.OH || d.W * 2 != a.OW) return false; }
    else if (d.H != a.OH * a.ostride || d.W != a.OW * a.ostride) return false;
  }
  // full-resolution maps only: small grids belong to the MFMA-bound kernels' occupancy rules
  return (long long)a.N * a.OH * a.OW >= 4096;
}

int thin_conv_rows(const ConvArgs& a) { return thin_geom(a.N, a.OH, a.OW, thin_th(a.Cin)).G; }

hipError_t launch_thin_conv(int dtype, const ConvArgs& a, hipStream_t st) {
  const ThinGeom g = thin_geom(a.N, a.OH, a.OW, thin_th(a.Cin));
  const int kind = thin_kind(a.tap_dy, a.tap_dx, a.ntaps, a.istride, a.ostride);
  ThinArgs ta;
  memset(&ta, 0, sizeof(ta));
  const SrcDesc& s = a.src[0];
  const DstDesc& d = a.dst[0];
  ta.x = (const char*)s.ptr; ta.scale = s.scale; ta.shift = s.shift; ta.relu = s.relu; ta.up = s.up; ta.sH = s.H; ta.sW = s.W;
  ta.N = a.N; ta.OH = a.OH; ta.OW = a.OW;
  thin_fill_taps(ta, a.tap_dy, a.tap_dx, a.tap_w, a.ntaps, kind == K11 ? 0 : 1);
  ta.w = a.Wmaster; ta.wO = a.wO; ta.wI = a.wI; ta.wtrans = a.wtrans; ta.wscale = a.wscale;
  ta.Cout = a.Cout; ta.bias = a.bias; ta.relu_out = a.relu_out;
  ta.slab = a.stat_slab; ta.slab_row0 = a.slab_row0;
  ta.y = (char*)d.ptr; ta.yC = d.C; ta.yH = d.H; ta.yW = d.W; ta.accum = (d.accum || a.out_mode == OUT_ACCUM) ? 1 : 0; ta.pool = d.pool;
  ta.head = a.out_mode == OUT_HEAD_NCHW ? 1 : 0;
  ta.os = a.ostride; ta.ooy = a.ooy; ta.oox = a.oox;
  ta.tiles_x = g.tiles_x; ta.tiles_y = g.tiles_y; ta.ntiles = g.ntiles;
  const int NB = a.Cout > 16 ? 2 : 1;
#define THIN_BY_KIND(T)                                                                  \
  switch (kind) {                                                                          \
    case K33: return thin_launch<T, K33>(ta, a.Cin, NB, g.G, st);                          \
    case K11: return thin_launch<T, K11>(ta, a.Cin, NB, g.G, st);                          \
    default: return thin_launch<T, KT22>(ta, a.Cin, NB, g.G, st);                          \
  }
  if (dtype == DT_F16) { THIN_BY_KIND(f16_t) }
  THIN_BY_KIND(bf16_t)
#undef THIN_BY_KIND
}

bool thin_wgrad_eligible(const WgradArgs& a, int dtype) {
  if (!thin_on() || dtype != DT_BF16 || deterministic_mode()) return false;
  if (a.dstride != 1 || a.doy != 0 || a.dox != 0 || a.nsrc != 1) return false;
  const int kind = thin_kind(a.tap_dy, a.tap_dx, a.ntaps, a.istride, 1);
  if (kind != K33 && kind != K11) return false;
  if (a.Cin != 16 && a.Cin != 32) return false;
  const SrcDesc& s = a.src[0];
  if (s.C != a.Cin || s.c0 != 0 || a.IH != a.OH || a.IW != a.OW || (s.H << s.up) != a.IH || (s.W << s.up) != a.IW) return false;
  if (a.DH != a.OH || a.DW != a.OW) return false;
  if (kind == K33) { if (a.dyC != 16 || a.Cout > 16) return false; }
  else if (!((a.dyC == 16 && a.Cout <= 16) || (a.dyC == 32 && a.Cout <= 32))) return false;
  return (long long)a.N * a.OH * a.OW >= 4096;
}

template <int KIND, int CIN, int DYB, int TH>
static hipError_t thin_wgrad_launch_k(const ThinArgs& ta, int G, hipStream_t st) {
  typedef ThinK<KIND, CIN> K;
  const size_t win = K::STEM ? (size_t)TH * THIN_TW * K::PIXB : (size_t)(TH + 2 * K::HALO) * K::RW * K::PIXB;
  const size_t stage = win + (size_t)TH * THIN_TW * ThinDy<DYB>::PITCH + (K::STEM ? (size_t)StemWin<TH>::BYTES : 0);
  const size_t sum = (size_t)K::NTAPS * DYB * K::NCB * 256 * sizeof(float);
  const size_t lds = stage > sum ? stage : sum;
  if (lds > 64 * 1024) {
    static bool set = false;
    if (!set) {
      hipError_t e = hipFuncSetAttribute((const void*)thin_wgrad_kernel<bf16_t, KIND, CIN, DYB, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      set = true;
    }
  }
  hipLaunchKernelGGL((thin_wgrad_kernel<bf16_t, KIND, CIN, DYB, TH>), dim3(G), dim3(THIN_NTHR), lds, st, ta);
  return hipGetLastError();
}

hipError_t launch_thin_wgrad(int dtype, const WgradArgs& a, hipStream_t st) {
  (void)dtype;
  ThinGeom g = thin_geom(a.N, a.OH, a.OW, thin_th(a.Cin));
  if (g.G > 512) g.G = 512;            // two workgroups per CU: fewer flush rounds, same streaming
  const int kind = thin_kind(a.tap_dy, a.tap_dx, a.ntaps, a.istride, 1);
  ThinArgs ta;
  memset(&ta, 0, sizeof(ta));
  const SrcDesc& s = a.src[0];
  ta.x = (const char*)s.ptr; ta.scale = s.scale; ta.shift = s.shift; ta.relu = s.relu; ta.up = s.up; ta.sH = s.H; ta.sW = s.W;
  ta.N = a.N; ta.OH = a.OH; ta.OW = a.OW;
  thin_fill_taps(ta, a.tap_dy, a.tap_dx, a.tap_w, a.ntaps, kind == K11 ? 0 : 1);
  ta.wO = a.Cout; ta.wI = a.Cin; ta.Cout = a.Cout;
  ta.dy = (const char*)a.dy; ta.dyC = a.dyC; ta.dW = a.dW;
  ta.os = 1;
  ta.tiles_x = g.tiles_x; ta.tiles_y = g.tiles_y; ta.ntiles = g.ntiles;
  if (kind == K33) return a.Cin == 16 ? thin_wgrad_launch_k<K33, 16, 1, 16>(ta, g.G, st) : thin_wgrad_launch_k<K33, 32, 1, 8>(ta, g.G, st);
  if (a.dyC == 16) return a.Cin == 16 ? thin_wgrad_launch_k<K11, 16, 1, 16>(ta, g.G, st) : thin_wgrad_launch_k<K11, 32, 1, 8>(ta, g.G, st);
  return a.Cin == 16 ? thin_wgrad_launch_k<K11, 16, 2, 16>(ta, g.G, st) : thin_wgrad_launch_k<K11, 32, 2, 8>(ta, g.G, st);
}

// ---- the ResNet stem (conv 7x7 stride 2 pad 3, 3 -> 64) on the NCHW f32 frame
bool thin_stem_eligible(int dtype) { return thin_ext_on() && dtype != DT_F32; }
static ThinGeom stem_geom(int N, int H, int W) {   // ~225 registers per lane: two workgroups per CU, one resident round
  static const int gcap = getenv("OCTSEG_STEM_G") ? atoi(getenv("OCTSEG_STEM_G")) : 512;   // experiments
  ThinGeom g = thin_geom(N, H / 2, W / 2, 4);
  if (g.G > gcap) g.G = gcap;
  return g;
}
int thin_stem_rows(int N, int H, int W) { return stem_geom(N, H, W).G; }

static void thin_stem_args(ThinArgs& ta, const StemArgs& s, const ThinGeom& g) {
  memset(&ta, 0, sizeof(ta));
  ta.img = s.img; ta.IH = s.H; ta.IW = s.W; ta.normalize = s.normalize;
  for (int i = 0; i < 3; ++i) { ta.mean[i] = s.normalize ? s.mean[i] : 0.f; ta.inv[i] = s.normalize ? 1.0f / s.stdv[i] : 1.f; }
  ta.N = s.N; ta.OH = s.H / 2; ta.OW = s.W / 2;
  ta.ntaps = 1;
  ta.w = s.w; ta.wO = 64; ta.wI = 160; ta.Cout = 64; ta.wscale = s.wscale;
  ta.bias = s.bias; ta.relu_out = s.relu_out; ta.slab = s.slab; ta.slab_row0 = s.slab_row0;
  ta.y = (char*)s.y; ta.yC = 64; ta.yH = s.H / 2; ta.yW = s.W / 2;
  ta.os = 1;
  ta.dy = (const char*)s.dy; ta.dyC = 64; ta.dW = s.dW;
  ta.tiles_x = g.tiles_x; ta.tiles_y = g.tiles_y; ta.ntiles = g.ntiles;
}
hipError_t launch_thin_stem_forward(int dtype, const StemArgs& s, hipStream_t st) {
  const ThinGeom g = stem_geom(s.N, s.H, s.W);
  ThinArgs ta;
  thin_stem_args(ta, s, g);
  if (dtype == DT_F16) return thin_launch_k<f16_t, KSTEM, 16, 4, 4>(ta, g.G, st);
  return thin_launch_k<bf16_t, KSTEM, 16, 4, 4>(ta, g.G, st);
}
hipError_t launch_thin_stem_wgrad(int dtype, const StemArgs& s, hipStream_t st) {
  if (dtype != DT_BF16) return hipErrorInvalidValue;
  const ThinGeom g = stem_geom(s.N, s.H, s.W);
  ThinArgs ta;
  thin_stem_args(ta, s, g);
  return thin_wgrad_launch_k<KSTEM, 16, 4, 4>(ta, g.G, st);
}

}  // namespace octseg
