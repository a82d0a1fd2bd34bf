// conv_mfma.hip -- im2col-free NHWC convolution on gfx950 matrix cores.
//
// One kernel family serves every convolution of the hot path:
//   * forward 1x1 / 3x3 / 4x4 convs, stride 1 or 2 (tap table + istride),
//   * data gradients (same kernel, transposed weight pack, mirrored tap table),
//   * ConvTranspose2d 4x4 s2 (four parity launches, 2x2 taps each, ostride 2),
//   * the segmentation head (epilogue writes NCHW f32 logits + bias).
// The input is a *virtual* tensor: up to 5 concatenated sources, each optionally
// nearest-x2 upsampled and lazily batch-normalised (relu(x*scale+shift)) while it
// is staged into LDS -- torch.cat / F.interpolate / BN-apply / ReLU never touch HBM.
//
// Tiling: a workgroup (4 waves) owns an 8x16 tile of output pixels x BN output
// channels.  Per 128-byte channel chunk the (8+halo)x(16+halo) input window is
// staged once into LDS; every tap then reads its A fragments from the same window
// at a shifted LDS address (no im2col buffer anywhere).  Weights stream per tap
// through a double-buffered LDS slab.  Wave tile 64 px x BN/2 channels built from
// 32x32 MFMA tiles: v_mfma_f32_32x32x16_bf16 (bf16) or 4x v_mfma_f32_32x32x2_f32
// (exact f32 fmaf chain, used by the 1e-4 parity path).
// The epilogue adds bias, emits per-channel (sum, sumsq) partials for the following
// BatchNorm (deterministic slab, reduced by bn_finalize), and stores / accumulates.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

namespace octseg {

constexpr int TH = 8;                 // output-grid tile height
constexpr int ROWB = 128;             // channel-chunk bytes per LDS row
constexpr int PITCH = ROWB + 16;      // padded row pitch (bank spread for ds_read_b128)

template <typename T, int BN>
__global__ __launch_bounds__(NTHR) void conv_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = Tr<T>::VEC;
  constexpr int KC = ROWB / (int)sizeof(T);
  constexpr int NT = BN / 64;            // 32-wide n tiles per wave
  constexpr int BROWS_PER_THR = BN * 8 / NTHR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + TH - 1) / TH;
  int mt_idx = blockIdx.x;
  const int n = mt_idx / (tiles_x * tiles_y);
  mt_idx -= n * tiles_x * tiles_y;
  const int tyi = mt_idx / tiles_x, txi = mt_idx - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int co0 = blockIdx.y * BN;

  // window geometry
  const bool single = a.ntaps == 1;
  const int lstride = single ? 1 : a.istride;   // LDS lookup stride
  const int smul = single ? a.istride : 1;      // staging coordinate multiplier
  const int RH = single ? TH : (TH - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const float inv_rw = 1.0f / (float)RW;
  const int gy0 = y0 * a.istride + a.min_dy, gx0 = x0 * a.istride + a.min_dx;

  char* ldsA = smem;
  char* ldsB = smem + ((npix * PITCH + 15) & ~15);
  constexpr int BBYTES = BN * PITCH;

  f32x16_t acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  int abase[2], bbase[NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int ty = wm * 4 + mt * 2 + (r >> 4), tx = r & 15;
    abase[mt] = ((ty * lstride) * RW + tx * lstride) * PITCH + h * 16;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (wn * (BN / 2) + nt * 32 + r) * PITCH + h * 16;

  const int nchunks = (a.Cin + KC - 1) / KC;
  const char* Wp = (const char*)a.W;

  // weight slab loader: rows = output channels, 8 vectors of 16 B per row
  auto loadB = [&](int tapw, int chunk, uint4* regs) {
#pragma unroll
    for (int i = 0; i < BROWS_PER_THR; ++i) {
      const int v = tid + i * NTHR;
      const int row = v >> 3, cv = v & 7;
      const int co = co0 + row, c = chunk * KC + cv * VEC;
      regs[i] = make_uint4(0, 0, 0, 0);
      if (co < a.Cout && c < a.Cin)
        regs[i] = *(const uint4*)(Wp + (((size_t)tapw * a.Cout + co) * a.Cin + c) * sizeof(T));
    }
  };
  auto writeB = [&](char* dstb, const uint4* regs) {
#pragma unroll
    for (int i = 0; i < BROWS_PER_THR; ++i) {
      const int v = tid + i * NTHR;
      *(uint4*)(dstb + (v >> 3) * PITCH + (v & 7) * 16) = regs[i];
    }
  };

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();  // every wave is done with the previous window / weight slabs
    stage_window<T, ROWB>(ldsA, a.src, a.nsrc, a.Cin, chunk, n, gy0, gx0, smul, RW, npix, inv_rw, a.IH, a.IW, tid);
    {
      uint4 regs[BROWS_PER_THR];
      loadB(a.tap_w[0], chunk, regs);
      writeB(ldsB, regs);
    }
    __syncthreads();
    for (int t = 0; t < a.ntaps; ++t) {
      uint4 nregs[BROWS_PER_THR];
      const bool more = t + 1 < a.ntaps;
      if (more) loadB(a.tap_w[t + 1], chunk, nregs);  // global loads fly under the MFMAs below
      const int toff = single ? 0 : ((a.tap_dy[t] - a.min_dy) * RW + (a.tap_dx[t] - a.min_dx)) * PITCH;
      const char* bsl = ldsB + (t & 1) * BBYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint4 af[2], bf[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[mt] = *(const uint4*)(ldsA + abase[mt] + toff + ks * 32);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const uint4*)(bsl + bbase[nt] + ks * 32);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af[mt], bf[nt], acc[mt][nt]);
      }
      if (more) writeB(ldsB + ((t + 1) & 1) * BBYTES, nregs);
      __syncthreads();
    }
  }

  // ---------------- epilogue ----------------
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] = 0.f; s2[nt] = 0.f;
    const int co = co0 + wn * (BN / 2) + nt * 32 + r;
    const bool cok = co < a.Cout;
    // destination slice of this channel
    char* dptr = (char*)a.dst[0].ptr; int dC = a.dst[0].C, dc0 = a.dst[0].c0, dH = a.dst[0].H, dW = a.dst[0].W;
#pragma unroll
    for (int i = 1; i < MAX_SRC; ++i)
      if (i < a.ndst && co >= a.dst[i].c0) {
        dptr = (char*)a.dst[i].ptr; dC = a.dst[i].C; dc0 = a.dst[i].c0; dH = a.dst[i].H; dW = a.dst[i].W;
      }
    const float bias = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int p = mt * 32 + rr;
        const int gy = y0 + wm * 4 + (p >> 4), gx = x0 + (p & 15);
        if (cok && gy < a.OH && gx < a.OW) {
          const float val = acc[mt][nt][i] + bias;
          s1[nt] += val; s2[nt] += val * val;
          const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
          if (a.out_mode == OUT_HEAD_NCHW) {
            ((float*)dptr)[(((size_t)n * a.Cout + co) * dH + oy) * dW + ox] = val;
          } else {
            const size_t e = (((size_t)n * dH + oy) * dW + ox) * dC + (co - dc0);
            if (a.out_mode == OUT_ACCUM) Tr<T>::store(dptr, e, Tr<T>::load(dptr, e) + val);
            else Tr<T>::store(dptr, e, val);
          }
        }
      }
    }
  }
  if (a.stat_slab != nullptr) {
    __syncthreads();  // LDS is free again
    float* red = (float*)smem;  // [wm][BN][2]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s1[nt] += __shfl_xor(s1[nt], 32);
      s2[nt] += __shfl_xor(s2[nt], 32);
      if (h == 0) {
        const int cl = wn * (BN / 2) + nt * 32 + r;
        red[(wm * BN + cl) * 2 + 0] = s1[nt];
        red[(wm * BN + cl) * 2 + 1] = s2[nt];
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int co = co0 + tid;
      if (co < a.Cout) {
        float* slab = a.stat_slab + ((size_t)(a.slab_row0 + blockIdx.x) * a.Cout + co) * 2;
        slab[0] = red[tid * 2 + 0] + red[(BN + tid) * 2 + 0];
        slab[1] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
size_t conv_lds_bytes(const ConvArgs& a, int BN) {
  const bool single = a.ntaps == 1;
  const int RH = single ? TH : (TH - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  size_t abytes = ((size_t)RH * RW * PITCH + 15) & ~(size_t)15;
  size_t total = abytes + 2 * (size_t)BN * PITCH;
  size_t red = (size_t)2 * BN * 2 * sizeof(float);
  return total > red ? total : red;
}

int conv_num_mtiles(const ConvArgs& a) {
  return a.N * ((a.OH + TH - 1) / TH) * ((a.OW + TW - 1) / TW);
}

template <typename T, int BN>
static hipError_t launch_conv_t(const ConvArgs& a, hipStream_t st) {
  dim3 grid(conv_num_mtiles(a), (a.Cout + BN - 1) / BN);
  size_t lds = conv_lds_bytes(a, BN);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<T, BN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_mfma_kernel<T, BN>), grid, dim3(NTHR), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t st) {
  if (a.ntaps <= 0) return hipSuccess;
  const bool wide = a.Cout > 64;
  if (dtype == DT_F32) return wide ? launch_conv_t<float, 128>(a, st) : launch_conv_t<float, 64>(a, st);
  return wide ? launch_conv_t<bf16_t, 128>(a, st) : launch_conv_t<bf16_t, 64>(a, st);
}

}  // namespace octseg
