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
// Tiling: a workgroup of WM x WN waves owns a (4*WM) x 16 tile of output pixels x BN output
// channels; every wave computes 64 pixels x NT*32 channels from 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16, or 4x v_mfma_f32_32x32x2_f32 = exact f32 fmaf chain for the parity
// path).  Per channel chunk (RB bytes of K) the (tile + halo) input window is staged ONCE into LDS
// and every tap reads its A fragments from it at a shifted address (no im2col buffer anywhere);
// weights are pre-packed (pack_weight_image) into the exact, XOR-swizzled LDS image of every
// (tap, chunk, N-tile) slab and stream through a 2-slab LDS ring by LDS-DMA (global_load_lds, no
// VGPRs / ds_write / address math).  Software pipeline: while the MFMAs of (chunk c, tap t) run, the
// DMA of the next slab and the register-staged load of a slice of chunk c+1's
// window are in flight; one barrier per tap.  The epilogue adds bias, emits per-channel (sum, sumsq) partials for the
// following BatchNorm (deterministic slab, reduced by bn_finalize), transposes the accumulators
// through LDS and stores / accumulates whole 16-byte channel vectors.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <type_traits>

namespace octseg {

#ifdef OCTSEG_STAMP
// diagnostic build: s_memtime stamps around the phases of the tap loop (never in the shipped library)
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

template <int RB> struct ConvCfg { static constexpr int PITCH = RB + 16, KSTEPS = RB / 32, VPR = RB / 16; };

template <typename T, int NT, int WN, int WM, int RB>
__global__ __launch_bounds__(64 * WM * WN) void conv_mfma_kernel(const ConvArgs a, const int mode) {
  const int dbuf = mode & 1;
  const bool resident = (mode & 2) != 0;   // single K chunk + small slabs: every tap's weights stay in LDS, no per-tap DMA / barrier
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int BN = NT * 32 * WN;
  constexpr int TH = 4 * WM;
  constexpr int BM = TH * TW;
  constexpr int VEC = Tr<T>::VEC;
  constexpr int KC = RB / (int)sizeof(T);
  constexpr int PITCH = ConvCfg<RB>::PITCH, KSTEPS = ConvCfg<RB>::KSTEPS, VPR = ConvCfg<RB>::VPR;
  constexpr int NWAVES = WM * WN;
  constexpr int BBYTES = BN * RB;                             // one weight slab = its packed image
  constexpr int NDMA = BBYTES / 1024;                         // 1 KiB LDS-DMA pieces per slab
  constexpr int DPW = (NDMA + NWAVES - 1) / NWAVES;           // pieces issued per wave
  constexpr int SWZ_DIV = 256 / RB;                           // rows per 256-byte LDS bank line
  constexpr int MAXP = 4;                                     // window passes prefetched per tap
  typedef WindowStager<T, RB, NTHREADS> Stager;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + TH - 1) / TH;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with a
  // private 4 MiB L2.  Every workgroup streams the whole weight set of its N tile, so the N tile is chosen
  // by XCD: an XCD then re-reads ONE N tile's weights (<= 4 MiB for every layer but the 3072-channel one)
  // from its own L2 instead of every N tile's from MALL/HBM.  Speed only -- any placement is correct.
  const int n_mt = gridDim.x, n_nt = gridDim.y;
  const int lid = blockIdx.x + blockIdx.y * n_mt;
  int mt_idx, nt_idx;
  {
    const int total = n_mt * n_nt;
    const int xcd = lid & 7, seq = lid >> 3;                 // position inside this XCD's stream
    const int per_xcd = (total + 7) >> 3;
    // XCD x owns the global work range [x * per_xcd, (x+1) * per_xcd) of the N-major order (nt outer, mt inner)
    int w = xcd * per_xcd + seq;
    if (w >= total) w = lid;                                  // ragged tail: fall back to the plain order
    const bool exact = (total & 7) == 0;
    if (!exact) w = lid;                                      // keep the map a bijection when 8 does not divide the grid
    nt_idx = w / n_mt;
    mt_idx = w - nt_idx * n_mt;
  }
  const int n = mt_idx / (tiles_x * tiles_y);
  mt_idx -= n * tiles_x * tiles_y;
  const int tyi = mt_idx / tiles_x, txi = mt_idx - tyi * tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int co0 = nt_idx * BN;
  const int w_mt = n * tiles_x * tiles_y + mt_idx;   // M-tile index (BN-stat slab row)

  // window geometry
  const bool single = a.ntaps == 1;
  const int lstride = single ? 1 : a.istride;   // LDS lookup stride
  const int smul = single ? a.istride : 1;      // staging coordinate multiplier
  const int RH = single ? TH : (TH - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const int npass = (npix + Stager::PSTEP - 1) / Stager::PSTEP;
  const float inv_rw = 1.0f / (float)RW;
  const int gy0 = y0 * a.istride + a.min_dy, gx0 = x0 * a.istride + a.min_dx;

  const int abytes = npass * Stager::PSTEP * PITCH;   // rows padded to whole passes (unconditional stores)
  char* ldsA = smem;                                  // [1 or 2] windows
  char* ldsB = smem + (dbuf ? 2 : 1) * abytes;        // [2] weight slab ring

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
  int bswz[NT];  // XOR swizzle of the 16-byte chunk index inside a slab row (matches pack_weight_image)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wn * NT * 32 + nt * 32 + r;
    bbase[nt] = row * RB;
    bswz[nt] = (row / SWZ_DIV) & (VPR - 1);
  }

  const int nchunks = (a.Cin + KC - 1) / KC;
  const char* Wp = (const char*)a.W;

  const int ntiles_n = gridDim.y;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ldsB_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsB;  // LDS byte address
  // LDS-DMA of a packed slab into ring slot `slot`.  Issued through inline asm: hipcc cannot prove that
  // a builtin LDS-DMA write does not alias the ds_reads that follow (runtime ring offsets) and would drain
  // vmcnt(0) right behind it; the asm form is invisible to its waitcnt pass and is retired by the explicit
  // s_waitcnt in front of the barrier instead.  Per-lane source = precomputed base + scalar slab offset.
  const char* dma_src0 = Wp + (size_t)nt_idx * BBYTES + (size_t)wave_u * DPW * 1024 + lane * 16;
  const size_t slab_stride = (size_t)ntiles_n * BBYTES;  // between consecutive (tap, chunk) slabs
  const bool dma_wave = NDMA % NWAVES == 0 || wave_u * DPW < NDMA;
  auto dmaB = [&](int slab_idx, int slot) {   // slab_idx = tapw * nchunks + chunk
    if (dma_wave) {
      const char* gsrc0 = dma_src0 + (size_t)slab_idx * slab_stride;
#pragma unroll
      for (int j = 0; j < DPW; ++j) {
        const char* gsrc = gsrc0 + j * 1024;
        const unsigned dst = ldsB_addr + slot * BBYTES + (wave_u * DPW + j) * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      }
    }
  };
  // tap tables in VGPR lanes (lane i = tap i), fetched per iteration with v_readlane: a kernarg s_load in
  // the loop costs its full scalar-cache latency every iteration
  int v_toff = 0, v_tapw = 0;
  if (lane < a.ntaps) {
    v_toff = single ? 0 : ((a.tap_dy[lane] - a.min_dy) * RW + (a.tap_dx[lane] - a.min_dx)) * PITCH;
    v_tapw = a.tap_w[lane];
  }
  auto stage_full = [&](const Stager& sg, char* dst) {
    for (int p = 0; p < npass; p += MAXP) {
      uint4 v[MAXP];
      bool ok[MAXP];
#pragma unroll
      for (int u = 0; u < MAXP; ++u) v[u] = sg.load(min(p + u, npass - 1), n, gy0, gx0, smul, RW, npix, inv_rw, a.IH, a.IW, ok[u]);
#pragma unroll
      for (int u = 0; u < MAXP; ++u) sg.write(dst, min(p + u, npass - 1), v[u], ok[u]);
    }
  };
  // MFMAs of one tap.  The LDS fragment reads run two k-steps ahead of the MFMAs that consume them
  // (three register sets): LDS latency under load is several hundred cycles, a k-step of MFMAs is ~128.
  auto mma_tap = [&](const char* awin, const char* bsl, int toff) {
    uint4 af[3][2], bf[3][NT];
    auto frag_load = [&](int buf, int ks) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[buf][mt] = *(const uint4*)(awin + abase[mt] + toff + ks * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[buf][nt] = *(const uint4*)(bsl + bbase[nt] + (((ks * 2 + h) ^ bswz[nt]) * 16));
    };
    frag_load(0, 0);
    if (KSTEPS > 1) frag_load(1, 1);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 2 < KSTEPS) frag_load((ks + 2) % 3, ks + 2);
      // pin the stage order: left alone, hipcc sinks every read next to its MFMA (lgkmcnt(1) in front of
      // almost every MFMA pair) and the LDS latency is paid k-step by k-step
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af[ks % 3][mt], bf[ks % 3][nt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---------------- prologue: window of chunk 0 + first weight slab ----------------
  {
    Stager cur;
    cur.setup(a.src, a.nsrc, a.Cin, 0, tid);
    if (resident) {
      for (int t = 0; t < a.ntaps; ++t) dmaB(a.tap_w[t] * nchunks, t);
    } else {
      dmaB(a.tap_w[0] * nchunks, 0);
    }
    stage_full(cur, ldsA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- main loop ----------------
  // PPT = window passes of the NEXT chunk prefetched per tap (1 for multi-tap convs, 4 for 1x1);
  // every load and every LDS store of the pipeline is unconditional: indices are clamped instead
  // (re-staging the last pass / the last chunk again is harmless).
#ifdef OCTSEG_STAMP
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};
#endif
  const int IHl = a.IH, IWl = a.IW, ntaps = a.ntaps;
  // window-pixel cursor of the prefetch: pass p covers pixels p*PSTEP + p0; advancing by one pass is
  // (hy, hx) += (PSTEP / RW, PSTEP % RW) with one carry -- no division in the loop
  const int p0w = tid / VPR;
  const int hy_first = (int)(((float)p0w + 0.5f) * inv_rw), hx_first = p0w - hy_first * RW;
  const int dq = Stager::PSTEP / RW, dr = Stager::PSTEP - dq * RW;
  auto run = [&](auto ppt_c, auto dbuf_c) {
    constexpr int PPT = decltype(ppt_c)::value;
    constexpr bool DBUF = decltype(dbuf_c)::value;
    int it = 0;
    int tap2 = ntaps == 1 ? 0 : 1, chunk2 = ntaps == 1 ? min(1, nchunks - 1) : 0;  // slab of iteration 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool has_next = chunk + 1 < nchunks;
      Stager nxt;
      nxt.setup(a.src, a.nsrc, a.Cin, has_next ? chunk + 1 : chunk, tid);
      nxt.bind_image(n);
      const char* awin = ldsA + ((DBUF && (chunk & 1)) ? abytes : 0);
      char* anext = ldsA + ((chunk & 1) ? 0 : abytes);
      int hy = hy_first, hx = hx_first, hp = p0w;      // cursor of the pass to prefetch next
      char* wrow = anext + p0w * PITCH;
      for (int t = 0; t < ntaps; ++t, ++it) {
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5;
        STAMP(s0);
        // window slice of the next chunk (register load) and the LDS-DMA of the next iteration's slab:
        // both fly under the MFMAs below and are retired in front of the barrier.
        uint4 av[PPT];
        bool aok[PPT];
        char* wr[PPT];
        if constexpr (DBUF) {
#pragma unroll
          for (int u = 0; u < PPT; ++u) {
            av[u] = nxt.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, aok[u]);
            wr[u] = wrow;
            // advance to the next pass unless this was the last one (then it is simply staged again)
            const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
            if (adv) {
              hp += Stager::PSTEP; wrow += Stager::PSTEP * PITCH;
              hy += dq; hx += dr;
              if (hx >= RW) { hx -= RW; hy += 1; }
            }
          }
        }
        dmaB(__builtin_amdgcn_readlane(v_tapw, tap2) * nchunks + chunk2, (it + 1) & 1);
        const int toff = __builtin_amdgcn_readlane(v_toff, t);
        STAMP(s1);
        mma_tap(awin, ldsB + (it & 1) * BBYTES, toff);
        STAMP(s2);
        // keep the consumers of the prefetched registers (BN affine, LDS stores) behind the MFMA block
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBUF) {
#pragma unroll
          for (int u = 0; u < PPT; ++u) nxt.write_at(wr[u], av[u], aok[u]);
        }
        STAMP(s3);
        // the slab of iteration it+1 (and this wave's LDS stores) must have landed before anyone passes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(s4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(s5);
#ifdef OCTSEG_STAMP
        tsum[0] += s1 - s0; tsum[1] += s2 - s1; tsum[2] += s3 - s2; tsum[3] += s4 - s3; tsum[4] += s5 - s4; tsum[5] += 1;
#endif
        // (tap, chunk) cursor of the next iteration's slab (clamped at the very end)
        if (++tap2 == ntaps) { tap2 = 0; chunk2 = min(chunk2 + 1, nchunks - 1); }
      }
      if constexpr (!DBUF) {
        if (has_next) {  // window does not fit twice: restage in place (all waves passed the barrier)
          stage_full(nxt, ldsA);
          __syncthreads();
        }
      }
    }
  };
  if (resident) {
    for (int t = 0; t < ntaps; ++t) mma_tap(ldsA, ldsB + t * BBYTES, __builtin_amdgcn_readlane(v_toff, t));
    __syncthreads();   // the epilogue reuses the LDS
  } else if (dbuf) {
    if (single) run(std::integral_constant<int, 4>{}, std::true_type{});
    else run(std::integral_constant<int, 1>{}, std::true_type{});
  } else {
    run(std::integral_constant<int, 1>{}, std::false_type{});
  }

#ifdef OCTSEG_STAMP
  if (a.stamp != nullptr && lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(a.stamp + i, tsum[i]);
#endif

  // ---------------- epilogue ----------------
  // (all waves are past the last barrier: LDS is free)
  constexpr int OPITCH = BN * (int)sizeof(T) + 16;       // transposed-tile row pitch
  constexpr int OVPR = BN * (int)sizeof(T) / 16;         // 16-byte vectors per pixel row
  char* otile = smem;                                    // [BM][BN] T
  float* red = (float*)(smem + BM * OPITCH);             // [WM][BN][2] stat partials
  const bool head = a.out_mode == OUT_HEAD_NCHW;
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] = 0.f; s2[nt] = 0.f;
    const int cl = wn * NT * 32 + nt * 32 + r;
    const int co = co0 + cl;
    const bool cok = co < a.Cout;
    const float bias = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
        const int p = mt * 32 + rr;                       // pixel inside the wave's 64
        const int ty = wm * 4 + (p >> 4), tx = p & 15;
        const int gy = y0 + ty, gx = x0 + tx;
        const float val = acc[mt][nt][i] + bias;
        if (cok && gy < a.OH && gx < a.OW) {
          s1[nt] += val; s2[nt] += val * val;
          if (head) {
            const DstDesc& d = a.dst[0];
            const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
            ((float*)d.ptr)[(((size_t)n * a.Cout + co) * d.H + oy) * d.W + ox] = val;
          }
        }
        if (!head) {
          if (sizeof(T) == 4) *(float*)(otile + (ty * TW + tx) * OPITCH + cl * 4) = val;
          else { __bf16 b = (__bf16)val; *(unsigned short*)(otile + (ty * TW + tx) * OPITCH + cl * 2) = __builtin_bit_cast(unsigned short, b); }
        }
      }
    }
  }
  if (a.stat_slab != nullptr) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s1[nt] += __shfl_xor(s1[nt], 32);
      s2[nt] += __shfl_xor(s2[nt], 32);
      if (h == 0) {
        const int cl = wn * NT * 32 + nt * 32 + r;
        red[(wm * BN + cl) * 2 + 0] = s1[nt];
        red[(wm * BN + cl) * 2 + 1] = s2[nt];
      }
    }
  }
  __syncthreads();
  if (a.stat_slab != nullptr && tid < BN) {
    const int co = co0 + tid;
    if (co < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      float* slab = a.stat_slab + ((size_t)(a.slab_row0 + (w_mt)) * a.Cout + co) * 2;
      slab[0] = t1; slab[1] = t2;
    }
  }
  if (!head) {
    // cooperative store: every thread moves whole 16-byte channel vectors of one pixel
    for (int v = tid; v < BM * OVPR; v += NTHREADS) {
      const int p = v / OVPR, cvv = v % OVPR;
      const int ty = p >> 4, tx = p & 15;
      const int gy = y0 + ty, gx = x0 + tx;
      const int co = co0 + cvv * VEC;
      if (gy >= a.OH || gx >= a.OW || co >= a.Cout) continue;
      char* dptr = (char*)a.dst[0].ptr; int dC = a.dst[0].C, dc0 = a.dst[0].c0, dH = a.dst[0].H, dW = a.dst[0].W, dacc = a.dst[0].accum;
#pragma unroll
      for (int i = 1; i < MAX_SRC; ++i)
        if (i < a.ndst && co >= a.dst[i].c0) {
          dptr = (char*)a.dst[i].ptr; dC = a.dst[i].C; dc0 = a.dst[i].c0; dH = a.dst[i].H; dW = a.dst[i].W; dacc = a.dst[i].accum;
        }
      const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
      uint4* gp = (uint4*)(dptr + ((((size_t)n * dH + oy) * dW + ox) * dC + (co - dc0)) * sizeof(T));
      uint4 val = *(const uint4*)(otile + p * OPITCH + cvv * 16);
      if (a.out_mode == OUT_ACCUM || dacc) {
        const uint4 old = *gp;
        if (sizeof(T) == 4) {
          val.x = __float_as_uint(__uint_as_float(val.x) + __uint_as_float(old.x));
          val.y = __float_as_uint(__uint_as_float(val.y) + __uint_as_float(old.y));
          val.z = __float_as_uint(__uint_as_float(val.z) + __uint_as_float(old.z));
          val.w = __float_as_uint(__uint_as_float(val.w) + __uint_as_float(old.w));
        } else {
          unsigned nv[4] = {val.x, val.y, val.z, val.w};
          const unsigned ov[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = __uint_as_float(nv[i] << 16) + __uint_as_float(ov[i] << 16);
            const float hi = __uint_as_float(nv[i] & 0xffff0000u) + __uint_as_float(ov[i] & 0xffff0000u);
            nv[i] = pack_bf16(lo, hi);
          }
          val = make_uint4(nv[0], nv[1], nv[2], nv[3]);
        }
      }
      *gp = val;
    }
  }
}

// ---------------------------------------------------------------------------------------------
namespace {

struct Variant { int NT, WN, WM, RB; };

size_t variant_lds(const ConvArgs& a, const Variant& v, int esz, int dbuf, int* npass_out) {
  const int TH = 4 * v.WM, BN = v.NT * 32 * v.WN, BM = TH * TW, PITCH = v.RB + 16;
  (void)PITCH;
  const bool single = a.ntaps == 1;
  const int RH = single ? TH : (TH - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const int nthreads = 64 * v.WM * v.WN;
  const int pstep = nthreads / (v.RB / 16);
  if (npass_out) *npass_out = (npix + pstep - 1) / pstep;
  const int npass = (npix + pstep - 1) / pstep;
  const size_t abytes = (size_t)npass * pstep * PITCH;
  const size_t main_loop = (dbuf ? 2 : 1) * abytes + 2 * (size_t)BN * v.RB;
  const size_t epi = (size_t)BM * (BN * esz + 16) + (size_t)v.WM * BN * 2 * sizeof(float);
  return main_loop > epi ? main_loop : epi;
}

template <typename T, int NT, int WN, int WM, int RB>
hipError_t launch_variant(const ConvArgs& a, int dbuf, size_t lds, hipStream_t st) {
  constexpr int BN = NT * 32 * WN, TH = 4 * WM;
  const int mtiles = a.N * ((a.OH + TH - 1) / TH) * ((a.OW + TW - 1) / TW);
  dim3 grid(mtiles, (a.Cout + BN - 1) / BN);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<T, NT, WN, WM, RB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_mfma_kernel<T, NT, WN, WM, RB>), grid, dim3(64 * WM * WN), lds, st, a, dbuf);
  return hipGetLastError();
}

// Tile choice: N tile from Cout, K chunk from Cin, M tile (16x16 or 8x16 pixels) from tile utilisation
// and LDS fit (double-buffered window preferred).
struct Choice { Variant v; int dbuf; size_t lds; int resident; };

Choice choose(const ConvArgs& a, int esz) {
  Choice c;
  int NT, WN;
  if (a.Cout > 64) { NT = 2; WN = 2; } else if (a.Cout > 32) { NT = 1; WN = 2; } else { NT = 1; WN = 1; }
  const int kc128 = 128 / esz;
  const int RB = a.Cin <= kc128 / 2 ? 64 : 128;
  auto util = [&](int TH) {
    const double ty = (a.OH + TH - 1) / TH, tx = (a.OW + TW - 1) / TW;
    return (double)a.OH * a.OW / (ty * TH * tx * TW);
  };
  const size_t cap = 160 * 1024;
  const int wm_first = util(8) > 1.15 * util(16) ? 2 : 4;
  const int order[2] = {wm_first, wm_first == 4 ? 2 : 4};
  const int nchunks_c = (a.Cin + RB / esz - 1) / (RB / esz);
  // a single K chunk never restages its window: a second buffer would only cost occupancy
  for (int pref_dbuf = nchunks_c > 1 ? 1 : 0; pref_dbuf >= 0; --pref_dbuf)
    for (int k = 0; k < 2; ++k) {
      Variant v{NT, WN, order[k], RB};
      int npass = 0;
      const size_t lds = variant_lds(a, v, esz, pref_dbuf, &npass);
      // the pipeline prefetches 1 pass per tap (4 for 1x1): the next window must fit that budget
      const bool fits_pipe = a.ntaps == 1 ? npass <= 4 : npass <= a.ntaps;
      if (lds <= cap && (!pref_dbuf || fits_pipe)) {
        c.v = v; c.dbuf = pref_dbuf; c.lds = lds; c.resident = 0;
        // thin layers: one K chunk and slabs small enough to keep all taps in LDS -> no per-tap DMA wait / barrier
        const size_t slab = (size_t)v.NT * 32 * v.WN * v.RB;
        if (nchunks_c == 1 && a.ntaps > 1 && a.ntaps * slab <= 40 * 1024) {
          const size_t extra = (size_t)(a.ntaps - 2) * slab;
          if (lds + extra <= 64 * 1024) { c.resident = 1; c.lds = lds + extra; }
        }
        return c;
      }
    }
  c.v = Variant{NT, WN, 2, RB}; c.dbuf = 0; c.resident = 0; c.lds = variant_lds(a, c.v, esz, 0, nullptr);
  return c;
}

template <typename T>
hipError_t dispatch(const ConvArgs& a, hipStream_t st) {
  const Choice c = choose(a, (int)sizeof(T));
  if (c.lds > 160 * 1024) return hipErrorInvalidValue;
  const Variant& v = c.v;
#define OCTSEG_CASE(NT_, WN_, WM_, RB_)                                             \
  if (v.NT == NT_ && v.WN == WN_ && v.WM == WM_ && v.RB == RB_)                     \
    return launch_variant<T, NT_, WN_, WM_, RB_>(a, c.dbuf | (c.resident << 1), c.lds, st);
  OCTSEG_CASE(2, 2, 4, 128)
  OCTSEG_CASE(2, 2, 2, 128)
  OCTSEG_CASE(1, 2, 4, 128)
  OCTSEG_CASE(1, 2, 2, 128)
  OCTSEG_CASE(1, 1, 4, 128)
  OCTSEG_CASE(1, 1, 2, 128)
  OCTSEG_CASE(2, 2, 4, 64)
  OCTSEG_CASE(2, 2, 2, 64)
  OCTSEG_CASE(1, 2, 4, 64)
  OCTSEG_CASE(1, 2, 2, 64)
  OCTSEG_CASE(1, 1, 4, 64)
  OCTSEG_CASE(1, 1, 2, 64)
#undef OCTSEG_CASE
  return hipErrorInvalidValue;
}

}  // namespace

ConvPackInfo conv_pack_info(const ConvArgs& a, int dtype) {
  const Choice c = choose(a, (int)dtype_size(dtype));
  ConvPackInfo p;
  p.BN = c.v.NT * 32 * c.v.WN;
  p.RB = c.v.RB;
  const int KC = p.RB / (int)dtype_size(dtype);
  p.nchunks = (a.Cin + KC - 1) / KC;
  p.ntiles = (a.Cout + p.BN - 1) / p.BN;
  return p;
}

int conv_num_mtiles(const ConvArgs& a, int dtype) {
  const Choice c = choose(a, (int)dtype_size(dtype));
  const int TH = 4 * c.v.WM;
  return a.N * ((a.OH + TH - 1) / TH) * ((a.OW + TW - 1) / TW);
}

// A stride-1 1x1 convolution does not care about image geometry: present the N*H*W pixels as one image of
// 16-pixel rows so that every tile is full (a 22x22 map otherwise fills 47 % of its 16x16 tiles).
static bool flatten_1x1(ConvArgs& a) {
  if (a.ntaps != 1 || a.istride != 1 || a.ostride != 1 || a.out_mode == OUT_HEAD_NCHW) return false;
  if (a.tap_dy[0] != 0 || a.tap_dx[0] != 0) return false;
  const long long npix = (long long)a.N * a.OH * a.OW;
  if (npix % TW != 0 || a.IH != a.OH || a.IW != a.OW) return false;
  for (int i = 0; i < a.nsrc; ++i)
    if (a.src[i].up || a.src[i].H != a.IH || a.src[i].W != a.IW) return false;
  for (int i = 0; i < a.ndst; ++i)
    if (a.dst[i].H != a.OH || a.dst[i].W != a.OW) return false;
  const int rows = (int)(npix / TW);
  a.N = 1; a.IH = a.OH = rows; a.IW = a.OW = TW;
  for (int i = 0; i < a.nsrc; ++i) { a.src[i].H = rows; a.src[i].W = TW; }
  for (int i = 0; i < a.ndst; ++i) { a.dst[i].H = rows; a.dst[i].W = TW; }
  return true;
}

int conv_num_mtiles_flat(const ConvArgs& a0, int dtype) {
  ConvArgs a = a0;
  flatten_1x1(a);
  return conv_num_mtiles(a, dtype);
}

hipError_t launch_conv(int dtype, const ConvArgs& a0, hipStream_t st) {
  if (a0.ntaps <= 0) return hipSuccess;
  ConvArgs a = a0;
  flatten_1x1(a);
  if (dtype == DT_F32) return dispatch<float>(a, st);
  return dispatch<bf16_t>(a, st);
}

}  // namespace octseg
