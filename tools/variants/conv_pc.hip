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
__global__ __launch_bounds__(64 * (WM * WN + (WM * WN >= 4 ? WM * WN / 2 : 2))) void conv_mfma_kernel(const ConvArgs a, const int dbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NC = WM * WN;                 // consumer waves: ds_read + MFMA only
  constexpr int NP = NC >= 4 ? NC / 2 : 2;    // producer waves: address math, loads, LDS-DMA, BN affine, LDS stores
  constexpr int NG = NP / 2;                  // ... in two groups that alternate (issue for it+2 | finish it+1)
  constexpr int NTHREADS = 64 * (NC + NP);
  constexpr int NPT = 64 * NG;                // threads of one producer group
  constexpr int BN = NT * 32 * WN;
  constexpr int TH = 4 * WM;
  constexpr int BM = TH * TW;
  constexpr int VEC = Tr<T>::VEC;
  constexpr int KC = RB / (int)sizeof(T);
  constexpr int PITCH = ConvCfg<RB>::PITCH, KSTEPS = ConvCfg<RB>::KSTEPS, VPR = ConvCfg<RB>::VPR;
  constexpr int BBYTES = BN * RB;                             // one weight slab = its packed image
  constexpr int NDMA = BBYTES / 1024;                         // 1 KiB LDS-DMA pieces per slab
  constexpr int DPW = (NDMA + NC - 1) / NC;                   // pieces issued per consumer wave (LDS-DMA rate is per wave)
  constexpr int SWZ_DIV = 256 / RB;                           // rows per 256-byte LDS bank line
  constexpr int PMAX = 16;                                    // window vectors a producer thread keeps in flight
  typedef WindowStager<T, RB, NPT> Stager;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < NC;

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
  const int ntaps = a.ntaps;
  const bool single = ntaps == 1;
  const int lstride = single ? 1 : a.istride;   // LDS lookup stride
  const int smul = single ? a.istride : 1;      // staging coordinate multiplier
  const int RH = single ? TH : (TH - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const int npass = (npix + Stager::PSTEP - 1) / Stager::PSTEP;   // producer passes covering the window
  const float inv_rw = 1.0f / (float)RW;
  const int gy0 = y0 * a.istride + a.min_dy, gx0 = x0 * a.istride + a.min_dx;
  const int IHl = a.IH, IWl = a.IW;

  const int abytes = npass * Stager::PSTEP * PITCH;   // rows padded to whole passes
  char* ldsA = smem;                                  // [1 or 2] windows
  char* ldsB = smem + (dbuf ? 2 : 1) * abytes;        // [4] weight slab ring
  const int nchunks = (a.Cin + KC - 1) / KC;
  const int niter = nchunks * ntaps;

  // tap tables in VGPR lanes (lane i = tap i), fetched per iteration with v_readlane: a kernarg s_load in
  // the loop costs its full scalar-cache latency every iteration
  int v_toff = 0, v_tapw = 0;
  if (lane < ntaps) {
    v_toff = single ? 0 : ((a.tap_dy[lane] - a.min_dy) * RW + (a.tap_dx[lane] - a.min_dx)) * PITCH;
    v_tapw = a.tap_w[lane];
  }

#ifdef OCTSEG_STAMP
  unsigned long long ts[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  f32x16_t acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  const int wm = consumer ? wave / WN : 0, wn = consumer ? wave % WN : 0;
  const int r = lane & 31, h = lane >> 5;

  // LDS-DMA of a packed weight slab into ring slot `slot`, issued by the consumer waves (DPW pieces of 1 KiB
  // each; the DMA rate is per issuing wave, so the slab is spread over all of them) through inline asm: a
  // builtin LDS-DMA makes hipcc drain vmcnt(0) before later ds accesses it cannot prove disjoint.
  const char* Wp = (const char*)a.W;
  const unsigned ldsB_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsB;  // LDS byte address
  const char* dma_src0 = Wp + (size_t)nt_idx * BBYTES + (size_t)wave * DPW * 1024 + lane * 16;
  const size_t slab_stride = (size_t)gridDim.y * BBYTES;   // between consecutive (tap, chunk) slabs
  const bool dma_wave = consumer && (NDMA % NC == 0 || wave * DPW < NDMA);
  auto dmaB = [&](int slab_idx, int slot) {                // slab_idx = tapw * nchunks + chunk
    if (dma_wave) {
      const char* gsrc0 = dma_src0 + (size_t)slab_idx * slab_stride;
#pragma unroll
      for (int j = 0; j < DPW; ++j) {
        const char* gsrc = gsrc0 + j * 1024;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ldsB_addr + slot * BBYTES + (wave * DPW + j) * 1024);
        unsigned keep;
#ifndef OCTSEG_NODMA
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
#else
        (void)gsrc; (void)dst; keep = 0; (void)keep;
#endif
      }
    }
  };
  // slab index of iteration k (clamped: past the end the last slab is fetched again into a free slot, which
  // keeps the number of outstanding DMAs per wave constant for the counted s_waitcnt below)
  auto slab_of = [&](int k) {
    k = min(k, niter - 1);
    const int kc = k / ntaps, kt = k - kc * ntaps;
    return __builtin_amdgcn_readlane(v_tapw, kt) * nchunks + kc;
  };

  if (consumer) {
    // =========================== consumer waves: fragments + MFMA ===========================
    int abase[2], bbase[NT], bswz[NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int ty = wm * 4 + mt * 2 + (r >> 4), tx = r & 15;
      abase[mt] = ((ty * lstride) * RW + tx * lstride) * PITCH + h * 16;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = wn * NT * 32 + nt * 32 + r;
      bbase[nt] = row * RB;
      bswz[nt] = (row / SWZ_DIV) & (VPR - 1);   // XOR swizzle of the 16-byte chunk index (matches pack_weight_image)
    }
    dmaB(slab_of(0), 0);
    dmaB(slab_of(1), 1);
    dmaB(slab_of(2), 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory");   // slab 0 landed
    __builtin_amdgcn_s_barrier();               // ... and the first window staged by the producers
    asm volatile("" ::: "memory");
    int slot = 0, it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const char* awin = ldsA + ((dbuf && (chunk & 1)) ? abytes : 0);
      for (int t = 0; t < ntaps; ++t, ++it) {
        unsigned long long c0 = 0, c1 = 0, c2 = 0; (void)c0; (void)c1; (void)c2;
        STAMP(c0);
        dmaB(slab_of(it + 3), (slot + 3) & 3);   // three iterations ahead; its slot was last read in iteration it-1
        const int toff = __builtin_amdgcn_readlane(v_toff, t);
        const char* bsl = ldsB + slot * BBYTES;
        slot = (slot + 1) & 3;
        // fragment reads run two k-steps ahead of the MFMAs that consume them (three register sets)
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
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af[ks % 3][mt], bf[ks % 3][nt], acc[mt][nt]);
          __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(c1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPW) : "memory");   // slab of iteration it+1 landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(c2);
#ifdef OCTSEG_STAMP
        ts[0] += c1 - c0; ts[1] += c2 - c1; ts[2] += 1;
#endif
      }
      if (!dbuf && chunk + 1 < nchunks) {  // producers restage the single window between these two barriers
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail DMAs must not land in the epilogue's LDS tile
  } else {
    // =========================== producer waves: everything that moves data ===========================
    // Two groups take turns: in iteration j the group ((j+1) & 1) FINISHes iteration j+1 (s_waitcnt vmcnt(0)
    // on what it issued on its previous turn, BN affine, LDS stores) and then ISSUEs the loads of iteration
    // j+3 (LDS-DMA of the weight slab into ring slot (j+3) & 3, window-slice vectors into registers).  Every
    // load so gets two full iterations to land (memory latency under load is ~2500 cycles, an iteration
    // ~1100), and a wave only ever waits for ALL of its own outstanding loads -- no counted vmcnt.
    // producers are the youngest waves of the workgroup and would lose every issue-arbitration round to the
    // MFMA-heavy consumers (priority, then age): give them static priority
    __builtin_amdgcn_s_setprio(3);
    const int pw = wave - NC;
    const int grp = pw / NG;                       // producer group
    const int gtid = tid - (NC + grp * NG) * 64;    // thread inside the group
    const int ppt = (npass + ntaps - 1) / ntaps;        // window passes (of a group) per slice, <= PMAX (host check)
    Stager sg;                                          // stager of the chunk this group is currently filling
    int sg_chunk = -1;
    uint4 uv[PMAX];
    bool uok[PMAX];
    int u_first = 0, u_cnt = 0;                         // passes held in uv[]
    char* u_dst = ldsA;
    // iteration k needs: slab(k) and, before its chunk starts, the whole window of chunk(k).  Slice s of the
    // window of chunk c+1 is finished during iteration (c, s), i.e. it belongs to "k = c * ntaps + s + 1".
    auto issue = [&](int k) {          // loads for iteration k (k >= 1)
      const int pc = (k - 1) / ntaps, ps = (k - 1) - pc * ntaps;   // slice ps of the window of chunk pc + 1
      u_cnt = 0;
      if (dbuf && pc + 1 < nchunks && k <= niter) {
        unsigned long long q0 = 0, q1 = 0; (void)q0; (void)q1;
        STAMP(q0);
        if (sg_chunk != pc + 1) { sg.setup(a.src, a.nsrc, a.Cin, pc + 1, gtid); sg.bind_image(n); sg_chunk = pc + 1; }
        STAMP(q1);
#ifdef OCTSEG_STAMP
        ts[7] += q1 - q0;
#endif
        u_first = ps * ppt;
        u_cnt = min(ppt, npass - u_first);
        u_dst = ldsA + (((pc + 1) & 1) ? abytes : 0);
#pragma unroll
        for (int u = 0; u < PMAX; ++u)
          if (u < u_cnt) {
            const int hp = (u_first + u) * Stager::PSTEP + sg.p0;
            const int hy = (int)(((float)hp + 0.5f) * inv_rw), hx = hp - hy * RW;
#ifndef OCTSEG_NOALOAD
            uv[u] = sg.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, uok[u]);
#else
            uv[u] = make_uint4(hy, hx, 0, 0); uok[u] = true;
#endif
          }
      }
    };
    auto finish = [&]() {              // retire what this group issued one iteration ago
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < PMAX; ++u)
        if (u < u_cnt) sg.write_at(u_dst + ((u_first + u) * Stager::PSTEP + sg.p0) * PITCH, uv[u], uok[u]);
      u_cnt = 0;
    };
    // synchronous staging of a whole window by this group's threads (prologue / single-buffer restage)
    auto stage_window_now = [&](int chunk, char* dst, int p_begin, int p_step) {
      Stager cur;
      cur.setup(a.src, a.nsrc, a.Cin, chunk, gtid);
      cur.bind_image(n);
      for (int p = p_begin; p < npass; p += p_step) {
        const int hp = p * Stager::PSTEP + cur.p0;
        const int hy = (int)(((float)hp + 0.5f) * inv_rw), hx = hp - hy * RW;
        bool ok;
        const uint4 v = cur.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, ok);
        cur.write_at(dst + hp * PITCH, v, ok);
      }
    };
    // ---- prologue: window of chunk 0 (both groups, interleaved passes) + slab 0; group 1 then issues iteration 1
    stage_window_now(0, ldsA, grp, 2);
    if (grp == 1) issue(1); else issue(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int j = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      for (int t = 0; t < ntaps; ++t, ++j) {
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0; (void)p0; (void)p1; (void)p2; (void)p3;
        const bool act = ((j + 1) & 1) == grp;
        STAMP(p0);
        if (act) finish();
        STAMP(p1);
        if (act) issue(j + 3);
        STAMP(p2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(p3);
#ifdef OCTSEG_STAMP
        if (act) { ts[3] += p1 - p0; ts[4] += p2 - p1; ts[5] += p3 - p2; ts[6] += 1; }

#endif
      }
      if (!dbuf && chunk + 1 < nchunks) {   // the window does not fit twice: restage in place while the consumers wait
        stage_window_now(chunk + 1, ldsA, grp, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this wave may still be landing in LDS
  }
#ifdef OCTSEG_STAMP
  if (a.stamp != nullptr && lane == 0)
    for (int i = 0; i < 9; ++i) atomicAdd(a.stamp + i, ts[i]);
#endif
  __syncthreads();

  // ---------------- epilogue ----------------
  // (all waves are past the last barrier: LDS is free)
  constexpr int OPITCH = BN * (int)sizeof(T) + 16;       // transposed-tile row pitch
  constexpr int OVPR = BN * (int)sizeof(T) / 16;         // 16-byte vectors per pixel row
  char* otile = smem;                                    // [BM][BN] T
  float* red = (float*)(smem + BM * OPITCH);             // [WM][BN][2] stat partials
  const bool head = a.out_mode == OUT_HEAD_NCHW;
  if (consumer) {
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
      char* dptr = (char*)a.dst[0].ptr; int dC = a.dst[0].C, dc0 = a.dst[0].c0, dH = a.dst[0].H, dW = a.dst[0].W;
#pragma unroll
      for (int i = 1; i < MAX_SRC; ++i)
        if (i < a.ndst && co >= a.dst[i].c0) {
          dptr = (char*)a.dst[i].ptr; dC = a.dst[i].C; dc0 = a.dst[i].c0; dH = a.dst[i].H; dW = a.dst[i].W;
        }
      const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
      uint4* gp = (uint4*)(dptr + ((((size_t)n * dH + oy) * dW + ox) * dC + (co - dc0)) * sizeof(T));
      uint4 val = *(const uint4*)(otile + p * OPITCH + cvv * 16);
      if (a.out_mode == OUT_ACCUM) {
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
  const int nc = v.WM * v.WN, np = nc >= 4 ? nc / 2 : 2;
  const int nthreads = 64 * (np / 2);             // threads of one producer group
  const int pstep = nthreads / (v.RB / 16);
  if (npass_out) *npass_out = (npix + pstep - 1) / pstep;
  const int npass = (npix + pstep - 1) / pstep;
  const size_t abytes = (size_t)npass * pstep * PITCH;
  const size_t main_loop = (dbuf ? 2 : 1) * abytes + 4 * (size_t)BN * v.RB;
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
  hipLaunchKernelGGL((conv_mfma_kernel<T, NT, WN, WM, RB>), grid, dim3(64 * (WM * WN + (WM * WN >= 4 ? WM * WN / 2 : 2))), lds, st, a, dbuf);
  return hipGetLastError();
}

// Tile choice: N tile from Cout, K chunk from Cin, M tile (16x16 or 8x16 pixels) from tile utilisation
// and LDS fit (double-buffered window preferred).
struct Choice { Variant v; int dbuf; size_t lds; };

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
  for (int pref_dbuf = 1; pref_dbuf >= 0; --pref_dbuf)
    for (int k = 0; k < 2; ++k) {
      Variant v{NT, WN, order[k], RB};
      int npass = 0;
      const size_t lds = variant_lds(a, v, esz, pref_dbuf, &npass);
      // a window slice (passes per tap) must fit the producer's register set (PMAX = 16 vectors)
      const bool fits_regs = !pref_dbuf || (npass + a.ntaps - 1) / a.ntaps <= 16;
      if (lds <= cap && fits_regs) { c.v = v; c.dbuf = pref_dbuf; c.lds = lds; return c; }
    }
  c.v = Variant{NT, WN, 2, RB}; c.dbuf = 0; c.lds = variant_lds(a, c.v, esz, 0, nullptr);
  return c;
}

template <typename T>
hipError_t dispatch(const ConvArgs& a, hipStream_t st) {
  const Choice c = choose(a, (int)sizeof(T));
  if (c.lds > 160 * 1024) return hipErrorInvalidValue;
  const Variant& v = c.v;
#define OCTSEG_CASE(NT_, WN_, WM_, RB_)                                             \
  if (v.NT == NT_ && v.WN == WN_ && v.WM == WM_ && v.RB == RB_)                     \
    return launch_variant<T, NT_, WN_, WM_, RB_>(a, c.dbuf, c.lds, st);
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

hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t st) {
  if (a.ntaps <= 0) return hipSuccess;
  if (dtype == DT_F32) return dispatch<float>(a, st);
  return dispatch<bf16_t>(a, st);
}

}  // namespace octseg
