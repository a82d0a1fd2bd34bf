// NOT BUILT.  conv_pipe_kernel as it stood at the end of round 1 (removed from conv_mfma.hip in round 2):
// parity-green, measured slower than conv_mfma_kernel (numbers in the header below).  Kept as a negative result.
// It needs the helpers of conv_mfma.hip (map_tile, conv_epilogue<.., GROUPED = true>, RowMap<true>) to compile.
// =============================================================================================
// conv_pipe_kernel -- the same tile and MFMA schedule with a DEEP, DMA-only prefetch pipeline.
//
// The kernel above retires every prefetch in front of every tap's barrier (vmcnt(0)): a tap's MFMAs take
// ~1000 cycles, a load under full-chip traffic ~2500, so each tap waits ~1500 cycles for its own prefetch.
// Here nothing the loop fetches has a VGPR destination, so the queue can stay in flight across barriers:
//   * weight slabs: LDS-DMA into a ring of D (3..4) slots, issued D-1 taps ahead;
//   * the next chunk's window: LDS-DMA of the RAW source bytes straight into the other window buffer
//     (one 64-pixel pass per tap), and L = D-2 taps later an in-place LDS->LDS pass applies the lazy
//     BN/ReLU and zeroes the halo -- every thread touches only the 16 bytes its own lane fetched;
//   * the BN scale/shift of the chunk after next: two 4-byte-per-lane LDS-DMAs by wave 0.
// Every wave counts what it issued per tap and waits with `s_waitcnt vmcnt(issued in the last L taps)`.
// The window is pitch-128 with the 16-byte chunk index XOR-swizzled by (pixel >> 1) & 7 (applied on the
// DMA's SOURCE side: lane l fetches chunk (l & 7) ^ swz), read through RowMap<true>: conflict-free.
// Restrictions (host-checked): multi-tap, stride-1 lookups, RB = 128, npass + L <= ntaps.
//
// STATUS (round 1): parity-green, NOT the default.  Measured on 16 x 352^2, 512 -> 128, 3x3, bf16:
//   conv_mfma_kernel 881 TFLOP/s;  this kernel, fillers after the MFMA block: D=3 824, D=4 832;  fillers
//   spread between the MFMAs (as below): 746.  s_memtime stamps per wave-tap (D=4): MFMA stream 1895 cycles
//   against 1024 for the two waves' 32 MFMAs, barrier skew 425.  The prefetch waits are gone (vmcnt wait
//   275 -> 0), what remains is instruction issue: ~250 VALU/SALU instructions per wave-tap (window source
//   address with clamps, 64-bit slab addresses, the bf16 affine, SGPR-spill readlanes) against the ~90 that
//   fit beside 16 MFMAs.  Next: interior-tile fast path with precomputed per-pass offsets, packed-math
//   affine, kernel arguments by pointer to cut the SGPR spills.
template <typename T, int NT, int WN, int WM>
__global__ __launch_bounds__(64 * WM * WN) void conv_pipe_kernel(const ConvArgs a, const int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = 128;
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int BN = NT * 32 * WN;
  constexpr int TH = 4 * WM;
  constexpr int VEC = Tr<T>::VEC;
  constexpr int KC = RB / (int)sizeof(T);
  constexpr int PITCH = 128, KSTEPS = RB / 32, VPR = 8;
  constexpr int NWAVES = WM * WN;
  constexpr int BBYTES = BN * RB;
  constexpr int NDMA = BBYTES / 1024;
  constexpr int DPW = NDMA / NWAVES;
  static_assert(NDMA % NWAVES == 0, "slab pieces must divide over the waves");
  constexpr int SWZ_DIV = 256 / RB;
  constexpr int PSTEP = NTHREADS / VPR;                       // window pixels per pass
  constexpr int PASS_BYTES = PSTEP * PITCH;
  constexpr int SCSH_BYTES = 512;                             // scale[64] shift[64] floats of one chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int L = D - 2;

  unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0;
  (void)k0; (void)k1; (void)k2; (void)k3;
  STAMP(k0);
  const TilePos tp = map_tile<TH, BN>(a);
  const int n = tp.n, y0 = tp.y0, x0 = tp.x0, nt_idx = tp.nt_idx;

  const int RH = (TH - 1) + a.span_y, RW = (TW - 1) + a.span_x;
  const int npix = RH * RW;
  const int npass = (npix + PSTEP - 1) / PSTEP;
  const float inv_rw = 1.0f / (float)RW;
  const int gy0 = y0 + a.min_dy, gx0 = x0 + a.min_dx;
  const int nchunks = (a.Cin + KC - 1) / KC;
  const int ntaps = a.ntaps;
  const int total = nchunks * ntaps;

  const int npix_pad = (npix + 7) & ~7;                       // rows the DMA pieces (8 pixels per wave) cover
  const int abytes = npix_pad * PITCH;
  char* ldsA = smem;                                          // [1 or 2] windows
  char* ldsB = smem + (nchunks > 1 ? 2 : 1) * abytes;         // [D] weight slab ring
  char* ldsS = ldsB + D * BBYTES;                             // [2] scale/shift of a chunk
  char* ldsX = ldsS + 2 * SCSH_BYTES;                         // 16 bytes per thread: target of the no-op DMAs / transforms

  f32x16_t acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

  int pbase[2], bbase[NT], bswz[NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) pbase[mt] = (wm * 4 + mt * 2 + RowMap<true>::ty(r)) * RW + RowMap<true>::tx(r);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wn * NT * 32 + nt * 32 + r;
    bbase[nt] = row * RB;
    bswz[nt] = (row / SWZ_DIV) & (VPR - 1);
  }

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ldsA_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsA;
  const unsigned ldsB_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsB;
  const unsigned ldsS_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsS;
  const unsigned ldsX_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsX;
  const char* Wp = (const char*)a.W;
  const char* dma_src0 = Wp + (size_t)nt_idx * BBYTES + (size_t)wave_u * DPW * 1024 + lane * 16;
  const size_t slab_stride = (size_t)gridDim.y * BBYTES;
  auto glds16 = [&](const char* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  auto glds4 = [&](const char* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  auto dmaB = [&](int slab_idx, int slot) {
    const char* gsrc0 = dma_src0 + (size_t)slab_idx * slab_stride;
#pragma unroll
    for (int j = 0; j < DPW; ++j)
      glds16(gsrc0 + j * 1024, __builtin_amdgcn_readfirstlane(ldsB_addr + slot * BBYTES + (wave_u * DPW + j) * 1024));
  };
  // scale / shift of chunk `ch` -> ldsS[ch & 1]: lane l fetches channel ch*KC + l (lanes past the chunk or
  // sources without an affine fetch a valid dummy word; they are never read)
  auto dmaS = [&](int ch) {
    const int c = ch * KC + (lane < KC ? lane : 0);
    const bool cv = c < a.Cin;
    const SrcSel s = select_src(a.src, a.nsrc, cv ? c : 0);
    const bool aff = cv && s.scale != nullptr;
    const char* g1 = aff ? (const char*)(s.scale + s.cl) : Wp;
    const char* g2 = aff ? (const char*)(s.shift + s.cl) : Wp;
    const unsigned dst = __builtin_amdgcn_readfirstlane(ldsS_addr + (ch & 1) * SCSH_BYTES);
    glds4(g1, dst);
    glds4(g2, dst + 256);
  };

  // tap tables in VGPR lanes (lane i = tap i): offset of the tap inside the window, in pixels
  int v_toff = 0, v_tapw = 0;
  if (lane < ntaps) {
    v_toff = (a.tap_dy[lane] - a.min_dy) * RW + (a.tap_dx[lane] - a.min_dx);
    v_tapw = a.tap_w[lane];
  }

  // this thread's staging identity: pixel p0 of every pass, LDS chunk slot q, channel vector q ^ swz(p0)
  const int p0 = tid >> 3, q = tid & 7;
  const int cvec = q ^ ((p0 >> 1) & 7);
  struct Src {   // source view of one chunk for this thread's channel vector
    const char* img_base; int row_bytes, pix_bytes, up, relu; bool cvalid, has_aff;
  };
  auto bind = [&](int chunk) {
    Src b;
    const int c = chunk * KC + cvec * VEC;
    b.cvalid = c < a.Cin;
    const SrcSel s = select_src(a.src, a.nsrc, b.cvalid ? c : 0);
    b.has_aff = b.cvalid && s.scale != nullptr;
    b.up = s.up; b.relu = s.relu;
    b.img_base = s.ptr + ((size_t)n * s.H * s.W * s.C + s.cl) * sizeof(T);
    b.pix_bytes = s.C * (int)sizeof(T);
    b.row_bytes = s.W * b.pix_bytes;
    return b;
  };
  const int IHl = a.IH, IWl = a.IW;
  auto src_addr = [&](const Src& b, int hy, int hx, bool in_window, bool& ok) {
    const int iy = gy0 + hy, ix = gx0 + hx;
    ok = b.cvalid && in_window && (unsigned)iy < (unsigned)IHl && (unsigned)ix < (unsigned)IWl;
    const int iyc = min(max(iy, 0), IHl - 1), ixc = min(max(ix, 0), IWl - 1);
    return b.img_base + (unsigned)((iyc >> b.up) * b.row_bytes + (ixc >> b.up) * b.pix_bytes);
  };

  // ---------------- prologue ----------------
  for (int i = 0; i < D - 1 && i < total; ++i) {   // slabs of iterations 0 .. D-2
    const int ch = i / ntaps, t = i - ch * ntaps;
    dmaB(a.tap_w[t] * nchunks + ch, i);
  }
  if (nchunks > 1 && wave_u == 0) dmaS(1);
  {
    // window of chunk 0 through registers (same layout as the DMA path produces)
    const Src b = bind(0);
    float sc[VEC], sh[VEC];
    if (b.has_aff) {
      const int c = cvec * VEC;
      const SrcSel s = select_src(a.src, a.nsrc, c);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { sc[i] = s.scale[s.cl + i]; sh[i] = s.shift[s.cl + i]; }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
    }
    constexpr int U = 4;
    for (int pb = 0; pb < npass; pb += U) {
      uint4 v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int hp = min(pb + u, npass - 1) * PSTEP + p0;
        const int hy = (int)(((float)hp + 0.5f) * inv_rw), hx = hp - hy * RW;
        v[u] = *(const uint4*)src_addr(b, hy, hx, hp < npix, ok[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint4 w = v[u];
        if (b.has_aff) w = Tr<T>::affine(w, sc, sh, b.relu);
        if (!ok[u]) w = make_uint4(0, 0, 0, 0);
        const int ps = min(pb + u, npass - 1);
        if (ps * PSTEP + p0 < npix_pad) *(uint4*)(ldsA + ps * PASS_BYTES + tid * 16) = w;
      }
    }
  }
  // builtin form: hipcc's own waitcnt bookkeeping must see that the prologue's register loads (tap tables,
  // chunk 0) are retired, or it drains vmcnt(0) at their first use INSIDE the loop on every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  STAMP(k1);

  // ---------------- main loop ----------------
  const int hy_first = (int)(((float)p0 + 0.5f) * inv_rw), hx_first = p0 - hy_first * RW;
  const int dq = PSTEP / RW, dr = PSTEP - dq * RW;
  int it = 0;
  int tapS = (D - 1) % ntaps, chunkS = (D - 1) / ntaps;     // (tap, chunk) of the slab issued next (iteration it + D - 1)
  int slotS = D - 1, slotC = 0;                              // ring slots of the slab issued next / consumed next
#ifdef OCTSEG_STAMP
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};
#endif
  // Every tap issues exactly GT = DPW + 1 vector-memory operations per wave (a slab past the end is fetched
  // again into a dead slot, a window pass that does not exist goes to the scratch rows), so the counted wait
  // is a constant: everything older than the last L taps has landed.
  constexpr int GT = DPW + 1;
  auto wait_landed = [&]() {
    if (L == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (L == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * GT) : "memory");
  };
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const bool has_next = chunk + 1 < nchunks;
    const Src nb = bind(has_next ? chunk + 1 : chunk);
    const char* awin = ldsA + ((chunk & 1) ? abytes : 0);
    const unsigned anext_addr = ldsA_addr + ((chunk & 1) ? 0 : abytes);
    char* anext = ldsA + ((chunk & 1) ? 0 : abytes);
    int hy = hy_first, hx = hx_first, hp = p0;               // window pixel of the pass issued next
    unsigned okbits = 0;                                     // bit j: validity of the pass issued j taps ago
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
    const float relu_lo = (nb.has_aff && nb.relu) ? 0.f : -3.402823466e38f;
    for (int t = 0; t < ntaps; ++t, ++it) {
      unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
      (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5;
      STAMP(s0);
      // rare, outside the MFMA stream: scale/shift DMA of the chunk after next (wave 0; its two extra operations
      // only make the constant wait stricter), scale/shift registers of the next chunk at its first transform
      if (t == 0 && chunk + 2 < nchunks && wave_u == 0) dmaS(chunk + 2);
      if (t == L && has_next && nb.has_aff) {
        const float* sl = (const float*)(ldsS + ((chunk + 1) & 1) * SCSH_BYTES) + cvec * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { sc[i] = sl[i]; sh[i] = sl[64 + i]; }
      }
      const int toffp = __builtin_amdgcn_readlane(v_toff, t);
      const char* bsl = ldsB + slotC * BBYTES;
      if (++slotC == D) slotC = 0;
      // per-tap scalars of the fillers
      const bool w_real = has_next && t < npass && (t * PSTEP + wave_u * 8 < npix);   // this wave's window piece exists
      const int tp_ = t - L;                                                           // pass transformed in this tap
      const bool x_real = has_next && tp_ >= 0 && tp_ < npass;
      STAMP(s1);

      uint4 af[3][2], bf[3][NT];
      int arow[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int p = pbase[mt] + toffp;
        arow[mt] = p * PITCH + ((((p >> 1) & 7) ^ h) << 4);
      }
      auto frag_a = [&](int buf, int ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[buf][mt] = *(const uint4*)(awin + (arow[mt] ^ (ks * 32)));
      };
      auto frag_b = [&](int buf, int ks) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[buf][nt] = *(const uint4*)(bsl + bbase[nt] + (((ks * 2 + h) ^ bswz[nt]) * 16));
      };
      const char* wsrc = nullptr;
      uint4 xw = make_uint4(0, 0, 0, 0);
      uint4* xslot = nullptr;
      bool xok = true;
      // the 16 filler slots of a tap, spread between its MFMAs (one MFMA issues in 8 of its 32 cycles and the
      // partner wave of the SIMD owns every other MFMA slot: ~50 cycles of issue time per slot)
      auto filler = [&](auto gc) {
        constexpr int G = decltype(gc)::value;
        if constexpr (G == 0) frag_a(2, 2);
        if constexpr (G == 1) frag_b(2, 2);
        if constexpr (G == 2) {   // weight slab of iteration it + D - 1 -> the slot freed by the last barrier
          const char* gsrc0 = dma_src0 + (size_t)(__builtin_amdgcn_readlane(v_tapw, tapS) * nchunks + chunkS) * slab_stride;
          const unsigned dst0 = ldsB_addr + slotS * BBYTES + wave_u * DPW * 1024;
#pragma unroll
          for (int j = 0; j < (DPW + 1) / 2; ++j) glds16(gsrc0 + j * 1024, __builtin_amdgcn_readfirstlane(dst0 + j * 1024));
        }
        if constexpr (G == 3) {
          const char* gsrc0 = dma_src0 + (size_t)(__builtin_amdgcn_readlane(v_tapw, tapS) * nchunks + chunkS) * slab_stride;
          const unsigned dst0 = ldsB_addr + slotS * BBYTES + wave_u * DPW * 1024;
#pragma unroll
          for (int j = (DPW + 1) / 2; j < DPW; ++j) glds16(gsrc0 + j * 1024, __builtin_amdgcn_readfirstlane(dst0 + j * 1024));
          if (it + D < total) {   // advance the slab cursor (clamped at the end: the last slab is fetched again)
            if (++tapS == ntaps) { tapS = 0; ++chunkS; }
          }
          if (++slotS == D) slotS = 0;
        }
        if constexpr (G == 4) frag_a(0, 3);
        if constexpr (G == 5) frag_b(0, 3);
        if constexpr (G == 6) {   // source address of this thread's 16 bytes of window pass t
          bool ok;
          wsrc = src_addr(nb, hy, hx, hp < npix, ok);
          okbits = (okbits << 1) | (ok ? 1u : 0u);
        }
        if constexpr (G == 7) {
          const unsigned dst = w_real ? anext_addr + t * PASS_BYTES + wave_u * 1024 : ldsX_addr + wave_u * 1024;
          glds16(wsrc, __builtin_amdgcn_readfirstlane(dst));
        }
        if constexpr (G == 8) {
          if (t < npass) {
            hp += PSTEP; hy += dq; hx += dr;
            if (hx >= RW) { hx -= RW; hy += 1; }
          }
        }
        if constexpr (G == 9) {   // everything issued L taps ago has landed: slab it + 1, window pass t - L
          wait_landed();
          const bool mine = x_real && tp_ * PSTEP + p0 < npix_pad;
          xslot = (uint4*)(mine ? anext + tp_ * PASS_BYTES + tid * 16 : ldsX + tid * 16);
          xok = mine ? (((okbits >> L) & 1u) != 0) : true;
          xw = *xslot;
        }
        if constexpr (G == 11) {
          xw = Tr<T>::affine_lo(xw, sc, sh, relu_lo);
          if (!xok) xw = make_uint4(0, 0, 0, 0);
        }
        if constexpr (G == 13) *xslot = xw;
      };
      frag_a(0, 0); frag_b(0, 0);
      frag_a(1, 1); frag_b(1, 1);
      constexpr int NMMA = KSTEPS * 2 * NT;            // MFMAs per tap (x4 for f32)
      constexpr int SPM = 16 / NMMA;                   // filler slots behind each MFMA
      auto run_fillers = [&](auto mc) {
        constexpr int M = decltype(mc)::value;
        if constexpr (SPM >= 1) {
          filler(std::integral_constant<int, M * SPM>{});
          if constexpr (SPM >= 2) filler(std::integral_constant<int, M * SPM + 1>{});
        }
      };
      auto kstep = [&](auto kc) {
        constexpr int ks = decltype(kc)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            __builtin_amdgcn_sched_barrier(0);
            Tr<T>::mma(af[ks % 3][mt], bf[ks % 3][nt], acc[mt][nt]);
            __builtin_amdgcn_sched_barrier(0);
            if (mt == 0 && nt == 0) run_fillers(std::integral_constant<int, ks * 2 * NT + 0>{});
            if (mt == 0 && nt == 1) run_fillers(std::integral_constant<int, ks * 2 * NT + 1>{});
            if (mt == 1 && nt == 0) run_fillers(std::integral_constant<int, ks * 2 * NT + NT>{});
            if (mt == 1 && nt == 1) run_fillers(std::integral_constant<int, ks * 2 * NT + NT + 1>{});
          }
      };
      kstep(std::integral_constant<int, 0>{});
      kstep(std::integral_constant<int, 1>{});
      kstep(std::integral_constant<int, 2>{});
      kstep(std::integral_constant<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
      STAMP(s2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      STAMP(s4);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      STAMP(s5);
#ifdef OCTSEG_STAMP
      tsum[0] += s1 - s0; tsum[1] += s2 - s1; tsum[3] += s4 - s2; tsum[4] += s5 - s4; tsum[5] += 1;
#endif
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(k2);
  conv_epilogue<T, NT, WN, WM, true>(a, smem, acc, tp);
  STAMP(k3);
#ifdef OCTSEG_STAMP
  if (a.stamp != nullptr && lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(a.stamp + i, tsum[i]);
    atomicAdd(a.stamp + 8, k1 - k0); atomicAdd(a.stamp + 9, k2 - k1); atomicAdd(a.stamp + 10, k3 - k2); atomicAdd(a.stamp + 11, 1ull);
  }
#endif
}

