// Winograd F(2x2, 3x3) for the stride-1 3x3 pad-1 convolutions (forward and input gradient), fused into ONE kernel on the
// gfx950 fp32 matrix cores.  Included by conv.hip inside its anonymous namespace (uses Geo / Shape / dma16 / dma_pad).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d: 4x4 input tile (stride 2), g: 3x3 filter, Y: 2x2 outputs
//
// 16 multiplies per 2x2 outputs and channel pair instead of 36: the matrix cores see 2.25x fewer FLOPs than the direct form
// (fp32 MFMA is the binding unit of these layers: 157 TF against 8 TB/s).  The 16 "frequencies" f are 16 independent GEMMs
// [tiles x Cin] x [Cin x Cout]; everything between the global loads and the global stores stays on chip:
//
//   * the input patch of a 256-pixel tile (64 Winograd tiles) goes global -> LDS by LDS-DMA per 8-channel chunk, two buffers,
//     one barrier per chunk (the scheme of conv_dma_kernel);
//   * a lane owns (tile = lane % 16, channel = lane / 16) -- exactly the A-operand layout of v_mfma_f32_16x16x4_f32 -- reads ITS
//     4x4 patch from LDS and transforms it in registers: the 16 results are the A operands of the 16 frequency MFMAs
//     (no second trip through LDS for the transformed input);
//   * the filter chunk is transformed by the workgroup itself while it is staged (one (channel, output channel) pair per
//     thread: 9 loads, 28 flops, 16 floats into LDS, frequency-contiguous so that a lane fetches its B operands as four
//     16-byte reads): no derived filter tensors, no extra launch;
//   * accumulators: 16 frequencies x NB output-channel blocks x 4 registers; a lane ends up holding all 16 frequencies of
//     (tile 4q + r, channel j), so the output transform is register-only too: 24 adds per tile, then bias / residual and
//     16-byte stores of two x-adjacent tiles.
//
// dgrad is the same kernel reading the filter transposed and spatially flipped.  Numerics: the transforms add and halve in
// fp32; measured on the CPU oracle (tests/golden fixtures re-run with every 3x3 layer in this form) the step's losses move by
// 1e-6 .. 3e-5 relative -- the size of a change in summation order.
#pragma once

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The 4x4 patch of a lane as twelve ds_read_b64 (row r: bytes RS*r + {0, 8, 16} from a per-lane LDS byte address) and the wait
// for them, as ONE instruction block: the compiler turns a float2 LDS load whose halves are not both used into dword reads
// (ds_read2_b32: banks mod 32, 2-way conflicts here), and it does not track the completion of loads it did not issue -- so
// the wait sits inside the block, after which the outputs are ordinary registers.  (Also drains the caller's earlier LDS reads.)
template <int RS>
__device__ __forceinline__ void lds_read_patch(uint32_t addr, f32x2 (&p)[4][3]) {
  asm volatile(
      "ds_read_b64 %0, %12 offset:%13\n\tds_read_b64 %1, %12 offset:%13+8\n\tds_read_b64 %2, %12 offset:%13+16\n\t"
      "ds_read_b64 %3, %12 offset:%14\n\tds_read_b64 %4, %12 offset:%14+8\n\tds_read_b64 %5, %12 offset:%14+16\n\t"
      "ds_read_b64 %6, %12 offset:%15\n\tds_read_b64 %7, %12 offset:%15+8\n\tds_read_b64 %8, %12 offset:%15+16\n\t"
      "ds_read_b64 %9, %12 offset:%16\n\tds_read_b64 %10, %12 offset:%16+8\n\tds_read_b64 %11, %12 offset:%16+16\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(p[0][0]), "=&v"(p[0][1]), "=&v"(p[0][2]), "=&v"(p[1][0]), "=&v"(p[1][1]), "=&v"(p[1][2]),
        "=&v"(p[2][0]), "=&v"(p[2][1]), "=&v"(p[2][2]), "=&v"(p[3][0]), "=&v"(p[3][1]), "=&v"(p[3][2])
      : "v"(addr), "i"(0), "i"(RS), "i"(2 * RS), "i"(3 * RS)
      : "memory");
}

template <int N, class F, int... I>
__device__ __forceinline__ void wino_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void wino_static_for(F&& f) { wino_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <class G> struct WinoMap;
// wave w, tile i (0..15) of the wave's 16-tile strip -> image inside the tile group, top-left OUTPUT pixel (y0, x0) inside the
// pixel tile.  Tiles 4q .. 4q+3 (one lane's accumulator rows) are x-adjacent pairs (0,1) (2,3) in every geometry.
template <> struct WinoMap<GX> {
  __device__ __forceinline__ static void at(int w, int i, int& img, int& y0, int& x0) { img = 0; y0 = 2 * w; x0 = 2 * i; }
};
template <> struct WinoMap<G16> {
  __device__ __forceinline__ static void at(int w, int i, int& img, int& y0, int& x0) { img = 0; y0 = 2 * (2 * w + (i >> 3)); x0 = 2 * (i & 7); }
};
template <> struct WinoMap<G8> {
  __device__ __forceinline__ static void at(int w, int i, int& img, int& y0, int& x0) { img = w; y0 = 2 * (i >> 2); x0 = 2 * (i & 3); }
};
template <> struct WinoMap<G4> {
  __device__ __forceinline__ static void at(int w, int i, int& img, int& y0, int& x0) { img = 4 * w + (i >> 2); y0 = 2 * ((i >> 1) & 1); x0 = 2 * (i & 1); }
};

// K-split geometries (small planes): ONE 16-tile strip per workgroup, shared by the four waves, each of which takes one k-step
// (4 channels) of a 16-channel chunk; their partial outputs are summed through LDS at the end (conv_wino_dma_kernel<.., KS = true>).
template <> struct WinoMap<G8k> {
  __device__ __forceinline__ static void at(int, int i, int& img, int& y0, int& x0) { img = 0; y0 = 2 * (i >> 2); x0 = 2 * (i & 3); }
};
template <> struct WinoMap<G4k> {
  __device__ __forceinline__ static void at(int, int i, int& img, int& y0, int& x0) { img = i >> 2; y0 = 2 * ((i >> 1) & 1); x0 = 2 * (i & 1); }
};

constexpr int WINO_CK = 8;      // input channels per staged chunk (two k-steps of the 16x16x4 MFMA); 4 for a 4-channel input
constexpr int WINO_US = 20;     // floats per (channel, output channel) slot of the transformed filter: 16 + 4 of padding
                                // (a ds_read_b128 lane group -- 8 lanes of one k, 8 of the next -- starts on 16 distinct 16-byte slots)

// LDS image of the input patch, filled by LDS-DMA exactly like conv_dma_kernel's (DPatch: rows of whole 16-byte chunks, four
// columns beyond the tile on either side, everything outside the image an out-of-range lane that reads as zero), but with a
// per-channel stride == 32 (mod 64) dwords.  A lane reads its 4x4 patch as 8-byte pairs at even dword offsets (columns
// x0+2 .. x0+7 of the LDS row; it needs x0+3 .. x0+6): the 16 tiles of a k are 128 contiguous bytes, the other k of the
// 32-lane group sits on the other half of the 64 banks -- no conflicts, where dword reads (banks mod 32, every lane on an
// even column) ran 2-way.
template <class G> struct WinoPatch {
  static constexpr int PH = G::TH + 2, PWS = G::TW + 8, QR = PWS / 4;
  static constexpr int IMG = PH * PWS, RAW = G::NI * IMG;
  static constexpr int CIS = RAW + ((96 - RAW % 64) % 64);       // channel stride, == 32 (mod 64)
  static constexpr int CPC = CIS / 4;                            // 16-byte chunks per channel (pad chunks included)
  static_assert(CIS % 64 == 32 && RAW % 4 == 0, "channel stride");
};

template <class G, int NB, bool DGRAD, int CK = WINO_CK, bool KS = false>
__global__ void __launch_bounds__(CT_THREADS, KS ? (NB == 1 ? 2 : 1) : (G::NI > 1) ? 1 : (NB == 1) ? 3 : 2)
conv_wino_dma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                     const float* __restrict__ residual, float* __restrict__ y, Shape s, int flags) {
  using P = WinoPatch<G>;
  constexpr int NT = 16 * NB, US = WINO_US, CIS = P::CIS;
  static_assert(CK % 4 == 0, "whole k-steps");
  static_assert(!KS || (CK == 16 && G::NPIX == 64), "K-split: 16 tiles per workgroup, one k-step of a 16-channel chunk per wave");
  constexpr int FPT = (CK * NT + CT_THREADS - 1) / CT_THREADS;     // filter pairs per thread
  constexpr int PCH = CK * P::CPC, PCHP = dma_pad(PCH);           // 16-byte chunks of a patch buffer, padded to whole wave pieces
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, UBUF = CK * NT * US;
  __shared__ __attribute__((aligned(16))) float pl[2 * PBUF];     // two buffers each: chunk c+1 arrives while chunk c is read
  __shared__ __attribute__((aligned(16))) float ul[2 * UBUF];

  const int lane = threadIdx.x & 63, wave = wave_index();
  int bid = blockIdx.x;
  if (flags & 1) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);       // an XCD (block id mod 8) works on neighbouring tiles
  const TileCoord tc = decode_tile<G>(bid, s.H, s.W);
  const int co0 = blockIdx.y * NT;
  const uint32_t HWp = (uint32_t)(s.H * s.W);

  // ---- per-lane buffer offsets (bytes) of the patch chunks, fixed for the whole tile
  uint32_t poff[NVP];
#pragma unroll
  for (int i = 0; i < NVP; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int ci = e / P::CPC, rem = e % P::CPC;
    const int row = rem / P::QR, q = rem % P::QR;
    const int pimg = row / P::PH, r = row % P::PH;
    const int hh = tc.h0 + r - 1, ww = tc.w0 - 4 + 4 * q;
    const bool ok = (e < PCH) && (rem < P::RAW / 4) && (tc.b0 + pimg < s.B) && (hh >= 0) && (hh < s.H) && (ww >= 0) && (ww < s.W);
    poff[i] = ok ? (__umul24(__umul24(pimg, s.Cin) + ci, HWp) + __umul24(hh, s.W) + ww) << 2 : DMA_OOB;
  }
  const char* xb = reinterpret_cast<const char*>(x) + ((int64_t)tc.b0 * s.Cin * HWp) * 4;
  int64_t xbytes = (int64_t)(s.B - tc.b0) * s.Cin * HWp * 4;
  const int64_t xstep = (int64_t)CK * HWp * 4;
  auto issue_patch = [&](float* buf) {          // one chunk of CK channels, global -> LDS; then the descriptor moves on
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NVP; ++i)
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP)      // (scalar)
        dma16(rx, buf + (i * CT_THREADS + wave * 64) * 4, poff[i]);
    xb += xstep; xbytes -= xstep;
  };

  const int mi = lane & 15, kk = lane >> 4;
  int img, y0, x0;
  WinoMap<G>::at(wave, mi, img, y0, x0);
  const int kbase = KS ? 4 * wave : 0;                                    // K-split: this wave's k-step of every chunk
  const int a_off = (img * P::PH + y0) * P::PWS + x0 + 2 + (kbase + kk) * CIS;      // (even) LDS offset of the first pair of this lane's patch
  const int b_off = ((kbase + kk) * NT + mi) * US;                                  // its filter slot

  // filter pairs of this thread, output channel fastest: consecutive lanes store 80 bytes apart (conflict-free ds_write_b128)
  float g9[FPT][9];
  auto load_filter = [&](int c0) {
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const int e = t * CT_THREADS + threadIdx.x;
      const int fn = e % NT, fk = e / NT;
      const bool ok = (e < CK * NT) && (co0 + fn < s.Cout) && (c0 + fk < s.Cin);
      const float* src = DGRAD ? w + ((int64_t)(c0 + fk) * s.Cout + (co0 + fn)) * 9 : w + ((int64_t)(co0 + fn) * s.Cin + (c0 + fk)) * 9;
#pragma unroll
      for (int i = 0; i < 9; ++i) g9[t][i] = ok ? src[DGRAD ? 8 - i : i] : 0.f;
    }
  };
  auto store_filter = [&](int buf) {        // U = G g G^T, 16 floats, frequency-contiguous
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const int e = t * CT_THREADS + threadIdx.x;
      if (e < CK * NT) {
        const float* g = g9[t];
        float u[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float s02 = g[c] + g[6 + c];
          u[0][c] = g[c];
          u[1][c] = 0.5f * (s02 + g[3 + c]);
          u[2][c] = 0.5f * (s02 - g[3 + c]);
          u[3][c] = g[6 + c];
        }
        float* dst = ul + buf * UBUF + e * US;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s02 = u[r][0] + u[r][2];
          *reinterpret_cast<float4*>(dst + 4 * r) = make_float4(u[r][0], 0.5f * (s02 + u[r][1]), 0.5f * (s02 - u[r][1]), u[r][2]);
        }
      }
    }
  };

  f32x4 acc[16][NB];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[f][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // chunk 0 -> buffer 0; chunk 1 requested into buffer 1
  issue_patch(pl);
  load_filter(0);
  store_filter(0);                                              // (waits for its own filter loads)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (CK < s.Cin) {
    issue_patch(pl + PBUF);
    load_filter(CK);
  }

  int buf = 0;
  for (int c0 = 0; c0 < s.Cin; c0 += CK, buf ^= 1) {
    const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)(pl + buf * PBUF + a_off);   // LDS byte address
    const float* pb = ul + buf * UBUF + b_off;
#pragma unroll
    for (int ks = 0; ks < (KS ? 1 : CK / 4); ++ks) {
      f32x4 ub[NB][4];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) ub[nb][j] = *reinterpret_cast<const f32x4*>(pb + (ks * 4 * NT + nb * 16) * US + 4 * j);
      float d[4][4];
      {
        f32x2 p[4][3];
        lds_read_patch<P::PWS * 4>(pa + ks * 4 * CIS * 4, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) { d[r][0] = p[r][0].y; d[r][1] = p[r][1].x; d[r][2] = p[r][1].y; d[r][3] = p[r][2].x; }
      }
      float v[4][4];                      // V = B^T d B
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c], t2 = d[2][c] - d[1][c], t3 = d[1][c] - d[3][c];
        d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r][0] = d[r][0] - d[r][2];
        v[r][1] = d[r][1] + d[r][2];
        v[r][2] = d[r][2] - d[r][1];
        v[r][3] = d[r][1] - d[r][3];
      }
#pragma unroll
      for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[f][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[f >> 2][f & 3], ub[nb][f >> 2][f & 3], acc[f][nb], 0, 0, 0);
    }
    if (c0 + CK < s.Cin) {
      store_filter(buf ^ 1);                                    // chunk c0 + CK: its filter values (registers) into the other buffer ...
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // ... and this wave's share of its patch has landed there
      __syncthreads();                                          // everybody's has; everybody is done reading THIS buffer
      if (c0 + 2 * CK < s.Cin) {                                // which chunk c0 + 2 CK may now overwrite, under the next chunk's MFMAs
        issue_patch(pl + buf * PBUF);
        load_filter(c0 + 2 * CK);
      }
    }
  }

  // ---- epilogue: lane (q = lane / 16, j = lane % 16) holds tiles 4q .. 4q+3 of output channel co0 + 16 nb + j
  const int q = kk, j = mi;
  const uint32_t HW = (uint32_t)(s.H * s.W);
  const int Wr = s.W >> 1;
  const uint32_t HWr = HW >> 2;
  // output transform of tile pair (2 pr2, 2 pr2 + 1) of block nb: two rows of four x-adjacent pixels
  auto out_rows = [&](int nb, int pr2, float (&o)[2][4]) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int r = 2 * pr2 + tt;
      float m[4][4];
#pragma unroll
      for (int f = 0; f < 16; ++f) m[f >> 2][f & 3] = acc[f][nb][r];
      float sa[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sa[0][c] = m[0][c] + m[1][c] + m[2][c];
        sa[1][c] = m[1][c] - m[2][c] - m[3][c];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        o[a][2 * tt] = sa[a][0] + sa[a][1] + sa[a][2];
        o[a][2 * tt + 1] = sa[a][1] - sa[a][2] - sa[a][3];
      }
    }
  };
  // bias / residual / store of one row of four pixels
  auto finish_row = [&](int nb, int pr2, int a, float4 o) {
    const int co = co0 + nb * 16 + j;
    int ti, ty, tx;
    WinoMap<G>::at(wave, 4 * q + 2 * pr2, ti, ty, tx);
    const int b = tc.b0 + ti;
    if (co >= s.Cout || b >= s.B) return;
    const float bv = bias ? bias[co] : 0.f;
    o.x += bv; o.y += bv; o.z += bv; o.w += bv;
    const int hh = tc.h0 + ty + a, ww = tc.w0 + tx;
    const int64_t plane = (int64_t)b * s.Cout + co;
    if (residual) {
      if (s.res_up) {
        const float2 rr = *reinterpret_cast<const float2*>(residual + plane * HWr + (int64_t)(hh >> 1) * Wr + (ww >> 1));
        o.x += rr.x; o.y += rr.x; o.z += rr.y; o.w += rr.y;
      } else {
        const float4 rr = *reinterpret_cast<const float4*>(residual + plane * HW + (int64_t)hh * s.W + ww);
        o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
      }
    }
    *reinterpret_cast<float4*>(y + plane * HW + (int64_t)hh * s.W + ww) = o;
  };
  if constexpr (!KS) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int pr2 = 0; pr2 < 2; ++pr2) {
        float o[2][4];
        out_rows(nb, pr2, o);
#pragma unroll
        for (int a = 0; a < 2; ++a) finish_row(nb, pr2, a, make_float4(o[a][0], o[a][1], o[a][2], o[a][3]));
      }
  } else {
    // K-split: the four waves hold partial sums of the SAME outputs.  Each transforms its own (the transform is linear), leaves
    // the rows in LDS -- red[wave][row slot][lane], over the filter buffers, which nobody reads any more -- and finishes the row
    // slots s == wave (mod 4), summing the four partials in wave order.
    constexpr int SLOTS = NB * 4;
    static_assert(4 * SLOTS * 64 * 4 <= 2 * UBUF, "the reduction buffer fits the filter buffers");
    float4* red = reinterpret_cast<float4*>(ul);
    __syncthreads();                              // every wave is past its last read of the filter buffers
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int pr2 = 0; pr2 < 2; ++pr2) {
        float o[2][4];
        out_rows(nb, pr2, o);
#pragma unroll
        for (int a = 0; a < 2; ++a) red[(wave * SLOTS + (nb * 4 + pr2 * 2 + a)) * 64 + lane] = make_float4(o[a][0], o[a][1], o[a][2], o[a][3]);
      }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int slot = nb * 4 + wave;             // (pr2, a) = (wave >> 1, wave & 1) of block nb
      float4 o = red[slot * 64 + lane];
#pragma unroll
      for (int ww2 = 1; ww2 < 4; ++ww2) {
        const float4 t = red[(ww2 * SLOTS + slot) * 64 + lane];
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      }
      finish_row(nb, wave >> 1, wave & 1, o);
    }
  }
}

// =========================================================================== AvgPool2d(2) o conv3x3 (and the up-conv's input gradient)
// out[i][j] = 1/4 sum_{a,b in {0,1}} conv3x3(x)[2i+a][2j+b]  =  1/4 (A 1)^T [ (G g G^T) .* (B^T d B) ] (A 1)  over the SAME 4x4
// stride-2 input tiles as above.  A 1 = (1, 2, 0, -1): frequency row / column 2 drops out -- 9 multiplies per output and channel
// pair where the 4x4 stride-2 form (conv_upT_dma_kernel) does 16 and the unfused pair 36 -- and the output transform is a
// plain weighted sum, so it folds into the filter: the 9 frequency products accumulate into ONE accumulator (a GEMM with
// K = 9 Cin), no output transform at all.  The filter arrives as the 4x4 stride-2 form w4[n][k][u][v] = sum_{r in S(u), s in S(v)}
// c g[r][s], S = {0},{0,1},{1,2},{2} (tg_poolconv3x3_weights: c = 1/4; tg_upconv3x3_weights_t: the transposed flipped filter,
// c = 1), from which U' = c (A1 A1^T) .* (G g G^T) is sums of its entries at rows / columns {0, 2, 3} -- no subtraction of
// nearly equal numbers:   U'[0] = (w00, w00+w02, -w03)   U'[1] = (w00+w20, w00+w02+w20+w22, -(w03+w23))   U'[3] = (-w30, -(w30+w32), w33).
// Geometry: G tiles the HIGH-resolution input plane (2H x 2W); a workgroup makes 64 outputs x 16 NB channels.
constexpr int WINO_PS = 12;     // floats per (channel, output channel) slot of U': 9 + 3 of padding (48 bytes: conflict-free b128)

template <class G, int NB>
__global__ void __launch_bounds__(CT_THREADS, (G::NI > 1) ? 1 : 3)
conv_poolwino_dma_kernel(const float* __restrict__ x, const float* __restrict__ w4, const float* __restrict__ bias,
                         const float* __restrict__ residual, float* __restrict__ y, Shape s /* H x W: the LOW-resolution plane */,
                         int flags) {
  using P = WinoPatch<G>;
  constexpr int CK = WINO_CK, NT = 16 * NB, US = WINO_PS, CIS = P::CIS;
  constexpr int PCH = CK * P::CPC, PCHP = dma_pad(PCH);
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, UBUF = CK * NT * US;
  constexpr int FPT = (CK * NT + CT_THREADS - 1) / CT_THREADS;     // filter pairs per thread
  __shared__ __attribute__((aligned(16))) float pl[2 * PBUF];
  __shared__ __attribute__((aligned(16))) float ul[2 * UBUF];

  const int lane = threadIdx.x & 63, wave = wave_index();
  const int Hh = 2 * s.H, Wh = 2 * s.W;                            // the input plane
  int bid = blockIdx.x;
  if (flags & 1) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const TileCoord tc = decode_tile<G>(bid, Hh, Wh);                // (high-resolution coordinates)
  const int co0 = blockIdx.y * NT;
  const uint32_t HWp = (uint32_t)(Hh * Wh);

  uint32_t poff[NVP];
#pragma unroll
  for (int i = 0; i < NVP; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int ci = e / P::CPC, rem = e % P::CPC;
    const int row = rem / P::QR, q = rem % P::QR;
    const int pimg = row / P::PH, r = row % P::PH;
    const int hh = tc.h0 + r - 1, ww = tc.w0 - 4 + 4 * q;
    const bool ok = (e < PCH) && (rem < P::RAW / 4) && (tc.b0 + pimg < s.B) && (hh >= 0) && (hh < Hh) && (ww >= 0) && (ww < Wh);
    poff[i] = ok ? (__umul24(__umul24(pimg, s.Cin) + ci, HWp) + __umul24(hh, Wh) + ww) << 2 : DMA_OOB;
  }
  const char* xb = reinterpret_cast<const char*>(x) + ((int64_t)tc.b0 * s.Cin * HWp) * 4;
  int64_t xbytes = (int64_t)(s.B - tc.b0) * s.Cin * HWp * 4;
  const int64_t xstep = (int64_t)CK * HWp * 4;
  auto issue_patch = [&](float* buf) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NVP; ++i)
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP)      // (scalar)
        dma16(rx, buf + (i * CT_THREADS + wave * 64) * 4, poff[i]);
    xb += xstep; xbytes -= xstep;
  };

  const int mi = lane & 15, kk = lane >> 4;
  int img, y0, x0;
  WinoMap<G>::at(wave, mi, img, y0, x0);
  const int a_off = (img * P::PH + y0) * P::PWS + x0 + 2 + kk * CIS;
  const int b_off = (kk * NT + mi) * US;

  float g9[FPT][9];                       // w4 at rows / columns {0, 2, 3}
  auto load_filter = [&](int c0) {
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const int e = t * CT_THREADS + threadIdx.x;
      const int fn = e % NT, fk = e / NT;
      const bool ok = (e < CK * NT) && (co0 + fn < s.Cout) && (c0 + fk < s.Cin);
      const float* src = w4 + ((int64_t)(co0 + fn) * s.Cin + (c0 + fk)) * 16;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g9[t][a * 3 + b] = ok ? src[(a == 0 ? 0 : a + 1) * 4 + (b == 0 ? 0 : b + 1)] : 0.f;
    }
  };
  auto store_filter = [&](int buf) {
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const int e = t * CT_THREADS + threadIdx.x;
      if (e < CK * NT) {
        const float* g = g9[t];           // g[a*3+b] = w4[{0,2,3}[a]][{0,2,3}[b]]
        float* dst = ul + buf * UBUF + e * US;
        const float r0 = g[0] + g[1], r1 = g[3] + g[4];
        *reinterpret_cast<float4*>(dst) = make_float4(g[0], r0, -g[2], g[0] + g[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(r0 + r1, -(g[2] + g[5]), -g[6], -(g[6] + g[7]));
        dst[8] = g[8];
      }
    }
  };

  f32x4 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_patch(pl);
  load_filter(0);
  store_filter(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (CK < s.Cin) {
    issue_patch(pl + PBUF);
    load_filter(CK);
  }

  int buf = 0;
  for (int c0 = 0; c0 < s.Cin; c0 += CK, buf ^= 1) {
    const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)(pl + buf * PBUF + a_off);
    const float* pb = ul + buf * UBUF + b_off;
#pragma unroll
    for (int ks = 0; ks < CK / 4; ++ks) {
      f32x4 ua[NB], ubb[NB];
      float uc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float* q = pb + (ks * 4 * NT + nb * 16) * US;
        ua[nb] = *reinterpret_cast<const f32x4*>(q);
        ubb[nb] = *reinterpret_cast<const f32x4*>(q + 4);
        uc[nb] = q[8];
      }
      float d[4][4];
      {
        f32x2 p[4][3];
        lds_read_patch<P::PWS * 4>(pa + ks * 4 * CIS * 4, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) { d[r][0] = p[r][0].y; d[r][1] = p[r][1].x; d[r][2] = p[r][1].y; d[r][3] = p[r][2].x; }
      }
      float t[3][4], v[9];                // rows / columns {0, 1, 3} of B^T d B
#pragma unroll
      for (int c = 0; c < 4; ++c) { t[0][c] = d[0][c] - d[2][c]; t[1][c] = d[1][c] + d[2][c]; t[2][c] = d[1][c] - d[3][c]; }
#pragma unroll
      for (int a = 0; a < 3; ++a) { v[a * 3] = t[a][0] - t[a][2]; v[a * 3 + 1] = t[a][1] + t[a][2]; v[a * 3 + 2] = t[a][1] - t[a][3]; }
#pragma unroll
      for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float u = f < 4 ? ua[nb][f] : f < 8 ? ubb[nb][f - 4] : uc[nb];
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[f], u, acc[nb], 0, 0, 0);
        }
    }
    if (c0 + CK < s.Cin) {
      store_filter(buf ^ 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (c0 + 2 * CK < s.Cin) {
        issue_patch(pl + buf * PBUF);
        load_filter(c0 + 2 * CK);
      }
    }
  }

  // ---- epilogue: lane (q = lane / 16, j = lane % 16) holds outputs 4q .. 4q+3 of its wave's strip -- four x-adjacent low-resolution pixels
  const int q = kk, j = mi;
  int ti, ty, tx;
  WinoMap<G>::at(wave, 4 * q, ti, ty, tx);
  const int b = tc.b0 + ti;
  if (b >= s.B) return;
  const int hh = (tc.h0 + ty) >> 1, ww = (tc.w0 + tx) >> 1;
  const uint32_t HW = (uint32_t)(s.H * s.W);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int co = co0 + nb * 16 + j;
    if (co >= s.Cout) continue;
    const float bv = bias ? bias[co] : 0.f;
    const int64_t off = ((int64_t)b * s.Cout + co) * HW + (int64_t)hh * s.W + ww;
    float4 o = make_float4(acc[nb][0] + bv, acc[nb][1] + bv, acc[nb][2] + bv, acc[nb][3] + bv);
    if (residual) {
      const float4 rr = *reinterpret_cast<const float4*>(residual + off);
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    *reinterpret_cast<float4*>(y + off) = o;
  }
}
