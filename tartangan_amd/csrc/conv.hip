// fp32 convolution (3x3 pad 1 / 1x1, stride 1, NCHW) as LDS-staged implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32,
// bit-identical to an fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
//   forward / dgrad : M = Cout tile, N = output pixels, K = (ci, tap).  The input patch (with
//                     halo) and the filter slice are staged once per Cin chunk; A/B fragments
//                     are single ds_read_b32 with compile-time offsets.  dgrad is the same
//                     kernel reading the filter transposed + spatially flipped.
//   wgrad           : M = Cout tile, N = (ci, tap) columns, K = pixels; each workgroup keeps its
//                     gw slice in MFMA accumulators while it walks its share of pixel tiles,
//                     then writes ONE partial; a second kernel sums the partials in a fixed
//                     order (deterministic, no float atomics).
//
// Staging: 16-byte global loads (whole rows of the NCHW planes are 16-byte aligned when W % 4 == 0)
// into registers one chunk AHEAD of the MFMA loop, ds_write_b128 into a patch whose interior starts
// at a 16-byte boundary (halo columns by scalar loads).  Shapes that break the alignment
// assumptions take a scalar path with identical LDS layout.
//
// Pixel tiles:
//   256 pixels, waves split pixels : G4 (16 4x4 images) G8 (4 8x8) G16 (one 16x16) GX (8 rows x 32)
//    64 pixels, waves split K       : G4k G8k G16k  -- small-spatial layers still fill the chip
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CT_THREADS = 256;

template <int TH_, int TW_, int NI_>
struct Geo {
  static constexpr int TH = TH_, TW = TW_, NI = NI_, NPIX = TH_ * TW_ * NI_;
  static_assert(NPIX == 512 || NPIX == 256 || NPIX == 64, "a pixel tile is 512 / 256 pixels (waves split pixels) or 64 (waves split K)");
  static constexpr int PPW = (NPIX == 64) ? 64 : NPIX / 4;     // pixels per wave
  static_assert(TW_ % 4 == 0, "rows are staged as float4");
};
using G4 = Geo<4, 4, 16>;
using G8 = Geo<8, 8, 4>;
using G16 = Geo<16, 16, 1>;
using GX = Geo<8, 32, 1>;
using G4k = Geo<4, 4, 4>;
using G8k = Geo<8, 8, 1>;
using G16k = Geo<4, 16, 1>;
using GX2 = Geo<16, 32, 1>;     // 512-pixel tiles of the LDS-DMA kernel (conv_dma_kernel)
using G16x2 = Geo<16, 16, 2>;

// Kernel-size codes: 1 = 1x1; 3 = 3x3 pad 1; 2 = a 2x2 window of taps INSIDE the 3x3 halo (one output phase of
// conv3x3(nearest_up2x(.)), see tg_upconv3x3_*): patch geometry of the 3x3 kernel, 4 taps starting at (oy, ox).
template <int KS> struct KGeom {
  static constexpr int HALO = (KS == 1) ? 1 : 3;
  static constexpr int T = (KS == 2) ? 2 : KS;                   // taps per side
};

template <class G, int KS>
struct Patch {
  static constexpr int PAD = KGeom<KS>::HALO / 2;
  static constexpr int PH = G::TH + KGeom<KS>::HALO - 1;
  static constexpr int IOFF = (KGeom<KS>::HALO == 3) ? 4 : 0;    // interior column 0 sits at a 16-byte boundary
  static constexpr int ORG = IOFF - PAD;                         // LDS column of patch column 0 (= image column w0 - PAD)
  static constexpr int PWS = G::TW + 2 * IOFF;                   // row stride (multiple of 4)
  static constexpr int IMG = PH * PWS;
  static constexpr int RAW = G::NI * IMG;
  static constexpr int CIS = ((RAW + 31) / 32) * 32 + 16;        // per-channel stride; %32 == 16 keeps the k-halves of a
                                                                 // 32-lane LDS group on disjoint banks
  static constexpr int ROWS_PER_CI = G::NI * PH;
};

struct Shape {
  int B, Cin, Cout, H, W;
  int oy = 0, ox = 0;      // KS code 2: first tap row / column inside the 3x3 halo
  int os = 1, py = 0, px = 0;   // output written to an (os*H) x (os*W) plane at rows os*h + py, columns os*w + px
  int prio = 0;                 // LDS-DMA kernels: serial sections at wave priority 3, MFMA blocks at 0 (see conv_dma_kernel)
  int res_up = 0;               // 1: `residual` is (B, Cout, H/2, W/2) and is added nearest-neighbour upsampled (os == 1, H, W even)
};

struct TileCoord {
  int b0, h0, w0;
};

template <class G>
__device__ __forceinline__ TileCoord decode_tile(int t, int H, int W) {
  const int tw_tiles = (W + G::TW - 1) / G::TW, th_tiles = (H + G::TH - 1) / G::TH;
  const int per = tw_tiles * th_tiles;
  const int grp = t / per, rem = t - grp * per;
  const int tr = rem / tw_tiles;
  TileCoord c;
  c.b0 = grp * G::NI;
  c.h0 = tr * G::TH;
  c.w0 = (rem - tr * tw_tiles) * G::TW;
  return c;
}

template <class G>
static inline int num_tiles(int B, int H, int W) {
  return ((B + G::NI - 1) / G::NI) * ((H + G::TH - 1) / G::TH) * ((W + G::TW - 1) / G::TW);
}

// LDS offset (inside one channel's patch) of the top-left tap of tile pixel p
template <class G, int KS>
__device__ __forceinline__ int pix_off(int p) {
  using P = Patch<G, KS>;
  const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
  const int r = rem / G::TW, c = rem % G::TW;
  return (img * P::PH + r) * P::PWS + c + P::ORG;
}

// ------------------------------------------------------------------ patch stager
// x[b0.., c0..c0+CK, h0-PAD.., w0-PAD..] (zero outside the tensor) -> lds[ci][img][r][ORG + c]
template <class G, int KS, int CK>
struct PatchStager {
  using P = Patch<G, KS>;
  static constexpr int Q = G::TW / 4;
  static constexpr int ROWS = CK * P::ROWS_PER_CI;
  static constexpr int NV = (ROWS * Q + CT_THREADS - 1) / CT_THREADS;
  static constexpr int NHALO = (KGeom<KS>::HALO == 3) ? ROWS * 2 : 0;
  static constexpr int NH = (NHALO + CT_THREADS - 1) / CT_THREADS;
  float4 v[NV];
  float hv[NH > 0 ? NH : 1];

  // Addressing: ONE wave-uniform 64-bit base per (tile, chunk) -- the element at (b0, c0, h0 - PAD, w0 - PAD), computed
  // on the scalar unit -- plus a per-lane 32-bit BYTE offset built from 24-bit multiplies (full rate; the host checks
  // the extents, see check_shape).  64-bit per-lane products (v_mad_u64_u32 / v_mul_lo_u32 are quarter rate) used to
  // cost the 16/32-channel layers as many issue cycles as their MFMAs.
  struct RowRef { bool ok; uint32_t boff; };
  __device__ __forceinline__ static RowRef row_ref(int row, const Shape& s, int C, int c0, const TileCoord& tc) {
    const int r = row % P::PH;
    const int t = row / P::PH;
    const int img = t % G::NI;
    const int ci = t / G::NI;
    const int b = tc.b0 + img, cc = c0 + ci, hh = tc.h0 + r - P::PAD;
    RowRef o;
    o.ok = (b < s.B) && (cc < C) && (hh >= 0) && (hh < s.H);
    o.boff = (__umul24(__umul24(img, C) + ci, s.H * s.W) + __umul24(r, s.W)) << 2;
    return o;
  }
  __device__ __forceinline__ static const char* tile_base(const float* __restrict__ x, const Shape& s, int C, int c0, const TileCoord& tc) {
    return reinterpret_cast<const char*>(x) + ((((int64_t)tc.b0 * C + c0) * s.H + (tc.h0 - P::PAD)) * s.W + (tc.w0 - P::PAD)) * 4;
  }
  template <class T>
  __device__ __forceinline__ static T ld(const char* base, uint32_t boff) { return *reinterpret_cast<const T*>(base + boff); }

  __device__ __forceinline__ void load(const float* __restrict__ x, const Shape& s, int C, int c0, const TileCoord& tc, bool vec) {
    const char* base = tile_base(x, s, C, c0, tc);
    if (vec) {                                            // W % 4 == 0: a quad is all-in or all-out
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = threadIdx.x + i * CT_THREADS;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < ROWS * Q) {
          const int q = e % Q;
          const RowRef rr = row_ref(e / Q, s, C, c0, tc);
          if (rr.ok && tc.w0 + 4 * q < s.W) val = ld<float4>(base, rr.boff + 4 * (P::PAD + 4 * q));
        }
        v[i] = val;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int e = threadIdx.x + i * CT_THREADS;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < ROWS * Q) {
          const int q = e % Q;
          const RowRef rr = row_ref(e / Q, s, C, c0, tc);
          const int ww = tc.w0 + 4 * q;
          const uint32_t off = rr.boff + 4 * (P::PAD + 4 * q);
          if (rr.ok) {
            if (ww < s.W) val.x = ld<float>(base, off);
            if (ww + 1 < s.W) val.y = ld<float>(base, off + 4);
            if (ww + 2 < s.W) val.z = ld<float>(base, off + 8);
            if (ww + 3 < s.W) val.w = ld<float>(base, off + 12);
          }
        }
        v[i] = val;
      }
    }
    if constexpr (KGeom<KS>::HALO == 3) {
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int e = threadIdx.x + i * CT_THREADS;
        float val = 0.f;
        if (e < NHALO) {
          const RowRef rr = row_ref(e >> 1, s, C, c0, tc);
          const int ww = (e & 1) ? tc.w0 + G::TW : tc.w0 - 1;
          if (rr.ok && ww >= 0 && ww < s.W) val = ld<float>(base, rr.boff + ((e & 1) ? 4 * (G::TW + 1) : 0));
        }
        hv[i] = val;
      }
    }
  }

  // CISX: per-channel stride of the LDS image (the Winograd kernel keeps its own: wino.h)
  template <int CISX = P::CIS>
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      if (e < ROWS * Q) {
        const int q = e % Q, row = e / Q;
        const int ci = row / P::ROWS_PER_CI, lrow = row % P::ROWS_PER_CI;
        *reinterpret_cast<float4*>(lds + ci * CISX + lrow * P::PWS + P::IOFF + 4 * q) = v[i];
      }
    }
    if constexpr (KGeom<KS>::HALO == 3) {
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        const int e = threadIdx.x + i * CT_THREADS;
        if (e < NHALO) {
          const int row = e >> 1;
          const int ci = row / P::ROWS_PER_CI, lrow = row % P::ROWS_PER_CI;
          lds[ci * CISX + lrow * P::PWS + ((e & 1) ? P::IOFF + G::TW : P::IOFF - 1)] = hv[i];
        }
      }
    }
  }
};

// ------------------------------------------------------------------ filter-slice stager
// LDS: wl[(ci*KK + tap)][co], row stride CTS.  ROWLEN contiguous floats per source row:
//   forward: row = co,  w[co][ci0..ci0+CK][tap]          (element k = ci*KK + tap)
//   dgrad  : row = ci,  w[ci][co0..co0+CT][KK-1-tap]     (element k = co*KK + tp), w stored [Cin_eff][Cout_eff][KK]
// (KS code 4: the 16-tap filter rows of the stride-2 transpose kernel, conv_upT_kernel; it only uses the filter stager)
template <int KS, bool WK> struct FwdCfg { static constexpr int CK = (KS == 3) ? (WK ? 16 : 8) : (KS == 2) ? (WK ? 32 : 16) : (KS == 4) ? (WK ? 16 : 4) : 32; };

template <int KS, int CT, bool WK>
struct WTile {
  static constexpr int KK = KGeom<KS>::T * KGeom<KS>::T;
  static constexpr int CK = FwdCfg<KS, WK>::CK;
  static constexpr int CTS = CT + 1;            // odd stride: transposed staging writes and fragment reads both spread over banks
  static constexpr int SIZE = CK * KK * CTS;
};

template <int KS, int CT, bool DGRAD, bool WK>
struct WeightStager {
  using WT = WTile<KS, CT, WK>;
  static constexpr int KK = WT::KK, CK = WT::CK, CTS = WT::CTS;
  static constexpr int NROWS = DGRAD ? CK : CT;
  static constexpr int ROWLEN = DGRAD ? CT * KK : CK * KK;
  static_assert(ROWLEN % 4 == 0, "filter rows are staged as float4");
  static constexpr int Q = ROWLEN / 4;
  static constexpr int NV = (NROWS * Q + CT_THREADS - 1) / CT_THREADS;
  float4 v[NV];

  __device__ __forceinline__ void load(const float* __restrict__ w, int Cin, int Cout, int ci0, int co0, bool vec) {
    // valid elements per row (ragged last chunk / last channel tile)
    const int nvalid = (DGRAD ? min(CT, Cout - co0) : min(CK, Cin - ci0)) * KK;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NROWS * Q) {
        const int q = e % Q, row = e / Q;
        const bool row_ok = DGRAD ? (ci0 + row < Cin) : (co0 + row < Cout);
        if (row_ok) {
          // wave-uniform base (first row of the slice) + 32-bit lane offset
          const float* base = w + (DGRAD ? ((int64_t)ci0 * Cout + co0) * KK : ((int64_t)co0 * Cin + ci0) * KK);
          const float* src = base + (__umul24(row, (DGRAD ? Cout : Cin) * KK) + 4 * q);
          if (vec && 4 * q + 3 < nvalid) {
            val = *reinterpret_cast<const float4*>(src);
          } else {
            if (4 * q < nvalid) val.x = src[0];
            if (4 * q + 1 < nvalid) val.y = src[1];
            if (4 * q + 2 < nvalid) val.z = src[2];
            if (4 * q + 3 < nvalid) val.w = src[3];
          }
        }
      }
      v[i] = val;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ wl) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      if (e < NROWS * Q) {
        const int q = e % Q, row = e / Q;
        const float vals[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = 4 * q + u;
          if (!DGRAD) {
            wl[k * CTS + row] = vals[u];                                   // k = ci*KK + tap, row = co
          } else {
            const int co = k / KK, tp = k % KK;
            wl[(row * KK + (KK - 1 - tp)) * CTS + co] = vals[u];           // row = ci
          }
        }
      }
    }
  }
};

// =========================================================================== forward / dgrad
template <class G, int KS, int MF, int MT, bool WK>
struct FwdCore {
  using P = Patch<G, KS>;
  static constexpr int CT = MF * MT;
  using WT = WTile<KS, CT, WK>;
  static constexpr int KK = WT::KK, CK = WT::CK, CTS = WT::CTS;
  static constexpr int NT = G::PPW / MF;             // pixel sub-tiles per wave
  static constexpr int KG = (MF == 32) ? 2 : 4;      // k values consumed per MFMA
  static constexpr int NREG = (MF == 32) ? 16 : 4;
  static constexpr int NG = CK / KG;                 // k-groups per chunk
  static constexpr int GSTEP = WK ? 4 : 1;           // WK: wave w takes k-groups w, w+4, ...
  static_assert(!WK || NG % 4 == 0, "K-split needs a multiple of 4 k-groups per chunk");
  using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;

  __device__ __forceinline__ static void zero(acc_t (&acc)[MT][NT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < NREG; ++r) acc[m][n][r] = 0.f;
  }

  // One staged Cin chunk.  No per-k-group branch here: channels past Cin are zero in LDS (a ragged last chunk just
  // multiplies zeros).  A wave-uniform `if` around the MFMA block made hipcc shuttle all accumulators between
  // VGPRs and AGPRs on both sides of it (64 v_accvgpr moves + s_nop 15 per 18 MFMAs).
  __device__ __forceinline__ static void mfma_chunk(acc_t (&acc)[MT][NT], const float* __restrict__ pl, const float* __restrict__ wl,
                                                    int lane_a, const int (&lane_b)[NT]) {
#pragma unroll
    for (int gi = 0; gi < NG / GSTEP; ++gi) {
      const int g = gi * GSTEP;               // + g0 (in the lane bases)
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        const int kh = tap / KGeom<KS>::T, kw = tap % KGeom<KS>::T;    // (+ the window offset, folded into lane_b)
        float a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = wl[lane_a + ((g * KG) * KK + tap) * CTS + m * MF];
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = pl[lane_b[n] + (g * KG) * P::CIS + kh * P::PWS + kw];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (MF == 32)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
            else
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
          }
      }
    }
  }

  // D[row = co][col = pixel] + bias (+ residual) -> y   (WK: each wave stores a quarter of the registers)
  // Same addressing scheme as the stagers: a wave-uniform base per tile, 32-bit lane byte offsets, the per-register
  // channel step (a compile-time constant times H*W) on the scalar unit.
  __device__ __forceinline__ static void epilogue(const acc_t (&acc)[MT][NT], const float* __restrict__ bias,
                                                  const float* __restrict__ residual, float* __restrict__ y, const Shape& s,
                                                  const TileCoord& tc, int co0, int pix0, int j, int h, int wave) {
    const int Wo = s.W * s.os;
    const uint32_t HW = (uint32_t)(s.H * s.os * Wo);            // output plane (os = 2: one phase of an upsampling conv)
    // ---- wide path: a 4 x 4 transpose inside each lane quad (registers = 4 consecutive channel rows, lanes = 4
    // consecutive pixels of a row) turns four 4-byte stores per register group into ONE 16-byte store per lane (and the
    // residual / bias reads likewise): these short-K kernels otherwise end on a store-issue-bound tail
    // (cdna_hip_programming.md T21).  Needs whole pixel quads in a row (W % 4 == 0, 16-byte aligned planes).
    constexpr int NRW = WK ? NREG / 4 : NREG;
    if constexpr (NRW % 4 == 0) {
      const bool wide = (s.os == 1) && (s.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                        (residual == nullptr || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
      if (wide) {
        const int q = j & 3;
        const int64_t tile0w = ((((int64_t)tc.b0 * s.Cout + co0) * s.H + tc.h0) * s.W + tc.w0) * 4;
        char* ybw = reinterpret_cast<char*>(y) + tile0w;
        // res_up: the residual lives at half the resolution (tile origins are even, so the halved coordinates split the same way)
        const int Wr = s.W >> 1;
        const uint32_t HWr = HW >> 2;
        const int64_t tile0r = ((((int64_t)tc.b0 * s.Cout + co0) * (s.H >> 1) + (tc.h0 >> 1)) * Wr + (tc.w0 >> 1)) * 4;
        const char* rbw = reinterpret_cast<const char*>(residual) + (s.res_up ? tile0r : tile0w);
        constexpr int NQW = WK ? 4 : 1;
#pragma unroll
        for (int qq = 0; qq < NQW; ++qq) {
          if (WK && qq != wave) continue;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int a4 = 0; a4 < NRW / 4; ++a4) {
              const int r0 = qq * NRW + 4 * a4;                                   // first register of the group
              const int row = m * MF + ((MF == 32) ? 8 * (r0 >> 2) : 0) + 4 * h + q;     // this lane's channel after the transpose
              const bool row_ok = co0 + row < s.Cout;
              const float bvw = (bias && row_ok) ? bias[co0 + row] : 0.f;
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                float t0 = acc[m][n][r0], t1 = acc[m][n][r0 + 1], t2 = acc[m][n][r0 + 2], t3 = acc[m][n][r0 + 3];
                {   // lanes differing in bit 0 exchange on the register pairs (0,1), (2,3)
                  const float x01 = (q & 1) ? t0 : t1, x23 = (q & 1) ? t2 : t3;
                  const float y01 = __shfl_xor(x01, 1, 64), y23 = __shfl_xor(x23, 1, 64);
                  if (q & 1) { t0 = y01; t2 = y23; } else { t1 = y01; t3 = y23; }
                }
                {   // lanes differing in bit 1 exchange on the pairs (0,2), (1,3)
                  const float x02 = (q & 2) ? t0 : t2, x13 = (q & 2) ? t1 : t3;
                  const float y02 = __shfl_xor(x02, 2, 64), y13 = __shfl_xor(x13, 2, 64);
                  if (q & 2) { t0 = y02; t1 = y13; } else { t2 = y02; t3 = y13; }
                }
                const int p = pix0 + n * MF + (j & ~3);                            // first pixel of the quad
                const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
                const int pr = rem / G::TW, pc = rem % G::TW;
                if (!row_ok || tc.b0 + img >= s.B || tc.h0 + pr >= s.H || tc.w0 + pc >= s.W) continue;
                const uint32_t off = (__umul24(__umul24(img, s.Cout) + row, HW) + __umul24(pr, s.W) + pc) << 2;
                float4 o = make_float4(t0 + bvw, t1 + bvw, t2 + bvw, t3 + bvw);
                if (residual) {
                  if (s.res_up) {
                    const uint32_t offr = (__umul24(__umul24(img, s.Cout) + row, HWr) + __umul24(pr >> 1, Wr) + (pc >> 1)) << 2;
                    const float2 rr = *reinterpret_cast<const float2*>(rbw + offr);
                    o.x += rr.x; o.y += rr.x; o.z += rr.y; o.w += rr.y;
                  } else {
                    const float4 rr = *reinterpret_cast<const float4*>(rbw + off);
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                  }
                }
                *reinterpret_cast<float4*>(ybw + off) = o;
              }
            }
        }
        return;
      }
    }
    const int64_t tile0 = ((((int64_t)tc.b0 * s.Cout + co0) * (s.H * s.os) + tc.h0 * s.os + s.py) * Wo + tc.w0 * s.os + s.px) * 4;
    char* ybase = reinterpret_cast<char*>(y) + tile0;
    const int Wr = s.W >> 1;
    const uint32_t HWr = (uint32_t)(s.H >> 1) * Wr;
    const int64_t tile0r = ((((int64_t)tc.b0 * s.Cout + co0) * (s.H >> 1) + (tc.h0 >> 1)) * Wr + (tc.w0 >> 1)) * 4;
    const char* rbase = reinterpret_cast<const char*>(residual) + (s.res_up ? tile0r : tile0);
    const char* bbase = reinterpret_cast<const char*>(bias + co0);
    const uint32_t lane_row = __umul24(4 * h, HW);
    constexpr int NR = WK ? NREG / 4 : NREG;      // registers this wave stores per (m, n): WK waves take a quarter each
    constexpr int NQ = WK ? 4 : 1;
    auto row_of = [](int m, int r) { return m * MF + ((MF == 32) ? ((r & 3) + 8 * (r >> 2)) : r); };   // + 4*h
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      if (WK && qq != wave) continue;             // register indices stay compile-time constants
      // All loads of a group are issued before the first use (one wait per group, not one per register): the bias
      // values of this lane's rows first, then per pixel sub-tile the residual values, then the stores.
      float bv[MT][NR];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < NR; ++k) bv[m][k] = 0.f;
      if (bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const int row = min(row_of(m, qq * NR + k) + 4 * h, s.Cout - 1 - co0);     // clamped: always a valid address
            bv[m][k] = *reinterpret_cast<const float*>(bbase + (uint32_t)(row << 2));
          }
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int p = pix0 + n * MF + j;
        const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
        const int pr = rem / G::TW, pc = rem % G::TW;
        if (tc.b0 + img >= s.B || tc.h0 + pr >= s.H || tc.w0 + pc >= s.W) continue;
        const uint32_t lane_off = __umul24(__umul24(img, s.Cout), HW) + __umul24(pr * s.os, Wo) + pc * s.os + lane_row;
        float o[MT][NR];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int k = 0; k < NR; ++k) o[m][k] = 0.f;
        if (residual) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int k = 0; k < NR; ++k) {
              const int rc = row_of(m, qq * NR + k);
              if (co0 + rc + 4 * h < s.Cout) {
                const uint32_t ro = s.res_up ? __umul24(__umul24(img, s.Cout) + 4 * h + rc, HWr) + __umul24(pr >> 1, Wr) + (pc >> 1)
                                             : lane_off + (uint32_t)rc * HW;
                o[m][k] = *reinterpret_cast<const float*>(rbase + (ro << 2));
              }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const int rc = row_of(m, qq * NR + k);
            if (co0 + rc + 4 * h < s.Cout)
              *reinterpret_cast<float*>(ybase + ((lane_off + (uint32_t)rc * HW) << 2)) = (acc[m][n][qq * NR + k] + bv[m][k]) + o[m][k];
          }
      }
    }
  }
};

// Waves per SIMD the register allocator is asked to fit (512 unified registers / SIMD lane): the 32-channel tiles sit
// just above the 128-register line without the hint.
template <class G, int KS, int MF, int MT> struct FwdOcc {   // (the multi-image 4x4 / 8x8 geometries would spill)
  static constexpr int W = (KS != 1 && MF * MT == 32 && G::NI == 1 && G::NPIX == 256) ? 4 : 1;
};

template <class G, int KS, int MF, int MT, bool DGRAD>
__global__ void __launch_bounds__(CT_THREADS, (FwdOcc<G, KS, MF, MT>::W))
conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                const float* __restrict__ residual /*nullable, same shape as y*/, float* __restrict__ y, Shape s, int vec_x, int vec_w) {
  using P = Patch<G, KS>;
  constexpr bool WK = (G::NPIX == 64);        // waves split K (same pixels) instead of pixels
  using Core = FwdCore<G, KS, MF, MT, WK>;
  using WT = typename Core::WT;
  constexpr int CT = Core::CT, KK = Core::KK, CK = Core::CK, CTS = Core::CTS, NT = Core::NT, KG = Core::KG, NREG = Core::NREG;
  constexpr int STAGE = CK * P::CIS + WT::SIZE;
  constexpr int REDF = WK ? 4 * MT * NT * NREG * 64 : 0;
  __shared__ __attribute__((aligned(16))) float lds[STAGE > REDF ? STAGE : REDF];
  float* pl = lds;
  float* wl = lds + CK * P::CIS;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % MF, h = lane / MF;     // MF=32: h in {0,1}; MF=16: h in {0..3}
  const TileCoord tc = decode_tile<G>(blockIdx.x, s.H, s.W);
  const int co0 = blockIdx.y * CT;
  const int pix0 = WK ? 0 : wave * 64;
  const int g0 = WK ? wave : 0;

  int lane_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) lane_b[n] = (h + g0 * KG) * P::CIS + pix_off<G, KS>(pix0 + n * MF + j) + s.oy * P::PWS + s.ox;
  const int lane_a = ((h + g0 * KG) * KK) * CTS + j;

  typename Core::acc_t acc[MT][NT];
  Core::zero(acc);

  PatchStager<G, KS, CK> ps;
  WeightStager<KS, CT, DGRAD, WK> ws;
  ps.load(x, s, s.Cin, 0, tc, vec_x);
  ws.load(w, s.Cin, s.Cout, 0, co0, vec_w);

  for (int ci0 = 0; ci0 < s.Cin; ci0 += CK) {
    __syncthreads();                          // every wave is done reading the previous chunk
    ps.store(pl);
    ws.store(wl);
    __syncthreads();
    if (ci0 + CK < s.Cin) {                   // next chunk's global loads fly under this chunk's MFMAs
      ps.load(x, s, s.Cin, ci0 + CK, tc, vec_x);
      ws.load(w, s.Cin, s.Cout, ci0 + CK, co0, vec_w);
    }
    Core::mfma_chunk(acc, pl, wl, lane_a, lane_b);
  }

  if constexpr (WK) {
    // sum the four waves' partial accumulators through LDS (fixed order)
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < NREG; ++r) red[(((wave * MT + m) * NT + n) * NREG + r) * 64 + lane] = acc[m][n][r];
    __syncthreads();
    constexpr int PW_ = MT * NT * NREG * 64;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const int e = ((m * NT + n) * NREG + r) * 64 + lane;
          acc[m][n][r] = (red[e] + red[PW_ + e]) + (red[2 * PW_ + e] + red[3 * PW_ + e]);
        }
  }
  Core::epilogue(acc, bias, residual, y, s, tc, co0, pix0, j, h, wave);
}

// =========================================================================== 3x3 forward / dgrad, LDS-DMA staged
// Same implicit GEMM as conv_fwd_kernel, but the patch and the filter slice go global -> LDS directly
// (buffer_load_dwordx4 ... lds: 1 KiB per wave-instruction, no VGPR staging, no ds_write, no per-element address
// or bounds arithmetic in the loop) into TWO buffers, one barrier per channel chunk:
//     DMA(chunk 0);  for c: { vmcnt(0); barrier; DMA(chunk c + 1 -> other buffer); MFMA(chunk c) }
// What makes the LDS images lane-linear (an LDS-DMA writes base + lane * 16, not a scatter):
//   * patch rows are staged as WHOLE 16-byte chunks, four columns beyond the tile on either side (the halo column is
//     the last / first float of those chunks): a row is (TW + 8) / 4 chunks, a channel PH rows, channels back to back
//     (+ pad chunks so that the channel stride is 16 mod 32 banks);
//   * everything outside the image, beyond the batch or in a pad chunk is a lane whose buffer offset is out of range:
//     the buffer bounds check returns zeros for it and touches no memory -- zero padding costs nothing;
//   * the filter slice is copied as it lies in memory, [rows][row length] with the fragment reads indexing it
//     (forward: rows = output channels, (ci, tap) along the row; dgrad: rows = the gy channels of the chunk,
//     (output channel, tap) along the row, taps read flipped) -- no transposing store.
// Per-lane buffer offsets are computed once per tile; a chunk step moves the (wave-uniform) descriptor base.
// Preconditions (host): W % 4 == 0, 16-byte aligned x / w, channel counts % 4 == 0, tensors < 2 GiB.
template <class G> struct DPatch {
  static constexpr int PH = G::TH + 2, PWS = G::TW + 8, QR = PWS / 4;
  static constexpr int IMG = PH * PWS, RAW = G::NI * IMG;
  static constexpr int CIS = RAW + ((48 - RAW % 32) % 32);       // channel stride, == 16 (mod 32)
  static constexpr int CPC = CIS / 4;                            // 16-byte chunks per channel (pad chunks included)
  static_assert(CIS % 32 == 16 && CIS % 4 == 0, "channel stride");
  __device__ __forceinline__ static int pix(int p) {             // LDS offset (inside a channel) of pixel p's top-left tap
    const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
    return (img * PH + rem / G::TW) * PWS + rem % G::TW + 3;
  }
};
template <int CT, int CK, bool DGRAD> struct DFilt {
  static constexpr int ROWS = DGRAD ? CK : CT;
  static constexpr int ROWLEN = (DGRAD ? CT : CK) * 9;
  static constexpr int RS = ROWLEN + 4;                          // + one pad chunk: rows do not all start on one bank
  static constexpr int QW = RS / 4;
  static constexpr int SIZE = ROWS * RS;
};
constexpr uint32_t DMA_OOB = 0x80000000u;                        // >= any descriptor's num_records (tensors < 2 GiB)

// LDS regions filled by LDS-DMA are padded to whole pieces (64 chunks = one wave instruction): every lane of an issuing
// wave takes part -- lanes past the region's last chunk carry DMA_OOB and write zeros into the pad -- and whether a wave
// issues piece i at all is a SCALAR condition on the wave index.  An LDS-DMA must never sit under a per-lane condition:
// the compiler may merge such calls across the divergent branch (operands become per-lane phis), and the LDS base, which
// is wave-uniform by contract and taken from the first active lane, then belongs to another piece for part of the lanes
// (ROCm 7.2, conv_dma_kernel<G16, 32, 1, 8, fwd>: filter rows half loaded, the neighbouring buffer overwritten).
constexpr int dma_pad(int chunks) { return (chunks + 63) & ~63; }
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

#include "wino.h"

#ifdef TG_DIAG_STAMPS
// Diagnostic build (make -C tartangan_amd/csrc diag -> libtartangan_amd_diag.so, loaded with TG_LIBRARY=...): every workgroup
// of conv_dma_kernel leaves (shader cycles, 100 MHz ticks) of its main loop here; nothing else reads this buffer and no
// output depends on it.  tools/clock_probe.py turns it into the clock the chip holds under the kernel.
#define TG_DIAG_SLOTS 8192
__device__ unsigned long long tg_diag_stamps[2 * TG_DIAG_SLOTS];
extern "C" int tg_diag_read_stamps(unsigned long long* host, int n_slots) {
  if (n_slots > TG_DIAG_SLOTS) return -1;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tg_diag_stamps), sizeof(unsigned long long) * 2 * n_slots);
}
__device__ unsigned long long tg_diag_phases[16 * TG_DIAG_SLOTS];
extern "C" int tg_diag_read_phases(unsigned long long* host, int n_slots) {
  if (n_slots > TG_DIAG_SLOTS) return -1;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tg_diag_phases), sizeof(unsigned long long) * 16 * n_slots);
}
__device__ unsigned long long tg_diag_life[16 * TG_DIAG_SLOTS];
extern "C" int tg_diag_read_life(unsigned long long* host, int n_slots) {
  if (n_slots > TG_DIAG_SLOTS) return -1;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tg_diag_life), sizeof(unsigned long long) * 16 * n_slots);
}
extern "C" int tg_diag_clear_stamps() {
  static unsigned long long zeros[2 * TG_DIAG_SLOTS];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(tg_diag_stamps), zeros, sizeof(zeros));
}
#endif

template <class G, int MF, int MT, int CK, bool DGRAD, bool DB = true>
__global__ void __launch_bounds__(CT_THREADS)
conv_dma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                const float* __restrict__ residual, float* __restrict__ y, Shape s, int xcd_swizzle) {
  using P = DPatch<G>;
  constexpr int CT = MF * MT;
  using F = DFilt<CT, CK, DGRAD>;
  constexpr bool WK = (G::NPIX == 64);            // 64-pixel tiles: the waves share the pixels and split the k-groups of a chunk
  using Core = FwdCore<G, 3, MF, MT, WK>;
  constexpr int NT = Core::NT, KG = Core::KG, NG = CK / KG, NREG = Core::NREG;
  static_assert(!WK || NG % 4 == 0, "K-split needs a multiple of 4 k-groups per chunk");
  constexpr int PCH = CK * P::CPC, WCH = F::ROWS * F::QW;        // chunks per buffer: patch, filter
  constexpr int PCHP = dma_pad(PCH), WCHP = dma_pad(WCH);        // ... padded to whole wave pieces
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS, NVW = (WCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, WBUF = WCHP * 4;                // floats
  constexpr int BUF = PBUF + WBUF;
  constexpr int REDF = WK ? 4 * MT * NT * NREG * 64 : 0;         // cross-wave reduction of the K-split partial sums
  constexpr int NBUF = DB ? 2 : 1;     // DB = false: one buffer, load -> barrier -> MFMA per chunk; the overlap then comes from the
                                       // other workgroups of the CU (half the LDS: more of them, and exact occupancy rounds)
  __shared__ __attribute__((aligned(16))) float lds[(NBUF * BUF > REDF) ? NBUF * BUF : REDF];

#ifdef TG_DIAG_STAMPS
  const uint64_t st_begin = __builtin_amdgcn_s_memtime();
#endif
  // TG_DMA_PRIO=1 (off by default): the short serial sections (setup, DMA issue, barrier, epilogue) at wave priority 3, the
  // MFMA blocks at 0.  A young wave otherwise gets the vector issue port only when no older wave has an MFMA waiting
  // (stamps: 8 k cycles of setup for ~200 instructions on the 128^2 layers, 4 k with this) -- but the MFMA phases stretch by
  // what the serial sections gain: 0 ... +1.4 % over the layer set depending on the device, not a win worth a default.
  const bool prio = (xcd_swizzle & 32) != 0;
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wave = wave_index();
  const int j = lane % MF, h = lane / MF;
  int bid = blockIdx.x;
  if (xcd_swizzle & 1) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);     // an XCD (block id mod 8) works on neighbouring tiles
  const TileCoord tc = decode_tile<G>(bid, s.H, s.W);
  const int co0 = blockIdx.y * CT;
  const int pix0 = WK ? 0 : wave * G::PPW;
  const int g0 = WK ? wave : 0;                                  // WK: wave w takes k-groups w, w + 4, ...
  const uint32_t HW = (uint32_t)(s.H * s.W);

  // ---- per-lane buffer offsets (bytes), fixed for the whole tile
  uint32_t poff[NVP], woff[NVW];
#pragma unroll
  for (int i = 0; i < NVP; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int ci = e / P::CPC, rem = e % P::CPC;
    const int row = rem / P::QR, q = rem % P::QR;
    const int img = row / P::PH, r = row % P::PH;
    const int hh = tc.h0 + r - 1, ww = tc.w0 - 4 + 4 * q;
    const bool ok = (e < PCH) && (rem < P::RAW / 4) && (tc.b0 + img < s.B) && (hh >= 0) && (hh < s.H) && (ww >= 0) && (ww < s.W);
    poff[i] = ok ? (__umul24(__umul24(img, s.Cin) + ci, HW) + __umul24(hh, s.W) + ww) << 2 : DMA_OOB;
  }
  // filter: forward rows = output channels co0.., row = w[co][ci0 .. ci0+CK][9]; dgrad rows = the chunk's gy channels,
  // row = w[kc][co0 .. co0+CT][9] (w is the forward filter [s.Cin (gy channels)][s.Cout (gx channels)][9])
  const int row_floats = (DGRAD ? s.Cout : s.Cin) * 9;           // floats per row of w as stored
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int row = e / F::QW, q = e % F::QW;
    bool ok = (e < WCH) && (4 * q < F::ROWLEN);
    if (DGRAD) ok = ok && (4 * q < (s.Cout - co0) * 9);          // (columns past the last output channel)
    else ok = ok && (co0 + row < s.Cout);
    woff[i] = ok ? (uint32_t)(row * row_floats + 4 * q) << 2 : DMA_OOB;
  }

  // ---- wave-uniform descriptor bases; a chunk step advances them
  const char* xb = reinterpret_cast<const char*>(x) + ((int64_t)tc.b0 * s.Cin * HW) * 4;
  int64_t xbytes = (int64_t)(s.B - tc.b0) * s.Cin * HW * 4;
  const int64_t xstep = (int64_t)CK * HW * 4;
  const char* wb = reinterpret_cast<const char*>(w) + (DGRAD ? (int64_t)co0 * 9 * 4 : (int64_t)co0 * s.Cin * 9 * 4);
  int64_t wbytes = (int64_t)s.Cin * s.Cout * 9 * 4 - (wb - reinterpret_cast<const char*>(w));
  const int64_t wstep = DGRAD ? (int64_t)CK * s.Cout * 9 * 4 : (int64_t)CK * 9 * 4;

  // one 1-KiB piece (a wave instruction) of a chunk: pieces 0 .. NVP-1 the patch, NVP .. NVP+NVW-1 the filter slice
  constexpr int NPIECE = NVP + NVW;
  auto issue_piece = [&](int i, float* buf, const __amdgpu_buffer_rsrc_t rx, const __amdgpu_buffer_rsrc_t rw, int cvalid) {
    if (i < NVP) {
      uint32_t off = poff[i];
      if (cvalid < CK) off = ((i * CT_THREADS + (int)threadIdx.x) / P::CPC < cvalid) ? off : DMA_OOB;
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP)      // (scalar)
        dma16(rx, buf + (i * CT_THREADS + wave * 64) * 4, off);
    } else {
      i -= NVP;
      uint32_t off = woff[i];
      if (cvalid < CK) {
        const int e = i * CT_THREADS + (int)threadIdx.x;
        const bool in = DGRAD ? (e / F::QW < cvalid) : (4 * (e % F::QW) < cvalid * 9);
        off = in ? off : DMA_OOB;
      }
      if ((i + 1) * CT_THREADS <= WCHP || i * CT_THREADS + wave * 64 < WCHP)
        dma16(rw, buf + PBUF + (i * CT_THREADS + wave * 64) * 4, off);
    }
  };
  auto advance = [&]() {
    xb += xstep; xbytes -= xstep;
    wb += wstep; wbytes -= wstep;
  };
  auto issue = [&](int c0, float* buf) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, (int)wbytes, 0x00020000);
    const int cvalid = s.Cin - c0;                              // channels of this chunk that exist (>= CK: all)
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_piece(i, buf, rx, rw, cvalid);
    advance();
  };

  // ---- fragment addresses
  int lane_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) lane_b[n] = (h + g0 * KG) * P::CIS + P::pix(pix0 + n * MF + j);
  // forward: A[m = co][k = (ci, tap)] = wl[co * RS + ci * 9 + tap];  dgrad: A[m][k = (kc, tap)] = wl[kc * RS + m * 9 + 8 - tap]
  const int lane_a = DGRAD ? ((h + g0 * KG) * F::RS + j * 9 + 8) : (j * F::RS + (h + g0 * KG) * 9);

  typename Core::acc_t acc[MT][NT];
  Core::zero(acc);

#ifdef TG_DIAG_STAMPS
  asm volatile("" :: "v"(poff[0]), "v"(woff[0]), "v"(lane_b[0]), "v"(lane_a));
  const uint64_t st_setup = __builtin_amdgcn_s_memtime();
#endif
  if (DB) issue(0, lds);
  int buf = 0;
#ifdef TG_DIAG_STAMPS
  const uint64_t st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  uint64_t st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_prev = 0, st_last = 0;
#endif
  for (int c0 = 0; c0 < s.Cin; c0 += CK) {
    if constexpr (!DB) {
      __syncthreads();                                         // everybody is done reading the previous chunk
      issue(c0, lds);
    }
#ifdef TG_DIAG_STAMPS
    const uint64_t ph0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's DMAs of chunk c0 have landed ...
#ifdef TG_DIAG_STAMPS
    const uint64_t ph1 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                                           // ... and everybody's; everybody is done reading the other buffer
#ifdef TG_DIAG_STAMPS
    const uint64_t ph2 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (DB) {
      if (c0 + CK < s.Cin) issue(c0 + CK, lds + (buf ^ 1) * BUF);
    }
#ifdef TG_DIAG_STAMPS
    const uint64_t ph3 = __builtin_amdgcn_s_memtime();
    st_wait += ph1 - ph0; st_bar += ph2 - ph1; st_issue += ph3 - ph2; st_last = ph3;
    if (c0 > 0) st_mfma += ph0 - st_prev;
    st_prev = ph3;
#endif
    const float* pl = lds + buf * BUF;
    const float* wl = pl + PBUF;
#ifdef TG_DIAG_STAMPS
    if (xcd_swizzle & 4) { buf ^= 1; continue; }               // TG_DMA_DIAG=4 (stage isolation, wrong results): no MFMAs
#endif
    if (prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int gi = 0; gi < NG / (WK ? 4 : 1); ++gi) {
      const int g = gi * (WK ? 4 : 1);                           // (+ g0, in the lane bases)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap % 3;
        float a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          a[m] = DGRAD ? wl[lane_a + (g * KG) * F::RS + m * MF * 9 - tap] : wl[lane_a + m * MF * F::RS + (g * KG) * 9 + tap];
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = pl[lane_b[n] + (g * KG) * P::CIS + kh * P::PWS + kw];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (MF == 32)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
            else
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
          }
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    if (DB) buf ^= 1;
  }
#ifdef TG_DIAG_STAMPS
  {   // diagnostic build only (make diag): shader clock held over the main loop = d(memtime) / d(memrealtime) x 100 MHz
    const uint64_t st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && slot < TG_DIAG_SLOTS) { tg_diag_stamps[2 * slot] = st_c1 - st_c0; tg_diag_stamps[2 * slot + 1] = st_r1 - st_r0; }
    st_mfma += st_c1 - st_last;                                 // per wave: cycles in (vmcnt wait, barrier, DMA issue, MFMA phase)
    if (lane == 0 && slot < TG_DIAG_SLOTS) {
      unsigned long long* ph = tg_diag_phases + (slot * 4 + wave) * 4;
      ph[0] = st_wait; ph[1] = st_bar; ph[2] = st_issue; ph[3] = st_mfma;
    }
  }
#endif
  if constexpr (WK) {
    // sum the four waves' partial accumulators through LDS (fixed order), as conv_fwd_kernel does
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < NREG; ++r) red[(((wave * MT + m) * NT + n) * NREG + r) * 64 + lane] = acc[m][n][r];
    __syncthreads();
    constexpr int PW_ = MT * NT * NREG * 64;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const int e = ((m * NT + n) * NREG + r) * 64 + lane;
          acc[m][n][r] = (red[e] + red[PW_ + e]) + (red[2 * PW_ + e] + red[3 * PW_ + e]);
        }
  }
#ifdef TG_DIAG_STAMPS
  if ((xcd_swizzle & 2) && acc[0][0][0] != 12345.678f) return;   // TG_DMA_DIAG=2 (stage isolation, wrong results): no epilogue
#endif
  Core::epilogue(acc, bias, residual, y, s, tc, co0, pix0, j, h, wave);
#ifdef TG_DIAG_STAMPS
  {   // workgroup life: entry -> loop, loop -> stores issued, ... -> stores acknowledged
    const uint64_t e0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t e1 = __builtin_amdgcn_s_memtime();
    const int slot = blockIdx.y * gridDim.x + blockIdx.x;
    if (lane == 0 && slot < TG_DIAG_SLOTS) {
      unsigned long long* ph = tg_diag_life + (slot * 4 + wave) * 4;
      ph[0] = st_c0 - st_begin; ph[1] = e0 - st_c0; ph[2] = e1 - e0; ph[3] = st_setup - st_begin;
    }
  }
#endif
}

// =========================================================================== 1x1, few channels: direct kernel
// The 1x1 layers with at most 8 output channels (to-RGB 16->3, the attention's 32->4 maps, dgrad of from-RGB) are pure
// streaming: a few FMAs per loaded float.  The tiled MFMA kernel above spends their whole (short) life on its
// patch staging and barriers; here a thread owns PX consecutive pixels of one image, reads each input channel once
// (float4 per lane, 1 KiB per wave-load), keeps all Cout accumulators in registers and reads the filter as LDS
// broadcasts.  Same kernel for dgrad with the filter indexed transposed.
template <int CO, int PX>
__global__ void __launch_bounds__(256)
conv1x1_direct_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                      const float* __restrict__ residual, float* __restrict__ y, int Cin, int Cout, int HW, int w_so, int w_si) {
  __shared__ __attribute__((aligned(16))) float wl[1024];                  // [ci][CO]; the host checks Cin * CO <= 1024
  for (int e = threadIdx.x; e < Cin * CO; e += 256) {
    const int ci = e / CO, co = e - ci * CO;
    wl[e] = co < Cout ? w[co * w_so + ci * w_si] : 0.f;
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int p = (blockIdx.x * 256 + threadIdx.x) * PX;
  if (p >= HW) return;
  const float* xb = x + (int64_t)b * Cin * HW + p;
  float acc[CO][PX];
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int u = 0; u < PX; ++u) acc[co][u] = 0.f;
#pragma unroll 4
  for (int ci = 0; ci < Cin; ++ci) {
    float xv[PX];
    if (PX == 4) {
      const float4 t = *reinterpret_cast<const float4*>(xb + (int64_t)ci * HW);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
      xv[0] = xb[(int64_t)ci * HW];
    }
#pragma unroll
    for (int c4 = 0; c4 < CO / 4; ++c4) {
      const float4 wv = *reinterpret_cast<const float4*>(&wl[ci * CO + 4 * c4]);     // same address in every lane: broadcast
      const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < PX; ++u) acc[4 * c4 + k][u] = fmaf(ww[k], xv[u], acc[4 * c4 + k][u]);
    }
  }
  const int64_t ob = (int64_t)b * Cout * HW + p;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (co >= Cout) break;
    const float bv = bias ? bias[co] : 0.f;
    float o[PX];
#pragma unroll
    for (int u = 0; u < PX; ++u) o[u] = acc[co][u] + bv;
    const int64_t off = ob + (int64_t)co * HW;
    if (PX == 4) {
      if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + off);
        o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
      }
      *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      if (residual) o[0] += residual[off];
      y[off] = o[0];
    }
  }
}

template <int CO>
static int launch_conv1x1_direct(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s,
                                 bool dgrad, hipStream_t st) {
  const int HW = s.H * s.W;
  // filter element (out channel o, in channel i): forward w[o][i]; dgrad reads the forward filter [Cout_fwd = Cin here][Cin_fwd = Cout here]
  const int w_so = dgrad ? 1 : s.Cin, w_si = dgrad ? s.Cout : 1;
  const bool vec = (HW % 4 == 0) && tg_aligned16(x) && tg_aligned16(y) && (!residual || tg_aligned16(residual));
  if (vec) {
    dim3 grid((HW / 4 + 255) / 256, s.B);
    conv1x1_direct_kernel<CO, 4><<<grid, 256, 0, st>>>(x, w, bias, residual, y, s.Cin, s.Cout, HW, w_so, w_si);
  } else {
    dim3 grid((HW + 255) / 256, s.B);
    conv1x1_direct_kernel<CO, 1><<<grid, 256, 0, st>>>(x, w, bias, residual, y, s.Cin, s.Cout, HW, w_so, w_si);
  }
  return tg_launch_status();
}
static inline int conv1x1_co(int Cout) { return Cout <= 4 ? 4 : 8; }
// Measured at batch 64: with <= 8 output channels it halves the time (16->3 @128^2: 13 us vs 29 us, 32->4 @64^2: 8 vs
// 11 us); from 16 output channels up the MFMA kernel is faster (3->16 @128^2: 30 us vs 41 us), so it keeps those.
static inline bool conv1x1_direct_ok(const Shape& s) { return s.Cout <= 8 && s.Cin * conv1x1_co(s.Cout) <= 1024 && s.B <= 65535; }
static int launch_conv1x1(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, bool dgrad,
                          hipStream_t st) {
  if (s.Cout <= 4) return launch_conv1x1_direct<4>(x, w, bias, residual, y, s, dgrad, st);
  return launch_conv1x1_direct<8>(x, w, bias, residual, y, s, dgrad, st);
}

// =========================================================================== 1x1, streaming MFMA kernel (no LDS)
// A 1x1 convolution moves 4 (Cin + Cout) bytes per pixel for 2 Cin Cout FLOP: every 1x1 layer of the step is
// bandwidth- or latency-bound, and the tiled kernel above (patch staging through LDS, barriers, a tile decode per
// workgroup) gave them 10-16 us each however small the tensor.  Here nothing is staged: a wave owns 64 consecutive
// pixels (16 lanes x one float4; the flattened (image, pixel / 4) space, so small planes waste nothing) and ALL output
// channels in 16-channel blocks.  Per k-step of 4 input channels a lane loads ONE float4 of x (lane group h = channel
// k0 + h, lanes j = 16 neighbouring pixel quads: 1 KiB per wave-load, rows of 256 contiguous bytes) and feeds its four
// components to four 16x16x4 MFMAs as the B operand -- MFMA r covers the pixels {4 j + r}; the A operand is the filter
// element (co = 16 m + j, ci = k0 + h), straight from L1.  With lane (j, h) holding D[row 4h + i][col j], the four
// results of a lane for output channel 16 m + 4 h + i are the pixels 4 j .. 4 j + 3: ONE 16-byte store (and residual
// load).  No LDS, no barrier, no divergence; the input-gradient form is the same kernel with the filter read transposed.
// Channel segments: the channels of the input and / or the output may be spread over up to three tensors (SelfAttention2d's
// theta / phi / g projections read ONE x: their outputs are three tensors, the inputs of their joint input gradient likewise).
// Channel c of a segmented operand lives in tensor k with start[k] <= c < start[k + 1], as its channel c - start[k].
struct ChanSegs {
  float* p[3];
  int start[4];                     // start[0] = 0, start[n] = all channels; unused entries = start[n]
  __device__ __forceinline__ int find(int c) const { return c >= start[2] ? 2 : (c >= start[1] ? 1 : 0); }
  __device__ __forceinline__ int width(int k) const { return start[k + 1] - start[k]; }
};
static inline ChanSegs one_seg(const float* p, int channels) {
  ChanSegs g;
  g.p[0] = g.p[1] = g.p[2] = const_cast<float*>(p);
  g.start[0] = 0; g.start[1] = g.start[2] = g.start[3] = channels;
  return g;
}
static inline ChanSegs three_segs(const float* p0, const float* p1, const float* p2, int c0, int c1, int c2) {
  ChanSegs g;
  g.p[0] = const_cast<float*>(p0); g.p[1] = const_cast<float*>(p1); g.p[2] = const_cast<float*>(p2);
  g.start[0] = 0; g.start[1] = c0; g.start[2] = c0 + c1; g.start[3] = c0 + c1 + c2;
  return g;
}

template <int NCO>
__global__ void __launch_bounds__(256)
conv1x1_mfma_kernel(ChanSegs xs, const float* __restrict__ w, const float* __restrict__ bias,
                    const float* __restrict__ residual, ChanSegs ys, int Cin, int Cout, int HW, int64_t quads,
                    int w_so, int w_si) {
  const int lane = threadIdx.x & 63, j = lane & 15, h = lane >> 4;
  const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + j;      // this lane's pixel quad
  const bool live = q < quads;
  const int qpp = HW >> 2;                                                          // quads per plane
  const int64_t b = live ? q / qpp : 0;
  const int p = live ? (int)(q - b * qpp) * 4 : 0;
  f32x4 acc[NCO][4];
#pragma unroll
  for (int m = 0; m < NCO; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  // filter rows of this lane: co = 16 m + j (clamped: rows past Cout multiply into accumulators that are never stored)
  int wrow[NCO];
#pragma unroll
  for (int m = 0; m < NCO; ++m) wrow[m] = min(16 * m + j, Cout - 1) * w_so;
  constexpr int U = 4;                                                              // k-steps whose loads are in flight together
  for (int k0 = 0; k0 < Cin; k0 += 4 * U) {
    float4 xv[U];
    float a[U][NCO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ci = k0 + 4 * u + h;
      const bool ok = live && ci < Cin;
      const int cc = min(ci, Cin - 1);
      const int k = xs.find(cc);
      const float* src = xs.p[k] + ((b * xs.width(k) + (cc - xs.start[k])) * HW + p);
      xv[u] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < NCO; ++m) a[u][m] = (ci < Cin) ? w[wrow[m] + cc * w_si] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float bx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
      for (int m = 0; m < NCO; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], bx[r], acc[m][r], 0, 0, 0);
    }
  }
  if (!live) return;
  const float* rb = residual ? residual + (b * Cout) * HW + p : nullptr;
#pragma unroll
  for (int m = 0; m < NCO; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 16 * m + 4 * h + i;
      if (co >= Cout) continue;
      const float bv = bias ? bias[co] : 0.f;
      float4 o = make_float4(acc[m][0][i] + bv, acc[m][1][i] + bv, acc[m][2][i] + bv, acc[m][3][i] + bv);
      if (rb) {
        const float4 t = *reinterpret_cast<const float4*>(rb + (int64_t)co * HW);
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      }
      const int k = ys.find(co);
      *reinterpret_cast<float4*>(ys.p[k] + ((b * ys.width(k) + (co - ys.start[k])) * HW + p)) = o;
    }
}

// planes of whole pixel quads, 16-byte aligned tensors, at most 128 output channels (8 accumulator blocks)
static inline bool conv1x1_mfma_ok(const float* x, const float* residual, const float* y, const Shape& s) {
  return (s.H * s.W) % 4 == 0 && s.Cout <= 128 && tg_aligned16(x) && tg_aligned16(y) && (!residual || tg_aligned16(residual));
}
static int launch_conv1x1_segs(ChanSegs xs, const float* w, const float* bias, const float* residual, ChanSegs ys, Shape s, bool dgrad,
                               hipStream_t st) {
  const int HW = s.H * s.W;
  const int64_t quads = (int64_t)s.B * (HW / 4);
  const int w_so = dgrad ? 1 : s.Cin, w_si = dgrad ? s.Cout : 1;      // (as launch_conv1x1_direct)
  const unsigned grid = (unsigned)((quads + 63) / 64);                 // 4 waves x 16 quads per workgroup
  const int nco = (s.Cout + 15) / 16;
#define TG_1X1(N) conv1x1_mfma_kernel<N><<<grid, 256, 0, st>>>(xs, w, bias, residual, ys, s.Cin, s.Cout, HW, quads, w_so, w_si)
  if (nco <= 1) TG_1X1(1);
  else if (nco == 2) TG_1X1(2);
  else if (nco <= 4) TG_1X1(4);
  else TG_1X1(8);
#undef TG_1X1
  return tg_launch_status();
}
static int launch_conv1x1_mfma(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, bool dgrad,
                               hipStream_t st) {
  return launch_conv1x1_segs(one_seg(x, s.Cin), w, bias, residual, one_seg(y, s.Cout), s, dgrad, st);
}

// =========================================================================== wgrad
template <int KS> struct WgCfg {
  static constexpr int CKW = (KS == 3) ? 16 : 32;      // input channels per workgroup
  static constexpr int KK = KS * KS;
  static constexpr int NCOL = CKW * KK;                // (ci, tap) columns
  static constexpr int NT = (NCOL + 15) / 16;
};
constexpr int WG_GYS = 256 + 4;    // gy tile row stride (multiple of 4 for ds_write_b128; %32 == 4: 2-way at worst on A reads)

// gy tile stager: gy[b0.., co0..co0+CT, h0.., w0..] -> gl[co][p], p = (img*TH + r)*TW + c
template <class G, int CT>
struct GyStager {
  static constexpr int Q = G::TW / 4;
  static constexpr int ROWS_PER_CO = G::NI * G::TH;
  static constexpr int ROWS = CT * ROWS_PER_CO;
  static constexpr int NV = (ROWS * Q + CT_THREADS - 1) / CT_THREADS;
  float4 v[NV];

  __device__ __forceinline__ void load(const float* __restrict__ gy, const Shape& s, int co0, const TileCoord& tc, bool vec) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < ROWS * Q) {
        const int q = e % Q, row = e / Q;
        const int r = row % G::TH;
        const int t = row / G::TH;
        const int img = t % G::NI, co = t / G::NI;
        const int b = tc.b0 + img, hh = tc.h0 + r, ww = tc.w0 + 4 * q;
        if (b < s.B && co0 + co < s.Cout && hh < s.H && ww < s.W) {
          // wave-uniform tile base + 32-bit lane offset (see PatchStager)
          const float* base = gy + ((((int64_t)tc.b0 * s.Cout + co0) * s.H + tc.h0) * s.W + tc.w0);
          const float* src = base + (__umul24(__umul24(img, s.Cout) + co, s.H * s.W) + __umul24(r, s.W) + 4 * q);
          if (vec) {
            val = *reinterpret_cast<const float4*>(src);
          } else {
            val.x = src[0];
            if (ww + 1 < s.W) val.y = src[1];
            if (ww + 2 < s.W) val.z = src[2];
            if (ww + 3 < s.W) val.w = src[3];
          }
        }
      }
      v[i] = val;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ gl) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      if (e < ROWS * Q) {
        const int q = e % Q, row = e / Q;
        const int co = row / ROWS_PER_CO, prow = row % ROWS_PER_CO;
        *reinterpret_cast<float4*>(gl + co * WG_GYS + prow * G::TW + 4 * q) = v[i];
      }
    }
  }

  // bias gradient for free: a 256-pixel tile is 64 float4 per channel, so register i of wave w holds
  // nothing but channel 4*i + w -- one wavefront reduction per register adds the tile's sum_p gy[co][p].
  static_assert(ROWS_PER_CO * Q == 64, "one channel of a gy tile = one wave's worth of float4");
  __device__ __forceinline__ void add_channel_sums(float* acc /*[NV]*/) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] += wave_sum((v[i].x + v[i].y) + (v[i].z + v[i].w));
  }
};

template <class G, int KS, int MTW>
__global__ void __launch_bounds__(CT_THREADS)
conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part,
                  float* __restrict__ bias_part /*nullable: [S][Cout]*/, Shape s, int ntiles, int S, int vec_x, int vec_gy) {
  using P = Patch<G, KS>;
  using C = WgCfg<KS>;
  constexpr int CKW = C::CKW, KK = C::KK, NT = C::NT, CT = 16 * MTW;
  constexpr int PATCH = CKW * P::CIS, GYT = CT * WG_GYS;
  constexpr int RED = 4 * MTW * NT * 4 * 64;           // cross-wave reduction: 4 waves x (MTW*NT*4 regs) x 64 lanes
  constexpr int LDS_FLOATS = (PATCH + GYT) > RED ? (PATCH + GYT) : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* pl = lds;
  float* gl = lds + PATCH;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, h = lane >> 4;
  const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CKW;
  const int split = blockIdx.x;

  // B operand: column j of N-tile n -> (ci, tap) -> patch offset; plus this lane's pixel within the k-group
  int colbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n * 16 + j;
    const int ci = col / KK, tap = col % KK;
    colbase[n] = (col < C::NCOL) ? (ci * P::CIS + (tap / KS) * P::PWS + (tap % KS)) : 0;
  }
  const int lane_b = pix_off<G, KS>(wave * 64) + h;                // + const(g) below
  const int lane_a = j * WG_GYS + wave * 64 + h;                    // + m*16*GYS + 4g

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  PatchStager<G, KS, CKW> ps;
  GyStager<G, CT> gs;
  const bool do_bias = (bias_part != nullptr) && (blockIdx.z == 0);
  float bsum[GyStager<G, CT>::NV];
#pragma unroll
  for (int i = 0; i < GyStager<G, CT>::NV; ++i) bsum[i] = 0.f;
  if (split < ntiles) {
    const TileCoord tc = decode_tile<G>(split, s.H, s.W);
    ps.load(x, s, s.Cin, ci0, tc, vec_x);
    gs.load(gy, s, co0, tc, vec_gy);
  }
  for (int t = split; t < ntiles; t += S) {
    __syncthreads();
    ps.store(pl);
    gs.store(gl);
    if (do_bias) gs.add_channel_sums(bsum);
    __syncthreads();
    if (t + S < ntiles) {
      const TileCoord tn = decode_tile<G>(t + S, s.H, s.W);
      ps.load(x, s, s.Cin, ci0, tn, vec_x);
      gs.load(gy, s, co0, tn, vec_gy);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      // pixel group 64*wave + 4g .. +3: same row (TW % 4 == 0); offset relative to the wave's first pixel
      constexpr int PPI = G::TH * G::TW;
      const int pg = 4 * g;                                   // < 64
      const int goff = ((pg / PPI) * P::PH + (pg % PPI) / G::TW) * P::PWS + (pg % G::TW);
      float a[MTW], b[NT];
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[m] = gl[lane_a + m * 16 * WG_GYS + 4 * g];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = pl[colbase[n] + lane_b + goff];
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
    }
  }

  if (do_bias && lane == 0) {
#pragma unroll
    for (int i = 0; i < GyStager<G, CT>::NV; ++i) {
      const int co = co0 + 4 * i + wave;
      if (4 * i + wave < CT && co < s.Cout) bias_part[(int64_t)split * s.Cout + co] = bsum[i];
    }
  }
  // cross-wave reduction through LDS (fixed order), then one partial write per workgroup
  __syncthreads();
  float* red = lds;                                            // [wave][m][n][r][lane]
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * MTW + m) * NT + n) * 4 + r) * 64 + lane] = acc[m][n][r];
  __syncthreads();
  constexpr int PER_WAVE = MTW * NT * 4 * 64;
  for (int e = threadIdx.x; e < PER_WAVE; e += CT_THREADS) {
    const float v = (red[e] + red[PER_WAVE + e]) + (red[2 * PER_WAVE + e] + red[3 * PER_WAVE + e]);
    const int l = e & 63, q = e >> 6;
    const int r = q & 3, n = (q >> 2) % NT, m = (q >> 2) / NT;
    const int col = n * 16 + (l & 15);
    const int co = co0 + m * 16 + (l >> 4) * 4 + r;
    const int ci = ci0 + col / KK, tap = col % KK;
    if (col < C::NCOL && co < s.Cout && ci < s.Cin)
      part[(((int64_t)split * s.Cout + co) * s.Cin + ci) * KK + tap] = v;
  }
}

// ---- the same weight-gradient kernel with LDS-DMA staging (see conv_dma_kernel): the patch and the gy tile of a pixel
// tile go global -> LDS by buffer_load ... lds, nothing passes through VGPRs, no per-element address / bounds code.
// Same work split (split, co tile, ci chunk), same accumulation order and the same partial-sum layout as
// conv_wgrad_kernel, so the two are bit-identical and share the plan, the workspace and the reduce kernels.
// DB: two LDS buffers, the next pixel tile's DMA flies under this tile's MFMAs (one barrier per tile); used where a
// workgroup walks several tiles and few workgroups share a CU.  The bias gradient is the sum of the A fragments
// (every gy value of the tile passes through them exactly once).
constexpr int WGD_GYQ = 65;                     // chunks per gy row: 256 pixels + one pad chunk (== WG_GYS / 4)
template <class G, int MTW, int CKW, bool DB>
__global__ void __launch_bounds__(CT_THREADS)
conv_wgrad_dma_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part,
                      float* __restrict__ bias_part /*nullable: [S][Cout]*/, Shape s, int ntiles, int S) {
  static_assert(G::NPIX == 256 && WG_GYS == 4 * WGD_GYQ, "256-pixel tiles");
  using P = DPatch<G>;
  constexpr int KK = 9, CT = 16 * MTW, NCOL = CKW * KK, NT = (NCOL + 15) / 16;
  constexpr int PCH = CKW * P::CPC, GCH = CT * WGD_GYQ;
  constexpr int PCHP = dma_pad(PCH), GCHP = dma_pad(GCH);        // regions of whole wave pieces (see dma_pad)
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS, NVG = (GCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, BUF = PBUF + GCHP * 4;
  constexpr int RED = 4 * MTW * NT * 4 * 64 + 4 * CT;
  constexpr int LDSF = ((DB ? 2 : 1) * BUF > RED) ? (DB ? 2 : 1) * BUF : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];

  const int lane = threadIdx.x & 63, wave = wave_index();
  if (s.prio) __builtin_amdgcn_s_setprio(3);
  const int j = lane & 15, h = lane >> 4;
  const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CKW;
  const int split = blockIdx.x;
  const uint32_t HW = (uint32_t)(s.H * s.W);

  int colbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n * 16 + j;
    const int ci = col / KK, tap = col % KK;
    colbase[n] = (col < NCOL) ? (ci * P::CIS + (tap / 3) * P::PWS + (tap % 3)) : 0;
  }
  const int lane_b = P::pix(wave * 64) + h;
  const int lane_a = j * WG_GYS + wave * 64 + h;

  f32x4 acc[MTW][NT];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = (bias_part != nullptr) && (blockIdx.z == 0);
  float bsum[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) bsum[m] = 0.f;

  const int cvalid = min(CKW, s.Cin - ci0);
  auto issue = [&](int t, float* buf) {
    const TileCoord tc = decode_tile<G>(t, s.H, s.W);
    const char* xb = reinterpret_cast<const char*>(x) + ((int64_t)tc.b0 * s.Cin + ci0) * HW * 4;
    const int64_t xbytes = ((int64_t)(s.B - tc.b0) * s.Cin - ci0) * HW * 4;
    const char* gb = reinterpret_cast<const char*>(gy) + ((int64_t)tc.b0 * s.Cout + co0) * HW * 4;
    const int64_t gbytes = ((int64_t)(s.B - tc.b0) * s.Cout - co0) * HW * 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gb), 0, (int)gbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
      const int e = i * CT_THREADS + threadIdx.x;
      const int ci = e / P::CPC, rem = e % P::CPC;
      const int row = rem / P::QR, q = rem % P::QR;
      const int img = row / P::PH, r = row % P::PH;
      const int hh = tc.h0 + r - 1, ww = tc.w0 - 4 + 4 * q;
      const bool ok = (rem < P::RAW / 4) && (ci < cvalid) && (tc.b0 + img < s.B) && (hh >= 0) && (hh < s.H) && (ww >= 0) && (ww < s.W);
      const uint32_t off = ok ? (__umul24(__umul24(img, s.Cin) + ci, HW) + __umul24(hh, s.W) + ww) << 2 : DMA_OOB;
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP) dma16(rx, buf + (i * CT_THREADS + wave * 64) * 4, off);
    }
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
      const int e = i * CT_THREADS + threadIdx.x;
      const int co = e / WGD_GYQ, q = e % WGD_GYQ;
      const int p = 4 * q;
      const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
      const int hh = tc.h0 + rem / G::TW, ww = tc.w0 + rem % G::TW;
      const bool ok = (e < GCH) && (q < 64) && (co0 + co < s.Cout) && (tc.b0 + img < s.B) && (hh < s.H) && (ww < s.W);
      const uint32_t off = ok ? (__umul24(__umul24(img, s.Cout) + co, HW) + __umul24(hh, s.W) + ww) << 2 : DMA_OOB;
      if ((i + 1) * CT_THREADS <= GCHP || i * CT_THREADS + wave * 64 < GCHP) dma16(rg, buf + PBUF + (i * CT_THREADS + wave * 64) * 4, off);
    }
  };
  auto compute = [&](const float* buf) {
    const float* pl = buf;
    const float* gl = buf + PBUF;
    if (s.prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      constexpr int PPI = G::TH * G::TW;
      const int pg = 4 * g;
      const int goff = ((pg / PPI) * P::PH + (pg % PPI) / G::TW) * P::PWS + (pg % G::TW);
      float a[MTW], b[NT];
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[m] = gl[lane_a + m * 16 * WG_GYS + 4 * g];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = pl[colbase[n] + lane_b + goff];
#pragma unroll
      for (int m = 0; m < MTW; ++m) bsum[m] += a[m];
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    if (s.prio) __builtin_amdgcn_s_setprio(3);
  };

  if constexpr (DB) {
    if (split < ntiles) issue(split, lds);
    int buf = 0;
    for (int t = split; t < ntiles; t += S) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + S < ntiles) issue(t + S, lds + (buf ^ 1) * BUF);
      compute(lds + buf * BUF);
      buf ^= 1;
    }
  } else {
    for (int t = split; t < ntiles; t += S) {
      __syncthreads();                          // everybody is done reading the previous tile
      issue(t, lds);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(lds);
    }
  }

  // cross-wave reduction through LDS (fixed order), one partial per workgroup -- as conv_wgrad_kernel
  __syncthreads();
  float* red = lds;
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(((wave * MTW + m) * NT + n) * 4 + r) * 64 + lane] = acc[m][n][r];
  constexpr int PER_WAVE = MTW * NT * 4 * 64;
  float* bred = lds + 4 * PER_WAVE;               // [wave][CT]
  if (do_bias) {
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      float v = bsum[m];                          // lane (j = channel, h): this lane's pixels; add the four h groups
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (h == 0) bred[wave * CT + m * 16 + j] = v;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < PER_WAVE; e += CT_THREADS) {
    const float v = (red[e] + red[PER_WAVE + e]) + (red[2 * PER_WAVE + e] + red[3 * PER_WAVE + e]);
    const int l = e & 63, q = e >> 6;
    const int r = q & 3, n = (q >> 2) % NT, m = (q >> 2) / NT;
    const int col = n * 16 + (l & 15);
    const int co = co0 + m * 16 + (l >> 4) * 4 + r;
    const int ci = ci0 + col / KK, tap = col % KK;
    if (col < NCOL && co < s.Cout && ci < s.Cin)
      part[(((int64_t)split * s.Cout + co) * s.Cin + ci) * KK + tap] = v;
  }
  if (do_bias && threadIdx.x < CT && co0 + (int)threadIdx.x < s.Cout) {
    const int c = threadIdx.x;
    bias_part[(int64_t)split * s.Cout + co0 + c] = (bred[c] + bred[CT + c]) + (bred[2 * CT + c] + bred[3 * CT + c]);
  }
}

// ---- 1x1 weight gradient, streaming (no operand staging): gw[co][ci] = sum over (image, pixel) of gy[co][p] x[ci][p] is a
// contraction over pixels with both operands read exactly once -- bandwidth-bound, and the tiled kernel above gave each of
// the step's two dozen 1x1 layers ~15 us.  Here a wave walks its share of 16-pixel groups; per group a lane loads ONE float4
// per operand row (lane j = row of the 16-channel block, lane group h = pixels 4h .. 4h+3 of the group: 16 rows x 64
// contiguous bytes per wave-load, consecutive groups continue the same rows) and the four components feed four 16x16x4
// MFMAs (k-step r contracts the pixels {4h + r}).  Per-workgroup partials in the layout of conv_wgrad_kernel (same plan,
// same reduce kernels, fixed summation order => deterministic); the bias gradient is the sum of the A operands.
template <int MT, int NT>
__global__ void __launch_bounds__(256)
conv1x1_wgrad_stream_kernel(const float* __restrict__ x, ChanSegs gys, float* __restrict__ part,
                            float* __restrict__ bias_part /*nullable: [S][Cout]*/, int Cin, int Cout, int HW, int64_t groups, int S) {
  __shared__ float red[4 * MT * NT * 4 * 64 + 4 * 16 * MT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, h = lane >> 4;
  const int co0 = blockIdx.y * 16 * MT, ci0 = blockIdx.z * 16 * NT, split = blockIdx.x;
  const int64_t g_begin = groups * split / S, g_end = groups * (split + 1) / S;
  const int64_t n = g_end - g_begin;
  const int64_t w_begin = g_begin + n * wave / 4, w_end = g_begin + n * (wave + 1) / 4;
  const int gpp = HW >> 4;                                   // 16-pixel groups per plane
  int64_t arow[MT], brow[NT], aimg[MT];                      // row offsets (clamped: rows past the tensor are never written)
  const float* abase[MT];                                    // gy rows may come from up to three tensors (ChanSegs)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int co = min(co0 + 16 * m + j, Cout - 1), k = gys.find(co);
    abase[m] = gys.p[k];
    aimg[m] = (int64_t)gys.width(k) * HW;
    arow[m] = (int64_t)(co - gys.start[k]) * HW + 4 * h;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) brow[t] = (int64_t)min(ci0 + 16 * t + j, Cin - 1) * HW + 4 * h;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) bsum[m] = 0.f;
  constexpr int U = 4;                                       // groups whose loads are in flight together
  for (int64_t g0 = w_begin; g0 < w_end; g0 += U) {
    float4 ga[U][MT], xb[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t g = g0 + u;
      const bool ok = g < w_end;
      const int64_t b = ok ? g / gpp : 0;
      const int64_t p = ok ? (g - b * gpp) * 16 : 0;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        ga[u][m] = ok ? *reinterpret_cast<const float4*>(abase[m] + b * aimg[m] + arow[m] + p) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        xb[u][t] = ok ? *reinterpret_cast<const float4*>(x + b * Cin * HW + brow[t] + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int m = 0; m < MT; ++m) bsum[m] += (ga[u][m].x + ga[u][m].y) + (ga[u][m].z + ga[u][m].w);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float av[4] = {ga[u][m].x, ga[u][m].y, ga[u][m].z, ga[u][m].w};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv[4] = {xb[u][t].x, xb[u][t].y, xb[u][t].z, xb[u][t].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[r], acc[m][t], 0, 0, 0);
        }
      }
    }
  }
  // cross-wave sums through LDS, fixed order
  constexpr int PER_WAVE = MT * NT * 4 * 64;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave * PER_WAVE + ((m * NT + t) * 4 + i) * 64 + lane] = acc[m][t][i];
  float* bred = red + 4 * PER_WAVE;                          // [wave][16 MT]
  const bool do_bias = bias_part != nullptr && blockIdx.z == 0;
  if (do_bias) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v = bsum[m];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (h == 0) bred[wave * 16 * MT + 16 * m + j] = v;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < PER_WAVE; e += 256) {
    const float v = (red[e] + red[PER_WAVE + e]) + (red[2 * PER_WAVE + e] + red[3 * PER_WAVE + e]);
    const int l = e & 63, q = e >> 6;
    const int i = q & 3, t = (q >> 2) % NT, m = (q >> 2) / NT;
    const int co = co0 + 16 * m + 4 * (l >> 4) + i, ci = ci0 + 16 * t + (l & 15);      // lane (j, h): D[row 4h + i][col j]
    if (co < Cout && ci < Cin) part[((int64_t)split * Cout + co) * Cin + ci] = v;
  }
  if (do_bias && threadIdx.x < 16 * MT && co0 + (int)threadIdx.x < Cout) {
    const int c = threadIdx.x;
    bias_part[(int64_t)split * Cout + co0 + c] = (bred[c] + bred[16 * MT + c]) + (bred[2 * 16 * MT + c] + bred[3 * 16 * MT + c]);
  }
}

static int conv1x1_mode();
static inline bool conv1x1_wgrad_stream_ok(const float* x, const float* gy, const Shape& s) {
  return conv1x1_mode() != 0 && (s.H * s.W) % 16 == 0 && tg_aligned16(x) && tg_aligned16(gy);
}
static int launch_conv1x1_wgrad_segs(const float* x, ChanSegs gy, float* part, float* bias_part, Shape s, int S, hipStream_t st) {
  const int HW = s.H * s.W;
  const int64_t groups = (int64_t)s.B * (HW / 16);
  const int mt = s.Cout > 16 ? 2 : 1, nt = s.Cin > 16 ? 2 : 1;
  dim3 grid(S, (s.Cout + 16 * mt - 1) / (16 * mt), (s.Cin + 16 * nt - 1) / (16 * nt));
  if (mt == 1 && nt == 1) conv1x1_wgrad_stream_kernel<1, 1><<<grid, 256, 0, st>>>(x, gy, part, bias_part, s.Cin, s.Cout, HW, groups, S);
  else if (mt == 1) conv1x1_wgrad_stream_kernel<1, 2><<<grid, 256, 0, st>>>(x, gy, part, bias_part, s.Cin, s.Cout, HW, groups, S);
  else if (nt == 1) conv1x1_wgrad_stream_kernel<2, 1><<<grid, 256, 0, st>>>(x, gy, part, bias_part, s.Cin, s.Cout, HW, groups, S);
  else conv1x1_wgrad_stream_kernel<2, 2><<<grid, 256, 0, st>>>(x, gy, part, bias_part, s.Cin, s.Cout, HW, groups, S);
  return tg_launch_status();
}
static int launch_conv1x1_wgrad_stream(const float* x, const float* gy, float* part, float* bias_part, Shape s, int S, hipStream_t st) {
  return launch_conv1x1_wgrad_segs(x, one_seg(gy, s.Cout), part, bias_part, s, S, st);
}

// gw[e] (+)= sum_s part[s][e], fixed order.  64 consecutive e per workgroup (coalesced), the S partials
// are split over the 4 waves and combined through LDS: no serial chain of S dependent loads.
// Workgroups past ceil(E/64) reduce the bias partials the same way.
__device__ __forceinline__ void wgrad_reduce_block(int blk, const float* __restrict__ part, float* __restrict__ gw, int64_t E, int S,
                                                   const float* __restrict__ bias_part, float* __restrict__ gbias, int Cout,
                                                   int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int eblocks = (int)((E + 63) / 64);
  const float* src = part;
  float* dst = gw;
  int64_t n = E, e = blk * 64ll + lane;
  if (blk >= eblocks) {
    src = bias_part; dst = gbias; n = Cout;
    e = (blk - eblocks) * 64ll + lane;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (e < n) {
    int sidx = wave;
    for (; sidx + 12 < S; sidx += 16) {
      a0 += src[(int64_t)sidx * n + e];
      a1 += src[(int64_t)(sidx + 4) * n + e];
      a2 += src[(int64_t)(sidx + 8) * n + e];
      a3 += src[(int64_t)(sidx + 12) * n + e];
    }
    for (; sidx < S; sidx += 4) a0 += src[(int64_t)sidx * n + e];
  }
  red[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && e < n) {
    const float r = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    dst[e] = accumulate ? dst[e] + r : r;
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int64_t E, int S,
                                                           const float* __restrict__ bias_part, float* __restrict__ gbias, int Cout,
                                                           int accumulate) {
  wgrad_reduce_block(blockIdx.x, part, gw, E, S, bias_part, gbias, Cout, accumulate);
}

// The same reduction for MANY layers in one launch (a whole backward pass' worth of partials): the per-layer
// reduce is ~5 us of launch latency for ~1 us of work, ~100 times per training step.  Items travel in the kernel
// argument block (no table in memory to keep alive or copy), a workgroup finds its item by scanning <= MAX_ITEMS
// block offsets.
constexpr int RB_MAX_ITEMS = 40;
struct ReduceItem {
  const float* part; float* gw; float* gbias;   // gbias null: no bias gradient; bias partials sit at part + S*E
  int64_t E;
  int S, Cout, accumulate, pad;
};
struct ReduceBatch {
  int n;
  int block_end[RB_MAX_ITEMS];                  // exclusive prefix of workgroups per item
  ReduceItem item[RB_MAX_ITEMS];
};
__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(ReduceBatch rb) {
  int i = 0;
  while (i + 1 < rb.n && (int)blockIdx.x >= rb.block_end[i]) ++i;     // uniform per workgroup
  const ReduceItem it = rb.item[i];
  const int blk = blockIdx.x - (i > 0 ? rb.block_end[i - 1] : 0);
  wgrad_reduce_block(blk, it.part, it.gw, it.E, it.S, it.gbias ? it.part + (int64_t)it.S * it.E : nullptr, it.gbias, it.Cout,
                     it.accumulate);
}

// =========================================================================== conv3x3(nearest_up2x(a)) by phases
// Output pixel (2i + dy, 2j + dx) of a 3x3 convolution over a nearest x2 upsampled image only sees a 2x2 block of
// SOURCE pixels: rows {i-1, i} for dy = 0, {i, i+1} for dy = 1 (columns alike), with the filter rows that land on the
// same source row summed.  Four 2x2-tap convolutions at the low resolution (16 multiply-adds per source pixel and
// channel pair) replace the nine-tap convolution at the high resolution (36): 2.25x fewer FLOPs, and the upsampled
// tensor is never needed.
//   wp[dy][dx][co][ci][ty][tx]:  rows  dy=0: {w0 | w1+w2}   dy=1: {w0+w1 | w2};  columns alike.
__global__ void __launch_bounds__(256) upconv_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int n /*Cout*Cin*/) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float k[3][3];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i / 3][i % 3] = w[(int64_t)e * 9 + i];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      float r[2][3];                      // rows combined
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        r[0][c] = dy == 0 ? k[0][c] : k[0][c] + k[1][c];
        r[1][c] = dy == 0 ? k[1][c] + k[2][c] : k[2][c];
      }
      float* o = wp + ((int64_t)(dy * 2 + dx) * n + e) * 4;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        o[ty * 2 + 0] = dx == 0 ? r[ty][0] : r[ty][0] + r[ty][1];
        o[ty * 2 + 1] = dx == 0 ? r[ty][1] + r[ty][2] : r[ty][2];
      }
    }
}

// All four phases in ONE kernel: the low-resolution patch is staged once, every (row, column) position of the 3x3 halo
// is read once per k-group and feeds the phases whose 2x2 window covers it (16 (phase, tap) products per channel), each
// phase into its own accumulators; the epilogue writes the two horizontally adjacent phases as one float2.  16-channel
// output tiles (64 accumulator registers per lane).
// WK (64-pixel tiles for planes that give too few 256-pixel ones): the waves share the pixels and split the k-groups of a
// chunk; the partial sums meet in LDS two phases at a time and wave n stores pixel sub-tile n.
template <class G, bool WK>
__global__ void __launch_bounds__(CT_THREADS)
conv_upfwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                  const float* __restrict__ residual, float* __restrict__ y, Shape s, int vec_x, int vec_w) {
  constexpr int KS = 2, CT = 16, NT = 4, KG = 4;
  using P = Patch<G, KS>;
  using WT = WTile<KS, CT, WK>;
  constexpr int KK = WT::KK, CK = WT::CK, CTS = WT::CTS, NG = CK / KG, GSTEP = WK ? 4 : 1;
  static_assert(G::NPIX == (WK ? 64 : 256) && KK == 4 && NG % GSTEP == 0, "256-pixel tiles (or 64, K split over the waves), 2x2 windows");
  constexpr int STAGE = CK * P::CIS + 4 * WT::SIZE, REDF = WK ? 4 * 2 * NT * 4 * 64 : 0;
  __shared__ __attribute__((aligned(16))) float lds[STAGE > REDF ? STAGE : REDF];
  float* pl = lds;
  float* wl = lds + CK * P::CIS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, h = lane >> 4;
  const TileCoord tc = decode_tile<G>(blockIdx.x, s.H, s.W);
  const int co0 = blockIdx.y * CT;
  const int pix0 = WK ? 0 : wave * 64;
  const int k0 = WK ? wave * KG : 0;                  // WK: wave w takes k-groups w, w + 4, ...
  int lane_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) lane_b[n] = (h + k0) * P::CIS + pix_off<G, KS>(pix0 + n * 16 + j);
  const int lane_a = ((h + k0) * KK) * CTS + j;
  f32x4 acc[4][NT];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[ph][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  PatchStager<G, KS, CK> ps;
  WeightStager<KS, CT, false, WK> ws[4];
  const size_t phase_stride = (size_t)s.Cout * s.Cin * KK;
  ps.load(x, s, s.Cin, 0, tc, vec_x);
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) ws[ph].load(wp + ph * phase_stride, s.Cin, s.Cout, 0, co0, vec_w);
  for (int ci0 = 0; ci0 < s.Cin; ci0 += CK) {
    __syncthreads();
    ps.store(pl);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) ws[ph].store(wl + ph * WT::SIZE);
    __syncthreads();
    if (ci0 + CK < s.Cin) {
      ps.load(x, s, s.Cin, ci0 + CK, tc, vec_x);
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) ws[ph].load(wp + ph * phase_stride, s.Cin, s.Cout, ci0 + CK, co0, vec_w);
    }
#pragma unroll
    for (int gi = 0; gi < NG / GSTEP; ++gi) {
      const int g = gi * GSTEP;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float b[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) b[n] = pl[lane_b[n] + (g * KG) * P::CIS + kh * P::PWS + kw];
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const int ty = kh - dy, tx = kw - dx;            // this position inside phase (dy, dx)'s window?
              if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
              const float a = wl[(dy * 2 + dx) * WT::SIZE + lane_a + ((g * KG) * KK + ty * 2 + tx) * CTS];
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[dy * 2 + dx][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[dy * 2 + dx][n], 0, 0, 0);
            }
        }
    }
  }

  if constexpr (WK) {
    // two phases (one output row parity) per pass through LDS; wave n keeps the sums of pixel sub-tile n
    float* red = lds;
    constexpr int PW_ = 2 * NT * 4 * 64;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      __syncthreads();
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave * PW_ + ((dx * NT + n) * 4 + r) * 64 + lane] = acc[dy * 2 + dx][n][r];
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n != wave) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = ((dx * NT + n) * 4 + r) * 64 + lane;
            acc[dy * 2 + dx][n][r] = (red[e] + red[PW_ + e]) + (red[2 * PW_ + e] + red[3 * PW_ + e]);
          }
      }
    }
  }

  // epilogue: y[b][co][2(h0+pr)+dy][2(w0+pc)+dx]; the dx pair of a lane is one aligned float2
  const int W2 = 2 * s.W;
  const uint32_t HW2 = (uint32_t)(2 * s.H * W2);
  const int64_t tile0 = ((((int64_t)tc.b0 * s.Cout + co0) * (2 * s.H) + 2 * tc.h0) * W2 + 2 * tc.w0) * 4;
  char* ybase = reinterpret_cast<char*>(y) + tile0;
  const char* rbase = reinterpret_cast<const char*>(residual) + tile0;
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = bias ? bias[min(co0 + 4 * h + r, s.Cout - 1)] : 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (WK && n != wave) continue;
    const int p = pix0 + n * 16 + j;
    const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
    const int pr = rem / G::TW, pc = rem % G::TW;
    if (tc.b0 + img >= s.B || tc.h0 + pr >= s.H || tc.w0 + pc >= s.W) continue;
    const uint32_t lane_off = __umul24(__umul24(img, s.Cout) + 4 * h, HW2) + __umul24(2 * pr, W2) + 2 * pc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (co0 + 4 * h + r >= s.Cout) continue;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const uint32_t off = (lane_off + (uint32_t)r * HW2 + (uint32_t)(dy * W2)) << 2;
        float2 o = make_float2(acc[dy * 2][n][r] + bv[r], acc[dy * 2 + 1][n][r] + bv[r]);
        if (residual) {
          const float2 rr = *reinterpret_cast<const float2*>(rbase + off);
          o.x += rr.x; o.y += rr.y;
        }
        *reinterpret_cast<float2*>(ybase + off) = o;
      }
    }
  }
}

// Transpose of the above (gradient w.r.t. the low-resolution input; also what AvgPool2d(2) o conv3x3 is up to a filter
// transform): a 4x4-tap STRIDE-2 convolution over the high-resolution tensor,
//     ga[b][ci][i][j] = sum_co sum_{u,v in 0..3} w4t[ci][co][u][v] * gy[b][co][2i-1+u][2j-1+v],
//     w4t[ci][co][u][v] = sum_{kh in S(u), kw in S(v)} w[co][ci][kh][kw],   S = {2}, {1,2}, {0,1}, {0}.
// 16 products per source pixel and channel pair instead of 36 (3x3 dgrad at the high resolution) + a 2x2 sum.
// Tile = 256 low-resolution pixels; the staged patch is the (2 TH + 2) x (2 TW + 2) high-resolution window.
__global__ void __launch_bounds__(256) upconvT_weights_kernel(const float* __restrict__ w, float* __restrict__ w4t, int Cout, int Cin) {
  const int e = blockIdx.x * 256 + threadIdx.x;          // e = ci * Cout + co
  if (e >= Cout * Cin) return;
  const int ci = e / Cout, co = e - ci * Cout;
  float k[3][3];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i / 3][i % 3] = w[((int64_t)co * Cin + ci) * 9 + i];
  float r[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { r[0][c] = k[2][c]; r[1][c] = k[1][c] + k[2][c]; r[2][c] = k[0][c] + k[1][c]; r[3][c] = k[0][c]; }
  float* o = w4t + (int64_t)e * 16;
#pragma unroll
  for (int u = 0; u < 4; ++u) { o[u * 4 + 0] = r[u][2]; o[u * 4 + 1] = r[u][1] + r[u][2]; o[u * 4 + 2] = r[u][0] + r[u][1]; o[u * 4 + 3] = r[u][0]; }
}

// Both derived layouts of an up-conv filter in one launch (a training step needs both: wp for the forward, w4t for the input
// gradient): thread = (co, ci), the 3x3 filter read once.  Same arithmetic as the two kernels above.
__global__ void __launch_bounds__(256) upconv_weights_pair_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ w4t,
                                                                  int Cout, int Cin) {
  const int e = blockIdx.x * 256 + threadIdx.x;          // e = co * Cin + ci
  const int n = Cout * Cin;
  if (e >= n) return;
  const int co = e / Cin, ci = e - co * Cin;
  float k[3][3];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i / 3][i % 3] = w[(int64_t)e * 9 + i];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      float r[2][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        r[0][c] = dy == 0 ? k[0][c] : k[0][c] + k[1][c];
        r[1][c] = dy == 0 ? k[1][c] + k[2][c] : k[2][c];
      }
      float* o = wp + ((int64_t)(dy * 2 + dx) * n + e) * 4;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        o[ty * 2 + 0] = dx == 0 ? r[ty][0] : r[ty][0] + r[ty][1];
        o[ty * 2 + 1] = dx == 0 ? r[ty][1] + r[ty][2] : r[ty][2];
      }
    }
  float r[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { r[0][c] = k[2][c]; r[1][c] = k[1][c] + k[2][c]; r[2][c] = k[0][c] + k[1][c]; r[3][c] = k[0][c]; }
  float* o = w4t + ((int64_t)ci * Cout + co) * 16;
#pragma unroll
  for (int u = 0; u < 4; ++u) { o[u * 4 + 0] = r[u][2]; o[u * 4 + 1] = r[u][1] + r[u][2]; o[u * 4 + 2] = r[u][0] + r[u][1]; o[u * 4 + 3] = r[u][0]; }
}

template <class G>
struct Patch2x {                       // high-resolution windows of a low-resolution tile (one per image of the tile)
  static constexpr int PH = 2 * G::TH + 2, TW2 = 2 * G::TW;
  static constexpr int IOFF = 4, ORG = IOFF - 1;
  static constexpr int PWS = TW2 + 2 * IOFF;
  static constexpr int IMG = PH * PWS;
  static constexpr int RAW = G::NI * IMG;
  static constexpr int CIS = ((RAW + 31) / 32) * 32 + 16;
  static constexpr int PPI = G::TH * G::TW;            // low-resolution pixels per image
  static_assert(PPI % 64 == 0, "a wave's 64 pixels stay inside one image");
  // LDS offset (inside one channel) of the top-left tap of low-resolution pixel p
  __device__ __forceinline__ static int pix(int p) {
    const int img = p / PPI, rem = p % PPI;
    return img * IMG + 2 * (rem / G::TW) * PWS + 2 * (rem % G::TW) + ORG;
  }
};

template <class G, int CK>
struct PatchStager2x {
  using P = Patch2x<G>;
  static constexpr int Q = P::TW2 / 4;
  static constexpr int ROWS = CK * G::NI * P::PH;
  static constexpr int NV = (ROWS * Q + CT_THREADS - 1) / CT_THREADS;
  static constexpr int NHALO = ROWS * 2;
  static constexpr int NH = (NHALO + CT_THREADS - 1) / CT_THREADS;
  float4 v[NV];
  float hv[NH];
  // gy: (B, C, 2H, 2W); tile (b0, h0, w0) in low-resolution coordinates; window rows 2 h0 - 1 .., columns 2 w0 - 1 ..
  __device__ __forceinline__ void load(const float* __restrict__ gy, int B, int C, int H2, int W2, int c0, const TileCoord& tc, bool vec) {
    const char* base = reinterpret_cast<const char*>(gy) + ((((int64_t)tc.b0 * C + c0) * H2 + (2 * tc.h0 - 1)) * W2 + (2 * tc.w0 - 1)) * 4;
    const uint32_t HW2 = (uint32_t)(H2 * W2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < ROWS * Q) {
        const int q = e % Q, row = e / Q;
        const int r = row % P::PH, t = row / P::PH;
        const int img = t % G::NI, ci = t / G::NI;
        const int hh = 2 * tc.h0 - 1 + r, ww = 2 * tc.w0 + 4 * q;
        const uint32_t off = (__umul24(__umul24(img, C) + ci, HW2) + __umul24(r, W2) + 1 + 4 * q) << 2;
        if (tc.b0 + img < B && c0 + ci < C && hh >= 0 && hh < H2 && ww < W2) {
          const char* p = base + off;
          if (vec) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = *reinterpret_cast<const float*>(p);
            if (ww + 1 < W2) val.y = *reinterpret_cast<const float*>(p + 4);
            if (ww + 2 < W2) val.z = *reinterpret_cast<const float*>(p + 8);
            if (ww + 3 < W2) val.w = *reinterpret_cast<const float*>(p + 12);
          }
        }
      }
      v[i] = val;
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      float val = 0.f;
      if (e < NHALO) {
        const int row = e >> 1;
        const int r = row % P::PH, t = row / P::PH;
        const int img = t % G::NI, ci = t / G::NI;
        const int hh = 2 * tc.h0 - 1 + r;
        const int ww = (e & 1) ? 2 * tc.w0 + P::TW2 : 2 * tc.w0 - 1;
        if (tc.b0 + img < B && c0 + ci < C && hh >= 0 && hh < H2 && ww >= 0 && ww < W2)
          val = *reinterpret_cast<const float*>(base + ((__umul24(__umul24(img, C) + ci, HW2) + __umul24(r, W2) + ((e & 1) ? P::TW2 + 1 : 0)) << 2));
      }
      hv[i] = val;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      if (e < ROWS * Q) {
        const int q = e % Q, row = e / Q;
        const int r = row % P::PH, t = row / P::PH;
        const int img = t % G::NI, ci = t / G::NI;
        *reinterpret_cast<float4*>(lds + ci * P::CIS + img * P::IMG + r * P::PWS + P::IOFF + 4 * q) = v[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int e = threadIdx.x + i * CT_THREADS;
      if (e < NHALO) {
        const int row = e >> 1;
        const int r = row % P::PH, t = row / P::PH;
        const int img = t % G::NI, ci = t / G::NI;
        lds[ci * P::CIS + img * P::IMG + r * P::PWS + ((e & 1) ? P::IOFF + P::TW2 : P::IOFF - 1)] = hv[i];
      }
    }
  }
};

// WK (64-pixel tiles, used when 256-pixel tiles would leave CUs idle): the four waves share the pixels and split the
// 16 channels of a chunk, partial sums meet in LDS (as in conv_fwd_kernel).
template <class G, bool WK>
__global__ void __launch_bounds__(CT_THREADS)
conv_upT_kernel(const float* __restrict__ gy, const float* __restrict__ w4t, const float* __restrict__ bias,
                const float* __restrict__ residual, float* __restrict__ ga, Shape s /*Cin = gy channels, Cout = ga channels,
                H x W = ga plane*/, int vec_x, int vec_w) {
  constexpr int CT = 16, NT = 4, KG = 4;
  using P = Patch2x<G>;
  using WT = WTile<4, CT, WK>;                        // 16 taps per (out, in) channel pair
  constexpr int KK = WT::KK, CK = WT::CK, CTS = WT::CTS, NG = CK / KG;
  static_assert(KK == 16 && G::NPIX == (WK ? 64 : 256) && (!WK || NG == 4), "16 taps; 256-pixel tiles, or 64 with one k-group per wave");
  constexpr int STAGE = CK * P::CIS + WT::SIZE, REDF = WK ? 4 * NT * 4 * 64 : 0;
  __shared__ __attribute__((aligned(16))) float lds[STAGE > REDF ? STAGE : REDF];
  float* pl = lds;
  float* wl = lds + CK * P::CIS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, h = lane >> 4;
  const TileCoord tc = decode_tile<G>(blockIdx.x, s.H, s.W);
  const int co0 = blockIdx.y * CT;
  const int pix0 = WK ? 0 : wave * 64;
  const int k0 = WK ? wave * KG : 0;                  // first channel (inside a chunk) of this wave's k-group
  int lane_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    lane_b[n] = (h + k0) * P::CIS + P::pix(pix0 + n * 16 + j);
  }
  const int lane_a = ((h + k0) * KK) * CTS + j;
  using Core = FwdCore<G, 3, 16, 1, WK>;
  typename Core::acc_t acc[1][NT];
  Core::zero(acc);

  PatchStager2x<G, CK> ps;
  WeightStager<4, CT, false, WK> ws;
  const int H2 = 2 * s.H, W2 = 2 * s.W;
  ps.load(gy, s.B, s.Cin, H2, W2, 0, tc, vec_x);
  ws.load(w4t, s.Cin, s.Cout, 0, co0, vec_w);
  for (int ci0 = 0; ci0 < s.Cin; ci0 += CK) {
    __syncthreads();
    ps.store(pl);
    ws.store(wl);
    __syncthreads();
    if (ci0 + CK < s.Cin) {
      ps.load(gy, s.B, s.Cin, H2, W2, ci0 + CK, tc, vec_x);
      ws.load(w4t, s.Cin, s.Cout, ci0 + CK, co0, vec_w);
    }
#pragma unroll
    for (int g = 0; g < (WK ? 1 : NG); ++g)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float a = wl[lane_a + ((g * KG) * KK + t) * CTS];
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pl[lane_b[n] + (g * KG) * P::CIS + (t >> 2) * P::PWS + (t & 3)], acc[0][n], 0, 0, 0);
      }
  }
  if constexpr (WK) {
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * NT + n) * 4 + r) * 64 + lane] = acc[0][n][r];
    __syncthreads();
    constexpr int PW_ = NT * 4 * 64;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = (n * 4 + r) * 64 + lane;
        acc[0][n][r] = (red[e] + red[PW_ + e]) + (red[2 * PW_ + e] + red[3 * PW_ + e]);
      }
  }
  Core::epilogue(acc, bias, residual, ga, s, tc, co0, pix0, j, h, wave);
}

// ---- the two stride-2 kernels with LDS-DMA staging (the scheme of conv_dma_kernel: whole 16-byte chunks, bounds-checked
// zero padding, filter rows copied as they lie, two LDS buffers, one barrier per channel chunk).  Same arithmetic and
// accumulation order as conv_upT_kernel / conv_upfwd_kernel.
template <class G> struct DPatch2x {            // high-resolution windows of a low-resolution tile, rows of 2 TW + 8 floats
  static constexpr int PH = 2 * G::TH + 2, PWS = 2 * G::TW + 8, QR = PWS / 4;
  static constexpr int IMG = PH * PWS, RAW = G::NI * IMG;
  static constexpr int CIS = RAW + ((48 - RAW % 32) % 32);
  static constexpr int CPC = CIS / 4;
  static constexpr int PPI = G::TH * G::TW;
  __device__ __forceinline__ static int pix(int p) {             // top-left tap (row 2r - 1, column 2c - 1) of low-res pixel p
    const int img = p / PPI, rem = p % PPI;
    return img * IMG + 2 * (rem / G::TW) * PWS + 2 * (rem % G::TW) + 3;
  }
};

// WK (64-pixel tiles): the four waves share the pixels, wave w takes k-group w of every 16-channel chunk.
template <class G, bool WK, bool DB = true>
__global__ void __launch_bounds__(CT_THREADS)
conv_upT_dma_kernel(const float* __restrict__ gy, const float* __restrict__ w4t, const float* __restrict__ bias,
                    const float* __restrict__ residual, float* __restrict__ ga, Shape s /*Cin = gy channels, Cout = ga channels,
                    H x W = ga plane*/, int xcd_swizzle) {
  constexpr int CT = 16, NT = 4, KG = 4, KK = 16, CK = WK ? 16 : 4;
  using P = DPatch2x<G>;
  using Core = FwdCore<G, 3, 16, 1, WK>;
  constexpr int RS = CK * KK + 4, QW = RS / 4;
  constexpr int PCH = CK * P::CPC, WCH = CT * QW;
  constexpr int PCHP = dma_pad(PCH), WCHP = dma_pad(WCH);        // regions of whole wave pieces (see dma_pad)
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS, NVW = (WCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, BUF = PBUF + WCHP * 4;
  constexpr int REDF = WK ? 4 * NT * 4 * 64 : 0, NBUF = DB ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[(NBUF * BUF > REDF) ? NBUF * BUF : REDF];
  const int lane = threadIdx.x & 63, wave = wave_index();
  if (s.prio) __builtin_amdgcn_s_setprio(3);
  const int j = lane & 15, h = lane >> 4;
  int bid = blockIdx.x;
  if (xcd_swizzle) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const TileCoord tc = decode_tile<G>(bid, s.H, s.W);
  const int co0 = blockIdx.y * CT;
  const int pix0 = WK ? 0 : wave * 64;
  const int k0 = WK ? wave * KG : 0;
  const int H2 = 2 * s.H, W2 = 2 * s.W;
  const uint32_t HW2 = (uint32_t)(H2 * W2);

  uint32_t poff[NVP], woff[NVW];
#pragma unroll
  for (int i = 0; i < NVP; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int ci = e / P::CPC, rem = e % P::CPC;
    const int row = rem / P::QR, q = rem % P::QR;
    const int img = row / P::PH, r = row % P::PH;
    const int hh = 2 * tc.h0 - 1 + r, ww = 2 * tc.w0 - 4 + 4 * q;
    const bool ok = (e < PCH) && (rem < P::RAW / 4) && (tc.b0 + img < s.B) && (hh >= 0) && (hh < H2) && (ww >= 0) && (ww < W2);
    poff[i] = ok ? (__umul24(__umul24(img, s.Cin) + ci, HW2) + __umul24(hh, W2) + ww) << 2 : DMA_OOB;
  }
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int row = e / QW, q = e % QW;
    const bool ok = (e < WCH) && (4 * q < CK * KK) && (co0 + row < s.Cout);
    woff[i] = ok ? (uint32_t)(row * s.Cin * KK + 4 * q) << 2 : DMA_OOB;
  }
  const char* xb = reinterpret_cast<const char*>(gy) + ((int64_t)tc.b0 * s.Cin * HW2) * 4;
  int64_t xbytes = (int64_t)(s.B - tc.b0) * s.Cin * HW2 * 4;
  const int64_t xstep = (int64_t)CK * HW2 * 4;
  const char* wb = reinterpret_cast<const char*>(w4t) + (int64_t)co0 * s.Cin * KK * 4;
  int64_t wbytes = (int64_t)(s.Cout - co0) * s.Cin * KK * 4;
  const int64_t wstep = (int64_t)CK * KK * 4;
  auto issue = [&](int c0, float* buf) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, (int)wbytes, 0x00020000);
    const int cvalid = s.Cin - c0;
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
      uint32_t off = poff[i];
      if (cvalid < CK) off = ((i * CT_THREADS + (int)threadIdx.x) / P::CPC < cvalid) ? off : DMA_OOB;
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP) dma16(rx, buf + (i * CT_THREADS + wave * 64) * 4, off);
    }
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
      uint32_t off = woff[i];
      if (cvalid < CK) off = (4 * ((i * CT_THREADS + (int)threadIdx.x) % QW) < cvalid * KK) ? off : DMA_OOB;
      if ((i + 1) * CT_THREADS <= WCHP || i * CT_THREADS + wave * 64 < WCHP) dma16(rw, buf + PBUF + (i * CT_THREADS + wave * 64) * 4, off);
    }
    xb += xstep; xbytes -= xstep;
    wb += wstep; wbytes -= wstep;
  };
  int lane_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) lane_b[n] = (h + k0) * P::CIS + P::pix(pix0 + n * 16 + j);
  const int lane_a = j * RS + (h + k0) * KK;
  typename Core::acc_t acc[1][NT];
  Core::zero(acc);

  if (DB) issue(0, lds);
  int buf = 0;
  for (int c0 = 0; c0 < s.Cin; c0 += CK) {
    if constexpr (!DB) {
      __syncthreads();
      issue(c0, lds);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (DB) {
      if (c0 + CK < s.Cin) issue(c0 + CK, lds + (buf ^ 1) * BUF);
    }
    const float* pl = lds + buf * BUF;
    const float* wl = pl + PBUF;
    if (s.prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = wl[lane_a + t];
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pl[lane_b[n] + (t >> 2) * P::PWS + (t & 3)], acc[0][n], 0, 0, 0);
    }
    if (s.prio) __builtin_amdgcn_s_setprio(3);
    if (DB) buf ^= 1;
  }
  if constexpr (WK) {
    __syncthreads();
    float* red = lds;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * NT + n) * 4 + r) * 64 + lane] = acc[0][n][r];
    __syncthreads();
    constexpr int PW_ = NT * 4 * 64;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = (n * 4 + r) * 64 + lane;
        acc[0][n][r] = (red[e] + red[PW_ + e]) + (red[2 * PW_ + e] + red[3 * PW_ + e]);
      }
  }
  Core::epilogue(acc, bias, residual, ga, s, tc, co0, pix0, j, h, wave);
}

template <class G, bool WK, bool DB = true>
__global__ void __launch_bounds__(CT_THREADS)
conv_upfwd_dma_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                      const float* __restrict__ residual, float* __restrict__ y, Shape s, int xcd_swizzle) {
  constexpr int CT = 16, NT = 4, KG = 4, KK = 4, CK = WK ? 32 : 8, NG = CK / KG, GSTEP = WK ? 4 : 1;
  using P = DPatch<G>;
  constexpr int RS = CK * KK + 4, QW = RS / 4, WSZ = CT * RS;        // one phase's filter image
  constexpr int PCH = CK * P::CPC, WCH = 4 * CT * QW;
  constexpr int PCHP = dma_pad(PCH), WCHP = dma_pad(WCH);        // regions of whole wave pieces (see dma_pad)
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS, NVW = (WCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, BUF = PBUF + WCHP * 4;
  constexpr int REDF = WK ? 4 * 2 * NT * 4 * 64 : 0, NBUF = DB ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float lds[(NBUF * BUF > REDF) ? NBUF * BUF : REDF];
  const int lane = threadIdx.x & 63, wave = wave_index();
  if (s.prio) __builtin_amdgcn_s_setprio(3);
  const int j = lane & 15, h = lane >> 4;
  int bid = blockIdx.x;
  if (xcd_swizzle) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const TileCoord tc = decode_tile<G>(bid, s.H, s.W);
  const int co0 = blockIdx.y * CT;
  const int pix0 = WK ? 0 : wave * 64;
  const int k0 = WK ? wave * KG : 0;
  const uint32_t HW = (uint32_t)(s.H * s.W);
  const int64_t phase_floats = (int64_t)s.Cout * s.Cin * KK;

  uint32_t poff[NVP], woff[NVW];
#pragma unroll
  for (int i = 0; i < NVP; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int ci = e / P::CPC, rem = e % P::CPC;
    const int row = rem / P::QR, q = rem % P::QR;
    const int img = row / P::PH, r = row % P::PH;
    const int hh = tc.h0 + r - 1, ww = tc.w0 - 4 + 4 * q;
    const bool ok = (e < PCH) && (rem < P::RAW / 4) && (tc.b0 + img < s.B) && (hh >= 0) && (hh < s.H) && (ww >= 0) && (ww < s.W);
    poff[i] = ok ? (__umul24(__umul24(img, s.Cin) + ci, HW) + __umul24(hh, s.W) + ww) << 2 : DMA_OOB;
  }
#pragma unroll
  for (int i = 0; i < NVW; ++i) {
    const int e = i * CT_THREADS + threadIdx.x;
    const int ph = e / (CT * QW), rem = e % (CT * QW);
    const int row = rem / QW, q = rem % QW;
    const bool ok = (e < WCH) && (4 * q < CK * KK) && (co0 + row < s.Cout);
    woff[i] = ok ? (uint32_t)(ph * phase_floats + (int64_t)row * s.Cin * KK + 4 * q) << 2 : DMA_OOB;
  }
  const char* xb = reinterpret_cast<const char*>(x) + ((int64_t)tc.b0 * s.Cin * HW) * 4;
  int64_t xbytes = (int64_t)(s.B - tc.b0) * s.Cin * HW * 4;
  const int64_t xstep = (int64_t)CK * HW * 4;
  const char* wb = reinterpret_cast<const char*>(wp) + (int64_t)co0 * s.Cin * KK * 4;
  int64_t wbytes = 4 * phase_floats * 4 - (int64_t)co0 * s.Cin * KK * 4;
  const int64_t wstep = (int64_t)CK * KK * 4;
  auto issue = [&](int c0, float* buf) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, (int)wbytes, 0x00020000);
    const int cvalid = s.Cin - c0;
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
      uint32_t off = poff[i];
      if (cvalid < CK) off = ((i * CT_THREADS + (int)threadIdx.x) / P::CPC < cvalid) ? off : DMA_OOB;
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP) dma16(rx, buf + (i * CT_THREADS + wave * 64) * 4, off);
    }
#pragma unroll
    for (int i = 0; i < NVW; ++i) {
      uint32_t off = woff[i];
      if (cvalid < CK) off = (4 * ((i * CT_THREADS + (int)threadIdx.x) % QW) < cvalid * KK) ? off : DMA_OOB;
      if ((i + 1) * CT_THREADS <= WCHP || i * CT_THREADS + wave * 64 < WCHP) dma16(rw, buf + PBUF + (i * CT_THREADS + wave * 64) * 4, off);
    }
    xb += xstep; xbytes -= xstep;
    wb += wstep; wbytes -= wstep;
  };
  int lane_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) lane_b[n] = (h + k0) * P::CIS + P::pix(pix0 + n * 16 + j);
  const int lane_a = j * RS + (h + k0) * KK;
  f32x4 acc[4][NT];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[ph][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (DB) issue(0, lds);
  int buf = 0;
  for (int c0 = 0; c0 < s.Cin; c0 += CK) {
    if constexpr (!DB) {
      __syncthreads();
      issue(c0, lds);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (DB) {
      if (c0 + CK < s.Cin) issue(c0 + CK, lds + (buf ^ 1) * BUF);
    }
    const float* pl = lds + buf * BUF;
    const float* wl = pl + PBUF;
    if (s.prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int gi = 0; gi < NG / GSTEP; ++gi) {
      const int g = gi * GSTEP;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float b[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) b[n] = pl[lane_b[n] + (g * KG) * P::CIS + kh * P::PWS + kw];
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const int ty = kh - dy, tx = kw - dx;
              if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
              const float a = wl[(dy * 2 + dx) * WSZ + lane_a + (g * KG) * KK + ty * 2 + tx];
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[dy * 2 + dx][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[dy * 2 + dx][n], 0, 0, 0);
            }
        }
    }
    if (s.prio) __builtin_amdgcn_s_setprio(3);
    if (DB) buf ^= 1;
  }
  if constexpr (WK) {
    float* red = lds;
    constexpr int PW_ = 2 * NT * 4 * 64;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      __syncthreads();
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave * PW_ + ((dx * NT + n) * 4 + r) * 64 + lane] = acc[dy * 2 + dx][n][r];
      __syncthreads();
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n != wave) continue;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = ((dx * NT + n) * 4 + r) * 64 + lane;
            acc[dy * 2 + dx][n][r] = (red[e] + red[PW_ + e]) + (red[2 * PW_ + e] + red[3 * PW_ + e]);
          }
      }
    }
  }
  const int W2 = 2 * s.W;
  const uint32_t HW2 = (uint32_t)(2 * s.H * W2);
  const int64_t tile0 = ((((int64_t)tc.b0 * s.Cout + co0) * (2 * s.H) + 2 * tc.h0) * W2 + 2 * tc.w0) * 4;
  char* ybase = reinterpret_cast<char*>(y) + tile0;
  const char* rbase = reinterpret_cast<const char*>(residual) + tile0;
  // (the residual add keeps the reference's order: (acc + bias) + residual)
  const bool wide = (s.W % 2 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                    (residual == nullptr || (reinterpret_cast<uintptr_t>(residual) & 15) == 0);
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = bias ? bias[min(co0 + 4 * h + r, s.Cout - 1)] : 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    if (WK && n != wave) continue;
    const int p = pix0 + n * 16 + j;
    const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
    const int pr = rem / G::TW, pc = rem % G::TW;
    if (tc.b0 + img >= s.B || tc.h0 + pr >= s.H || tc.w0 + pc >= s.W) continue;
    const uint32_t lane_off = __umul24(__umul24(img, s.Cout) + 4 * h, HW2) + __umul24(2 * pr, W2) + 2 * pc;
    if (wide) {
      // lane pairs (two horizontally adjacent low-resolution pixels) trade output rows: the even lane stores the four
      // columns of row 2 pr, the odd lane those of row 2 pr + 1 -- one 16-byte store per lane instead of two 8-byte ones
      const int odd = j & 1;
      const uint32_t off0 = lane_off - (odd ? 2 : 0) + (odd ? (uint32_t)W2 : 0u);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a00 = acc[0][n][r] + bv[r], a01 = acc[1][n][r] + bv[r], a10 = acc[2][n][r] + bv[r], a11 = acc[3][n][r] + bv[r];
        const float sx = odd ? a00 : a10, sy = odd ? a01 : a11;                 // what the partner needs
        const float rx = __shfl_xor(sx, 1, 64), ry = __shfl_xor(sy, 1, 64);
        if (co0 + 4 * h + r >= s.Cout) continue;
        float4 o = odd ? make_float4(rx, ry, a10, a11) : make_float4(a00, a01, rx, ry);
        const uint32_t off = (off0 + (uint32_t)r * HW2) << 2;
        if (residual) {
          const float4 rr = *reinterpret_cast<const float4*>(rbase + off);
          o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        *reinterpret_cast<float4*>(ybase + off) = o;
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (co0 + 4 * h + r >= s.Cout) continue;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const uint32_t off = (lane_off + (uint32_t)r * HW2 + (uint32_t)(dy * W2)) << 2;
        float2 o = make_float2(acc[dy * 2][n][r] + bv[r], acc[dy * 2 + 1][n][r] + bv[r]);
        if (residual) {
          const float2 rr = *reinterpret_cast<const float2*>(rbase + off);
          o.x += rr.x; o.y += rr.y;
        }
        *reinterpret_cast<float2*>(ybase + off) = o;
      }
    }
  }
}

// Bias gradient of the stride-2 forms, taken from the operands the weight-gradient kernel stages anyway (no pass of its own):
//   mode 0 (pooled conv, bias over the `lo` channels = gy): the sum of the A fragments -- every gy value of a tile passes
//          through them exactly once; by the workgroups of hi-chunk 0;
//   mode 1 (up-conv, bias over the `hi` channels = gy at the high resolution): the sum of the B fragments at the four inner taps
//          (u, v in {1, 2}): hi pixel (2r-1+u, 2c-1+v) is met once over all low-resolution pixels (r, c); by the lo-tile-0 workgroups.
// Per-workgroup partials [S][channels] (fixed order inside: lane groups, then the four waves), finished by the fold kernel.
template <int NT>
__device__ __forceinline__ void s2_bias_finish(float* red /*LDS, >= 4 * 16 floats, free*/, float bsum_a, const float (&bsum_b)[NT], int mode,
                                               float* __restrict__ bias_part, int split, int Clo, int Chi, int co0, int ci0, int lane,
                                               int wave) {
  const int j = lane & 15, h = lane >> 4;
  if (mode == 0) {
    float v = bsum_a;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (h == 0) red[wave * 16 + j] = v;
  } else {
    const bool inner = ((j >> 2) == 1 || (j >> 2) == 2) && ((j & 3) == 1 || (j & 3) == 2);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float v = inner ? bsum_b[n] : 0.f;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) red[wave * 16 + n] = v;
    }
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < 16) {
    const float v = (red[c] + red[16 + c]) + (red[32 + c] + red[48 + c]);
    if (mode == 0) {
      if (co0 + c < Clo) bias_part[(int64_t)split * Clo + co0 + c] = v;
    } else if (c < NT && ci0 + c < Chi) {
      bias_part[(int64_t)split * Chi + ci0 + c] = v;
    }
  }
}

// The third member of the family: T[o][i][u][v] = sum_{b, r, c} lo[b][o][r][c] * hi[b][i][2r-1+u][2c-1+v], the product both
// weight gradients reduce to (pooled conv: lo = gy, hi = x; up-conv: lo = a, hi = gy), 16 taps per low-resolution pixel and
// channel pair instead of 36 for the 3x3 weight gradient on the (materialised) high-resolution pair.  Same scheme as
// conv_wgrad_kernel: M = 16 `lo` channels, N = (S2_CKW `hi` channels x 16 taps) columns, K = pixels; per-workgroup partials.
constexpr int S2_CKW = 4, S2_NT = S2_CKW * 16 / 16;   // (measured: 4 beats 8 and 2 -- 38 KB of LDS, 4 workgroups per CU)
template <class G>
__global__ void __launch_bounds__(CT_THREADS)
conv_wgrad_s2_kernel(const float* __restrict__ hi, const float* __restrict__ lo, float* __restrict__ part, Shape s /*Cin = hi channels,
                     Cout = lo channels, H x W = lo plane*/, int ntiles, int S, int vec_hi, int vec_lo,
                     float* __restrict__ bias_part /*nullable*/, int bias_mode) {
  using P = Patch2x<G>;
  constexpr int CKW = S2_CKW, NT = S2_NT, CT = 16;
  constexpr int PATCH = CKW * P::CIS, GYT = CT * WG_GYS;
  constexpr int RED = 4 * NT * 4 * 64;
  constexpr int LDS_FLOATS = (PATCH + GYT) > RED ? (PATCH + GYT) : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* pl = lds;
  float* gl = lds + PATCH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, h = lane >> 4;
  const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CKW;
  const int split = blockIdx.x;
  int colbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) colbase[n] = n * P::CIS + (j >> 2) * P::PWS + (j & 3);      // column n*16 + j = (hi channel n, tap j)
  const int lane_b = P::pix(wave * 64) + 2 * h;                     // the wave's first pixel (its 64 pixels share an image)
  const int lane_a = j * WG_GYS + wave * 64 + h;
  f32x4 acc[NT];
  float bsum_a = 0.f, bsum_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { acc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; bsum_b[n] = 0.f; }
  const bool do_bias = bias_part != nullptr && (bias_mode == 0 ? blockIdx.z == 0 : blockIdx.y == 0);
  auto products = [&](const float* pl, const float* gl, auto with_sums) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int pg = 4 * g;
      const int goff = 2 * (pg / G::TW) * P::PWS + 2 * (pg % G::TW);
      const float a = gl[lane_a + 4 * g];
      float b[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = pl[colbase[n] + lane_b + goff];
      if constexpr (decltype(with_sums)::value) {
        bsum_a += a;
#pragma unroll
        for (int n = 0; n < NT; ++n) bsum_b[n] += b[n];
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
    }
  };
  PatchStager2x<G, CKW> ps;
  GyStager<G, CT> gs;
  const int H2 = 2 * s.H, W2 = 2 * s.W;
  if (split < ntiles) {
    const TileCoord tc = decode_tile<G>(split, s.H, s.W);
    ps.load(hi, s.B, s.Cin, H2, W2, ci0, tc, vec_hi);
    gs.load(lo, s, co0, tc, vec_lo);
  }
  for (int t = split; t < ntiles; t += S) {
    __syncthreads();
    ps.store(pl);
    gs.store(gl);
    __syncthreads();
    if (t + S < ntiles) {
      const TileCoord tn = decode_tile<G>(t + S, s.H, s.W);
      ps.load(hi, s.B, s.Cin, H2, W2, ci0, tn, vec_hi);
      gs.load(lo, s, co0, tn, vec_lo);
    }
    if (do_bias) products(pl, gl, std::true_type{});          // (block-uniform: the operand sums only where they are used)
    else products(pl, gl, std::false_type{});
  }
  __syncthreads();
  float* red = lds;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * NT + n) * 4 + r) * 64 + lane] = acc[n][r];
  __syncthreads();
  constexpr int PER_WAVE = NT * 4 * 64;
  for (int e = threadIdx.x; e < PER_WAVE; e += CT_THREADS) {
    const float v = (red[e] + red[PER_WAVE + e]) + (red[2 * PER_WAVE + e] + red[3 * PER_WAVE + e]);
    const int l = e & 63, q = e >> 6;
    const int r = q & 3, n = q >> 2;
    const int co = co0 + (l >> 4) * 4 + r, ci = ci0 + n, tap = l & 15;
    if (co < s.Cout && ci < s.Cin) part[(((int64_t)split * s.Cout + co) * s.Cin + ci) * 16 + tap] = v;
  }
  if (do_bias) {                                            // (block-uniform)
    __syncthreads();                                        // the partial sums above have been read out of `red`
    s2_bias_finish<NT>(red, bsum_a, bsum_b, bias_mode, bias_part, split, s.Cout, s.Cin, co0, ci0, lane, wave);
  }
}

// ---- conv_wgrad_s2_kernel with LDS-DMA staging (single buffer, as conv_wgrad_dma_kernel): same split, accumulation
// order and partial layout -> bit-identical partials.
template <class G>
__global__ void __launch_bounds__(CT_THREADS)
conv_wgrad_s2_dma_kernel(const float* __restrict__ hi, const float* __restrict__ lo, float* __restrict__ part, Shape s /*Cin = hi
                         channels, Cout = lo channels, H x W = lo plane*/, int ntiles, int S, float* __restrict__ bias_part /*nullable*/,
                         int bias_mode) {
  static_assert(G::NPIX == 256, "256-pixel tiles");
  using P = DPatch2x<G>;
  constexpr int CKW = S2_CKW, NT = S2_NT, CT = 16;
  constexpr int PCH = CKW * P::CPC, GCH = CT * WGD_GYQ;
  constexpr int PCHP = dma_pad(PCH), GCHP = dma_pad(GCH);        // regions of whole wave pieces (see dma_pad)
  constexpr int NVP = (PCHP + CT_THREADS - 1) / CT_THREADS, NVG = (GCHP + CT_THREADS - 1) / CT_THREADS;
  constexpr int PBUF = PCHP * 4, BUF = PBUF + GCHP * 4;
  constexpr int RED = 4 * NT * 4 * 64;
  __shared__ __attribute__((aligned(16))) float lds[BUF > RED ? BUF : RED];
  const int lane = threadIdx.x & 63, wave = wave_index();
  if (s.prio) __builtin_amdgcn_s_setprio(3);
  const int j = lane & 15, h = lane >> 4;
  const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CKW;
  const int split = blockIdx.x;
  const int H2 = 2 * s.H, W2 = 2 * s.W;
  const uint32_t HW = (uint32_t)(s.H * s.W), HW2 = (uint32_t)(H2 * W2);
  int colbase[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) colbase[n] = n * P::CIS + (j >> 2) * P::PWS + (j & 3);
  const int lane_b = P::pix(wave * 64) + 2 * h;
  const int lane_a = j * WG_GYS + wave * 64 + h;
  f32x4 acc[NT];
  float bsum_a = 0.f, bsum_b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { acc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; bsum_b[n] = 0.f; }
  const bool do_bias = bias_part != nullptr && (bias_mode == 0 ? blockIdx.z == 0 : blockIdx.y == 0);
  auto products = [&](const float* pl, const float* gl, auto with_sums) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int pg = 4 * g;
      const int goff = 2 * (pg / G::TW) * P::PWS + 2 * (pg % G::TW);
      const float a = gl[lane_a + 4 * g];
      float b[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = pl[colbase[n] + lane_b + goff];
      if constexpr (decltype(with_sums)::value) {
        bsum_a += a;
#pragma unroll
        for (int n = 0; n < NT; ++n) bsum_b[n] += b[n];
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc[n], 0, 0, 0);
    }
  };
  const int cvalid = min(CKW, s.Cin - ci0);
  for (int t = split; t < ntiles; t += S) {
    const TileCoord tc = decode_tile<G>(t, s.H, s.W);
    const char* xb = reinterpret_cast<const char*>(hi) + ((int64_t)tc.b0 * s.Cin + ci0) * HW2 * 4;
    const int64_t xbytes = ((int64_t)(s.B - tc.b0) * s.Cin - ci0) * HW2 * 4;
    const char* gb = reinterpret_cast<const char*>(lo) + ((int64_t)tc.b0 * s.Cout + co0) * HW * 4;
    const int64_t gbytes = ((int64_t)(s.B - tc.b0) * s.Cout - co0) * HW * 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gb), 0, (int)gbytes, 0x00020000);
    __syncthreads();                          // everybody is done reading the previous tile
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
      const int e = i * CT_THREADS + threadIdx.x;
      const int ci = e / P::CPC, rem = e % P::CPC;
      const int row = rem / P::QR, q = rem % P::QR;
      const int img = row / P::PH, r = row % P::PH;
      const int hh = 2 * tc.h0 - 1 + r, ww = 2 * tc.w0 - 4 + 4 * q;
      const bool ok = (rem < P::RAW / 4) && (ci < cvalid) && (tc.b0 + img < s.B) && (hh >= 0) && (hh < H2) && (ww >= 0) && (ww < W2);
      const uint32_t off = ok ? (__umul24(__umul24(img, s.Cin) + ci, HW2) + __umul24(hh, W2) + ww) << 2 : DMA_OOB;
      if ((i + 1) * CT_THREADS <= PCHP || i * CT_THREADS + wave * 64 < PCHP) dma16(rx, lds + (i * CT_THREADS + wave * 64) * 4, off);
    }
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
      const int e = i * CT_THREADS + threadIdx.x;
      const int co = e / WGD_GYQ, q = e % WGD_GYQ;
      const int p = 4 * q;
      const int img = p / (G::TH * G::TW), rem = p % (G::TH * G::TW);
      const int hh = tc.h0 + rem / G::TW, ww = tc.w0 + rem % G::TW;
      const bool ok = (e < GCH) && (q < 64) && (co0 + co < s.Cout) && (tc.b0 + img < s.B) && (hh < s.H) && (ww < s.W);
      const uint32_t off = ok ? (__umul24(__umul24(img, s.Cout) + co, HW) + __umul24(hh, s.W) + ww) << 2 : DMA_OOB;
      if ((i + 1) * CT_THREADS <= GCHP || i * CT_THREADS + wave * 64 < GCHP) dma16(rg, lds + PBUF + (i * CT_THREADS + wave * 64) * 4, off);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float* pl = lds;
    const float* gl = lds + PBUF;
    if (s.prio) __builtin_amdgcn_s_setprio(0);
    if (do_bias) products(pl, gl, std::true_type{});          // (block-uniform: the operand sums only where they are used)
    else products(pl, gl, std::false_type{});
    if (s.prio) __builtin_amdgcn_s_setprio(3);
  }
  __syncthreads();
  float* red = lds;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * NT + n) * 4 + r) * 64 + lane] = acc[n][r];
  __syncthreads();
  constexpr int PER_WAVE = NT * 4 * 64;
  for (int e = threadIdx.x; e < PER_WAVE; e += CT_THREADS) {
    const float v = (red[e] + red[PER_WAVE + e]) + (red[2 * PER_WAVE + e] + red[3 * PER_WAVE + e]);
    const int l = e & 63, q = e >> 6;
    const int r = q & 3, n = q >> 2;
    const int co = co0 + (l >> 4) * 4 + r, ci = ci0 + n, tap = l & 15;
    if (co < s.Cout && ci < s.Cin) part[(((int64_t)split * s.Cout + co) * s.Cin + ci) * 16 + tap] = v;
  }
  if (do_bias) {                                            // (block-uniform)
    __syncthreads();                                        // the partial sums above have been read out of `red`
    s2_bias_finish<NT>(red, bsum_a, bsum_b, bias_mode, bias_part, split, s.Cout, s.Cin, co0, ci0, lane, wave);
  }
}

// Sum of the S partials of T (fixed order) and fold onto the 3x3 taps, in one kernel: a workgroup owns 4 (lo, hi)
// channel pairs; thread = (pair, tap, S-slice), the 4 slices are combined through LDS, then 9 threads per pair fold.
//   mode 0 (pooled conv, T[co][ci]):  gw[kh][kw] (+)= 0.25 sum_{dy,dx} T[dy+kh][dx+kw];
//   mode 1 (up-conv, T[ci][co]):      gw[kh][kw] (+)= sum_{u: kh in S(u)} sum_{v: kw in S(v)} T[u][v], S = {2},{1,2},{0,1},{0}.
__device__ __forceinline__ void s2wgrad_reduce_fold(const float* __restrict__ part, float* __restrict__ gw, int S, int Clo, int Chi,
                                                    int Cout, int Cin, int mode, int accumulate,
                                                    const float* __restrict__ bias_part /*nullable: [S][Cout]*/, float* __restrict__ gbias,
                                                    int pair_blocks, int bx /*block index within this layer's blocks*/) {
  if (bx >= pair_blocks) {
    // the bias gradient: one wave per channel, lanes stride over the S partials, fixed-order wavefront sum
    const int c = (bx - pair_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= Cout) return;                        // (whole wave)
    float v = 0.f;
    for (int sidx = lane; sidx < S; sidx += 64) v += bias_part[(int64_t)sidx * Cout + c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) gbias[c] = accumulate ? gbias[c] + v : v;
    return;
  }
  __shared__ float red[4][4][16];                 // [slice][pair][tap]
  __shared__ float T[4][16];
  const int tap = threadIdx.x & 15, pr = (threadIdx.x >> 4) & 3, sl = threadIdx.x >> 6;
  const int64_t npairs = (int64_t)Clo * Chi;
  const int64_t pair = (int64_t)bx * 4 + pr;    // index into T's [lo][hi] layout
  float a0 = 0.f, a1 = 0.f;
  if (pair < npairs) {
    const float* src = part + pair * 16 + tap;
    const int64_t stride = npairs * 16;
    int s = sl;
    for (; s + 4 < S; s += 8) { a0 += src[(int64_t)s * stride]; a1 += src[(int64_t)(s + 4) * stride]; }
    for (; s < S; s += 4) a0 += src[(int64_t)s * stride];
  }
  red[sl][pr][tap] = a0 + a1;
  __syncthreads();
  if (sl == 0) T[pr][tap] = (red[0][pr][tap] + red[1][pr][tap]) + (red[2][pr][tap] + red[3][pr][tap]);
  __syncthreads();
  if (threadIdx.x < 36) {
    const int p = threadIdx.x / 9, k = threadIdx.x % 9, kh = k / 3, kw = k % 3;
    const int64_t pp = (int64_t)bx * 4 + p;
    if (pp < npairs) {
      const int lo = (int)(pp / Chi), hi = (int)(pp % Chi);
      const int co = mode == 0 ? lo : hi, ci = mode == 0 ? hi : lo;
      const int u0 = mode == 0 ? kh : 2 - kh, v0 = mode == 0 ? kw : 2 - kw;
      float r = (T[p][u0 * 4 + v0] + T[p][u0 * 4 + v0 + 1]) + (T[p][(u0 + 1) * 4 + v0] + T[p][(u0 + 1) * 4 + v0 + 1]);
      if (mode == 0) r *= 0.25f;
      float* o = gw + ((int64_t)co * Cin + ci) * 9 + k;
      *o = accumulate ? *o + r : r;
    }
  }
}

__global__ void __launch_bounds__(256) s2wgrad_reduce_fold_kernel(const float* __restrict__ part, float* __restrict__ gw, int S, int Clo,
                                                                  int Chi, int Cout, int Cin, int mode, int accumulate,
                                                                  const float* __restrict__ bias_part, float* __restrict__ gbias,
                                                                  int pair_blocks) {
  s2wgrad_reduce_fold(part, gw, S, Clo, Chi, Cout, Cin, mode, accumulate, bias_part, gbias, pair_blocks, (int)blockIdx.x);
}
// ... for the stride-2 layers of a whole backward pass in one launch (tg_s2_wgrad_reduce_batch)
constexpr int SB_MAX_ITEMS = 16;
struct FoldItem { const float* part; float* gw; const float* bias_part; float* gbias; int S, Clo, Chi, Cout, Cin, mode, accumulate, pair_blocks; };
struct FoldBatch { FoldItem item[SB_MAX_ITEMS]; int block_end[SB_MAX_ITEMS]; int n; };
__global__ void __launch_bounds__(256) s2wgrad_reduce_fold_batch_kernel(FoldBatch fb) {
  int i = 0;
  while (i + 1 < fb.n && (int)blockIdx.x >= fb.block_end[i]) ++i;            // (block-uniform)
  const int first = i == 0 ? 0 : fb.block_end[i - 1];
  const FoldItem& it = fb.item[i];
  s2wgrad_reduce_fold(it.part, it.gw, it.S, it.Clo, it.Chi, it.Cout, it.Cin, it.mode, it.accumulate, it.bias_part, it.gbias,
                      it.pair_blocks, (int)blockIdx.x - first);
}

// AvgPool2d(2) o conv3x3 = 0.25 * (transpose of the up-conv with the flipped, transposed filter): the same two kernels
// with the roles swapped.  w4[co][ci][u][v] = 0.25 * sum_{kh in S'(u), kw in S'(v)} w[co][ci][kh][kw], S' = {0},{0,1},{1,2},{2};
// its input gradient is the four-phase kernel with wp[dy][dx][ci][co][ty][tx], rows dy=0: {w2 | w1+w0}, dy=1: {w2+w1 | w0}, x 0.25.
__device__ __forceinline__ void poolconv_weights_one(const float* __restrict__ w, float* __restrict__ w4, float* __restrict__ wp, int Cout,
                                                     int Cin, int e /*co * Cin + ci*/) {
  const int n = Cout * Cin;
  if (e >= n) return;
  const int co = e / Cin, ci = e - co * Cin;
  float k[3][3];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i / 3][i % 3] = 0.25f * w[(int64_t)e * 9 + i];
  {
    float r[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { r[0][c] = k[0][c]; r[1][c] = k[0][c] + k[1][c]; r[2][c] = k[1][c] + k[2][c]; r[3][c] = k[2][c]; }
    float* o = w4 + (int64_t)e * 16;
#pragma unroll
    for (int u = 0; u < 4; ++u) { o[u * 4 + 0] = r[u][0]; o[u * 4 + 1] = r[u][0] + r[u][1]; o[u * 4 + 2] = r[u][1] + r[u][2]; o[u * 4 + 3] = r[u][2]; }
  }
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      float r[2][3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        r[0][c] = dy == 0 ? k[2][c] : k[2][c] + k[1][c];
        r[1][c] = dy == 0 ? k[1][c] + k[0][c] : k[0][c];
      }
      float* o = wp + ((int64_t)(dy * 2 + dx) * n + (int64_t)ci * Cout + co) * 4;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        o[ty * 2 + 0] = dx == 0 ? r[ty][2] : r[ty][2] + r[ty][1];
        o[ty * 2 + 1] = dx == 0 ? r[ty][1] + r[ty][0] : r[ty][0];
      }
    }
}
__global__ void __launch_bounds__(256) poolconv_weights_kernel(const float* __restrict__ w, float* __restrict__ w4, float* __restrict__ wp,
                                                               int Cout, int Cin) {
  poolconv_weights_one(w, w4, wp, Cout, Cin, blockIdx.x * 256 + threadIdx.x);
}
// ... of several layers in one launch (a network's stride-2 layers all need theirs at the start of a pass)
constexpr int FB_MAX_ITEMS = 16;
struct FormItem { const float* w; float* w4; float* wp; int Cout, Cin; };
struct FormBatch { FormItem item[FB_MAX_ITEMS]; int block_end[FB_MAX_ITEMS]; int n; };
__global__ void __launch_bounds__(256) poolconv_weights_batch_kernel(FormBatch fb) {
  int i = 0;
  while (i + 1 < fb.n && (int)blockIdx.x >= fb.block_end[i]) ++i;            // (block-uniform)
  const int first = i == 0 ? 0 : fb.block_end[i - 1];
  const FormItem& it = fb.item[i];
  poolconv_weights_one(it.w, it.w4, it.wp, it.Cout, it.Cin, ((int)blockIdx.x - first) * 256 + threadIdx.x);
}

// =========================================================================== from-RGB 1x1 composed into the first 3x3
// wc[co][c][tap] = sum_m w1c[m][c] * w3[co][m][tap],  w1c = [w1 | b1] (C x (Cimg + 1): the from-RGB bias rides on an all-ones input
// channel).  One thread per output; the backward (one launch): gw3[co][m][tap] = sum_c gwc[co][c][tap] w1c[m][c],
// gw1c[m][c] = sum_{co, tap} gwc[co][c][tap] w3[co][m][tap] -- a few thousand multiply-adds that used to be a cat, a broadcast copy,
// a batched GEMM and, backwards, two GEMMs, a sum and three accumulations.
__device__ __forceinline__ float rgb_w1c(const float* __restrict__ w1, const float* __restrict__ b1, int m, int c, int Cimg) {
  return c < Cimg ? w1[m * Cimg + c] : b1[m];
}
__global__ void __launch_bounds__(256) rgb_compose_fwd_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w3, float* __restrict__ wc, int Cout, int C,
                                                              int Cimg) {
  const int n = Cout * (Cimg + 1) * 9;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int tap = e % 9, c = (e / 9) % (Cimg + 1), co = e / (9 * (Cimg + 1));
    float acc = 0.f;
    for (int m = 0; m < C; ++m) acc = fmaf(rgb_w1c(w1, b1, m, c, Cimg), w3[(co * C + m) * 9 + tap], acc);
    wc[e] = acc;
  }
}
__global__ void __launch_bounds__(256) rgb_compose_bwd_kernel(const float* __restrict__ gwc, const float* __restrict__ w1,
                                                              const float* __restrict__ b1, const float* __restrict__ w3,
                                                              float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw3,
                                                              int Cout, int C, int Cimg, int accumulate) {
  const int n3 = Cout * C * 9, n1 = C * (Cimg + 1);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n3 + n1; e += gridDim.x * 256) {
    if (e < n3) {
      const int tap = e % 9, m = (e / 9) % C, co = e / (9 * C);
      float acc = 0.f;
      for (int c = 0; c <= Cimg; ++c) acc = fmaf(gwc[(co * (Cimg + 1) + c) * 9 + tap], rgb_w1c(w1, b1, m, c, Cimg), acc);
      gw3[e] = accumulate ? gw3[e] + acc : acc;
    } else {
      const int k = e - n3, c = k % (Cimg + 1), m = k / (Cimg + 1);
      float acc = 0.f;
      for (int co = 0; co < Cout; ++co)
        for (int tap = 0; tap < 9; ++tap) acc = fmaf(gwc[(co * (Cimg + 1) + c) * 9 + tap], w3[(co * C + m) * 9 + tap], acc);
      float* o = c < Cimg ? gw1 + m * Cimg + c : gb1 + m;
      *o = accumulate ? *o + acc : acc;
    }
  }
}

// =========================================================================== host dispatch
enum GeoId { GEO_4, GEO_8, GEO_16, GEO_X };
static inline GeoId pick_geo(int H, int W) {
  if (H == 4 && W == 4) return GEO_4;
  if (H == 8 && W == 8) return GEO_8;
  if (H == 16 && W == 16) return GEO_16;
  return GEO_X;
}
static inline int geo_tiles(GeoId g, int B, int H, int W) {
  switch (g) {
    case GEO_4: return num_tiles<G4>(B, H, W);
    case GEO_8: return num_tiles<G8>(B, H, W);
    case GEO_16: return num_tiles<G16>(B, H, W);
    default: return num_tiles<GX>(B, H, W);
  }
}

// 16-byte row loads need W % 4 == 0 and a 16-byte aligned base
static inline int plane_vec_ok(const void* p, int W) { return (W % 4 == 0) && tg_aligned16(p); }
// stride-2 family: low-resolution planes of 8x8 (4 images per tile), 16x16 and larger; workgroups needed to pay off
static inline bool s2_geo(GeoId g) { return g == GEO_8 || g == GEO_16 || g == GEO_X; }
static inline int64_t s2_min_wgs(GeoId g) { return g == GEO_8 ? 128 : 256; }
#define TG_S2_DISPATCH(g, KERNEL, ...)                                                   \
  do {                                                                                   \
    if ((g) == GEO_8) KERNEL<G8><<<grid, CT_THREADS, 0, st>>>(__VA_ARGS__);              \
    else if ((g) == GEO_16) KERNEL<G16><<<grid, CT_THREADS, 0, st>>>(__VA_ARGS__);       \
    else KERNEL<GX><<<grid, CT_THREADS, 0, st>>>(__VA_ARGS__);                           \
  } while (0)

// stride-2 transpose form: 8x8 planes that give fewer than 256 four-image tiles run as single-image K-split tiles
static bool s2_dma_ok(const void* x, const void* w, const Shape& s, int hi_channels);
static int dma_prio();
static bool s2_single_buffer();

static int wino_mode();
// AvgPool2d(2) o conv3x3 in its 9-frequency form (conv_poolwino_dma_kernel, wino.h) -> true when it took the launch
static bool try_launch_poolwino(const float* x, const float* w4, const float* bias, const float* residual, float* y, const Shape& s,
                                int vx, hipStream_t st) {
  const int mode = wino_mode();
  // (16-channel inputs: a workgroup's 64 outputs do not pay for its filter chunk -- 93 vs 63 us on 16 -> 16 @128^2, batch 128)
  if (!mode || !vx || s.Cin % WINO_CK != 0 || (mode != 2 && s.Cin < 32) || s.Cout < 16 || s.W % 4 != 0 || !tg_aligned16(y) || (residual && !tg_aligned16(residual)) ||
      (int64_t)s.B * s.Cin * s.H * s.W * 16 >= (1ll << 31))
    return false;
  const int Hh = 2 * s.H, Wh = 2 * s.W;
  const int cob = (s.Cout + 31) / 32;
  const int64_t min_wgs = (mode == 2) ? 1 : 256;
  if (Hh % 8 == 0 && Wh % 32 == 0) {
    const int tiles = num_tiles<GX>(s.B, Hh, Wh);
    if ((int64_t)tiles * cob < min_wgs) return false;
    conv_poolwino_dma_kernel<GX, 2><<<dim3(tiles, cob), CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, (tiles % 8 == 0) ? 1 : 0);
    return true;
  }
  if (Hh == 16 && Wh == 16) {
    const int tiles = num_tiles<G16>(s.B, Hh, Wh);
    if ((int64_t)tiles * cob < min_wgs) return false;
    conv_poolwino_dma_kernel<G16, 2><<<dim3(tiles, cob), CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, (tiles % 8 == 0) ? 1 : 0);
    return true;
  }
  return false;
}

static void launch_upT(GeoId g, const float* x, const float* w4, const float* bias, const float* residual, float* y, Shape s,
                       int vx, int vw, hipStream_t st) {
  if (try_launch_poolwino(x, w4, bias, residual, y, s, vx, st)) return;
  const int cot = (s.Cout + 15) / 16;
  if (vx && vw && s2_dma_ok(x, w4, s, s.Cin)) {            // LDS-DMA staged forms
    s.prio = dma_prio();
    if (g == GEO_8 && (int64_t)geo_tiles(g, s.B, s.H, s.W) * cot < 256) {
      const int t = num_tiles<G8k>(s.B, s.H, s.W);
      conv_upT_dma_kernel<G8k, true><<<dim3(t, cot), CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
      return;
    }
    const int t = geo_tiles(g, s.B, s.H, s.W);
    dim3 grid(t, cot);
    if (s2_single_buffer()) {
      if (g == GEO_8) conv_upT_dma_kernel<G8, false, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
      else if (g == GEO_16) conv_upT_dma_kernel<G16, false, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
      else conv_upT_dma_kernel<GX, false, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
      return;
    }
    if (g == GEO_8) conv_upT_dma_kernel<G8, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
    else if (g == GEO_16) conv_upT_dma_kernel<G16, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
    else conv_upT_dma_kernel<GX, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, t % 8 == 0);
    return;
  }
  if (g == GEO_8 && (int64_t)geo_tiles(g, s.B, s.H, s.W) * cot < 256) {
    conv_upT_kernel<G8k, true><<<dim3(num_tiles<G8k>(s.B, s.H, s.W), cot), CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, vx, vw);
    return;
  }
  dim3 grid(geo_tiles(g, s.B, s.H, s.W), cot);
  if (g == GEO_8) conv_upT_kernel<G8, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, vx, vw);
  else if (g == GEO_16) conv_upT_kernel<G16, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, vx, vw);
  else conv_upT_kernel<GX, false><<<grid, CT_THREADS, 0, st>>>(x, w4, bias, residual, y, s, vx, vw);
}

static void launch_upfwd(GeoId g, const float* x, const float* wp, const float* bias, const float* residual, float* y, Shape s,
                         int vx, int vw, hipStream_t st) {
  const int cot = (s.Cout + 15) / 16;
  if (vx && vw && s2_dma_ok(x, wp, s, 0)) {
    s.prio = dma_prio();
    if (g == GEO_8 && (int64_t)geo_tiles(g, s.B, s.H, s.W) * cot < 256) {
      const int t = num_tiles<G8k>(s.B, s.H, s.W);
      conv_upfwd_dma_kernel<G8k, true><<<dim3(t, cot), CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
      return;
    }
    const int t = geo_tiles(g, s.B, s.H, s.W);
    dim3 grid(t, cot);
    if (s2_single_buffer()) {
      if (g == GEO_8) conv_upfwd_dma_kernel<G8, false, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
      else if (g == GEO_16) conv_upfwd_dma_kernel<G16, false, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
      else conv_upfwd_dma_kernel<GX, false, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
      return;
    }
    if (g == GEO_8) conv_upfwd_dma_kernel<G8, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
    else if (g == GEO_16) conv_upfwd_dma_kernel<G16, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
    else conv_upfwd_dma_kernel<GX, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, t % 8 == 0);
    return;
  }
  if (g == GEO_8 && (int64_t)geo_tiles(g, s.B, s.H, s.W) * cot < 256) {
    conv_upfwd_kernel<G8k, true><<<dim3(num_tiles<G8k>(s.B, s.H, s.W), cot), CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, vx, vw);
    return;
  }
  dim3 grid(geo_tiles(g, s.B, s.H, s.W), cot);
  if (g == GEO_8) conv_upfwd_kernel<G8, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, vx, vw);
  else if (g == GEO_16) conv_upfwd_kernel<G16, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, vx, vw);
  else conv_upfwd_kernel<GX, false><<<grid, CT_THREADS, 0, st>>>(x, wp, bias, residual, y, s, vx, vw);
}

template <class G, int KS, bool DGRAD>
int launch_fwd_geo(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st) {
  const int tiles = num_tiles<G>(s.B, s.H, s.W);
  constexpr bool WK = (G::NPIX == 64);
  const int vx = plane_vec_ok(x, s.W);
  // filter rows start at multiples of (channels * KK) floats: 16-byte aligned iff that channel count is a multiple of 4
  const int vw = tg_aligned16(w) && ((DGRAD ? s.Cout : s.Cin) % 4 == 0);
  // MFMA flavour: 32x32x2 when the output-channel count fills (or nearly fills) 32-row tiles, else 16x16x4
  const bool use32 = (s.Cout % 32 == 0) || s.Cout > 48;
  if (use32) {
    bool done = false;
    if constexpr (!WK) {
      // 64-channel tiles halve the patch re-reads but also the workgroup count: only when the grid stays full
      if (s.Cout > 32 && (int64_t)tiles * ((s.Cout + 63) / 64) >= 768) {
        dim3 grid(tiles, (s.Cout + 63) / 64);
        conv_fwd_kernel<G, KS, 32, 2, DGRAD><<<grid, CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, vx, vw);
        done = true;
      }
    }
    if (!done) {
      dim3 grid(tiles, (s.Cout + 31) / 32);
      conv_fwd_kernel<G, KS, 32, 1, DGRAD><<<grid, CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, vx, vw);
    }
  } else {
    if (s.Cout > 16) {
      dim3 grid(tiles, (s.Cout + 31) / 32);
      conv_fwd_kernel<G, KS, 16, 2, DGRAD><<<grid, CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, vx, vw);
    } else {
      dim3 grid(tiles, 1);
      conv_fwd_kernel<G, KS, 16, 1, DGRAD><<<grid, CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, vx, vw);
    }
  }
  return tg_launch_status();
}

// ---- LDS-DMA kernel dispatch.  Tuning knobs are read once from the environment (development only; the defaults are the
// measured best): TG_CONV_DMA=0 disables the kernel, TG_DMA_TILE=256|512 and TG_DMA_CK=4|8 force a tile / chunk size.
struct DmaKnobs { int enable, tile, ck, ksplit, wgrad, wgrad_db, s2, db, diag, prio; };
static const DmaKnobs& dma_knobs() {
  static const DmaKnobs k = [] {
    DmaKnobs d{1, 0, 0, 1, 1, 0, 1, 1, 0, 0};
    if (const char* e = getenv("TG_DMA_PRIO")) d.prio = atoi(e);
#ifdef TG_DIAG_STAMPS
    if (const char* e = getenv("TG_DMA_DIAG")) d.diag = atoi(e) & 6;       // stage isolation: the diagnostic build only
#endif
    if (const char* e = getenv("TG_DMA_DB")) d.db = atoi(e);
    if (const char* e = getenv("TG_DMA_S2")) d.s2 = atoi(e);
    if (const char* e = getenv("TG_DMA_WGRAD")) d.wgrad = atoi(e);
    if (const char* e = getenv("TG_DMA_WGRAD_DB")) d.wgrad_db = atoi(e);
    if (const char* e = getenv("TG_DMA_KSPLIT")) d.ksplit = atoi(e);
    if (const char* e = getenv("TG_CONV_DMA")) d.enable = atoi(e);
    if (const char* e = getenv("TG_DMA_TILE")) d.tile = atoi(e);
    if (const char* e = getenv("TG_DMA_CK")) d.ck = atoi(e);
    return d;
  }();
  return k;
}

// stride-2 kernels: whole 16-byte chunks (the callers checked plane / filter alignment), channel counts that keep filter rows
// 16-byte aligned (x 16 or x 4 floats per channel pair: always), 32-bit buffer offsets; hi_channels > 0: the input is the
// high-resolution tensor (4x the plane)
static bool s2_dma_ok(const void* x, const void* w, const Shape& s, int hi_channels) {
  const DmaKnobs& k = dma_knobs();
  if (!k.enable || !k.s2) return false;
  const int64_t plane = (int64_t)s.H * s.W * (hi_channels ? 4 : 1);
  if ((int64_t)s.B * s.Cin * plane * 4 >= (1ll << 31) || (int64_t)s.Cin * s.Cout * 64 * 4 >= (1ll << 31)) return false;
  return s.W % 4 == 0 || hi_channels;        // low-resolution rows of whole chunks (high-resolution rows: 2 W, the caller's vx)
}

static bool s2_single_buffer() { return dma_knobs().db == 0; }
static int dma_prio() { return dma_knobs().prio; }

template <class G, int CK, bool DGRAD>
static bool launch_dma_geo(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st) {
  const int64_t tiles = num_tiles<G>(s.B, s.H, s.W);
  const int swz = ((tiles % 8 == 0) ? 1 : 0) | dma_knobs().diag | (dma_knobs().prio ? 32 : 0);
  const bool use32 = (s.Cout % 32 == 0) || s.Cout > 48;
  // output-channel tile: as wide as the grid allows (one workgroup per CU at the very least)
  const bool sb = dma_knobs().db == 0;
  if (use32 && s.Cout > 32 && tiles * ((s.Cout + 63) / 64) >= 512) {
    conv_dma_kernel<G, 32, 2, CK, DGRAD><<<dim3(tiles, (s.Cout + 63) / 64), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
  } else if (use32 && tiles * ((s.Cout + 31) / 32) >= 256) {
    if (sb) conv_dma_kernel<G, 32, 1, CK, DGRAD, false><<<dim3(tiles, (s.Cout + 31) / 32), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
    else conv_dma_kernel<G, 32, 1, CK, DGRAD><<<dim3(tiles, (s.Cout + 31) / 32), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
  } else if (tiles * ((s.Cout + 15) / 16) >= 192) {
    if (sb) conv_dma_kernel<G, 16, 1, CK, DGRAD, false><<<dim3(tiles, (s.Cout + 15) / 16), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
    else conv_dma_kernel<G, 16, 1, CK, DGRAD><<<dim3(tiles, (s.Cout + 15) / 16), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
  } else {
    return false;                      // too few tiles: the K-split variants of conv_fwd_kernel fill the chip better
  }
  return true;
}

template <class G, bool DGRAD>
static bool launch_dma_ksplit(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st) {
  const int64_t tiles = num_tiles<G>(s.B, s.H, s.W);
  const int swz = ((tiles % 8 == 0) ? 1 : 0) | (dma_knobs().prio ? 32 : 0);
  // 32-channel tiles only while they still give every CU more than one workgroup: with one (or less) per CU nothing
  // overlaps a workgroup's DMA waits, and twice as many 16-channel workgroups win (batch 64: 128->128 @4^2 15.9 -> 10.6 us,
  // 64->128 @8^2 dgrad 15.5 -> 10.2 us, 128->128 @8^2 16.6 -> 16.0 us; 256->256 @8^2, 512 workgroups, stays on 32)
  if ((s.Cout % 32 == 0 || s.Cout > 48) && tiles * ((s.Cout + 31) / 32) > 256)
    conv_dma_kernel<G, 32, 1, 16, DGRAD><<<dim3(tiles, (s.Cout + 31) / 32), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
  else
    conv_dma_kernel<G, 16, 1, 16, DGRAD><<<dim3(tiles, (s.Cout + 15) / 16), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, swz);
  return true;
}

// -> true when the LDS-DMA kernel took the launch
template <bool DGRAD>
static bool try_launch_dma(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st,
                           int* rc) {
  const DmaKnobs& k = dma_knobs();
  if (!k.enable) return false;
  if (s.W % 4 != 0 || !tg_aligned16(x) || !tg_aligned16(w) || s.Cin % 4 != 0 || s.Cout % 4 != 0) return false;
  if ((int64_t)s.B * s.Cin * s.H * s.W * 4 >= (1ll << 31) || (int64_t)s.Cin * s.Cout * 36 >= (1ll << 31)) return false;
  // channel chunk: 4 keeps more workgroups resident (17 KB of LDS) and wins on the short-K layers; 8 halves the barriers
  const bool ck4 = (k.ck == 4) || (k.ck == 0 && s.Cin <= 16);
  bool took = false;
  if (s.H % 16 == 0 && s.W % 32 == 0 && k.tile == 512) {
    took = ck4 ? launch_dma_geo<GX2, 4, DGRAD>(x, w, bias, residual, y, s, st) : launch_dma_geo<GX2, 8, DGRAD>(x, w, bias, residual, y, s, st);
  } else if (s.H % 8 == 0 && s.W % 32 == 0) {
    took = ck4 ? launch_dma_geo<GX, 4, DGRAD>(x, w, bias, residual, y, s, st) : launch_dma_geo<GX, 8, DGRAD>(x, w, bias, residual, y, s, st);
  } else if (s.H == 16 && s.W == 16) {
    took = ck4 ? launch_dma_geo<G16, 4, DGRAD>(x, w, bias, residual, y, s, st) : launch_dma_geo<G16, 8, DGRAD>(x, w, bias, residual, y, s, st);
  }
  if (!took && s.Cin >= 16 && k.ksplit) {
    // small planes with long K: 64-pixel tiles, the four waves split each 16-channel chunk (one workgroup per CU is the
    // normal case here, so the chunk DMA under the MFMAs is the only overlap there is)
    if (s.H == 8 && s.W == 8) took = launch_dma_ksplit<G8k, DGRAD>(x, w, bias, residual, y, s, st);
    else if (s.H == 4 && s.W == 4) took = launch_dma_ksplit<G4k, DGRAD>(x, w, bias, residual, y, s, st);
    else if (s.H == 16 && s.W == 16) took = launch_dma_ksplit<G16k, DGRAD>(x, w, bias, residual, y, s, st);
  }
  if (took) *rc = tg_launch_status();
  return took;
}

static int conv1x1_mode() {
  static const int v = [] { const char* e = getenv("TG_CONV_1X1"); return e ? atoi(e) : 1; }();
  return v;
}

// ---- Winograd F(2x2, 3x3) dispatch (wino.h).  TG_CONV_WINO=0 disables it, 2 takes every eligible shape (tests), 1 (default)
// only the launches that fill the chip: 64-tile workgroups on 8x8 / 4x4 planes give too few workgroups at the step's batch.
static int wino_mode() {
  static const int v = [] { const char* e = getenv("TG_CONV_WINO"); return e ? atoi(e) : 1; }();
  return v;
}
template <class G, bool DGRAD>
static int launch_wino_geo(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st,
                           int64_t min_wgs) {
  const int tiles = num_tiles<G>(s.B, s.H, s.W);
  const int flags = (tiles % 8 == 0) ? 1 : 0;
  if (s.Cin == 4) {                      // the composed from-RGB layer: one 4-channel chunk
    if (s.Cout > 16 || (int64_t)tiles < min_wgs) return -1;
    conv_wino_dma_kernel<G, 1, DGRAD, 4><<<dim3(tiles, 1), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, flags);
    return tg_launch_status();
  }
  if (s.Cout > 16 && (int64_t)tiles * ((s.Cout + 31) / 32) >= min_wgs)
    conv_wino_dma_kernel<G, 2, DGRAD><<<dim3(tiles, (s.Cout + 31) / 32), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, flags);
  else if ((int64_t)tiles * ((s.Cout + 15) / 16) >= min_wgs)
    conv_wino_dma_kernel<G, 1, DGRAD><<<dim3(tiles, (s.Cout + 15) / 16), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, flags);
  else
    return -1;
  return tg_launch_status();
}
// small planes (8x8: one image per workgroup, 4x4: four): the four waves split every 16-channel chunk, partial outputs summed in LDS
template <class G, bool DGRAD>
static int launch_wino_ksplit(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st,
                              int64_t min_wgs) {
  if (s.Cin % 16 != 0) return -1;
  const int tiles = num_tiles<G>(s.B, s.H, s.W);
  const int flags = (tiles % 8 == 0) ? 1 : 0;
  if (s.Cout > 16 && (int64_t)tiles * ((s.Cout + 31) / 32) >= 256 && s.B % 2 == 0)
    conv_wino_dma_kernel<G, 2, DGRAD, 16, true><<<dim3(tiles, (s.Cout + 31) / 32), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, flags);
  else if ((int64_t)tiles * ((s.Cout + 15) / 16) >= min_wgs)
    conv_wino_dma_kernel<G, 1, DGRAD, 16, true><<<dim3(tiles, (s.Cout + 15) / 16), CT_THREADS, 0, st>>>(x, w, bias, residual, y, s, flags);
  else
    return -1;
  return tg_launch_status();
}
// -> true when the Winograd kernel took the launch
template <bool DGRAD>
static bool try_launch_wino(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st,
                            int* rc) {
  const int mode = wino_mode();
  if (!mode) return false;
  // channel counts: whole 8-channel chunks (or the 4-channel image layer); >= 4 output channels (a 16-wide block is masked)
  if (s.os != 1 || (s.Cin % WINO_CK != 0 && s.Cin != 4) || s.Cout < 4 || s.W % 4 != 0 || !tg_aligned16(x) || !tg_aligned16(y) ||
      (residual && !tg_aligned16(residual)) || (int64_t)s.B * s.Cin * s.H * s.W * 4 >= (1ll << 31))
    return false;
  const int64_t min_wgs = (mode == 2) ? 1 : 256;
  // (64-tile workgroups on 8x8 / 4x4 planes give too few workgroups at the step's batch: mode 2 only)
  int r = -1;
  if (s.H % 8 == 0 && s.W % 32 == 0) r = launch_wino_geo<GX, DGRAD>(x, w, bias, residual, y, s, st, min_wgs);
  // (16x16 planes at batch 64 -- exactly one workgroup per CU -- measured neutral to slightly slower inside the step: from two per CU)
  else if (s.H == 16 && s.W == 16) r = launch_wino_geo<G16, DGRAD>(x, w, bias, residual, y, s, st, mode == 2 ? 1 : 512);
  // 8x8 / 4x4 planes, K-split form: measured SLOWER than the direct K-split kernels (128 -> 128 @8^2, batch 64 / 128: 20.5 / 39.1 us
  // against 17.1 / 28.8; the step 9.29 against 9.21 ms) -- a workgroup transforms a whole filter chunk for 16 tiles; mode 2 only
  else if (mode == 2 && s.H == 8 && s.W == 8) r = launch_wino_ksplit<G8k, DGRAD>(x, w, bias, residual, y, s, st, min_wgs);
  else if (mode == 2 && s.H == 4 && s.W == 4) r = launch_wino_ksplit<G4k, DGRAD>(x, w, bias, residual, y, s, st, min_wgs);
  if (r < 0) return false;
  *rc = r;
  return true;
}

template <int KS, bool DGRAD>
int launch_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, Shape s, hipStream_t st) {
  if constexpr (KS == 3) {
    int rc = 0;
    if (try_launch_wino<DGRAD>(x, w, bias, residual, y, s, st, &rc)) return rc;
    if (try_launch_dma<DGRAD>(x, w, bias, residual, y, s, st, &rc)) return rc;
  }
  // small layers: if 256-pixel tiles cannot even give every other CU a workgroup, use 64-pixel tiles whose waves
  // split K (4x the workgroups, each wave 1/4 of the k-groups).  Measured at batch 64: 128->128 @ 16^2 (128
  // such workgroups) runs 65 us unsplit vs 74 us split; @ 8^2 (32) 55 us vs 24 us.
  if constexpr (KS == 1) {
    // Measured at batch 64 (tools/bench_conv.py, us, tiled -> streaming): 3->16 @128^2 26.3 -> 15.0; 16->32 @128^2 (dgrad of
    // 32->16) 43.4 -> 31.7; 16->32 @64^2 15.6 -> 13.3; but 64->32 @64^2 22.3 -> 26.5 and 128->64 @32^2 21.1 -> 28.2: with
    // many input channels a wave's serial chain of k-steps is longer than the tiled kernel's four waves sharing a tile.
    // So: at most 32 input channels, and the few-output-channel VALU kernel keeps its shapes.  (The step as a whole does
    // not move -- these layers sit at the bandwidth their tensors allow; TG_CONV_1X1 = 0 / 2: never / wherever possible.)
    const int mode = conv1x1_mode();
    if (mode != 0 && conv1x1_mfma_ok(x, residual, y, s) && (mode == 2 || (!conv1x1_direct_ok(s) && s.Cin <= 32)))
      return launch_conv1x1_mfma(x, w, bias, residual, y, s, DGRAD, st);
    if (conv1x1_direct_ok(s)) return launch_conv1x1(x, w, bias, residual, y, s, DGRAD, st);
  }
  const GeoId g = pick_geo(s.H, s.W);
  const int64_t wgs256 = (int64_t)geo_tiles(g, s.B, s.H, s.W) * ((s.Cout + 63) / 64);
  const bool ksplit = (g != GEO_X) && wgs256 < 128 && s.Cin >= 16;
  switch (g) {
    case GEO_4: return ksplit ? launch_fwd_geo<G4k, KS, DGRAD>(x, w, bias, residual, y, s, st) : launch_fwd_geo<G4, KS, DGRAD>(x, w, bias, residual, y, s, st);
    case GEO_8: return ksplit ? launch_fwd_geo<G8k, KS, DGRAD>(x, w, bias, residual, y, s, st) : launch_fwd_geo<G8, KS, DGRAD>(x, w, bias, residual, y, s, st);
    case GEO_16: return ksplit ? launch_fwd_geo<G16k, KS, DGRAD>(x, w, bias, residual, y, s, st) : launch_fwd_geo<G16, KS, DGRAD>(x, w, bias, residual, y, s, st);
    default: return launch_fwd_geo<GX, KS, DGRAD>(x, w, bias, residual, y, s, st);
  }
}

struct WgPlan {
  int tiles, co_tiles, ci_chunks, S, mtw;
};
static inline WgPlan wgrad_plan(int B, int Cin, int Cout, int H, int W, int ks) {
  WgPlan p;
  p.tiles = geo_tiles(pick_geo(H, W), B, H, W);
  p.mtw = (Cout > 16 && H * W == 64) ? 2 : 1;   // 16-channel tiles keep LDS at 42 KB (3 workgroups/CU): faster except on the tiny 8x8 / 4x4 layers
  p.co_tiles = (Cout + 16 * p.mtw - 1) / (16 * p.mtw);
  const int ckw = (ks == 3) ? WgCfg<3>::CKW : WgCfg<1>::CKW;
  p.ci_chunks = (Cin + ckw - 1) / ckw;
  const int groups = p.co_tiles * p.ci_chunks;
  int S = 768 / groups;
  const int smax = p.tiles >= 4096 ? 512 : 256;      // measured: two workgroups per CU only pay off on the 128^2 planes
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  if (S > p.tiles) S = p.tiles;
  p.S = S;
  return p;
}

template <class G, int KS>
int launch_wgrad_geo(const float* x, const float* gy, float* part, float* bias_part, Shape s, const WgPlan& p, hipStream_t st) {
  dim3 grid(p.S, p.co_tiles, p.ci_chunks);
  const int vx = plane_vec_ok(x, s.W), vg = plane_vec_ok(gy, s.W);
  if constexpr (KS == 3 && G::NPIX == 256) {
    // LDS-DMA staging: needs whole 16-byte chunks (W % 4, aligned planes) and 32-bit buffer offsets
    const DmaKnobs& k = dma_knobs();
    const bool small = (int64_t)s.B * (s.Cin > s.Cout ? s.Cin : s.Cout) * s.H * s.W * 4 < (1ll << 31);
    constexpr int CKW = WgCfg<3>::CKW;
    constexpr int buf1 = (dma_pad(CKW * DPatch<G>::CPC) + dma_pad(16 * WGD_GYQ)) * 16, buf2 = (dma_pad(CKW * DPatch<G>::CPC) + dma_pad(32 * WGD_GYQ)) * 16;   // bytes, MTW 1 / 2
    if (k.enable && k.wgrad && vx && vg && small) {
      s.prio = k.prio;
      // (two buffers -- the next tile in flight under this tile's MFMAs -- measured slower at every shape of the 128 px step:
      // they halve the workgroups per CU; kept behind TG_DMA_WGRAD_DB=1 for wider layers)
      const bool db = k.wgrad_db == 1 || (k.wgrad_db < 0 && p.tiles >= 2 * p.S && (int64_t)p.S * p.co_tiles * p.ci_chunks <= 600);
      if (s.Cin <= 4 && p.mtw == 1) {
        // the composed from-RGB layer (3 image channels + the bias channel): 36 (ci, tap) columns, not 144
        conv_wgrad_dma_kernel<G, 1, 4, false><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S);
        return tg_launch_status();
      }
      if (p.mtw == 2) {
        if constexpr (2 * buf2 <= 160 * 1024) {
          if (db) { conv_wgrad_dma_kernel<G, 2, CKW, true><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S); return tg_launch_status(); }
        }
        if constexpr (buf2 <= 160 * 1024) {
          conv_wgrad_dma_kernel<G, 2, CKW, false><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S);
          return tg_launch_status();
        }
      } else {
        if constexpr (2 * buf1 <= 160 * 1024) {
          if (db) { conv_wgrad_dma_kernel<G, 1, CKW, true><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S); return tg_launch_status(); }
        }
        if constexpr (buf1 <= 160 * 1024) {
          conv_wgrad_dma_kernel<G, 1, CKW, false><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S);
          return tg_launch_status();
        }
      }
    }
  }
  if constexpr (KS == 1) {
    if (conv1x1_wgrad_stream_ok(x, gy, s)) return launch_conv1x1_wgrad_stream(x, gy, part, bias_part, s, p.S, st);
  }
  if (p.mtw == 2) conv_wgrad_kernel<G, KS, 2><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S, vx, vg);
  else conv_wgrad_kernel<G, KS, 1><<<grid, CT_THREADS, 0, st>>>(x, gy, part, bias_part, s, p.tiles, p.S, vx, vg);
  return tg_launch_status();
}

template <int KS>
int launch_wgrad(const float* x, const float* gy, float* part, float* bias_part, Shape s, const WgPlan& p, hipStream_t st) {
  switch (pick_geo(s.H, s.W)) {
    case GEO_4: return launch_wgrad_geo<G4, KS>(x, gy, part, bias_part, s, p, st);
    case GEO_8: return launch_wgrad_geo<G8, KS>(x, gy, part, bias_part, s, p, st);
    case GEO_16: return launch_wgrad_geo<G16, KS>(x, gy, part, bias_part, s, p, st);
    default: return launch_wgrad_geo<GX, KS>(x, gy, part, bias_part, s, p, st);
  }
}

static inline int check_shape(int B, int Cin, int Cout, int H, int W, int ks) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return TG_EINVAL;
  if (ks != 1 && ks != 3) return TG_EUNSUPPORTED;
  if ((int64_t)B * (Cin > Cout ? Cin : Cout) * H * W >= (1ll << 40)) return TG_EUNSUPPORTED;
  // the kernels address with a 64-bit base per tile + 32-bit lane byte offsets built from 24-bit multiplies:
  // a tile's image group (16 / 4 / 1 images for 4x4 / 8x8 / larger planes) must span < 2^30 elements
  const int64_t ni = (H == 4 && W == 4) ? 16 : (H == 8 && W == 8) ? 4 : 1;
  const int64_t cmax = Cin > Cout ? Cin : Cout;
  if ((int64_t)H * W >= (1 << 24) || cmax * 9 >= (1 << 24) || ni * cmax * H * W >= (1ll << 30)) return TG_EUNSUPPORTED;
  return TG_OK;
}

}  // namespace

extern "C" {

int tg_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int B, int Cin, int Cout,
                  int H, int W, int ks, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(w); TG_CHECK_PTR(y);
  if (int rc = check_shape(B, Cin, Cout, H, W, ks)) return rc;
  Shape s{B, Cin, Cout, H, W};
  return ks == 3 ? launch_fwd<3, false>(x, w, bias, residual, y, s, tg_stream(stream))
                 : launch_fwd<1, false>(x, w, bias, residual, y, s, tg_stream(stream));
}

int tg_conv2d_fwd_up2res(const float* x, const float* w, const float* bias, const float* residual_lo, float* y, int B, int Cin,
                         int Cout, int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(w); TG_CHECK_PTR(y); TG_CHECK_PTR(residual_lo);
  if (int rc = check_shape(B, Cin, Cout, H, W, 3)) return rc;
  if ((H & 1) || (W & 1)) return TG_EUNSUPPORTED;
  Shape s{B, Cin, Cout, H, W};
  s.res_up = 1;
  return launch_fwd<3, false>(x, w, bias, residual_lo, y, s, tg_stream(stream));
}

int tg_upconv3x3_weights(const float* w, float* wp, int Cout, int Cin, void* stream) {
  TG_CHECK_PTR(w); TG_CHECK_PTR(wp); TG_CHECK_POS(Cout); TG_CHECK_POS(Cin);
  const int n = Cout * Cin;
  upconv_weights_kernel<<<(n + 255) / 256, 256, 0, tg_stream(stream)>>>(w, wp, n);
  return tg_launch_status();
}

int tg_upconv3x3_fwd(const float* a, const float* wp, const float* bias, const float* residual, float* y, int B, int Cin, int Cout,
                     int H, int W, void* stream) {
  TG_CHECK_PTR(a); TG_CHECK_PTR(wp); TG_CHECK_PTR(y);
  if (int rc = check_shape(B, Cin, Cout, 2 * H, 2 * W, 3)) return rc;      // the output plane is the large one
  hipStream_t st = tg_stream(stream);
  {
    // one kernel for all four phases when the low-resolution plane gives enough 256-pixel tiles
    Shape s{B, Cin, Cout, H, W};
    const GeoId g = pick_geo(H, W);
    const int cot = (Cout + 15) / 16;
    const bool aligned8 = (((uintptr_t)y & 7) == 0) && (!residual || ((uintptr_t)residual & 7) == 0);
    if (s2_geo(g) && aligned8 && (int64_t)geo_tiles(g, B, H, W) * cot >= s2_min_wgs(g)) {
      const int vx = plane_vec_ok(a, W), vw = tg_aligned16(wp);
      launch_upfwd(g, a, wp, bias, residual, y, s, vx, vw, st);
      return tg_launch_status();
    }
  }
  for (int ph = 0; ph < 4; ++ph) {
    Shape s{B, Cin, Cout, H, W};
    s.oy = s.py = ph >> 1;
    s.ox = s.px = ph & 1;
    s.os = 2;
    if (int rc = launch_fwd<2, false>(a, wp + (size_t)ph * Cout * Cin * 4, bias, residual, y, s, st)) return rc;
  }
  return TG_OK;
}

int tg_upconv3x3_weights_t(const float* w, float* w4t, int Cout, int Cin, void* stream) {
  TG_CHECK_PTR(w); TG_CHECK_PTR(w4t); TG_CHECK_POS(Cout); TG_CHECK_POS(Cin);
  upconvT_weights_kernel<<<(Cout * Cin + 255) / 256, 256, 0, tg_stream(stream)>>>(w, w4t, Cout, Cin);
  return tg_launch_status();
}

int tg_upconv3x3_weights_pair(const float* w, float* wp, float* w4t, int Cout, int Cin, void* stream) {
  TG_CHECK_PTR(w); TG_CHECK_PTR(wp); TG_CHECK_PTR(w4t); TG_CHECK_POS(Cout); TG_CHECK_POS(Cin);
  upconv_weights_pair_kernel<<<(Cout * Cin + 255) / 256, 256, 0, tg_stream(stream)>>>(w, wp, w4t, Cout, Cin);
  return tg_launch_status();
}

int tg_upconv3x3_dgrad_supported(int B, int Cin, int Cout, int H, int W) {
  const GeoId g = pick_geo(H, W);
  return s2_geo(g) && check_shape(B, Cin, Cout, 2 * H, 2 * W, 3) == TG_OK &&
         (int64_t)geo_tiles(g, B, H, W) * ((Cin + 15) / 16) >= s2_min_wgs(g);
}

int tg_upconv3x3_dgrad(const float* gy, const float* w4t, float* ga, int B, int Cin, int Cout, int H, int W, void* stream) {
  TG_CHECK_PTR(gy); TG_CHECK_PTR(w4t); TG_CHECK_PTR(ga);
  if (!tg_upconv3x3_dgrad_supported(B, Cin, Cout, H, W)) return TG_EUNSUPPORTED;
  // a convolution whose input channels are the forward's Cout (gy) and output channels its Cin (ga)
  Shape s{B, Cout, Cin, H, W};
  const GeoId g = pick_geo(H, W);
  const int vx = plane_vec_ok(gy, 2 * W), vw = tg_aligned16(w4t);
  launch_upT(g, gy, w4t, nullptr, nullptr, ga, s, vx, vw, tg_stream(stream));
  return tg_launch_status();
}

static inline int s2_splits(int tiles, int lo_tiles, int hi_chunks) {
  int S = 768 / (lo_tiles * hi_chunks);
  const int smax = tiles >= 4096 ? 512 : 256;
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  if (S > tiles) S = tiles;
  return S;
}
// workspace: S partials of T, then S partials of the bias gradient (over the wider of the two channel counts)
static size_t s2_workspace(int B, int Clo, int Chi, int H, int W) {
  const GeoId g = pick_geo(H, W);
  const int S = s2_splits(geo_tiles(g, B, H, W), (Clo + 15) / 16, (Chi + S2_CKW - 1) / S2_CKW);
  return ((size_t)S * Clo * Chi * 16 + (size_t)S * (Clo > Chi ? Clo : Chi)) * sizeof(float);
}
struct S2Plan { int S, tiles, lo_tiles, hi_chunks; };
static S2Plan s2_plan(int B, int Clo, int Chi, int H, int W) {
  const GeoId g = pick_geo(H, W);
  S2Plan p;
  p.tiles = geo_tiles(g, B, H, W);
  p.lo_tiles = (Clo + 15) / 16;
  p.hi_chunks = (Chi + S2_CKW - 1) / S2_CKW;
  p.S = s2_splits(p.tiles, p.lo_tiles, p.hi_chunks);
  return p;
}
// stage 1: per-workgroup partials of T (and of the bias gradient, want_bias) into ws
static int s2_wgrad_stage1(const float* hi, const float* lo, float* ws, size_t ws_bytes, int B, int Clo, int Chi, int H, int W, int mode,
                           int want_bias, hipStream_t st) {
  const GeoId g = pick_geo(H, W);
  if (!s2_geo(g) || check_shape(B, Clo, Chi, 2 * H, 2 * W, 3) != TG_OK) return TG_EUNSUPPORTED;
  if (ws_bytes < s2_workspace(B, Clo, Chi, H, W)) return TG_EWORKSPACE;
  const S2Plan p = s2_plan(B, Clo, Chi, H, W);
  const int tiles = p.tiles, S = p.S;
  Shape s{B, Chi, Clo, H, W};
  dim3 grid(S, p.lo_tiles, p.hi_chunks);
  float* bias_part = want_bias ? ws + (size_t)S * Clo * Chi * 16 : nullptr;      // [S][Cout]: Cout == Clo (mode 0) or Chi (mode 1)
  const int vh = plane_vec_ok(hi, 2 * W), vl = plane_vec_ok(lo, W);
  if (vh && vl && dma_knobs().enable && dma_knobs().wgrad && (int64_t)B * (Chi > Clo ? Chi : Clo) * H * W * 16 < (1ll << 31)) {
    s.prio = dma_knobs().prio;
    TG_S2_DISPATCH(g, conv_wgrad_s2_dma_kernel, hi, lo, ws, s, tiles, S, bias_part, mode);
  } else {
    TG_S2_DISPATCH(g, conv_wgrad_s2_kernel, hi, lo, ws, s, tiles, S, vh, vl, bias_part, mode);
  }
  return tg_launch_status();
}
static FoldItem s2_fold_item(const float* ws, float* gw, float* gbias, int B, int Clo, int Chi, int H, int W, int Cout, int Cin, int mode,
                             int accumulate) {
  const S2Plan p = s2_plan(B, Clo, Chi, H, W);
  FoldItem it;
  it.part = ws; it.gw = gw; it.gbias = gbias;
  it.bias_part = gbias ? ws + (size_t)p.S * Clo * Chi * 16 : nullptr;
  it.S = p.S; it.Clo = Clo; it.Chi = Chi; it.Cout = Cout; it.Cin = Cin; it.mode = mode; it.accumulate = accumulate;
  it.pair_blocks = (Clo * Chi + 3) / 4;
  return it;
}
static int s2_wgrad(const float* hi, const float* lo, float* gw, float* gbias, float* ws, size_t ws_bytes, int B, int Clo, int Chi, int H,
                    int W, int Cout, int Cin, int mode, int accumulate, hipStream_t st) {
  if (int rc = s2_wgrad_stage1(hi, lo, ws, ws_bytes, B, Clo, Chi, H, W, mode, gbias != nullptr, st)) return rc;
  const FoldItem it = s2_fold_item(ws, gw, gbias, B, Clo, Chi, H, W, Cout, Cin, mode, accumulate);
  s2wgrad_reduce_fold_kernel<<<it.pair_blocks + (gbias ? (Cout + 3) / 4 : 0), 256, 0, st>>>(it.part, gw, it.S, Clo, Chi, Cout, Cin, mode,
                                                                                         accumulate, it.bias_part, gbias, it.pair_blocks);
  return tg_launch_status();
}

int tg_poolconv3x3_wgrad_partials(const float* x, const float* gy, float* workspace, size_t workspace_bytes, int B, int Cin, int Cout,
                                  int H, int W, int want_bias, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(gy); TG_CHECK_PTR(workspace);
  return s2_wgrad_stage1(x, gy, workspace, workspace_bytes, B, Cout, Cin, H, W, 0, want_bias, tg_stream(stream));
}
int tg_upconv3x3_wgrad_partials(const float* a, const float* gy, float* workspace, size_t workspace_bytes, int B, int Cin, int Cout,
                                int H, int W, int want_bias, void* stream) {
  TG_CHECK_PTR(a); TG_CHECK_PTR(gy); TG_CHECK_PTR(workspace);
  return s2_wgrad_stage1(gy, a, workspace, workspace_bytes, B, Cin, Cout, H, W, 1, want_bias, tg_stream(stream));
}
int tg_s2_wgrad_reduce_batch(const tg_host_i64* items, int n_items, void* stream) {
  if (n_items < 0) return TG_EINVAL;
  if (n_items == 0) return TG_OK;
  TG_CHECK_PTR(items);
  hipStream_t st = tg_stream(stream);
  for (int base = 0; base < n_items; base += SB_MAX_ITEMS) {
    FoldBatch fb;
    fb.n = n_items - base < SB_MAX_ITEMS ? n_items - base : SB_MAX_ITEMS;
    int blocks = 0;
    for (int i = 0; i < fb.n; ++i) {
      const tg_host_i64* it = items + (size_t)(base + i) * TG_WGRAD_ITEM_FIELDS;
      const int B = (int)it[3], Cin = (int)it[4], Cout = (int)it[5], H = (int)it[6], W = (int)it[7], mode = (int)it[8];
      if (it[0] == 0 || it[1] == 0 || (mode != 0 && mode != 1)) return TG_EINVAL;
      const int Clo = mode == 0 ? Cout : Cin, Chi = mode == 0 ? Cin : Cout;
      if (!s2_geo(pick_geo(H, W)) || check_shape(B, Clo, Chi, 2 * H, 2 * W, 3) != TG_OK) return TG_EUNSUPPORTED;
      fb.item[i] = s2_fold_item(reinterpret_cast<const float*>(it[0]), reinterpret_cast<float*>(it[1]), reinterpret_cast<float*>(it[2]),
                                B, Clo, Chi, H, W, Cout, Cin, mode, (int)it[9]);
      blocks += fb.item[i].pair_blocks + (fb.item[i].gbias ? (Cout + 3) / 4 : 0);
      fb.block_end[i] = blocks;
    }
    for (int i = fb.n; i < SB_MAX_ITEMS; ++i) { fb.item[i] = fb.item[0]; fb.block_end[i] = blocks; }
    s2wgrad_reduce_fold_batch_kernel<<<blocks, 256, 0, st>>>(fb);
  }
  return tg_launch_status();
}

size_t tg_poolconv3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  return s2_workspace(B, Cout, Cin, H, W);
}
int tg_poolconv3x3_wgrad(const float* x, const float* gy, float* gw, float* workspace, size_t workspace_bytes, int B, int Cin,
                         int Cout, int H, int W, int accumulate, float* gbias, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(gy); TG_CHECK_PTR(gw); TG_CHECK_PTR(workspace);
  return s2_wgrad(x, gy, gw, gbias, workspace, workspace_bytes, B, Cout, Cin, H, W, Cout, Cin, 0, accumulate, tg_stream(stream));
}
size_t tg_upconv3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  return s2_workspace(B, Cin, Cout, H, W);
}
int tg_upconv3x3_wgrad(const float* a, const float* gy, float* gw, float* workspace, size_t workspace_bytes, int B, int Cin,
                       int Cout, int H, int W, int accumulate, float* gbias, void* stream) {
  TG_CHECK_PTR(a); TG_CHECK_PTR(gy); TG_CHECK_PTR(gw); TG_CHECK_PTR(workspace);
  return s2_wgrad(gy, a, gw, gbias, workspace, workspace_bytes, B, Cin, Cout, H, W, Cout, Cin, 1, accumulate, tg_stream(stream));
}

int tg_rgb_compose_fwd(const float* w1, const float* b1, const float* w3, float* wc, int Cout, int C, int Cimg, void* stream) {
  TG_CHECK_PTR(w1); TG_CHECK_PTR(b1); TG_CHECK_PTR(w3); TG_CHECK_PTR(wc); TG_CHECK_POS(Cout); TG_CHECK_POS(C); TG_CHECK_POS(Cimg);
  const int n = Cout * (Cimg + 1) * 9;
  rgb_compose_fwd_kernel<<<(n + 255) / 256, 256, 0, tg_stream(stream)>>>(w1, b1, w3, wc, Cout, C, Cimg);
  return tg_launch_status();
}
int tg_rgb_compose_bwd(const float* gwc, const float* w1, const float* b1, const float* w3, float* gw1, float* gb1, float* gw3, int Cout,
                       int C, int Cimg, int accumulate, void* stream) {
  TG_CHECK_PTR(gwc); TG_CHECK_PTR(w1); TG_CHECK_PTR(b1); TG_CHECK_PTR(w3); TG_CHECK_PTR(gw1); TG_CHECK_PTR(gb1); TG_CHECK_PTR(gw3);
  TG_CHECK_POS(Cout); TG_CHECK_POS(C); TG_CHECK_POS(Cimg);
  const int n = Cout * C * 9 + C * (Cimg + 1);
  rgb_compose_bwd_kernel<<<(n + 255) / 256, 256, 0, tg_stream(stream)>>>(gwc, w1, b1, w3, gw1, gb1, gw3, Cout, C, Cimg, accumulate);
  return tg_launch_status();
}

int tg_poolconv3x3_weights_batch(const tg_host_i64* items, int n_items, void* stream) {
  if (n_items < 0) return TG_EINVAL;
  if (n_items == 0) return TG_OK;
  TG_CHECK_PTR(items);
  hipStream_t st = tg_stream(stream);
  for (int base = 0; base < n_items; base += FB_MAX_ITEMS) {
    FormBatch fb;
    fb.n = n_items - base < FB_MAX_ITEMS ? n_items - base : FB_MAX_ITEMS;
    int blocks = 0;
    for (int i = 0; i < fb.n; ++i) {
      const tg_host_i64* it = items + (size_t)(base + i) * TG_FORM_ITEM_FIELDS;
      const int Cout = (int)it[3], Cin = (int)it[4];
      if (it[0] == 0 || it[1] == 0 || it[2] == 0 || Cout <= 0 || Cin <= 0) return TG_EINVAL;
      fb.item[i] = FormItem{reinterpret_cast<const float*>(it[0]), reinterpret_cast<float*>(it[1]), reinterpret_cast<float*>(it[2]), Cout, Cin};
      blocks += (Cout * Cin + 255) / 256;
      fb.block_end[i] = blocks;
    }
    for (int i = fb.n; i < FB_MAX_ITEMS; ++i) { fb.item[i] = fb.item[0]; fb.block_end[i] = blocks; }
    poolconv_weights_batch_kernel<<<blocks, 256, 0, st>>>(fb);
  }
  return tg_launch_status();
}

int tg_poolconv3x3_weights(const float* w, float* w4, float* wp, int Cout, int Cin, void* stream) {
  TG_CHECK_PTR(w); TG_CHECK_PTR(w4); TG_CHECK_PTR(wp); TG_CHECK_POS(Cout); TG_CHECK_POS(Cin);
  poolconv_weights_kernel<<<(Cout * Cin + 255) / 256, 256, 0, tg_stream(stream)>>>(w, w4, wp, Cout, Cin);
  return tg_launch_status();
}

int tg_poolconv3x3_supported(int B, int Cin, int Cout, int H, int W) {
  const GeoId g = pick_geo(H, W);
  if (!s2_geo(g) || check_shape(B, Cin, Cout, 2 * H, 2 * W, 3) != TG_OK) return 0;
  const int64_t tiles = geo_tiles(g, B, H, W);
  return tiles * ((Cout + 15) / 16) >= s2_min_wgs(g) && tiles * ((Cin + 15) / 16) >= s2_min_wgs(g);
}

int tg_poolconv3x3_fwd(const float* x, const float* w4, const float* bias, const float* residual, float* y, int B, int Cin, int Cout,
                       int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(w4); TG_CHECK_PTR(y);
  if (!tg_poolconv3x3_supported(B, Cin, Cout, H, W)) return TG_EUNSUPPORTED;
  Shape s{B, Cin, Cout, H, W};
  const GeoId g = pick_geo(H, W);
  const int vx = plane_vec_ok(x, 2 * W), vw = tg_aligned16(w4);
  launch_upT(g, x, w4, bias, residual, y, s, vx, vw, tg_stream(stream));
  return tg_launch_status();
}

int tg_poolconv3x3_dgrad(const float* gy, const float* wp, float* gx, int B, int Cin, int Cout, int H, int W, void* stream) {
  TG_CHECK_PTR(gy); TG_CHECK_PTR(wp); TG_CHECK_PTR(gx);
  if (!tg_poolconv3x3_supported(B, Cin, Cout, H, W)) return TG_EUNSUPPORTED;
  if (((uintptr_t)gx & 7) != 0) return TG_EUNSUPPORTED;
  // the four-phase kernel with gy (Cout channels, H x W) as its low-resolution input and Cin output channels
  Shape s{B, Cout, Cin, H, W};
  const GeoId g = pick_geo(H, W);
  const int vx = plane_vec_ok(gy, W), vw = tg_aligned16(wp);
  launch_upfwd(g, gy, wp, nullptr, nullptr, gx, s, vx, vw, tg_stream(stream));
  return tg_launch_status();
}

int tg_conv2d_dgrad(const float* gy, const float* w, float* gx, int B, int Cin, int Cout, int H, int W, int ks, void* stream) {
  TG_CHECK_PTR(gy); TG_CHECK_PTR(w); TG_CHECK_PTR(gx);
  if (int rc = check_shape(B, Cin, Cout, H, W, ks)) return rc;
  // a forward convolution whose input channels are the original Cout and output channels the original Cin
  Shape s{B, Cout, Cin, H, W};
  return ks == 3 ? launch_fwd<3, true>(gy, w, nullptr, nullptr, gx, s, tg_stream(stream))
                 : launch_fwd<1, true>(gy, w, nullptr, nullptr, gx, s, tg_stream(stream));
}

size_t tg_conv2d_wgrad_workspace(int B, int Cin, int Cout, int H, int W, int ks) {
  if (check_shape(B, Cin, Cout, H, W, ks) != TG_OK) return 0;
  const WgPlan p = wgrad_plan(B, Cin, Cout, H, W, ks);
  return ((size_t)p.S * Cout * Cin * ks * ks + (size_t)p.S * Cout) * sizeof(float);
}

static int wgrad_partials(const float* x, const float* gy, float* workspace, size_t workspace_bytes, int B, int Cin, int Cout, int H,
                          int W, int ks, int want_bias, hipStream_t st, WgPlan* plan) {
  if (int rc = check_shape(B, Cin, Cout, H, W, ks)) return rc;
  if (workspace_bytes < tg_conv2d_wgrad_workspace(B, Cin, Cout, H, W, ks)) return TG_EWORKSPACE;
  const WgPlan p = wgrad_plan(B, Cin, Cout, H, W, ks);
  Shape s{B, Cin, Cout, H, W};
  const int64_t E = (int64_t)Cout * Cin * ks * ks;
  float* bias_part = want_bias ? workspace + (size_t)p.S * E : nullptr;
  *plan = p;
  return ks == 3 ? launch_wgrad<3>(x, gy, workspace, bias_part, s, p, st) : launch_wgrad<1>(x, gy, workspace, bias_part, s, p, st);
}

int tg_conv2d_wgrad(const float* x, const float* gy, float* gw, float* gbias, float* workspace, size_t workspace_bytes, int B,
                    int Cin, int Cout, int H, int W, int ks, int accumulate, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(gy); TG_CHECK_PTR(gw); TG_CHECK_PTR(workspace);
  hipStream_t st = tg_stream(stream);
  WgPlan p;
  if (int rc = wgrad_partials(x, gy, workspace, workspace_bytes, B, Cin, Cout, H, W, ks, gbias != nullptr, st, &p)) return rc;
  const int64_t E = (int64_t)Cout * Cin * ks * ks;
  float* bias_part = gbias ? workspace + (size_t)p.S * E : nullptr;
  const int blocks = (int)((E + 63) / 64) + (gbias ? (Cout + 63) / 64 : 0);
  wgrad_reduce_kernel<<<blocks, 256, 0, st>>>(workspace, gw, E, p.S, bias_part, gbias, Cout, accumulate);
  return tg_launch_status();
}

int tg_conv2d_wgrad_partials(const float* x, const float* gy, float* workspace, size_t workspace_bytes, int B, int Cin, int Cout,
                             int H, int W, int ks, int want_bias, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(gy); TG_CHECK_PTR(workspace);
  WgPlan p;
  return wgrad_partials(x, gy, workspace, workspace_bytes, B, Cin, Cout, H, W, ks, want_bias, tg_stream(stream), &p);
}

// ---- SelfAttention2d's three projections of one input (theta, phi, g) as one pass each way
static inline bool qkv_ok(int c0, int c1, int c2, int B, int Cin, int H, int W) {
  const int C = c0 + c1 + c2;
  // (at most 64 channels: beyond, a wave's serial chain of k-steps and its 8 accumulator blocks cost more than reading x once
  // saves -- measured on the 64:1 configurations, attention on 128 channels: 9840 -> 9568 img/s with the fused pass)
  return c0 > 0 && c1 > 0 && c2 > 0 && C <= 64 && Cin <= 64 && (H * W) % 16 == 0 && check_shape(B, Cin, C, H, W, 1) == TG_OK;
}
int tg_conv1x1_multi_supported(int c0, int c1, int c2, int B, int Cin, int H, int W) { return qkv_ok(c0, c1, c2, B, Cin, H, W) ? 1 : 0; }

int tg_conv1x1_multi_fwd(const float* x, const float* w, float* y0, float* y1, float* y2, int c0, int c1, int c2, int B, int Cin,
                         int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(w); TG_CHECK_PTR(y0); TG_CHECK_PTR(y1); TG_CHECK_PTR(y2);
  if (!qkv_ok(c0, c1, c2, B, Cin, H, W)) return TG_EUNSUPPORTED;
  if (!tg_aligned16(x) || !tg_aligned16(y0) || !tg_aligned16(y1) || !tg_aligned16(y2)) return TG_EUNSUPPORTED;
  Shape s{B, Cin, c0 + c1 + c2, H, W};
  return launch_conv1x1_segs(one_seg(x, Cin), w, nullptr, nullptr, three_segs(y0, y1, y2, c0, c1, c2), s, false, tg_stream(stream));
}

int tg_conv1x1_multi_dgrad(const float* gy0, const float* gy1, const float* gy2, const float* w, float* gx, int c0, int c1, int c2,
                           int B, int Cin, int H, int W, void* stream) {
  TG_CHECK_PTR(gy0); TG_CHECK_PTR(gy1); TG_CHECK_PTR(gy2); TG_CHECK_PTR(w); TG_CHECK_PTR(gx);
  if (!qkv_ok(c0, c1, c2, B, Cin, H, W)) return TG_EUNSUPPORTED;
  if (!tg_aligned16(gx) || !tg_aligned16(gy0) || !tg_aligned16(gy1) || !tg_aligned16(gy2)) return TG_EUNSUPPORTED;
  // as an operation: c0 + c1 + c2 input channels (the three gradients), Cin output channels, the forward filter read transposed
  Shape s{B, c0 + c1 + c2, Cin, H, W};
  return launch_conv1x1_segs(three_segs(gy0, gy1, gy2, c0, c1, c2), w, nullptr, nullptr, one_seg(gx, Cin), s, true, tg_stream(stream));
}

size_t tg_conv1x1_multi_wgrad_workspace(int c0, int c1, int c2, int B, int Cin, int H, int W) {
  return tg_conv2d_wgrad_workspace(B, Cin, c0 + c1 + c2, H, W, 1);
}

int tg_conv1x1_multi_wgrad(const float* x, const float* gy0, const float* gy1, const float* gy2, float* gw, float* workspace,
                           size_t workspace_bytes, int c0, int c1, int c2, int B, int Cin, int H, int W, int accumulate, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(gy0); TG_CHECK_PTR(gy1); TG_CHECK_PTR(gy2); TG_CHECK_PTR(gw); TG_CHECK_PTR(workspace);
  if (!qkv_ok(c0, c1, c2, B, Cin, H, W)) return TG_EUNSUPPORTED;
  if (!tg_aligned16(x) || !tg_aligned16(gy0) || !tg_aligned16(gy1) || !tg_aligned16(gy2)) return TG_EUNSUPPORTED;
  const int C = c0 + c1 + c2;
  if (workspace_bytes < tg_conv2d_wgrad_workspace(B, Cin, C, H, W, 1)) return TG_EWORKSPACE;
  const WgPlan p = wgrad_plan(B, Cin, C, H, W, 1);
  Shape s{B, Cin, C, H, W};
  hipStream_t st = tg_stream(stream);
  if (int rc = launch_conv1x1_wgrad_segs(x, three_segs(gy0, gy1, gy2, c0, c1, c2), workspace, nullptr, s, p.S, st)) return rc;
  const int64_t E = (int64_t)C * Cin;
  wgrad_reduce_kernel<<<(int)((E + 63) / 64), 256, 0, st>>>(workspace, gw, E, p.S, nullptr, nullptr, C, accumulate);
  return tg_launch_status();
}

int tg_conv2d_wgrad_reduce_batch(const tg_host_i64* items, int n_items, void* stream) {
  if (n_items < 0) return TG_EINVAL;
  if (n_items == 0) return TG_OK;
  TG_CHECK_PTR(items);
  hipStream_t st = tg_stream(stream);
  for (int base = 0; base < n_items; base += RB_MAX_ITEMS) {
    ReduceBatch rb;
    rb.n = n_items - base < RB_MAX_ITEMS ? n_items - base : RB_MAX_ITEMS;
    int blocks = 0;
    for (int i = 0; i < rb.n; ++i) {
      const tg_host_i64* it = items + (size_t)(base + i) * TG_WGRAD_ITEM_FIELDS;
      const int B = (int)it[3], Cin = (int)it[4], Cout = (int)it[5], H = (int)it[6], W = (int)it[7], ks = (int)it[8];
      if (it[0] == 0 || it[1] == 0) return TG_EINVAL;
      if (int rc = check_shape(B, Cin, Cout, H, W, ks)) return rc;
      ReduceItem& r = rb.item[i];
      r.part = reinterpret_cast<const float*>(it[0]);
      r.gw = reinterpret_cast<float*>(it[1]);
      r.gbias = reinterpret_cast<float*>(it[2]);
      r.E = (int64_t)Cout * Cin * ks * ks;
      r.S = wgrad_plan(B, Cin, Cout, H, W, ks).S;
      r.Cout = Cout;
      r.accumulate = (int)it[9];
      r.pad = 0;
      blocks += (int)((r.E + 63) / 64) + (r.gbias ? (Cout + 63) / 64 : 0);
      rb.block_end[i] = blocks;
    }
    for (int i = rb.n; i < RB_MAX_ITEMS; ++i) rb.block_end[i] = blocks;
    wgrad_reduce_batch_kernel<<<blocks, 256, 0, st>>>(rb);
  }
  return tg_launch_status();
}

}  // extern "C"
