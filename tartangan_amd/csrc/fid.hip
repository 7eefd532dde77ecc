// Fréchet-distance / Inception-score math of the reference (inception_utils.py:97-144, 205-246) on gfx950.
//
// The heavy part is sqrt_newton_schulz: 20 iterations of three 2048^3 fp32 products (1.03 TFLOP) -- a dense
// contraction that belongs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, 157 TFLOP/s chip peak, exact fp32).
// gemm_big_kernel is sized for exactly that: 128 x 128 output tiles (2048^2 / 128^2 = 256 workgroups = one per CU),
// four waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers), K in steps of 32 staged global -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds) into two buffers, one barrier per step:
//     DMA(step 0);  for s: { vmcnt(0); barrier; DMA(step s + 1 -> other buffer); 64 MFMAs per wave on step s }
// One wave per SIMD, so the DMA of the next step under the MFMAs of this one is the only overlap there is; out-of-range
// rows / columns / K tails are lanes whose buffer offset fails the bounds check (zeros, no memory touched).
// Operands are row-major; "TA" reads A^T (the covariance X_c^T X_c).  Epilogue: C = alpha * A B + diag * I
// (the Newton-Schulz step T = 1.5 I - 0.5 Z Y in one pass).
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GB_T = 256, GB_BM = 128, GB_BK = 32;
constexpr uint32_t GB_OOB = 0x80000000u;

__device__ __forceinline__ void gb_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_wave_base, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}

// LDS images: A (not transposed) [m][BK + 4] (one pad chunk per row: the fragment reads of 32 rows then spread over
// 8 banks instead of 2), A^T and B [k][128].
template <bool TA, int GB_BN>
__global__ void __launch_bounds__(GB_T)
gemm_big_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, int M, int N, int K, int lda,
                int ldb, int ldc, float alpha, float diag, int64_t bytesA, int64_t bytesB) {
  constexpr int ARS = TA ? GB_BM : GB_BK + 4;                    // A image row stride (floats)
  constexpr int AROWS = TA ? GB_BK : GB_BM, AQ = ARS / 4;         // rows, chunks per row
  constexpr int ACH = AROWS * AQ, BCH = GB_BK * (GB_BN / 4);
  constexpr int NVA = (ACH + GB_T - 1) / GB_T, NVB = (BCH + GB_T - 1) / GB_T;
  static_assert(ACH % 64 == 0 && BCH % 64 == 0, "LDS regions of whole wave pieces");
  constexpr int ABUF = ACH * 4, BBUF = BCH * 4, BUF = ABUF + BBUF;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  // (scalar wave index: an LDS-DMA may only sit under wave-uniform conditions, see dma_pad in conv.hip / DESIGN.md section 7)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // workgroup -> tile: an XCD (block id mod 8) takes a band of tile rows, so that its L2 keeps the A panels
  const int tiles_n = (N + GB_BN - 1) / GB_BN, tiles_m = (M + GB_BM - 1) / GB_BM;
  int bid = blockIdx.x;
  const int total = tiles_m * tiles_n;
  if (total % 8 == 0) bid = (bid & 7) * (total >> 3) + (bid >> 3);
  const int m0 = (bid / tiles_n) * GB_BM, n0 = (bid % tiles_n) * GB_BN;

  uint32_t aoff[NVA], boff[NVB];
  int arow[NVA], brow[NVB];                                      // k index (inside a step) of the row a lane loads, or -1
#pragma unroll
  for (int v = 0; v < NVA; ++v) {
    const int e = v * GB_T + threadIdx.x;
    const int row = e / AQ, q = e % AQ;
    bool ok = e < ACH;
    if (TA) { ok = ok && (m0 + 4 * q < M); arow[v] = row; }
    else { ok = ok && (4 * q < GB_BK) && (m0 + row < M); arow[v] = 4 * q; }
    aoff[v] = ok ? (uint32_t)(row * lda + 4 * q) << 2 : GB_OOB;
  }
#pragma unroll
  for (int v = 0; v < NVB; ++v) {
    const int e = v * GB_T + threadIdx.x;
    const int row = e / (GB_BN / 4), q = e % (GB_BN / 4);
    const bool ok = (e < BCH) && (n0 + 4 * q < N);
    brow[v] = row;
    boff[v] = ok ? (uint32_t)(row * ldb + 4 * q) << 2 : GB_OOB;
  }
  const char* ab = reinterpret_cast<const char*>(A) + (TA ? (int64_t)m0 * 4 : (int64_t)m0 * lda * 4);
  int64_t abytes = bytesA - (ab - reinterpret_cast<const char*>(A));
  const int64_t astep = TA ? (int64_t)GB_BK * lda * 4 : (int64_t)GB_BK * 4;
  const char* bb = reinterpret_cast<const char*>(Bm) + (int64_t)n0 * 4;
  int64_t bbytes = bytesB - (bb - reinterpret_cast<const char*>(Bm));
  const int64_t bstep = (int64_t)GB_BK * ldb * 4;

  auto issue = [&](int k0, float* buf) {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ab), 0, (int)abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bb), 0, (int)bbytes, 0x00020000);
    const int kvalid = K - k0;                                   // k rows of this step that exist (>= BK: all)
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
      uint32_t off = aoff[v];
      if (kvalid < GB_BK) off = (arow[v] < kvalid) ? off : GB_OOB;       // (K % 4 == 0: a chunk is all in or all out)
      if ((v + 1) * GB_T <= ACH || v * GB_T + wave * 64 < ACH) gb_dma16(ra, buf + (v * GB_T + wave * 64) * 4, off);
    }
#pragma unroll
    for (int v = 0; v < NVB; ++v) {
      uint32_t off = boff[v];
      if (kvalid < GB_BK) off = (brow[v] < kvalid) ? off : GB_OOB;
      if ((v + 1) * GB_T <= BCH || v * GB_T + wave * 64 < BCH) gb_dma16(rb, buf + ABUF + (v * GB_T + wave * 64) * 4, off);
    }
    ab += astep; abytes -= astep;
    bb += bstep; bbytes -= bstep;
  };

  constexpr int WN = GB_BN / 2, NT = WN / 32;                    // a wave's columns; its 32-column MFMA tiles (2 or 1)
  f32x16 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int la = TA ? (h * GB_BM + wm * 64 + i) : ((wm * 64 + i) * ARS + h);
  const int lb = h * GB_BN + wn * WN + i;

  issue(0, lds);
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += GB_BK) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k0 + GB_BK < K) issue(k0 + GB_BK, lds + (buf ^ 1) * BUF);
    const float* al = lds + buf * BUF;
    const float* bl = al + ABUF;
#pragma unroll
    for (int kk = 0; kk < GB_BK / 2; ++kk) {
      float a[2], b[NT];
#pragma unroll
      for (int t = 0; t < 2; ++t) a[t] = TA ? al[la + (2 * kk) * GB_BM + t * 32] : al[la + t * 32 * ARS + 2 * kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = bl[lb + (2 * kk) * GB_BN + t * 32];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[t], acc[s][t], 0, 0, 0);
    }
    buf ^= 1;
  }

  // D[row][col]: lane (col = i, h), register r -> row (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n0 + wn * WN + t * 32 + i;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) C[(int64_t)row * ldc + col] = alpha * acc[s][t][r] + (row == col ? diag : 0.f);
      }
    }
}

// X[n][d] -= mean[d]   (torch_cov's in-place centring, inception_utils.py:120)
__global__ void __launch_bounds__(256) center_rows_kernel(float* __restrict__ X, const float* __restrict__ mean, int64_t n, int D) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += stride) X[e] -= mean[e % D];
}

// out = sum_i A[i][i]
__global__ void __launch_bounds__(256) trace_kernel(const float* __restrict__ A, float* __restrict__ out, int D, int ld) {
  __shared__ double scratch[32];
  double s = 0.0;
  for (int i = threadIdx.x; i < D; i += 256) s += (double)A[(int64_t)i * ld + i];
  s = block_sum_d(s, scratch);
  if (threadIdx.x == 0) *out = (float)s;
}

// rows[n] = sum_c p[n][c] * (log p[n][c] - log mean[c])     (calculate_inception_score's KL term, :243)
__global__ void __launch_bounds__(256) is_kl_rows_kernel(const float* __restrict__ p, const float* __restrict__ mean, float* __restrict__ rows,
                                                         int N, int Cn) {
  __shared__ float scratch[32];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < Cn; c += 256) {
    const float v = p[(int64_t)n * Cn + c];
    s += v * (logf(v) - logf(mean[c]));
  }
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) rows[n] = s;
}

}  // namespace

extern "C" {

int tg_gemm_big_supported(int M, int N, int K, int lda, int ldb, int transA) {
  // 16-byte chunks: leading dimensions and K multiples of 4; 32-bit buffer offsets
  if (M <= 0 || N <= 0 || K <= 0 || lda % 4 || ldb % 4 || (!transA && K % 4)) return 0;   // (A^T and B are chunked along m / n)
  const int64_t rowsA = transA ? K : M;
  if (rowsA * lda * 4 >= (1ll << 31) || (int64_t)K * ldb * 4 >= (1ll << 31)) return 0;
  return 1;
}

int tg_gemm_big(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA, float alpha,
                float diag, void* stream) {
  TG_CHECK_PTR(A); TG_CHECK_PTR(Bm); TG_CHECK_PTR(C);
  if (!tg_gemm_big_supported(M, N, K, lda, ldb, transA) || !tg_aligned16(A) || !tg_aligned16(Bm)) return TG_EUNSUPPORTED;
  if (lda < (transA ? M : K) || ldb < N || ldc < N) return TG_EINVAL;
  const int64_t bytesA = (int64_t)(transA ? K : M) * lda * 4, bytesB = (int64_t)K * ldb * 4;
  hipStream_t st = tg_stream(stream);
  // 128 x 128 tiles when they give every CU two workgroups (the second hides the first one's barriers); else 128 x 64
  const int tm = (M + GB_BM - 1) / GB_BM;
  static const int force = [] { const char* e = getenv("TG_GEMM_BN"); return e ? atoi(e) : 0; }();    // (development knob)
  const bool wide = force ? force == 128 : (int64_t)tm * ((N + 127) / 128) >= 512;
  if (wide) {
    const int tiles = tm * ((N + 127) / 128);
    if (transA) gemm_big_kernel<true, 128><<<tiles, GB_T, 0, st>>>(A, Bm, C, M, N, K, lda, ldb, ldc, alpha, diag, bytesA, bytesB);
    else gemm_big_kernel<false, 128><<<tiles, GB_T, 0, st>>>(A, Bm, C, M, N, K, lda, ldb, ldc, alpha, diag, bytesA, bytesB);
  } else {
    const int tiles = tm * ((N + 63) / 64);
    if (transA) gemm_big_kernel<true, 64><<<tiles, GB_T, 0, st>>>(A, Bm, C, M, N, K, lda, ldb, ldc, alpha, diag, bytesA, bytesB);
    else gemm_big_kernel<false, 64><<<tiles, GB_T, 0, st>>>(A, Bm, C, M, N, K, lda, ldb, ldc, alpha, diag, bytesA, bytesB);
  }
  return tg_launch_status();
}

int tg_center_rows(float* X, const float* mean, int N, int D, void* stream) {
  TG_CHECK_PTR(X); TG_CHECK_PTR(mean); TG_CHECK_POS(N); TG_CHECK_POS(D);
  const int64_t n = (int64_t)N * D;
  center_rows_kernel<<<tg_ew_grid(n, 256), 256, 0, tg_stream(stream)>>>(X, mean, n, D);
  return tg_launch_status();
}

int tg_trace(const float* A, float* out, int D, int ld, void* stream) {
  TG_CHECK_PTR(A); TG_CHECK_PTR(out); TG_CHECK_POS(D);
  if (ld < D) return TG_EINVAL;
  trace_kernel<<<1, 256, 0, tg_stream(stream)>>>(A, out, D, ld);
  return tg_launch_status();
}

int tg_is_kl_rows(const float* p, const float* mean, float* rows, int N, int Cn, void* stream) {
  TG_CHECK_PTR(p); TG_CHECK_PTR(mean); TG_CHECK_PTR(rows); TG_CHECK_POS(N); TG_CHECK_POS(Cn);
  is_kl_rows_kernel<<<N, 256, 0, tg_stream(stream)>>>(p, mean, rows, N, Cn);
  return tg_launch_status();
}

}  // extern "C"
