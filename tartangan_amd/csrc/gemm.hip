// Generic batched fp32 GEMM  C[b] = op(A[b]) op(B[b]) (+ bias[n]).
// Used for nn.Linear (tiny) and the attention bmm's (K = C/8 or N/4).  LDS-tiled
// 64x64x16, 256 threads, 4x4 register micro-tile per thread, all edges guarded.
// These contractions are <1% (linear) / <8% (attention) of the step's FLOPs; the
// conv path (conv.hip) is where the MFMA work is.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, GT = 256;

template <bool TA, bool TB>
__global__ void __launch_bounds__(GT) gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                  const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc,
                                                  int64_t sA, int64_t sB, int64_t sC, float beta) {
  __shared__ float As[BK][BM + 4];
  __shared__ float Bs[BK][BN + 4];
  const int batch = blockIdx.z;
  A += batch * sA; Bm += batch * sB; C += batch * sC;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;    // 16 x 16 threads, each 4 (m) x 4 (n)
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += BK) {
    // stage A tile (BM x BK) and B tile (BK x BN); choose the thread->element map so that the
    // fastest-varying thread index follows the contiguous memory dimension
#pragma unroll
    for (int e = tid; e < BM * BK; e += GT) {
      int m, k;
      if (TA) { m = e % BM; k = e / BM; } else { k = e % BK; m = e / BK; }
      const int gm = m0 + m, gk = k0 + k;
      float v = 0.f;
      if (gm < M && gk < K) v = TA ? A[(int64_t)gk * lda + gm] : A[(int64_t)gm * lda + gk];
      As[k][m] = v;
    }
#pragma unroll
    for (int e = tid; e < BK * BN; e += GT) {
      int n, k;
      if (TB) { k = e % BK; n = e / BK; } else { n = e % BN; k = e / BN; }
      const int gn = n0 + n, gk = k0 + k;
      float v = 0.f;
      if (gn < N && gk < K) v = TB ? Bm[(int64_t)gn * ldb + gk] : Bm[(int64_t)gk * ldb + gn];
      Bs[k][n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn < N) {
        float r = acc[i][j] + (bias ? bias[gn] : 0.f);
        if (beta != 0.f) r += beta * C[(int64_t)gm * ldc + gn];
        C[(int64_t)gm * ldc + gn] = r;
      }
    }
  }
}

// Skinny product C (M x N) = A (M x K) B (K x N), M <= 16, both row-major, K long: the attention backward's
// dphi = theta ds / dg = go beta under double backward (M = 4 / 16, K = 1024, N = 256 at 128:3).  The 64x64 tile
// kernel above would use 4/64 of its rows; here a workgroup owns 64 columns, its 4 waves split K, every B element
// is read once (coalesced along N) and A is broadcast from LDS.
constexpr int SK_KC = 256;      // K chunk staged per iteration
template <int MM>
__global__ void __launch_bounds__(GT) gemm_skinny_nn_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                            const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc,
                                                            int64_t sA, int64_t sB, int64_t sC, float beta) {
  __shared__ float As[MM][SK_KC + 1];
  __shared__ float red[4][MM][64];
  const int batch = blockIdx.y;
  A += batch * sA; Bm += batch * sB; C += batch * sC;
  const int col = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  for (int k0 = 0; k0 < K; k0 += SK_KC) {
    __syncthreads();
    for (int e = threadIdx.x; e < MM * SK_KC; e += GT) {
      const int m = e / SK_KC, k = e - m * SK_KC;
      As[m][k] = (m < M && k0 + k < K) ? A[(int64_t)m * lda + k0 + k] : 0.f;
    }
    __syncthreads();
    const int kend = min(SK_KC, K - k0);
    for (int k = ks; k < kend; k += 4) {
      const float b = (n < N) ? Bm[(int64_t)(k0 + k) * ldb + n] : 0.f;
#pragma unroll
      for (int m = 0; m < MM; ++m) acc[m] = fmaf(As[m][k], b, acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < MM; ++m) red[ks][m][col] = acc[m];
  __syncthreads();
  for (int e = threadIdx.x; e < MM * 64; e += GT) {
    const int m = e >> 6, c = e & 63;
    const int gn = blockIdx.x * 64 + c;
    if (m < M && gn < N) {
      float r = (red[0][m][c] + red[1][m][c]) + (red[2][m][c] + red[3][m][c]) + (bias ? bias[gn] : 0.f);
      if (beta != 0.f) r += beta * C[(int64_t)m * ldc + gn];
      C[(int64_t)m * ldc + gn] = r;
    }
  }
}

}  // namespace

extern "C" int tg_gemm(const float* A, const float* Bm, float* C, const float* bias_n, int M, int N, int K, int lda, int ldb,
                       int ldc, int transA, int transB, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                       float beta, void* stream) {
  TG_CHECK_PTR(A); TG_CHECK_PTR(Bm); TG_CHECK_PTR(C);
  TG_CHECK_POS(M); TG_CHECK_POS(N); TG_CHECK_POS(K); TG_CHECK_POS(batch);
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return TG_EINVAL;
  if (batch > 65535) return TG_EUNSUPPORTED;
  hipStream_t st = tg_stream(stream);
  if (!transA && !transB && M <= 16 && K >= 256) {
    dim3 sgrid((N + 63) / 64, batch);
    if (M <= 4) gemm_skinny_nn_kernel<4><<<sgrid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
    else gemm_skinny_nn_kernel<16><<<sgrid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
    return tg_launch_status();
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  if (transA && transB) gemm_kernel<true, true><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  else if (transA) gemm_kernel<true, false><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  else if (transB) gemm_kernel<false, true><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  else gemm_kernel<false, false><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  return tg_launch_status();
}
