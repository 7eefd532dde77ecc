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

}  // namespace

extern "C" int tg_gemm(const float* A, const float* Bm, float* C, const float* bias_n, int M, int N, int K, int lda, int ldb,
                       int ldc, int transA, int transB, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                       float beta, void* stream) {
  TG_CHECK_PTR(A); TG_CHECK_PTR(Bm); TG_CHECK_PTR(C);
  TG_CHECK_POS(M); TG_CHECK_POS(N); TG_CHECK_POS(K); TG_CHECK_POS(batch);
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return TG_EINVAL;
  if (batch > 65535) return TG_EUNSUPPORTED;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  hipStream_t st = tg_stream(stream);
  if (transA && transB) gemm_kernel<true, true><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  else if (transA) gemm_kernel<true, false><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  else if (transB) gemm_kernel<false, true><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  else gemm_kernel<false, false><<<grid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
  return tg_launch_status();
}
