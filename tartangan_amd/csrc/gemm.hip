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

  // The next k-tile's global loads are issued before this tile's products (register prefetch): with one load -> LDS ->
  // barrier -> products round trip per 16 k-values, the small-grid products of the step (a Linear layer on 64..512 rows:
  // one to 32 workgroups) spent ~2.7 us per k-tile waiting on memory.
  // Thread -> element map: the fastest-varying thread index follows the contiguous memory dimension.
  constexpr int PER = BM * BK / GT;
  static_assert(BM == BN && PER * GT == BM * BK, "square tiles, whole elements per thread");
  float ra[PER], rb[PER];
  auto load = [&](int k0) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * GT;
      int m, k;
      if (TA) { m = e % BM; k = e / BM; } else { k = e % BK; m = e / BK; }
      const int gm = m0 + m, gk = k0 + k;
      ra[u] = (gm < M && gk < K) ? (TA ? A[(int64_t)gk * lda + gm] : A[(int64_t)gm * lda + gk]) : 0.f;
      int n, kb;
      if (TB) { kb = e % BK; n = e / BK; } else { n = e % BN; kb = e / BN; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[u] = (gn < N && gkb < K) ? (TB ? Bm[(int64_t)gn * ldb + gkb] : Bm[(int64_t)gkb * ldb + gn]) : 0.f;
    }
  };
  load(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * GT;
      if (TA) As[e / BM][e % BM] = ra[u]; else As[e % BK][e / BK] = ra[u];
      if (TB) Bs[e % BK][e / BK] = rb[u]; else Bs[e / BN][e % BN] = rb[u];
    }
    __syncthreads();
    if (k0 + BK < K) load(k0 + BK);
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
// dphi = theta ds / dg = go beta under double backward (M = 4 / 16, K = 1024, N = 256 at 128:3).  Pure streaming of B
// (64 MiB per call at batch 64): a workgroup owns 16 columns, its 256 threads are 16 columns x 16 K-slices, so a
// wave-load covers 4 rows x 64 contiguous bytes and the grid is N/16 x batch workgroups (enough loads in flight to
// stream at HBM rate; the first version, 64 columns per workgroup, had one workgroup per CU and ran at 0.8 TB/s).
// A is broadcast from LDS; the 16 K-slices are combined through LDS in a fixed order.
constexpr int SK_KC = 512;      // K chunk of A staged per iteration
constexpr int SK_COLS = 16, SK_SL = GT / SK_COLS;
template <int MM>
__global__ void __launch_bounds__(GT) gemm_skinny_nn_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                            const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc,
                                                            int64_t sA, int64_t sB, int64_t sC, float beta) {
  __shared__ __attribute__((aligned(16))) float As[SK_KC][MM];        // [k][m]: the MM values of one k are read as float4 broadcasts
  __shared__ float red[SK_SL][MM][SK_COLS];
  static_assert(MM % 4 == 0, "A rows are read four at a time");
  const int batch = blockIdx.y;
  A += batch * sA; Bm += batch * sB; C += batch * sC;
  const int col = threadIdx.x % SK_COLS, ks = threadIdx.x / SK_COLS;
  const int n = blockIdx.x * SK_COLS + col;
  float acc[MM];
#pragma unroll
  for (int m = 0; m < MM; ++m) acc[m] = 0.f;
  for (int k0 = 0; k0 < K; k0 += SK_KC) {
    __syncthreads();
    for (int e = threadIdx.x; e < MM * SK_KC; e += GT) {
      const int m = e / SK_KC, k = e - m * SK_KC;                     // coalesced along k
      As[k][m] = (m < M && k0 + k < K) ? A[(int64_t)m * lda + k0 + k] : 0.f;
    }
    __syncthreads();
    const int kend = min(SK_KC, K - k0);
#pragma unroll 4
    for (int k = ks; k < kend; k += SK_SL) {
      const float b = (n < N) ? Bm[(int64_t)(k0 + k) * ldb + n] : 0.f;
#pragma unroll
      for (int m4 = 0; m4 < MM / 4; ++m4) {
        const float4 a = *reinterpret_cast<const float4*>(&As[k][4 * m4]);
        acc[4 * m4] = fmaf(a.x, b, acc[4 * m4]);
        acc[4 * m4 + 1] = fmaf(a.y, b, acc[4 * m4 + 1]);
        acc[4 * m4 + 2] = fmaf(a.z, b, acc[4 * m4 + 2]);
        acc[4 * m4 + 3] = fmaf(a.w, b, acc[4 * m4 + 3]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MM; ++m) red[ks][m][col] = acc[m];
  __syncthreads();
  for (int e = threadIdx.x; e < MM * SK_COLS; e += GT) {
    const int m = e / SK_COLS, c = e % SK_COLS;
    const int gn = blockIdx.x * SK_COLS + c;
    if (m < M && gn < N) {
      float r = 0.f;
#pragma unroll
      for (int q = 0; q < SK_SL; ++q) r += red[q][m][c];
      r += bias ? bias[gn] : 0.f;
      if (beta != 0.f) r += beta * C[(int64_t)m * ldc + gn];
      C[(int64_t)m * ldc + gn] = r;
    }
  }
}

// Skinny product with B transposed: C (M x N) = A (M x K) B^T, B stored (N x K) row-major, M <= 16 -- the composed
// attention's o = g beta^T and dtheta = phi ds^T (M = 16 / 4, N = 1024, K = 256): again one pass over B.  On the matrix
// cores, transposed: C^T tile (16 rows of B) x (16 = padded M).  Lane (c, g) loads B[n0 + c][16 s + 4 g + (0..3)] as one
// float4 (a wave-load = 16 rows x 64 contiguous bytes) and uses component j as the A operand of k-step (s, j); the k
// index of that step is 16 s + 4 g + j, and A^T is loaded in the same order.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(GT) gemm_skinny_nt_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                                            const float* __restrict__ bias, int M, int N, int K, int lda, int ldb, int ldc,
                                                            int64_t sA, int64_t sB, int64_t sC, float beta) {
  const int batch = blockIdx.y;
  A += batch * sA; Bm += batch * sB; C += batch * sC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n0 = (blockIdx.x * 4 + wave) * 16;
  if (n0 >= N) return;                                   // whole wave; no barriers
  const int brow = min(n0 + c, N - 1), arow = min(c, M - 1);
  const float* bp = Bm + (int64_t)brow * ldb + 4 * g;
  const float* ap = A + (int64_t)arow * lda + 4 * g;
  f32x4_ acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {                    // host guarantees K % 32 == 0 and 16-byte aligned rows
    const float4 b0 = *reinterpret_cast<const float4*>(bp + k0), b1 = *reinterpret_cast<const float4*>(bp + k0 + 16);
    float4 a0 = *reinterpret_cast<const float4*>(ap + k0), a1 = *reinterpret_cast<const float4*>(ap + k0 + 16);
    if (c >= M) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.x, a0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.x, a1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.y, a0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.y, a1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.z, a0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.z, a1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0.w, a0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1.w, a1.w, acc1, 0, 0, 0);
  }
  // result register r of lane (c, g): C^T[n0 + 4g + r][m = c]
  if (c < M) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gn = n0 + 4 * g + r;
      if (gn < N) {
        float v = acc0[r] + acc1[r] + (bias ? bias[gn] : 0.f);
        if (beta != 0.f) v += beta * C[(int64_t)c * ldc + gn];
        C[(int64_t)c * ldc + gn] = v;
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
  hipStream_t st = tg_stream(stream);
  if (!transA && transB && M <= 16 && K % 32 == 0 && N >= 64 && lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0 &&
      tg_aligned16(A) && tg_aligned16(Bm)) {
    dim3 tgrid((N + 63) / 64, batch);
    gemm_skinny_nt_kernel<<<tgrid, GT, 0, st>>>(A, Bm, C, bias_n, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, beta);
    return tg_launch_status();
  }
  if (!transA && !transB && M <= 16 && K >= 256) {
    dim3 sgrid((N + SK_COLS - 1) / SK_COLS, batch);
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
