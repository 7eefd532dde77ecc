// Input pipeline of the reference on the device: ImageBytesDataset keeps the training set as one
// uint8 NHWC array (image_bytes_dataset.py:12-49) and pushes every image through
//   ToPILImage -> RandomCrop(size) -> ToTensor -> Normalize((.5,.5,.5), (.5,.5,.5))      (trainers/trainer.py:69-78)
// one image at a time on the host.  Here the archive is resident in HBM as it is on disk (1 byte per sample) and one
// kernel gathers a batch: out[b][c][y][x] = ((u8 / 255) - 0.5) / 0.5 in exactly that fp32 operation order (ToTensor's
// .div(255), Normalize's .sub_(mean).div_(std)), so the result is bit-identical to the reference's tensor.
// HBM-bound and tiny (49 KB of bytes in, 197 KB of floats out per 128 x 128 image): a thread owns 4 consecutive pixels
// of one row -- 12 contiguous bytes in, one float4 per channel plane out.
#include "common.h"

namespace {

__device__ __forceinline__ float to_unit(uint8_t v) {
  const float t = __fdiv_rn((float)v, 255.f);        // ToTensor: correctly rounded fp32 division
  return __fdiv_rn(t - 0.5f, 0.5f);                  // Normalize(mean .5, std .5)
}

template <int C>
__global__ void __launch_bounds__(256)
image_bytes_batch_kernel(const uint8_t* __restrict__ archive, const int64_t* __restrict__ index, const int* __restrict__ oy,
                         const int* __restrict__ ox, float* __restrict__ out, int B, int Hs, int Ws, int S, int64_t n_images) {
  const int qpr = (S + 3) / 4;                        // 4-pixel groups per output row
  const int64_t total = (int64_t)B * S * qpr;
  for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int q = (int)(e % qpr);
    const int y = (int)((e / qpr) % S);
    const int b = (int)(e / ((int64_t)qpr * S));
    int64_t img = index[b];
    img = img < 0 ? 0 : (img >= n_images ? n_images - 1 : img);          // (the host validates; never read out of bounds)
    const int y0 = oy ? oy[b] : 0, x0 = ox ? ox[b] : 0;
    const uint8_t* src = archive + ((img * Hs + (y0 + y)) * Ws + (x0 + 4 * q)) * C;
    const int npx = min(4, S - 4 * q);
    float v[C][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < C; ++c) v[c][p] = p < npx ? to_unit(src[p * C + c]) : 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float* dst = out + (((int64_t)b * C + c) * S + y) * S + 4 * q;
      if (npx == 4 && (S % 4 == 0)) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
      } else {
        for (int p = 0; p < npx; ++p) dst[p] = v[c][p];
      }
    }
  }
}

}  // namespace

extern "C" int tg_image_bytes_batch(const uint8_t* archive, const int64_t* index, const int* crop_y, const int* crop_x, float* out,
                                    int B, int64_t n_images, int H, int W, int channels, int size, void* stream) {
  TG_CHECK_PTR(archive); TG_CHECK_PTR(index); TG_CHECK_PTR(out);
  TG_CHECK_POS(B); TG_CHECK_POS(H); TG_CHECK_POS(W); TG_CHECK_POS(size);
  if (n_images <= 0 || size > H || size > W) return TG_EINVAL;
  if (channels != 3 && channels != 1) return TG_EUNSUPPORTED;
  if (!tg_aligned16(out)) return TG_EUNSUPPORTED;
  const int64_t total = (int64_t)B * size * ((size + 3) / 4);
  hipStream_t st = tg_stream(stream);
  if (channels == 3)
    image_bytes_batch_kernel<3><<<tg_ew_grid(total, 256), 256, 0, st>>>(archive, index, crop_y, crop_x, out, B, H, W, size, n_images);
  else
    image_bytes_batch_kernel<1><<<tg_ew_grid(total, 256), 256, 0, st>>>(archive, index, crop_y, crop_x, out, B, H, W, size, n_images);
  return tg_launch_status();
}
