"""Progress samples (reference components/image_sampler.py:12-57) for the HIP trainers.

What matters for a seeded run is preserved exactly: ``sample_z(32)`` is drawn when training begins (:15), the
interpolation grid's four corners by one ``sample_z(4)`` at the first output (:47-54), both from the CPU default
generator through the trainer; ``target_g`` / ``g`` run under ``no_grad`` in whatever mode they are in (train-mode
BatchNorm: the running statistics move, as in the reference).  The image grids are the arithmetic of
``torchvision.utils.save_image(..., normalize=True, range=(-1, 1))`` restated (torchvision is not a dependency):
``make_grid`` padding 2 / pad value 0, clamp to (-1, 1), scale to [0, 1], ``mul(255).add_(0.5).clamp_(0, 255)`` to
uint8; PNG encoding through Pillow.  Under data parallelism only rank 0 renders.
"""
import os

import numpy as np
import torch

from .base import TrainerComponent


def slerp(val, low, high):
    """utils/slerp.py:5-15"""
    omega = np.arccos(np.clip(np.dot(low / np.linalg.norm(low), high / np.linalg.norm(high)), -1, 1))
    so = np.sin(omega)
    if so == 0:
        return (1.0 - val) * low + val * high
    return np.sin((1.0 - val) * omega) / so * low + np.sin(val * omega) / so * high


def slerp_grid(top_left, top_right, bottom_left, bottom_right, nrows, ncols):
    """utils/slerp.py:18-33"""
    left_col = [slerp(x, top_left, bottom_left) for x in np.linspace(0, 1, nrows)]
    right_col = [slerp(x, top_right, bottom_right) for x in np.linspace(0, 1, nrows)]
    rows = []
    for left, right in zip(left_col, right_col):
        rows.append(torch.from_numpy(np.vstack([slerp(x, left, right) for x in np.linspace(0, 1, ncols)])))
    return torch.cat(rows, dim=0)


def image_grid_uint8(imgs, nrow=8, padding=2, value_range=(-1, 1)):
    """save_image(normalize=True, range=value_range)'s pixel arithmetic -> (H, W, C) uint8 array."""
    imgs = imgs.detach().float().cpu()
    lo, hi = value_range
    imgs = imgs.clamp(lo, hi).sub(lo).div(max(hi - lo, 1e-5))
    n, c, h, w = imgs.shape
    xmaps = min(nrow, n)
    ymaps = int(np.ceil(n / xmaps))
    H, W = h + padding, w + padding
    grid = torch.zeros(c, H * ymaps + padding, W * xmaps + padding)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = imgs[k]
            k += 1
    return grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()


class ImageSamplerComponent(TrainerComponent):
    def on_train_begin(self, steps, logs):
        os.makedirs(self.sample_root, exist_ok=True)
        self.progress_samples = self.trainer.sample_z(32)

    def on_train_end(self, steps, logs):
        self.output_samples(f'{self.sample_root}/sample_{steps}.png')

    def on_batch_end(self, steps, logs):
        if steps % self.trainer.args.gen_freq == 0:
            self.output_samples(f'{self.sample_root}/sample_{steps}.png')

    def render(self):
        """-> (samples (32,C,S,S): 16 target_g + 16 g, grid (25,C,S,S)) as the reference computes them (:23-38)."""
        with torch.no_grad():
            imgs = self.trainer.target_g(self.progress_samples)[:16]
            imgs_g = self.trainer.g(self.progress_samples)[:16]
            imgs = torch.cat([imgs, imgs_g], dim=0)
            if not hasattr(self, '_latent_grid_samples'):
                self._latent_grid_samples = self.sample_latent_grid(5, 5)
            grid_imgs = self.trainer.target_g(self._latent_grid_samples)
        return imgs, grid_imgs

    def output_samples(self, filename, n=None):
        imgs, grid_imgs = self.render()          # every rank renders (BatchNorm statistics stay in step); rank 0 writes
        dp = getattr(self.trainer, 'data_parallel', None)
        if dp is not None and dp.rank != 0:
            return
        from PIL import Image
        Image.fromarray(image_grid_uint8(imgs)).save(filename, format='png')
        grid_filename = os.path.join(os.path.dirname(filename), f'grid_{os.path.basename(filename)}')
        Image.fromarray(image_grid_uint8(grid_imgs, nrow=5)).save(grid_filename, format='png')

    def sample_latent_grid(self, nrows, ncols):
        top_left, top_right, bottom_left, bottom_right = map(lambda x: x.cpu().numpy(), self.trainer.sample_z(4))
        grid = slerp_grid(top_left, top_right, bottom_left, bottom_right, nrows, ncols)
        # (float32: what the reference's pinned numpy 1.x produces; numpy 2 would promote the interpolants to float64)
        return grid.to(self.trainer.device, torch.float32)

    @property
    def sample_root(self):
        return f'{self.trainer.output_root}/samples'

    @classmethod
    def add_args_to_parser(cls, parser):
        parser.add_argument('--gen-freq', type=int, default=200, help='Output samples every N batches')
