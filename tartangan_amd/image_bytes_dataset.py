"""Device-resident ``ImageBytesDataset`` (reference image_bytes_dataset.py:12-49) fused with the transform chain the
trainers put on it (``ToPILImage -> RandomCrop(img_size) -> ToTensor -> Normalize(.5, .5)``, trainers/trainer.py:69-78).

The reference keeps the uint8 (N, H, W, 3) archive in host memory and transforms one image at a time through PIL
inside a ``DataLoader(num_workers=0)`` -- at thousands of images per second that starves the step.  Here the archive
lives in HBM as stored (one byte per sample) and ``batch()`` gathers, crops, converts and normalises a whole batch in
one kernel (``tg_image_bytes_batch``), bit-identical to ToTensor + Normalize.  ``loader()`` mirrors
``DataLoader(dataset, batch_size, shuffle=True, drop_last=True)`` (trainers/trainer.py:84-86) including how a
``RandomSampler`` consumes the default CPU generator, and shards every global batch by rank under data parallelism.
"""
import numpy as np
import torch

from . import backend as _be


class ImageBytesDataset:
    """Store images as bytes (on the device) to reduce the memory footprint."""

    def __init__(self, images, crop_size=None, device='cuda'):
        images = torch.as_tensor(np.ascontiguousarray(images))
        if images.dtype != torch.uint8 or images.dim() != 4:
            raise ValueError('images must be a uint8 array of shape (N, H, W, C)')
        self.images = images.to(device)
        self.crop_size = crop_size
        self.device = device

    def __len__(self):
        return self.images.shape[0]

    @property
    def image_size(self):
        return self.crop_size or min(self.images.shape[1], self.images.shape[2])

    @classmethod
    def from_path(cls, path, crop_size=None, device='cuda'):
        """The reference's on-disk format: ``np.savez_compressed(outfile, images=data)`` (image_bytes_dataset.py:91-92) or a
        bare ``.npy`` (:43-49)."""
        images = np.load(path)
        if isinstance(images, np.lib.npyio.NpzFile):
            images = images['images']
        return cls(images, crop_size=crop_size, device=device)

    def _crop_offsets(self, n, size):
        """torchvision RandomCrop.get_params per image: nothing is drawn when the image already has the crop size."""
        H, W = self.images.shape[1:3]
        if H == size and W == size:
            return None, None
        ys, xs = [], []
        for _ in range(n):
            ys.append(int(torch.randint(0, H - size + 1, size=(1,)).item()))
            xs.append(int(torch.randint(0, W - size + 1, size=(1,)).item()))
        dev = self.images.device
        return (torch.tensor(ys, dtype=torch.int32, device=dev), torch.tensor(xs, dtype=torch.int32, device=dev))

    def batch(self, indices, crop_offsets=None):
        """-> (B, C, S, S) fp32 in [-1, 1]: the batch the reference's DataLoader would collate for these indices."""
        index = torch.as_tensor(indices, dtype=torch.int64)
        if index.numel() and (int(index.min()) < 0 or int(index.max()) >= len(self)):
            raise IndexError('image index out of range')
        index = index.to(self.images.device)
        N, H, W, C = self.images.shape
        S = self.image_size
        if S > H or S > W:
            raise ValueError(f'crop size {S} exceeds the stored images ({H} x {W})')
        oy, ox = crop_offsets if crop_offsets is not None else self._crop_offsets(index.numel(), S)
        out = torch.empty(index.numel(), C, S, S, dtype=torch.float32, device=self.images.device)
        _be.get().image_bytes_batch(self.images, index, oy, ox, out, index.numel(), N, H, W, C, S)
        return out

    def __getitem__(self, idx):
        return self.batch([int(idx)])[0]

    def loader(self, batch_size, shuffle=True, drop_last=True, rank=0, world=1):
        """One epoch of GLOBAL batches of ``batch_size * world`` images; yields this rank's rows.  The default CPU
        generator is consumed exactly like ``iter(DataLoader(...))`` of this torch does -- the iterator's base seed, then
        ``RandomSampler``'s seed for a private permutation generator, then per image the two crop offsets -- pinned against a
        real ``torch.utils.data.DataLoader`` in tests/test_input_and_components.py; every rank draws the same stream."""
        n = len(self)
        # creating a DataLoader iterator draws its ``_base_seed`` from the default generator (one int64 ``random_()``,
        # torch/utils/data/dataloader.py ``_BaseDataLoaderIter.__init__``) -- once per epoch, shuffled or not, BEFORE the
        # sampler draws anything; a loader that skipped it would diverge from the reference's stream at the first batch
        torch.empty((), dtype=torch.int64).random_()
        if shuffle:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            order = torch.randperm(n, generator=torch.Generator().manual_seed(seed))
        else:
            order = torch.arange(n)
        gb = batch_size * world
        stop = n - (n % gb) if drop_last else n
        for start in range(0, stop, gb):
            rows = order[start:start + gb]
            offs = self._crop_offsets(len(rows), self.image_size)          # global draw, then this rank's slice
            lo, hi = rank * batch_size, min((rank + 1) * batch_size, len(rows))
            if offs[0] is not None:
                offs = (offs[0][lo:hi].contiguous(), offs[1][lo:hi].contiguous())
            yield self.batch(rows[lo:hi], offs)
