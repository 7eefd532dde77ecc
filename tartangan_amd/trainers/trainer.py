"""Hot-path half of the reference's ``Trainer`` (trainers/trainer.py).

In scope: the latent / batch assembly helpers
``sample_z / sample_g / make_adversarial_batch / make_generator_batch``
(trainer.py:153-176) and the hyper-parameter flags of the step
(trainer.py:268-313).  The epoch loop, datasets, callbacks and filesystem
handling of the reference stay with the reference: a tartangan ``Trainer`` calls
``train_batch(images)`` once per DataLoader batch (trainer.py:96) and that call
is the drop-in boundary.
"""
import argparse

import torch

from .utils import set_device_from_args


class Trainer:
    def __init__(self, args, components=()):
        self.args = args
        if not hasattr(args, 'device'):
            set_device_from_args(args)
        self.components = list(components)
        self.steps = 0
        self.epoch = 1

    # ------------------------------------------------------------------ hot-path helpers
    @property
    def device(self):
        return self.args.device

    def sample_z(self, n=None):
        """Latents come from the CPU default generator, then move (trainer.py:153-156):
        a seed gives the same z on any device."""
        if n is None:
            n = self.args.batch_size
        return torch.randn(n, self.gan_config.latent_dims).to(self.device)

    def sample_g(self, n=None, target_g=False, **g_kwargs):
        z = self.sample_z(n)
        return (self.target_g if target_g else self.g)(z, **g_kwargs)

    def make_adversarial_batch(self, real_data, **g_kwargs):
        generated = self.sample_g(len(real_data), **g_kwargs)
        batch = torch.cat([real_data, generated], dim=0)
        labels = torch.zeros(len(batch), 1, device=self.device)
        labels[:len(labels) // 2] = 1
        return batch, labels

    def make_generator_batch(self, real_data, **g_kwargs):
        generated = self.sample_g(len(real_data), **g_kwargs)
        return generated, torch.ones(len(generated), 1, device=self.device)

    def build_models(self):
        raise NotImplementedError

    def train_batch(self, imgs):
        raise NotImplementedError

    # ------------------------------------------------------------------ flags of the step
    @classmethod
    def add_args_to_parser(cls, p):
        p.add_argument('--batch-size', type=int, default=128)
        p.add_argument('--lr-g', type=float, default=1e-4)
        p.add_argument('--lr-d', type=float, default=4e-4)
        p.add_argument('--lr-target-g', type=float, default=1e-3)
        p.add_argument('--no-cuda', action='store_true')
        p.add_argument('--grad-penalty', type=float, default=5.)
        p.add_argument('--config', default='64')
        p.add_argument('--model-scale', type=float, default=1.)
        p.add_argument('--g-base', default='mlp')
        p.add_argument('--norm', default='bn')
        p.add_argument('--activation', default='relu')

    @classmethod
    def default_args(cls, **overrides):
        p = argparse.ArgumentParser()
        cls.add_args_to_parser(p)
        args = p.parse_args([])
        for k, v in overrides.items():
            setattr(args, k, v)
        return args
