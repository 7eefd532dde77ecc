"""INTEGRATION.md section 1, executed: tartangan's own ``Trainer`` (imported from the reference tree, build container
only) subclassed exactly as the snippet shows, delegating ``build_models`` / ``train_batch`` to the HIP-engine trainer;
run over the emulator and compared with the fixture the reference's own CNNTrainer produced.  Skipped where the
reference tree is absent (it never travels to the GPU box)."""
import argparse
import os
import sys

import pytest
import torch

from conftest import REPO, load_golden
from emulator import Emulator
from oracle.procedural import procedural_state, synthetic_images

REFERENCE = os.environ.get('TARTANGAN_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'tartangan')), reason='reference tree not present')


def test_reference_trainer_subclass_delegating_to_the_hip_step(single_thread):
    sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
    import make_golden          # noqa: F401  registers the inert torchvision / smart_open / boto3 stand-ins + sys.path
    from tartangan.trainers.trainer import Trainer               # the REFERENCE's Trainer
    from tartangan_amd import backend
    from tartangan_amd.models.pluggan import GAN_CONFIGS
    from tartangan_amd.trainers.cnn import CNNTrainer as _HipStep

    class CNNTrainer(Trainer):                                   # ---- INTEGRATION.md section 1, verbatim
        def build_models(self):
            self._hip = _HipStep(self.args)
            self._hip.build_models()
            self.g, self.target_g, self.d = self._hip.g, self._hip.target_g, self._hip.d
            self.optimizer_g, self.optimizer_d = self._hip.optimizer_g, self._hip.optimizer_d
            self.gan_config = self._hip.gan_config

        def train_batch(self, imgs):
            return self._hip.train_batch(imgs)

        sample_z = lambda self, n=None: self._hip.sample_z(n)    # noqa: E731

    fx = load_golden('c32_cnn_b16')
    prev = backend._set_backend_for_testing(Emulator())
    try:
        tr = object.__new__(CNNTrainer)                          # Trainer.__init__ only touches the filesystem
        tr.args = argparse.Namespace(config=GAN_CONFIGS['32'], model_scale=1., norm='bn', g_base='mlp', activation='relu',
                                     lr_g=1e-4, lr_d=4e-4, lr_target_g=1e-3, batch_size=16, grad_penalty=5., device='cpu',
                                     run_id='t', output='/tmp/tg_integration')
        torch.manual_seed(0)
        tr.build_models()
        assert list(tr.g.state_dict().keys()) == fx['state_keys']['g']
        tr.g.load_state_dict(procedural_state(tr.g.state_dict(), fx['weight_seed']))
        tr.target_g.load_state_dict(procedural_state(tr.target_g.state_dict(), fx['weight_seed'] + 1))
        tr.d.load_state_dict(procedural_state(tr.d.state_dict(), fx['weight_seed'] + 2))
        torch.manual_seed(fx['rng_seed'])
        logs = tr.train_batch(synthetic_images(16, 32, fx['img_seed']))
        for name in ('g_loss', 'd_loss', 'gp'):
            want = fx['steps'][0][name]
            assert abs(logs[name] - want) <= 1e-4 * max(abs(want), 1e-6), (name, logs[name], want)
        # what the reference's components read keeps working through the reference Trainer's own helpers
        assert tr.sample_z(3).shape == (3, tr.gan_config.latent_dims)
        with torch.no_grad():
            assert tr.sample_g(2, target_g=True).shape == (2, 3, 32, 32)       # Trainer.sample_g (trainer.py:158-164)
        assert tr.g.max_size == 32 and tr.g.config.latent_dims == 128
    finally:
        backend._set_backend_for_testing(prev)
