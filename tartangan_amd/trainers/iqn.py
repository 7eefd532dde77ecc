"""SA-GAN-IQN trainer: drop-in for ``tartangan.trainers.iqn.IQNTrainer``
(reference trainers/iqn.py:28-156).  Identical to the CNN trainer except that the
discriminator is an ``IQNDiscriminator`` whose head returns ``(p_target, loss)``:
``d_loss = loss_real + loss_fake`` (iqn.py:118-120), the R1 penalty is taken on the
quantile-mean prediction (iqn.py:126) and ``g_loss`` is the head's loss against
all-ones targets (iqn.py:137).
"""
from .. import functional as TF
from ..models.blocks import IQNDiscriminatorOutput
from ..models.pluggan import IQNDiscriminator
from .cnn import CNNTrainer


class IQNTrainer(CNNTrainer):
    discriminator_class = IQNDiscriminator
    d_output_class = IQNDiscriminatorOutput
    activations = {k: v for k, v in CNNTrainer.activations.items() if k != 'elu'}      # iqn.py:41-44 has no 'elu'

    def _d_losses(self, real, fake, labels):
        bs = len(real)
        if self._d_pairable():
            p, d_loss = self.d(TF.Pair(real, fake), targets=labels)      # blocks and head on 2B images: loss_real + loss_fake
            return p.r, d_loss
        p_real, loss_real = self.d(real, targets=labels[:bs])     # taus drawn here (8B) ...
        _, loss_fake = self.d(fake, targets=labels[bs:])           # ... then here (8B)
        return p_real, TF.add(loss_real, loss_fake)

    def _rng_plan(self, bs):
        """z; taus of D(real), D(fake) (models/iqn.py:105-108 inside each head evaluation); z; taus of the G phase's D(fake)."""
        z = ('z', bs, self.gan_config.latent_dims)
        q = self.d.to_output.iqn.num_quantiles
        t = ('tau', q * bs, q)
        return [z, t, t, z, t]

    def _g_loss(self, fake, ones):
        _, g_loss = self.d(fake, targets=ones)
        return g_loss
