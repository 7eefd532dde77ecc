"""models/losses.py of the reference, on the HIP kernels.

``gradient_penalty`` (R1, reference models/losses.py:17-30) is the function that
forces double backward through the discriminator.  The hinge losses
(losses.py:7-14) are assigned by the trainers (cnn.py:86-87) but never called; they are
here for API parity, composed from the twice-differentiable primitives (relu = LeakyReLU(0)).
"""
import torch

from .. import functional as TF


def _mean(x):
    return TF._RowSum.apply(x, x.dim(), 1.0 / x.numel())


def discriminator_hinge_loss(real, fake):
    """(mean relu(1 - real), mean relu(1 + fake))  (losses.py:7-10)"""
    ones_r, ones_f = torch.ones_like(real), torch.ones_like(fake)
    loss_real = _mean(TF.leaky_relu(TF.add(ones_r, TF.scale(real, -1.0)), 0.0))
    loss_fake = _mean(TF.leaky_relu(TF.add(ones_f, fake), 0.0))
    return loss_real, loss_fake


def generator_hinge_loss(fake):
    """-mean(fake)  (losses.py:13-14)"""
    return _mean(TF.scale(fake, -1.0))


def gradient_penalty(preds, data):
    """mean_b sum_chw (d sum(preds) / d data)^2, differentiable w.r.t. the D parameters."""
    batch_size = data.size(0)
    total = TF._RowSum.apply(preds, preds.dim(), 1.0)           # preds.sum()
    with TF.input_grads_only():                                 # d/d(data) only: no parameter gradients in this pass
        grad_dout = torch.autograd.grad(
            outputs=total, inputs=data, create_graph=True, retain_graph=True, only_inputs=True)[0]
    assert grad_dout.size() == data.size()
    return TF.sumsq(grad_dout, 1.0 / batch_size)
