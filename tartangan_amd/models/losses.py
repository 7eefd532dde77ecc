"""models/losses.py of the reference, on the HIP kernels.

``gradient_penalty`` (R1, reference models/losses.py:17-30) is the function that
forces double backward through the discriminator.  The hinge losses
(losses.py:7-14) are assigned by the trainers but never called; they are kept
for API parity on top of differentiable primitives.
"""
import torch

from .. import functional as TF


def discriminator_hinge_loss(real, fake):
    raise NotImplementedError('hinge losses are assigned but never called by trainers.cnn/iqn (cnn.py:86-87)')


def generator_hinge_loss(fake):
    raise NotImplementedError('hinge losses are assigned but never called by trainers.cnn/iqn (cnn.py:86-87)')


def gradient_penalty(preds, data):
    """mean_b sum_chw (d sum(preds) / d data)^2, differentiable w.r.t. the D parameters."""
    batch_size = data.size(0)
    total = TF._RowSum.apply(preds, preds.dim(), 1.0)           # preds.sum()
    with TF.input_grads_only():                                 # d/d(data) only: no parameter gradients in this pass
        grad_dout = torch.autograd.grad(
            outputs=total, inputs=data, create_graph=True, retain_graph=True, only_inputs=True)[0]
    assert grad_dout.size() == data.size()
    return TF.sumsq(grad_dout, 1.0 / batch_size)
