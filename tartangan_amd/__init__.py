"""tartangan_amd: MI355X-native (gfx950) SA-GAN / SA-GAN-IQN training step.

A drop-in for the G+D hot path of awentzonline/tartangan
(``tartangan.trainers.{cnn,iqn}`` + the ``tartangan.models`` generator /
discriminator API): hand-written HIP kernels behind the C ABI in
``include/tartangan_amd.h``; PyTorch-ROCm tensors are storage and autograd
glue only.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = '0.1.0'
