"""Trainer components of the reference that a run behind the drop-in needs next to the step:
checkpoint / resume and the progress sampler.  Same class names, hooks and file layout as
``tartangan.trainers.components``; a tartangan ``Trainer`` can attach either family."""
from .base import TrainerComponent
from .image_sampler import ImageSamplerComponent
from .model_checkpoint import ModelCheckpointComponent
from .metrics import FIDComponent

__all__ = ['TrainerComponent', 'ImageSamplerComponent', 'ModelCheckpointComponent', 'FIDComponent']
