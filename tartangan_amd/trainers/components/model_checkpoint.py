"""Checkpoint / resume (reference components/model_checkpoint.py:11-117) for the HIP trainers.

File layout and protocol are the reference's: ``<output_root>/checkpoints/<steps>/{g,g_target,d,opt_d,opt_g}.pt`` hold
whole pickled objects (``torch.save(model)``, :35-45), ``trainer.json`` the trainer state; loading goes through
``cp_model.state_dict()`` -> ``model.load_state_dict()`` (:56-68), so a checkpoint written by the reference's stock
modules loads into the HIP modules and back (identical state_dict keys) -- only the optimiser files
differ in class (``FusedAdam`` keeps its moments as two flat buffers).  Under data parallelism only rank 0 writes.
torch >= 2.6 defaults ``torch.load`` to weights_only=True, which refuses pickled modules: these are the run's own
files, so they are loaded with weights_only=False.
"""
import json
import os

import torch

from .base import TrainerComponent


class ModelCheckpointComponent(TrainerComponent):
    """Saves the models at regular intervals."""
    FILES = (('g', 'g.pt'), ('target_g', 'g_target.pt'), ('d', 'd.pt'), ('optimizer_d', 'opt_d.pt'), ('optimizer_g', 'opt_g.pt'))

    def on_train_begin(self, steps, logs):
        self._loaded_from = None
        if getattr(self.trainer.args, 'resume_training_step', None):
            self.trainer.steps = self.trainer.args.resume_training_step
            self.load_checkpoint()
        elif getattr(self.trainer.args, 'resume_training_latest', False):
            self.resume_training_from_latest()

    def on_batch_end(self, steps, logs):
        if steps and steps % self.trainer.args.checkpoint_freq == 0:
            if self._loaded_from != steps:                       # no immediate re-save of what was just loaded
                self.save_checkpoint(steps)

    def on_train_end(self, steps, logs):
        self.save_checkpoint(steps)

    def _is_writer(self):
        dp = getattr(self.trainer, 'data_parallel', None)
        return dp is None or dp.rank == 0

    def save_checkpoint(self, steps):
        if not self._is_writer():
            return
        os.makedirs(self.checkpoint_root, exist_ok=True)
        for name, filename in self.FILES:
            torch.save(getattr(self.trainer, name), f'{self.checkpoint_root}/{filename}')
        with open(f'{self.checkpoint_root}/trainer.json', 'w') as outfile:
            json.dump(self.trainer.get_state(), outfile)

    def load_checkpoint(self):
        self._loaded_from = self.trainer.steps
        for name, filename in self.FILES:
            cp_model = torch.load(f'{self.checkpoint_root}/{filename}', map_location=self.trainer.device, weights_only=False)
            getattr(self.trainer, name).load_state_dict(cp_model.state_dict())
        with open(f'{self.checkpoint_root}/trainer.json') as infile:
            self.trainer.set_state(json.load(infile))

    def resume_training_from_latest(self):
        latest_id = self.latest_checkpoint_id()
        if latest_id is not None:
            self.trainer.steps = latest_id
            self.load_checkpoint()
        else:
            print('No checkpoints found to resume.')

    def latest_checkpoint_id(self):
        """The largest integer-named directory under the checkpoints root, or None."""
        try:
            subdirs = os.listdir(self.all_checkpoints_root)
        except FileNotFoundError:
            return None
        ids = [int(k) for k in subdirs if k.isdigit()]
        return max(ids) if ids else None

    @property
    def checkpoint_root(self):
        return f'{self.all_checkpoints_root}/{self.trainer.steps}'

    @property
    def all_checkpoints_root(self):
        return f'{self.trainer.output_root}/checkpoints'

    @classmethod
    def add_args_to_parser(cls, parser):
        parser.add_argument('--checkpoint-freq', type=int, default=100000, help='Output a checkpoint every N batches')
        parser.add_argument('--resume-training-step', type=lambda v: None if v in (None, 'None', 'none') else int(v), default=None)
        parser.add_argument('--resume-training-latest', action='store_true')
