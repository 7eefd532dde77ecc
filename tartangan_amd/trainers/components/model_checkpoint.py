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
    """Periodic checkpoints of g / target_g / d / both optimisers + resume (same hooks, flags and files as the reference)."""
    # trainer attribute -> file name inside <output_root>/checkpoints/<steps>/ (model_checkpoint.py:35-45)
    FILES = {'g': 'g.pt', 'target_g': 'g_target.pt', 'd': 'd.pt', 'optimizer_d': 'opt_d.pt', 'optimizer_g': 'opt_g.pt'}
    STATE_FILE = 'trainer.json'
    _loaded_from = None

    # ------------------------------------------------------------------ paths
    def _dir(self, step=None):
        """Directory of the checkpoint taken at ``step`` (default: the trainer's current step count)."""
        root = os.path.join(self.trainer.output_root, 'checkpoints')
        return root if step is ... else f'{root}/{self.trainer.steps if step is None else step}'

    checkpoint_root = property(lambda self: self._dir())
    all_checkpoints_root = property(lambda self: self._dir(...))

    # ------------------------------------------------------------------ hooks
    def on_train_begin(self, steps, logs):
        args = self.trainer.args
        wanted = getattr(args, 'resume_training_step', None)
        if not wanted and getattr(args, 'resume_training_latest', False):
            wanted = self.latest_checkpoint_id()
            if wanted is None:
                print('No checkpoints found to resume.')
        self._loaded_from = None
        if wanted:
            self._resume(wanted)

    def on_batch_end(self, steps, logs):
        due = bool(steps) and steps % self.trainer.args.checkpoint_freq == 0
        if due and steps != self._loaded_from:          # a checkpoint that was just loaded is not written straight back
            self.save_checkpoint(steps)

    def on_train_end(self, steps, logs):
        self.save_checkpoint(steps)

    # ------------------------------------------------------------------ save / load
    def _is_writer(self):
        dp = getattr(self.trainer, 'data_parallel', None)
        return dp is None or dp.rank == 0

    def save_checkpoint(self, steps):
        trainer = self.trainer
        getattr(trainer, 'flush', lambda: None)()       # a deferred generator update (data parallel) lands first
        if not self._is_writer():
            return
        where = self.checkpoint_root
        os.makedirs(where, exist_ok=True)
        for attr, filename in self.FILES.items():
            torch.save(getattr(trainer, attr), os.path.join(where, filename))
        with open(os.path.join(where, self.STATE_FILE), 'w') as f:
            json.dump(trainer.get_state(), f)

    def load_checkpoint(self):
        """Load ``checkpoint_root`` (the directory named by ``trainer.steps``) through ``state_dict()`` of the pickled
        objects.  Everything is read and checked first; the trainer is only modified once every file has loaded."""
        trainer, where = self.trainer, self.checkpoint_root
        getattr(trainer, 'flush', lambda: None)()
        states = {}
        for attr, filename in self.FILES.items():
            obj = torch.load(os.path.join(where, filename), map_location=trainer.device, weights_only=False)
            states[attr] = obj.state_dict()
        with open(os.path.join(where, self.STATE_FILE)) as f:
            trainer_state = json.load(f)
        for attr, state in states.items():
            getattr(trainer, attr).load_state_dict(state)
        self._loaded_from = trainer.steps
        trainer.set_state(trainer_state)

    def _resume(self, step):
        self.trainer.steps = int(step)
        self.load_checkpoint()

    def resume_training_from_latest(self):
        step = self.latest_checkpoint_id()
        if step is None:
            print('No checkpoints found to resume.')
            return
        self._resume(step)

    def latest_checkpoint_id(self):
        """The largest integer-named directory under the checkpoints root, or None."""
        try:
            names = os.listdir(self.all_checkpoints_root)
        except FileNotFoundError:
            return None
        return max((int(k) for k in names if k.isdigit()), default=None)

    @classmethod
    def add_args_to_parser(cls, parser):
        parser.add_argument('--checkpoint-freq', type=int, default=100000, help='Output a checkpoint every N batches')
        parser.add_argument('--resume-training-step', type=lambda v: None if v in (None, 'None', 'none') else int(v), default=None)
        parser.add_argument('--resume-training-latest', action='store_true')
