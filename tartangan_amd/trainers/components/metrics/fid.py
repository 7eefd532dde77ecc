"""Frechet Inception Distance / Inception score during training: the reference's ``FIDComponent``
(trainers/components/metrics/fid.py:10-55) on ``tartangan_amd.inception_utils``.

Same flags (``--inception-moments``, ``--n-inception-imgs``, ``--fid-freq``, ``--cleanup-inception-model``), same hook
protocol and the same call: every ``fid_freq`` batches ``get_inception_metrics(trainer.sample_g, n_inception_imgs,
num_splits=5)`` and the three values appended to ``logs``.  The Inception network is handed in (``net=`` or
``trainer.args.inception_net``: a loaded torchvision Inception3, a ``WrapInception``, or any module returning ``(pool,
logits)``) because its pretrained weights have to come from the network; without one, ``on_train_begin`` raises with
that explanation.  Under data parallelism every rank samples a share and the moments are reduced over the ranks
(``inception_utils.sharded_moments``)."""
import os
import shutil
import tempfile

from .... import inception_utils
from ..base import TrainerComponent


class FIDComponent(TrainerComponent):
    def __init__(self, args=None, net=None):
        super().__init__(args)
        self._net = net

    def on_train_begin(self, steps, logs):
        args = self.trainer.args
        self.model_path = None
        if getattr(args, 'cleanup_inception_model', False):
            self.model_path = tempfile.mkdtemp()
            os.environ['TORCH_HOME'] = self.model_path
        net = self._net if self._net is not None else getattr(args, 'inception_net', None)
        if net is not None and not hasattr(net, 'forward_samples') and hasattr(net, 'Mixed_7c'):
            net = inception_utils.WrapInception(net.eval())             # a bare torchvision Inception3
        dp = getattr(self.trainer, 'data_parallel', None)
        self.get_inception_metrics = inception_utils.prepare_inception_metrics(
            args.inception_moments, self.trainer.device, False, net=net, group=None if dp is None else dp.group)

    def on_train_end(self, steps, logs):
        if self.model_path is not None:
            shutil.rmtree(self.model_path, ignore_errors=True)

    def on_batch_end(self, steps, logs):
        if steps and steps % self.trainer.args.fid_freq == 0:
            is_mean, is_std, fid = self._calculate()
            for key, value in (('fid', fid), ('inception_score_mean', is_mean), ('inception_score_std', is_std)):
                logs.setdefault(key, []).append(value)      # (the reference's logs is a defaultdict(list))

    def _calculate(self):
        trainer = self.trainer
        getattr(trainer, 'flush', lambda: None)()
        is_mean, is_std, fid = self.get_inception_metrics(trainer.sample_g, trainer.args.n_inception_imgs, num_splits=5)
        print('Inception Score is %3.3f +/- %3.3f' % (is_mean, is_std))
        print('FID is %5.4f' % (fid,))
        return is_mean, is_std, fid

    @classmethod
    def add_args_to_parser(cls, parser):
        parser.add_argument('--inception-moments', type=lambda v: None if v in (None, 'None', 'none') else str(v), default=None,
                            help='Path to pre-calculated inception moments')
        parser.add_argument('--n-inception-imgs', default=1000, type=int)
        parser.add_argument('--cleanup-inception-model', action='store_true')
        parser.add_argument('--fid-freq', default=10000, type=int, help='Calculate test metrics every N batches')
