"""Component protocol of the HIP trainers.

A component is anything with the six event hooks ``tartangan.trainers.components.base.TrainerComponent`` defines
(base.py:4-40: ``on_train_begin/on_train_end/on_batch_begin/on_batch_end(steps, logs)``,
``on_epoch_begin/on_epoch_end(steps, epochs, logs)``), a ``trainer`` back-reference assigned by the container and an
``add_args_to_parser`` classmethod.  In the INTEGRATION.md §1 deployment the host's own base class is the base and the
components of this package are accepted by duck typing; this mixin only exists so they also run without tartangan
installed.  Hooks are generated from the table below rather than spelled out one by one.
"""

EVENTS = {
    # hook name -> positional arguments the trainer passes
    'on_train_begin': ('steps', 'logs'),
    'on_train_end': ('steps', 'logs'),
    'on_batch_begin': ('steps', 'logs'),
    'on_batch_end': ('steps', 'logs'),
    'on_epoch_begin': ('steps', 'epochs', 'logs'),
    'on_epoch_end': ('steps', 'epochs', 'logs'),
}


class Unattached(AttributeError):
    """``component.trainer`` was read before a trainer took the component (AttributeError, as the reference raises)."""


class TrainerComponent:
    _owner = None

    def __init__(self, args=None):
        self.args = args

    def _get_trainer(self):
        owner = self._owner
        if owner is None:
            raise Unattached(f'{type(self).__name__} has no trainer yet: Trainer.attach() / add_components() assigns it')
        return owner

    def _set_trainer(self, owner):
        self._owner = owner

    trainer = property(_get_trainer, _set_trainer, doc='the trainer this component was attached to')

    @classmethod
    def add_args_to_parser(cls, parser):
        return None


def _ignore_event(arity):
    def hook(self, *args):
        if len(args) != arity:
            raise TypeError(f'expected {arity} positional arguments, got {len(args)}')
    return hook


for _name, _args in EVENTS.items():
    _hook = _ignore_event(len(_args))
    _hook.__name__ = _name
    _hook.__doc__ = f'event hook ({", ".join(_args)}); default: nothing'
    setattr(TrainerComponent, _name, _hook)
del _name, _args, _hook
