"""Attribute-dict config used in place of OmegaConf.

The reference builds its model configs with ``OmegaConf.create()`` and toggles
read-only mode through ``set_config_readonly/writeable``
(reference: effdet/config/config_utils.py:4-9).  OmegaConf is not a dependency
here; ``Config`` gives the same attribute + item access, ``update``, ``in`` and
deepcopy behaviour the hot path relies on.
"""
from copy import deepcopy


class Config(dict):
    """dict with attribute access and an optional read-only latch."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        object.__setattr__(self, '_readonly', False)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __setitem__(self, name, value):
        if object.__getattribute__(self, '_readonly'):
            raise RuntimeError('config is read-only: cannot set %r' % (name,))
        if isinstance(value, dict) and not isinstance(value, Config):
            value = Config(value)
        super().__setitem__(name, value)

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __deepcopy__(self, memo):
        out = Config()
        for k, v in self.items():
            dict.__setitem__(out, k, deepcopy(v, memo))
        return out

    def __reduce__(self):
        return (Config, (dict(self),))

    @staticmethod
    def create(d=None):
        return Config(d or {})


def set_config_readonly(conf):
    object.__setattr__(conf, '_readonly', True)


def set_config_writeable(conf):
    object.__setattr__(conf, '_readonly', False)
