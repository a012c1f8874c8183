"""Stub of omegaconf: attribute dict with the few calls effdet/config makes."""
from copy import deepcopy


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, DictConfig):
            v = DictConfig(v)
        dict.__setitem__(self, k, v)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def __deepcopy__(self, memo):
        out = DictConfig()
        for k, v in self.items():
            dict.__setitem__(out, k, deepcopy(v, memo))
        return out


class OmegaConf(object):
    @staticmethod
    def create(d=None):
        return DictConfig(d or {})

    @staticmethod
    def set_readonly(conf, flag):
        pass

    @staticmethod
    def set_struct(conf, flag):
        pass
