"""Stub of absl.flags: FLAGS is a plain namespace the generator fills in."""


class _Flags(object):
    pass


FLAGS = _Flags()
# defaults of pretrain.py:43,60-62 / infer.py flags read inside effdet
FLAGS.pretrain_classes = 400
FLAGS.alpha = 0.15
FLAGS.gamma = 0.0
FLAGS.bbox_coeff = 50.0
FLAGS.supp_level_offset = 2         # infer.py:94, pretrain.py:63
FLAGS.multi_gpu = False


def _define(name, default, *a, **k):
    setattr(FLAGS, name, default)


DEFINE_integer = DEFINE_float = DEFINE_string = DEFINE_bool = DEFINE_boolean = _define
