"""Model construction entry points with the reference's signatures (effdet/factory.py:7-54):
`create_model(name, bench_task, num_classes, pretrained, checkpoint_path, checkpoint_ema, **kwargs)` and
`create_model_from_config(config, ...)`.

Order of operations (what the reference does, :20-54): config overrides -> EfficientDet -> pretrained weights ->
head reset for a different class count -> checkpoint -> optional task wrapper.  Build-specific behaviour:
* nothing can be fetched offline, so `pretrained_backbone` never triggers a download (weights come from
  `checkpoint_path`);
* `image_size=(H, W)` (extra keyword) runs the model at another resolution than the config default."""
from . import bench as _bench
from . import helpers as _helpers
from .config import get_efficientdet_config
from .efficientdet import EfficientDet

# keyword arguments that are copied onto the config when given (factory.py:30-34 of the reference)
_CONFIG_OVERRIDES = ('redundant_bias', 'label_smoothing', 'legacy_focal', 'jit_loss', 'soft_nms')
_WRAPPERS = {'train': lambda m, labeler: _bench.DetBenchTrain(m, create_labeler=labeler),
             'predict': lambda m, labeler: _bench.DetBenchPredict(m)}


def _apply_overrides(config, options):
    for name in _CONFIG_OVERRIDES:
        if options.get(name) is not None:
            setattr(config, name, options[name])
        options.pop(name, None)
    size = options.pop('image_size', None)
    if size is not None:
        config.image_size = tuple(size)


def create_model_from_config(config, bench_task='', num_classes=None, pretrained=False,
                             checkpoint_path='', checkpoint_ema=False, **kwargs):
    options = dict(kwargs)
    options.pop('pretrained_backbone', None)          # offline: the backbone is initialised locally, weights via checkpoint
    want_labeler = options.pop('bench_labeler', False)
    _apply_overrides(config, options)

    net = EfficientDet(config, pretrained_backbone=False, **options)
    if pretrained:
        _helpers.load_pretrained(net, config.url)
    if num_classes is not None and num_classes != config.num_classes:
        net.reset_head(num_classes=num_classes)
    if checkpoint_path:
        _helpers.load_checkpoint(net, checkpoint_path, use_ema=checkpoint_ema)
    wrap = _WRAPPERS.get(bench_task)
    return wrap(net, want_labeler) if wrap is not None else net


def create_model(model_name, bench_task='', num_classes=None, pretrained=False,
                 checkpoint_path='', checkpoint_ema=False, **kwargs):
    return create_model_from_config(get_efficientdet_config(model_name), bench_task=bench_task, num_classes=num_classes,
                                    pretrained=pretrained, checkpoint_path=checkpoint_path, checkpoint_ema=checkpoint_ema,
                                    **kwargs)
