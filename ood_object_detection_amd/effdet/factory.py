"""create_model / create_model_from_config (reference: effdet/factory.py:7-54)."""
from .bench import DetBenchPredict, DetBenchTrain
from .config import get_efficientdet_config
from .efficientdet import EfficientDet, HeadNet
from .helpers import load_checkpoint, load_pretrained


def create_model(model_name, bench_task='', num_classes=None, pretrained=False,
                 checkpoint_path='', checkpoint_ema=False, **kwargs):
    config = get_efficientdet_config(model_name)
    return create_model_from_config(
        config, bench_task=bench_task, num_classes=num_classes, pretrained=pretrained,
        checkpoint_path=checkpoint_path, checkpoint_ema=checkpoint_ema, **kwargs)


def create_model_from_config(config, bench_task='', num_classes=None, pretrained=False,
                             checkpoint_path='', checkpoint_ema=False, **kwargs):
    pretrained_backbone = kwargs.pop('pretrained_backbone', True)
    if pretrained or checkpoint_path:
        pretrained_backbone = False
    # Offline build: a pretrained backbone would need a network fetch.  The reference default
    # (pretrained_backbone=True) is therefore honoured only when the caller passes weights.
    if pretrained_backbone:
        pretrained_backbone = False
    overrides = ('redundant_bias', 'label_smoothing', 'legacy_focal', 'jit_loss', 'soft_nms')
    for ov in overrides:
        value = kwargs.pop(ov, None)
        if value is not None:
            setattr(config, ov, value)
    # extra (build-defined) override: run at another input resolution than the model default
    image_size = kwargs.pop('image_size', None)
    if image_size is not None:
        config.image_size = tuple(image_size)
    labeler = kwargs.pop('bench_labeler', False)
    model = EfficientDet(config, pretrained_backbone=pretrained_backbone, **kwargs)
    if pretrained:
        load_pretrained(model, config.url)
    if num_classes is not None and num_classes != config.num_classes:
        model.reset_head(num_classes=num_classes)
    if checkpoint_path:
        load_checkpoint(model, checkpoint_path, use_ema=checkpoint_ema)
    if bench_task == 'train':
        model = DetBenchTrain(model, create_labeler=labeler)
    elif bench_task == 'predict':
        model = DetBenchPredict(model)
    return model
