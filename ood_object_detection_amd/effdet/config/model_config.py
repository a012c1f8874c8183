"""EfficientDet model hyper-parameters.

Restates `default_detection_model_configs` / `get_efficientdet_config`
(reference: effdet/config/model_config.py:16-85, :579-586) and the
`tf_efficientdet_d0..d5` rows of its parameter table (:420-487).

The reference reads four values from absl FLAGS at call time
(`pretrain_classes`, `alpha`, `gamma`, `bbox_coeff`; model_config.py:30,67,69,77).
Here they come from `FLAG_DEFAULTS` (the pretrain.py defaults, pretrain.py:43,60-62)
and, when absl is importable and the script has defined + parsed those flags, from
absl - so a script written against the reference keeps its behaviour.
"""
from copy import deepcopy

from .config_utils import Config

# pretrain.py:43 (pretrain_classes), :60 (alpha), :61 (gamma), :62 (bbox_coeff)
FLAG_DEFAULTS = dict(pretrain_classes=400, alpha=0.15, gamma=0.0, bbox_coeff=50.0)


def _flag(name):
    try:
        from absl import flags  # optional
        return getattr(flags.FLAGS, name)
    except Exception:
        return FLAG_DEFAULTS[name]


def default_detection_model_configs():
    h = Config()
    h.name = 'tf_efficientdet_d1'
    h.backbone_name = 'tf_efficientnet_b1'
    h.backbone_args = None
    h.image_size = (640, 640)
    h.num_classes = _flag('pretrain_classes')

    # feature + anchor config
    h.min_level = 3
    h.max_level = 7
    h.num_levels = h.max_level - h.min_level + 1
    h.num_scales = 3
    h.aspect_ratios = [(1.0, 1.0), (1.4, 0.7), (0.7, 1.4)]
    h.anchor_scale = 4.0

    # FPN and head config
    h.pad_type = 'same'
    h.act_type = 'swish'
    h.norm_layer = None
    h.norm_kwargs = dict(eps=0.001, momentum=0.01)
    h.box_class_repeats = 3
    h.fpn_cell_repeats = 3
    h.fpn_channels = 88
    h.separable_conv = True
    h.apply_resample_bn = True
    h.conv_after_downsample = False
    h.conv_bn_relu_pattern = False
    h.use_native_resize_op = False
    h.downsample_type = 'max'
    h.upsample_type = 'nearest'
    h.redundant_bias = True
    h.head_bn_level_first = False
    h.head_act_type = None

    h.fpn_name = None
    h.fpn_config = None
    h.fpn_drop_path_rate = 0.

    # classification loss
    h.alpha = _flag('alpha')
    h.gamma = _flag('gamma')
    h.label_smoothing = 0.
    h.legacy_focal = False
    h.jit_loss = False

    # localization loss
    h.delta = 0.1
    h.box_loss_weight = _flag('bbox_coeff')

    # nms
    h.soft_nms = False
    h.max_detection_points = 5000
    h.max_det_per_image = 100
    return h


_URL = 'https://github.com/rwightman/efficientdet-pytorch/releases/download/v0.1/'

efficientdet_model_param_dict = dict(
    # PyTorch-trained variants on timm's non-tf EfficientNets (model_config.py:90-158): symmetric padding, no redundant biases
    efficientdet_d0=dict(
        name='efficientdet_d0', backbone_name='efficientnet_b0', image_size=(512, 512),
        fpn_channels=64, fpn_cell_repeats=3, box_class_repeats=3, pad_type='', redundant_bias=False,
        backbone_args=dict(drop_path_rate=0.), url=_URL + 'efficientdet_d0-f3276ba8.pth'),
    efficientdet_d1=dict(
        name='efficientdet_d1', backbone_name='efficientnet_b1', image_size=(640, 640),
        fpn_channels=88, fpn_cell_repeats=4, box_class_repeats=3, pad_type='', redundant_bias=False,
        backbone_args=dict(drop_path_rate=0.2), url=_URL + 'efficientdet_d1-bb7e98fe.pth'),
    efficientdet_d2=dict(
        name='efficientdet_d2', backbone_name='efficientnet_b2', image_size=(768, 768),
        fpn_channels=112, fpn_cell_repeats=5, box_class_repeats=3, pad_type='', redundant_bias=False,
        backbone_args=dict(drop_path_rate=0.2), url=''),
    efficientdet_d3=dict(
        name='efficientdet_d3', backbone_name='efficientnet_b3', image_size=(896, 896),
        fpn_channels=160, fpn_cell_repeats=6, box_class_repeats=4, pad_type='', redundant_bias=False,
        backbone_args=dict(drop_path_rate=0.2), url=''),
    efficientdet_d4=dict(
        name='efficientdet_d4', backbone_name='efficientnet_b4', image_size=(1024, 1024),
        fpn_channels=224, fpn_cell_repeats=7, box_class_repeats=4,
        backbone_args=dict(drop_path_rate=0.2)),
    efficientdet_d5=dict(
        name='efficientdet_d5', backbone_name='efficientnet_b5', image_size=(1280, 1280),
        fpn_channels=288, fpn_cell_repeats=7, box_class_repeats=4,
        backbone_args=dict(drop_path_rate=0.2), url=''),
    tf_efficientdet_d0=dict(
        name='tf_efficientdet_d0', backbone_name='tf_efficientnet_b0', image_size=(512, 512),
        fpn_channels=64, fpn_cell_repeats=3, box_class_repeats=3,
        backbone_args=dict(drop_path_rate=0.2),
        url=_URL + 'tf_efficientdet_d0_34-f153e0cf.pth'),
    tf_efficientdet_d1=dict(
        name='tf_efficientdet_d1', backbone_name='tf_efficientnet_b1', image_size=(640, 640),
        fpn_channels=88, fpn_cell_repeats=4, box_class_repeats=3,
        backbone_args=dict(drop_path_rate=0.2),
        url=_URL + 'tf_efficientdet_d1_40-a30f94af.pth'),
    tf_efficientdet_d2=dict(
        name='tf_efficientdet_d2', backbone_name='tf_efficientnet_b2', image_size=(768, 768),
        fpn_channels=112, fpn_cell_repeats=5, box_class_repeats=3,
        backbone_args=dict(drop_path_rate=0.2),
        url=_URL + 'tf_efficientdet_d2_43-8107aa99.pth'),
    tf_efficientdet_d3=dict(
        name='tf_efficientdet_d3', backbone_name='tf_efficientnet_b3', image_size=(896, 896),
        fpn_channels=160, fpn_cell_repeats=6, box_class_repeats=4,
        backbone_args=dict(drop_path_rate=0.2),
        url=_URL + 'tf_efficientdet_d3_47-0b525f35.pth'),
    tf_efficientdet_d4=dict(
        name='tf_efficientdet_d4', backbone_name='tf_efficientnet_b4', image_size=(1024, 1024),
        fpn_channels=224, fpn_cell_repeats=7, box_class_repeats=4,
        backbone_args=dict(drop_path_rate=0.2),
        url=_URL + 'tf_efficientdet_d4_49-f56376d9.pth'),
    tf_efficientdet_d5=dict(
        name='tf_efficientdet_d5', backbone_name='tf_efficientnet_b5', image_size=(1280, 1280),
        fpn_channels=288, fpn_cell_repeats=7, box_class_repeats=4,
        backbone_args=dict(drop_path_rate=0.2),
        url=_URL + 'tf_efficientdet_d5_51-c79f9be6.pth'),
)


def get_efficientdet_config(model_name='tf_efficientdet_d1'):
    """Default config for `model_name`; KeyError on an unknown name (model_config.py:582)."""
    h = default_detection_model_configs()
    h.update(efficientdet_model_param_dict[model_name])
    h.num_levels = h.max_level - h.min_level + 1
    return deepcopy(h)
