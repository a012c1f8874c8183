"""Configuration tables of the reference (`effdet/config/`): model hyper-parameters, BiFPN graphs, the attribute-dict."""
from . import config_utils as _cu
from . import fpn_config as _fpn
from . import model_config as _mc

Config = _cu.Config
set_config_readonly = _cu.set_config_readonly
set_config_writeable = _cu.set_config_writeable
get_fpn_config = _fpn.get_fpn_config
bifpn_config = _fpn.bifpn_config
get_efficientdet_config = _mc.get_efficientdet_config
default_detection_model_configs = _mc.default_detection_model_configs
FLAG_DEFAULTS = _mc.FLAG_DEFAULTS

__all__ = ['Config', 'set_config_readonly', 'set_config_writeable', 'get_fpn_config', 'bifpn_config',
           'get_efficientdet_config', 'default_detection_model_configs', 'FLAG_DEFAULTS']
