from .config_utils import Config, set_config_readonly, set_config_writeable
from .fpn_config import get_fpn_config, bifpn_config
from .model_config import get_efficientdet_config, default_detection_model_configs, FLAG_DEFAULTS
