"""`effdet` API surface of DavidPetrus/ood_object_detection on MI355X HIP kernels.

The reference's effdet/__init__.py is renamed away (a__init__.py), so scripts import
`effdet.factory`, `effdet.bench`, `effdet.anchors`, ... as submodules; the same works here, and the
usual names are also re-exported for convenience.
"""
from .anchors import Anchors, generate_detections, get_feat_sizes
from .bench import DetBenchPredict, DetBenchTrain, _post_process, unwrap_bench
from .config import get_efficientdet_config
from .efficientdet import EfficientDet
from .factory import create_model, create_model_from_config
