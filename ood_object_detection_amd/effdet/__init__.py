"""`effdet` API surface of DavidPetrus/ood_object_detection on MI355X HIP kernels.

The reference's effdet/__init__.py is renamed away (a__init__.py), so scripts import
`effdet.factory`, `effdet.bench`, `effdet.anchors`, ... as submodules; the same works here, and the
usual names are also re-exported for convenience.

Two ways in:
  * `import ood_object_detection_amd.effdet ...`                       (the package under its own name)
  * `sys.path.insert(0, '<repo>/ood_object_detection_amd'); import effdet ...`  - the reference's import names.  In that
    case this file runs as the top-level module `effdet`: it then loads the real package and registers it, with every
    submodule, under the `effdet.*` names, so both spellings are ONE set of module objects (no duplicated classes).
"""
import sys as _sys

if __name__ == 'effdet':
    import importlib as _importlib
    import os as _os
    import pkgutil as _pkgutil

    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    _real = _importlib.import_module('ood_object_detection_amd.effdet')
    for _m in _pkgutil.walk_packages(_real.__path__, 'ood_object_detection_amd.effdet.'):
        _importlib.import_module(_m.name)
    _prefix = 'ood_object_detection_amd.effdet'
    for _k, _v in list(_sys.modules.items()):
        if _k == _prefix or _k.startswith(_prefix + '.'):
            _sys.modules['effdet' + _k[len(_prefix):]] = _v
else:
    from .anchors import Anchors, generate_detections, get_feat_sizes
    from .bench import DetBenchPredict, DetBenchTrain, _post_process, unwrap_bench
    from .config import get_efficientdet_config
    from .efficientdet import EfficientDet
    from .factory import create_model, create_model_from_config
