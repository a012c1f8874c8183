"""bench.py on another build of the library: EFFDET_LIB_VARIANT=libeffdet_hip_<tag>.so python tools/variant_run.py [bench args].
For A/B timing of kernel variants inside one gpurun call (same box, same clocks); the product loads libeffdet_hip.so only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ood_object_detection_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['EFFDET_LIB_VARIANT'])
import bench
bench.main()
