"""bench.py on the phase-ablation build of the library (tools/ablate_gpu.sh); results are wrong by construction."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ood_object_detection_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libeffdet_hip_ablate.so')
import bench  # noqa: E402

bench.main()
