from .detection_evaluator import ObjectDetectionEvaluator, PascalDetectionEvaluator  # noqa: F401
