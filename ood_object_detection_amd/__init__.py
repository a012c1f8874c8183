"""MI355X-native EfficientDet inference + OOD-scoring path."""
__version__ = '0.1.0'
