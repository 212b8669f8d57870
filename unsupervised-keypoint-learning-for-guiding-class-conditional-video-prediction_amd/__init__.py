"""kpx -- MI355X-native detector_translator hot path.

Importing the package loads libkpx_hip.so and fails loudly if it has not been built (no CPU / PyTorch fallback).
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from . import layers, model_utils, networks, ops, variables  # noqa: F401
from .base_model import BaseModel  # noqa: F401
from .detector_translator_model import DetectorTranslatorModel  # noqa: F401
from .keypoint_model import KeypointModel  # noqa: F401
from .final_model import FinalModel  # noqa: F401
from .motion_generator_model import MotionGeneratorModel  # noqa: F401
from .vgg import Vgg19, synthetic_vgg19_weights  # noqa: F401
from . import data  # noqa: F401
from .data import ImagePairDataLoader, KeypointDataLoader, SequenceDataLoader  # noqa: F401

__all__ = ['BaseModel', 'DetectorTranslatorModel', 'KeypointModel', 'FinalModel', 'MotionGeneratorModel', 'Vgg19', 'synthetic_vgg19_weights', 'ImagePairDataLoader', 'KeypointDataLoader', 'SequenceDataLoader', 'data', 'layers', 'model_utils',
           'networks', 'ops', 'variables']
