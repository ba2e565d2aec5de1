"""partsbaseddetector_amd — MI355X (gfx950) inference path for PartsBasedDetector::detect().

Product code lives in csrc/ (HIP kernels + C ABI, built into libpbd_hip.so);
this package is the thin host-side mirror of the reference's interfaces.
"""
from .model import Model, make_person_model, make_face_like_model, make_tree_model, make_image, PERSON_TREE  # noqa: F401
from .detector import (PartsBasedDetector, HOGFeatures, SpatialConvolutionEngine, DynamicProgram, Candidate)  # noqa: F401
