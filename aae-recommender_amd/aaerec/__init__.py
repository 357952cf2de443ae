"""aaerec - MI355X-native drop-in for the hot path of lgalke/aae-recommender.

Mirrors the reference package's module names (base, datasets, transforms, evaluation,
condition, aae) for the AdversarialAutoEncoder training path; the arithmetic of the step
runs in libaaerec_hip.so (hand-written gfx950 kernels behind the C ABI of
include/aaerec_hip.h).
"""
__version__ = "0.1.0"
