"""aaerec - MI355X-native drop-in for the hot path of lgalke/aae-recommender.

Mirrors the reference package's module names (base, datasets, transforms, evaluation,
condition, aae) for the AdversarialAutoEncoder training path; the arithmetic of the step
runs in libaaerec_hip.so (hand-written gfx950 kernels behind the C ABI of
include/aaerec_hip.h).
"""
__version__ = "0.1.0"

# Modules of the reference that are NOT on the path this build replaces (svd.py, baselines.py, utils.py ... - DESIGN.md 6)
# are not restated here.  When a user's own checkout of the reference sits BEHIND this package on sys.path (INTEGRATION.md
# A), its aaerec/ directory becomes a second portion of this package's search path: `from aaerec.svd import SVDRecommender`
# and `from aaerec.baselines import ...` (reference main.py:14-15) then resolve to the user's files, while every module
# this build mirrors is found here first.  Nothing of the reference is read unless the user put it on sys.path.
from pkgutil import extend_path as _extend_path
__path__ = _extend_path(__path__, __name__)
