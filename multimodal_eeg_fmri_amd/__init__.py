"""MI355X-native (gfx950) EEG<->fMRI bridge trainer hot path.

Host-side mirror of the reference's model-class surface over hand-written HIP
kernels (``csrc/``) reached through the C-ABI library ``libmmeeg_hip.so``
(``include/mmeeg_hip.h``).  No CPU fallback exists in this package.
"""
__version__ = "0.1.0"
