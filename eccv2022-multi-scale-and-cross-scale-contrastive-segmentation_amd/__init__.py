"""mscs_amd -- MI355X-native hot path of the multi-scale / cross-scale dense pixel-contrastive
segmentation trainer (reference: RViMLab/ECCV2022-multi-scale-and-cross-scale-contrastive-segmentation).

Sub-packages mirror the reference's own package names so that its call surface is unchanged:
``mscs_amd.losses`` (LossWrapper, DenseContrastiveLossV2, DenseContrastiveLossV2_ms, TwoScaleLoss),
``mscs_amd.models``, ``mscs_amd.managers``, ``mscs_amd.utils``.  The compute path is the HIP
library ``libdcl_hip.so`` (C ABI in ``include/dcl_hip.h``) bound with ctypes in ``_lib``.
"""
__version__ = "0.1.0"
