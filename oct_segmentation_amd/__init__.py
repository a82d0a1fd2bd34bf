"""oct_segmentation_amd -- MI355X-native engine for the OCT segmentation hot path.

Host-side mirror of the reference's model API (``OCTSegmentationModel``,
``smp.create_model``, ``DiceLoss``, ``get_metrics``) over hand-written gfx950
HIP kernels behind a C ABI (``include/octseg.h``).
"""
__version__ = '0.1.0'
