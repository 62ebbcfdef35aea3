"""MI355X-native (gfx950) implementation of PanFusion's denoising hot path.

Mirrors the reference's operator API for that path (``models.pano``,
``external.Perspective_and_Equirectangular``, ``utils.pano``) on top of a
C-ABI library of hand-written HIP kernels (``include/panfusion_hip.h``).
"""
__version__ = "0.1.0"
