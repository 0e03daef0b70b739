"""MI355X-native hot path of the action-conditioned video-prediction GAN.

Python host (this package) -> C ABI (include/acgan_hip.h) -> hand-written HIP kernels for
gfx950 (csrc/).  See DESIGN.md.
"""
__version__ = '0.1.0'
