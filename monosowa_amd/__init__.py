"""MI355X-native MonoDETR forward/backward path for MonoSOWA (gfx950 HIP kernels behind the
reference's MultiScaleDeformableAttention operator boundary)."""
__version__ = "0.1.0"
