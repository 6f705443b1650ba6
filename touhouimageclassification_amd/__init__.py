"""MI355X-native fine-tuning hot path for TouhouIC: hand-written gfx950 HIP kernels behind the C ABI of include/tic_hip.h
(libtic_hip.so), with the reference's Python entry points (ViT / ResNet / ResMoE / utils) on top.  Nothing is imported eagerly:
the HIP library is loaded by ``_lib.lib()`` on first use and there is no CPU fallback."""
__version__ = "0.1.0"
