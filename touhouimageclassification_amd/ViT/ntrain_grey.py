"""``python -m touhouimageclassification_amd.ViT.ntrain_grey`` -- launcher of the reference's preset of the same name
(TIC/ViT/ntrain_grey.py); the arguments are the ``nViT_grey`` row of ``presets.PRESETS``."""
from .presets import run_preset

if __name__ == '__main__':
    run_preset("nViT_grey")
