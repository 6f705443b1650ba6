"""``python -m touhouimageclassification_amd.ViT.ntrain_unfiltered`` -- launcher of the reference's preset of the same name
(TIC/ViT/ntrain_unfiltered.py); the arguments are the ``nViT_unfiltered`` row of ``presets.PRESETS``."""
from .presets import run_preset

if __name__ == '__main__':
    run_preset("nViT_unfiltered")
