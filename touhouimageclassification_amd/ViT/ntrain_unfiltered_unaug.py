"""``python -m touhouimageclassification_amd.ViT.ntrain_unfiltered_unaug`` -- launcher of the reference's preset of the same name
(TIC/ViT/ntrain_unfiltered_unaug.py); the arguments are the ``nViT_unfiltered_unaug`` row of ``presets.PRESETS``."""
from .presets import run_preset

if __name__ == '__main__':
    run_preset("nViT_unfiltered_unaug")
