"""``python -m touhouimageclassification_amd.ViT.ntrain_but_mix`` -- launcher of the reference's preset of the same name
(TIC/ViT/ntrain_but_mix.py); the arguments are the ``nViT_but_mix`` row of ``presets.PRESETS``."""
from .presets import run_preset

if __name__ == '__main__':
    run_preset("nViT_but_mix")
