"""The ablation presets of the reference's report as ONE table.

The reference ships each preset as its own launcher module (``python -m TIC.ViT.ntrain_grey`` ...), every one a copy of the same
``train_main(...)`` call with one or two arguments changed (TIC/ViT/ntrain.py:250-267 and TIC/ViT/ntrain_*.py).  Here the shared
arguments live in ``COMMON`` and a preset is the handful of keys that differ; the launcher modules of the same names
(``python -m touhouimageclassification_amd.ViT.ntrain_grey`` ...) are one-liners over ``run_preset``.
"""
from __future__ import annotations

from typing import Dict

from ..utils.parameter import FILTERED_DATA_DIR, UNFILTERED_DATA_DIR

COMMON = dict(PRETRAINED=True, MODEL_NAME='google/vit-large-patch16-224', LR=1e-5, WEIGHT_DECAY=0.01, FULL_FINETUNE=True,
              BATCH_SIZE=8, NUM_WORKERS=4, TRAIN_SPLIT=0.8, DATA_DIR=FILTERED_DATA_DIR, MAX_EPOCHS=20,
              ENABLE_MIX_UP=True, ENABLE_AUGMENTATION=True)

# TRAIN_ID -> what differs from COMMON                                                   reference launcher
PRESETS: Dict[str, dict] = {
    "nViT": {},                                                                             # ntrain.py:250-267
    "nViT_but_gen": dict(ENABLE_GENERALIZATION=False),                                      # ntrain_but_gen.py:5-20
    "nViT_but_div": dict(ENABLE_DIVERSITY=False),                                           # ntrain_but_div.py:5-20
    "nViT_but_mix": dict(ENABLE_MIX_UP=False),                                              # ntrain_but_mix.py:5-19
    "nViT_grey": dict(ONLY_GREY_AUGMENTATION=True),                                         # ntrain_grey.py:5-35
    "nViT_grey_unmix": dict(ONLY_GREY_AUGMENTATION=True, ENABLE_MIX_UP=False),              # ntrain_grey_unmix.py:5-35
    "nViT_unfiltered": dict(DATA_DIR=UNFILTERED_DATA_DIR),                                  # ntrain_unfiltered.py:5-34
    "nViT_unfiltered_unmix": dict(DATA_DIR=UNFILTERED_DATA_DIR, ENABLE_MIX_UP=False),       # ntrain_unfiltered_unmix.py:5-33
    "nViT_unfiltered_unaug": dict(DATA_DIR=UNFILTERED_DATA_DIR, ENABLE_MIX_UP=False, ENABLE_AUGMENTATION=False,
                                  BATCH_SIZE=16, PATIENCE=-1),                              # ntrain_unfiltered_unaug.py:5-20
}


def preset_kwargs(train_id: str) -> dict:
    if train_id not in PRESETS:
        raise KeyError(f"unknown preset {train_id!r}; known: {sorted(PRESETS)}")
    return dict(COMMON, TRAIN_ID=train_id, **PRESETS[train_id])


def run_preset(train_id: str, argv=None):
    from .ntrain import train_main
    return train_main(**preset_kwargs(train_id), argv=argv)
