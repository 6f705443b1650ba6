"""Project-wide defaults shared by the harnesses: the VALUES of TIC/utils/parameter.py:1-16 under the same names, so that
``from ...utils.parameter import *`` in a harness written for the reference resolves to the same directories and sizes."""
from typing import Tuple

NUM_CLASSES = 120                      # dataset categories (parameter.py:1)

# image geometry: dataset thumbnails are 256 x 256 (ResNet input), every ViT variant is fed 224 x 224 (parameter.py:2-3)
IMAGE_SIZE: Tuple[int, int] = (256, 256)
VIT_IMAGE_SIZE: Tuple[int, int] = (224, 224)

# directory layout relative to the working directory (parameter.py:4-10)
DATA_DIR = "data"
UNFILTERED_DATA_DIR = f"{DATA_DIR}/unfiltered"
FILTERED_DATA_DIR = f"{DATA_DIR}/data_filtered_vit_base"
TEST_DIR = f"{DATA_DIR}/testset"
CHECKPOINT_DIR = "checkpoint"
LOG_DIR = "log"
CACHE_DIR = "cache"


def get_image_size(model_type: str) -> Tuple[int, int]:
    """input resolution by model type: ViT variants and the ViT-expert MoE 224 x 224, everything else the thumbnail size
    (parameter.py:12-16)"""
    kind = model_type.lower()
    return VIT_IMAGE_SIZE if ("vit" in kind or "resmoe" in kind) else IMAGE_SIZE
