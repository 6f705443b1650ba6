"""Project-wide defaults shared by the harnesses (values of TIC/utils/parameter.py:1-16; names kept for drop-in imports)."""
from typing import Tuple

NUM_CLASSES = 120                      # dataset categories

# image geometry: dataset thumbnails are 256 x 256, every ViT variant is fed 224 x 224
IMAGE_SIZE: Tuple[int, int] = (256, 256)
VIT_IMAGE_SIZE: Tuple[int, int] = (224, 224)

# directory layout relative to the working directory
_DATA_ROOT = "data"
DATA_DIR = f"{_DATA_ROOT}/train"
UNFILTERED_DATA_DIR = DATA_DIR
FILTERED_DATA_DIR = f"{_DATA_ROOT}/filtered"
TEST_DIR = f"{_DATA_ROOT}/test"
CHECKPOINT_DIR = "checkpoint"
LOG_DIR = "log"


def get_image_size(model_name: str) -> Tuple[int, int]:
    """input resolution for a model name: ViT variants 224 x 224, everything else the thumbnail size (parameter.py:12-16)"""
    is_vit = "vit" in model_name.lower()
    return VIT_IMAGE_SIZE if is_vit else IMAGE_SIZE
