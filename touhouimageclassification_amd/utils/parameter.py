"""Global constants of the reference (TIC/utils/parameter.py:1-16), unchanged values."""
NUM_CLASSES = 120
IMAGE_SIZE = (256, 256)
VIT_IMAGE_SIZE = (224, 224)
DATA_DIR = "data/train"
UNFILTERED_DATA_DIR = "data/train"
FILTERED_DATA_DIR = "data/filtered"
TEST_DIR = "data/test"
CHECKPOINT_DIR = "checkpoint"
LOG_DIR = "log"


def get_image_size(model_name: str):
    """ViT variants take 224x224, everything else the dataset's 256x256 thumbnails (parameter.py:12-16)."""
    return VIT_IMAGE_SIZE if "vit" in model_name.lower() else IMAGE_SIZE
