"""Counterpart of ``TIC/utils/serve.py``: model registry, checkpoint loading, single-image ``serve`` and the
directory evaluation ``full_judge`` that produces the reference's CSV (filename,predicted_class,confidence,actual_class,
correct,path) and top-1 accuracy -- the protocol behind the published accuracies (TIC/analysis/acc.py:30-55).

  get_model(model_type, num_classes)                         serve.py:24-45   ('resnet' | 'vit-base' | 'vit-large')
  load_model(model_type, num_classes, weights_path, device)  serve.py:47-81   tuple-or-dict checkpoints (finetune.py:249-258)
  serve(model, image_tensor, class_to_idx, device)           serve.py:83-114  -> (class name, softmax confidence)
  init(...) / full_judge(...)                                serve.py:116-230

The reference pushes ONE image at a time through PIL resize + the model; here a folder is evaluated in batches: raw
uint8 thumbnails -> one HIP resize/normalise kernel -> the HIP forward.  ``full_judge`` also reports top-5.
'resmoe' (serve.py:42) returns the dense MoE of ``ResMoE/train.get_model`` (its forward returns a tuple whose first element is the logits).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from ..ResNet.model import resnet152
from ..ViT.model import ViT
from .parameter import DATA_DIR, get_image_size
from .preprocess import get_class_to_idx, get_transforms

IMAGE_EXT = ('.jpg', '.jpeg', '.png', '.bmp', '.gif')
model_checkpoints = {
    'resnet': 'checkpoint/ResNet_model_final.pth',
    'vit-base': 'checkpoint/ViT_base_finetune_production_epoch10.pth',
    'vit-large': 'checkpoint/ViT_large_finetune_production_epoch25.pth',
}


def _canon(model_type: str) -> str:
    return model_type.lower().replace('_', '-')


def get_model(model_type: str, num_classes: int):
    kind = _canon(model_type)
    if kind == 'resnet':
        return resnet152(num_classes=num_classes)
    if kind == 'vit-base':
        return ViT(num_classes=num_classes, pretrained=False, model_name='google/vit-base-patch16-224-in21k', wrap_model_name=False)
    if kind == 'vit-large':
        return ViT(num_classes=num_classes, pretrained=False, model_name='google/vit-large-patch16-224-in21k', wrap_model_name=False)
    if kind == 'resmoe':
        from ..ResMoE import train as moet
        return moet.get_model()
    raise ValueError(f"Unsupported model type: {model_type}")


def load_model(model_type: str, num_classes: int, weights_path: Optional[str] = None, device: str = 'cuda'):
    kind = _canon(model_type)
    model = get_model(kind, num_classes)
    if weights_path is None:
        weights_path = model_checkpoints.get(kind)
        if weights_path is None:
            raise ValueError(f"No default checkpoint found for model type: {model_type}")
    ckpt = torch.load(weights_path, map_location='cpu', weights_only=False)
    model.load_state_dict(ckpt[0] if isinstance(ckpt, tuple) else ckpt)   # (model_sd, optim_sd[, sched_sd]) or a bare state_dict
    return model.to(device)


def _logits(model, x):
    out = model(x)
    if isinstance(out, tuple):      # MoEClassifier: (combined logits, gate weights, top-k indices)
        return out[0]
    return out.logits if hasattr(out, 'logits') else out


def serve(model, image_tensor: torch.Tensor, class_to_idx: Dict[str, int], device: str = 'cuda') -> Tuple[str, float]:
    """One preprocessed image [1,3,H,W] -> (predicted class name, softmax confidence)."""
    model.eval()
    idx_to_class = {v: k for k, v in class_to_idx.items()}
    with torch.no_grad():
        prob = torch.softmax(_logits(model, image_tensor.to(device)), dim=1)
        conf, idx = torch.max(prob, 1)
    return idx_to_class[idx.item()], conf.item()


def init(args=None, modelt=None, weights=None, device=None, data_dir=DATA_DIR):
    if args:
        modelt, weights, device = args.model, args.weights, args.device
    class_to_idx = get_class_to_idx(data_dir)
    model = load_model(modelt, len(class_to_idx), weights, device)
    transforms = get_transforms(data_dir, get_image_size(modelt))
    return model, transforms, class_to_idx


def _load_u8(path: str, staging: int):
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        im = im.convert('RGB')
        if im.size != (staging, staging):
            im = im.resize((staging, staging), Image.BILINEAR)
        return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())


def full_judge(model, transforms, class_to_idx, args=None, image=None, device=None, output=None, batch_size: int = 64, staging: int = 256):
    """Predict every image under `image` (class = directory name), optionally write the CSV; returns top-1 accuracy
    (single file: prints / returns the prediction)."""
    if args:
        image, device, output = args.image, args.device, args.output
    model.eval()
    idx_to_class = {v: k for k, v in class_to_idx.items()}
    if os.path.isfile(image):
        x = transforms(_load_u8(image, staging).unsqueeze(0).to(device))
        pred, conf = serve(model, x, class_to_idx, device)
        if not output:
            print(f"Prediction: {pred} (Confidence: {conf:.4f})")
        return pred, conf
    files = [(os.path.join(r, f), os.path.basename(r), f) for r, _, fs in os.walk(image) for f in sorted(fs)
             if os.path.splitext(f)[1].lower() in IMAGE_EXT]
    rows, correct, correct5 = [], 0, 0
    for s in range(0, len(files), batch_size):
        chunk = files[s:s + batch_size]
        raw = torch.stack([_load_u8(p, staging) for p, _, _ in chunk]).to(device)
        with torch.no_grad():
            prob = torch.softmax(_logits(model, transforms(raw)), dim=1)
        conf, idx = prob.max(1)
        top5 = prob.topk(min(5, prob.shape[1]), dim=1).indices
        for (path, label, fname), c, i, t5 in zip(chunk, conf.tolist(), idx.tolist(), top5.tolist()):
            pred = idx_to_class[i]
            correct += pred == label
            correct5 += label in [idx_to_class[j] for j in t5]
            rows.append(f"{fname},{pred},{c:.4f},{label},{pred == label},{path}")
    if output:
        with open(output, 'w') as f:
            f.write("filename,predicted_class,confidence,actual_class,correct,path\n" + "\n".join(rows) + ("\n" if rows else ""))
    n = max(len(files), 1)
    print(f"Total images processed: {len(files)}, Correct predictions: {correct}, Accuracy: {correct / n * 100:.2f}%, top-5: {correct5 / n * 100:.2f}%")
    return correct / n
