"""Drop-in for the reference's ``TIC/ViT/model.py`` -- same factory, MI355X-native underneath.

``ViT(num_classes, pretrained=True, model_name=None, wrap_model_name=True)`` (TIC/ViT/model.py:8-47)
returns an ``nn.Module`` with the surface the reference's callers rely on:
callable on a ``[B,3,224,224]`` tensor -> object with ``.logits`` (finetune.py:59-60, ntrain.py:47),
``.config.image_size`` (model.py:34), ``.base_model.parameters()`` (ntrain.py:35-37),
``.parameters()/.state_dict()/.load_state_dict()/.to()/.train()/.eval()``, transformers-5.x AND 4.x
state_dict keys.  Forward and backward run entirely in libtic_hip.so.

The reference resolves ``model_name`` through ``ensure()`` -> ``snapshot_download`` (utils/ensure.py:11-15).
This package never fetches: a name resolves to ``cache/<name>/`` if that directory exists locally,
otherwise to a built-in config preset (weights randomly initialised); ``pretrained=True`` without a
local checkpoint raises ``FileNotFoundError``.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from ..engine import VitEngine

VIT_IMAGE_SIZE = (224, 224)   # TIC/utils/parameter.py:3
CACHE_DIR = "cache"           # TIC/utils/ensure.py:8

_PRESETS = {
    "base": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072),
    "large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
    "tiny": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512),
    "micro": dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128, image_size=32),   # 5 tokens: harness tests
}


class ImageClassifierOutput:
    """Minimal stand-in for transformers' output object: ``.logits`` (and ``.loss``)."""

    def __init__(self, logits, loss=None):
        self.logits = logits
        self.loss = loss

    def __getitem__(self, i):
        return (self.logits,)[i] if self.loss is None else (self.loss, self.logits)[i]


class _Node(nn.Module):
    pass


class _VitFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, module):
        ctx.module = module
        logits, ctx.stamp = module._engine.forward_stamped(x)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.module._run_backward(dlogits, ctx.stamp)
        return None, None, None


def _old_to_new_key(k: str, strip_lightning: bool = True) -> str:
    """transformers-4.x checkpoint keys (what the authors' .pth files hold, finetune.py:250) and
    Lightning's extra ``vit.`` prefix (ntrain.py:27,192-193) -> 5.x keys."""
    if strip_lightning and (k.startswith("vit.vit.") or k.startswith("vit.classifier.")):
        k = k[4:]
    k = k.replace("vit.encoder.layer.", "vit.layers.")
    for old, new in ((".attention.attention.query.", ".attention.q_proj."), (".attention.attention.key.", ".attention.k_proj."),
                     (".attention.attention.value.", ".attention.v_proj."), (".attention.output.dense.", ".attention.o_proj."),
                     (".intermediate.dense.", ".mlp.fc1."), (".output.dense.", ".mlp.fc2.")):
        k = k.replace(old, new)
    return k


def _translate_keys_hook(module, state_dict, prefix, *_):
    """load_state_dict pre-hook: 4.x names under this module's prefix -> 5.x names, in place.  As a hook (not only an
    override of ``load_state_dict``) it also runs when the model is a CHILD of the module being loaded, e.g. a Lightning
    checkpoint restored through ``ViTLModule.load_state_dict`` (ntrain.py:27,232)."""
    for k in [k for k in state_dict if k.startswith(prefix)]:
        nk = prefix + _old_to_new_key(k[len(prefix):], strip_lightning=False)
        if nk != k:
            state_dict[nk] = state_dict.pop(k)


class TicViTForImageClassification(nn.Module):
    def __init__(self, config: SimpleNamespace, backend=None):
        super().__init__()
        self.config = config
        self.num_labels = config.num_labels
        if config.hidden_size != config.num_attention_heads * 64:
            raise ValueError("TIC HIP attention is built for head_dim 64 (hidden_size == 64 * num_attention_heads)")
        self._engine = VitEngine(config.hidden_size, config.num_attention_heads, config.intermediate_size,
                                 config.num_hidden_layers, config.num_labels, config.image_size, config.patch_size,
                                 config.num_channels, config.layer_norm_eps, backend=backend)
        self._table = self._engine.param_table()
        self._anchor = torch.zeros(1, requires_grad=True)
        self._bucket_hook = None
        self._build_tree()
        plist = list(self._params_by_name.values())
        # in-place updates through a Parameter (torch.optim.*, p.data.copy_, load_state_dict) bump THAT Parameter's version
        # counter; after `.to()` it is no longer the flat buffer's, so the bf16 operand copies follow the sum of both
        self._engine.version_probe = lambda: sum(p._version for p in plist)
        self.register_load_state_dict_pre_hook(_translate_keys_hook)
        self.reset_parameters()

    # ---- module tree with HF names; every Parameter is a view of the engine's flat buffer ----------------
    def _build_tree(self):
        self.vit = _Node()
        self.classifier = _Node()
        self._params_by_name: Dict[str, nn.Parameter] = {}
        for name, (off, shape) in self._table.items():
            parts = name.split(".")
            node = self
            for part in parts[:-1]:
                if isinstance(node, nn.ModuleList):
                    while len(node) <= int(part):
                        node.append(_Node())
                    node = node[int(part)]
                    continue
                if not hasattr(node, part):
                    setattr(node, part, nn.ModuleList() if part == "layers" else _Node())
                node = getattr(node, part)
            p = nn.Parameter(VitEngine.view(self._engine.params, off, shape))
            setattr(node, parts[-1], p)
            self._params_by_name[name] = p

    def _relink(self):
        e = self._engine
        for name, (off, shape) in self._table.items():
            p = self._params_by_name[name]
            p.data = VitEngine.view(e.params, off, shape)
            if p.grad is not None:
                p.grad = VitEngine.view(e.grads, off, shape) if e.grads is not None and e.grads.device == e.params.device else None
        e.mark_weights_dirty()

    def _apply(self, fn, recurse=True):
        new_flat = fn(self._engine.params)
        if new_flat.dtype != torch.float32:
            raise TypeError("TIC keeps fp32 master weights (the reference trains under autocast, never .half())")
        self._engine.to(new_flat.device)
        self._anchor = torch.zeros(1, requires_grad=True, device=new_flat.device)
        self._relink()
        return self

    @property
    def base_model(self):
        return self.vit

    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None):
        """``pretrained=False`` init (TIC/ViT/model.py:41-45 -> HF _init_weights): N(0, 0.02) Linear/Conv
        weights, zero biases, LayerNorm (1, 0), trunc-normal(0.02) cls / position embeddings."""
        g = None
        if seed is not None:
            g = torch.Generator().manual_seed(seed)
        std = self.config.initializer_range
        for name, p in self._params_by_name.items():
            if name.endswith("cls_token") or name.endswith("position_embeddings"):
                p.copy_(torch.empty(p.shape).normal_(0, std, generator=g).clamp_(-2 * std, 2 * std))
            elif "layernorm" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.copy_(torch.empty(p.shape).normal_(0, std, generator=g))
        self._engine.mark_weights_dirty()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = OrderedDict((_old_to_new_key(k), v) for k, v in state_dict.items())   # Lightning prefix; 4.x names: the pre-hook
        out = super().load_state_dict(sd, strict=strict, assign=False)
        self._engine.mark_weights_dirty()
        return out

    # ---- hot path -----------------------------------------------------------------------------------------
    def register_bucket_hook(self, hook):
        """hook(name, grad_slice) -- called during backward as each gradient bucket completes (DP)."""
        self._bucket_hook = hook

    def buckets(self):
        """[(name, start, end)] ranges of the flat gradient buffer in backward completion order (head, layer L-1 .. 0, embed)"""
        return self._engine.buckets()

    def broadcast_state(self, src: int = 0, group=None):
        """start-up broadcast of the flat fp32 master weights (one collective); the bf16 operand copies follow"""
        import torch.distributed as dist
        dist.broadcast(self._engine.params, src=src, group=group)
        self._engine.mark_weights_dirty()

    def forward(self, pixel_values: torch.Tensor, labels: Optional[torch.Tensor] = None, **kwargs):
        c = self.config
        if pixel_values.dim() != 4 or pixel_values.shape[1] != c.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration."
                             f" Expected {c.num_channels} but got {pixel_values.shape[1] if pixel_values.dim() == 4 else '?'}.")
        if pixel_values.shape[2] != c.image_size or pixel_values.shape[3] != c.image_size:
            raise ValueError(f"Input image size ({pixel_values.shape[2]}*{pixel_values.shape[3]}) doesn't match model"
                             f" ({c.image_size}*{c.image_size}).")
        x = pixel_values.to(torch.float32).contiguous()
        if x.device != self._engine.params.device:
            raise RuntimeError(f"pixel_values on {x.device} but the model is on {self._engine.params.device}")
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._params_by_name.values())
        if needs_grad:
            logits = _VitFunction.apply(self._anchor, x, self)
        else:   # no backward can follow: fc1 skips storing gelu'(u)
            logits = self._engine.forward_infer(x)
        loss = None
        if labels is not None:
            loss = torch.nn.functional.cross_entropy(logits, labels)
        return ImageClassifierOutput(logits, loss)

    def _run_backward(self, dlogits: torch.Tensor, stamp=None):
        e = self._engine
        params = self._params_by_name
        head_only = not any(p.requires_grad for n, p in params.items() if n.startswith("vit.embeddings") or n.startswith("vit.layers"))
        trainable = [p for p in params.values() if p.requires_grad]
        if any(p.grad is None for p in trainable):
            # zero_grad(set_to_none=True) (torch default) dropped the views: start from a clean buffer.  When EVERY gradient was dropped and the
            # whole network is trained, nothing in the buffer is to be kept: clear only the ranges that accumulate and let this backward store
            # the weight-matrix gradients (engine.zero_grads) -- what the fused step does; otherwise zero everything and accumulate.
            fresh = not head_only and all(p.grad is None for p in trainable) and len(trainable) == len(params)
            e.zero_grads(dlogits.shape[0], keep_matrices=fresh)
        hook = None
        if self._bucket_hook is not None:
            user = self._bucket_hook
            hook = lambda name, a, b: user(name, e.grads[a:b])   # noqa: E731
        e.backward(dlogits.to(torch.float32).contiguous(), bucket_hook=hook, head_only=head_only, stamp=stamp)
        for name, (off, shape) in self._table.items():
            p = params[name]
            if p.requires_grad and p.grad is None:
                p.grad = VitEngine.view(e.grads, off, shape)


def _config_from(model_name: Optional[str], num_classes: int, wrap_model_name: bool):
    """Local resolution only (never the network): <cache>/<name>/config.json, a directory path, or a preset."""
    cfg = dict(num_channels=3, image_size=224, patch_size=16, layer_norm_eps=1e-12, initializer_range=0.02,
               hidden_act="gelu", qkv_bias=True)
    local = None
    for cand in ((os.path.join(CACHE_DIR, model_name) if wrap_model_name else model_name), model_name):
        if cand and os.path.isdir(cand) and os.path.exists(os.path.join(cand, "config.json")):
            local = cand
            break
    if local:
        with open(os.path.join(local, "config.json")) as f:
            j = json.load(f)
        for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "image_size", "patch_size",
                  "num_channels", "layer_norm_eps", "initializer_range"):
            if k in j:
                cfg[k] = j[k]
    else:
        low = (model_name or "").lower()
        key = "micro" if "micro" in low else ("tiny" if "tiny" in low else ("base" if "base" in low else "large"))
        cfg.update(_PRESETS[key])
    cfg["num_labels"] = num_classes
    cfg["_local_dir"] = local
    cfg["_name"] = model_name
    return SimpleNamespace(**cfg)


def _load_local_checkpoint(model: TicViTForImageClassification, local: str) -> None:
    """from_pretrained(..., ignore_mismatched_sizes=True) semantics: load what matches by shape; a head of
    a different size stays freshly initialised."""
    sd = None
    st = os.path.join(local, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    elif os.path.exists(os.path.join(local, "pytorch_model.bin")):
        sd = torch.load(os.path.join(local, "pytorch_model.bin"), map_location="cpu")
    if sd is None:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {local}")
    own = model.state_dict()
    keep = {}
    for k, v in sd.items():
        nk = _old_to_new_key(k)
        if nk in own and tuple(own[nk].shape) == tuple(v.shape):
            keep[nk] = v
    model.load_state_dict(keep, strict=False)


def ViT(num_classes: int, pretrained: bool = True, model_name: str = None, wrap_model_name=True, backend=None):
    """Same signature and defaults as TIC/ViT/model.py:8 (default model: ViT-Large in21k, :20-22)."""
    if model_name is None:
        model_name = 'google/vit-large-patch16-224-in21k'
    config = _config_from(model_name, num_classes, wrap_model_name)
    model = TicViTForImageClassification(config, backend=backend)
    if pretrained:
        if config._local_dir is None:
            raise FileNotFoundError(
                f"pretrained weights for '{model_name}' are not available locally (looked for {os.path.join(CACHE_DIR, model_name)}/); "
                "this package never downloads. Place the checkpoint there or pass pretrained=False.")
        _load_local_checkpoint(model, config._local_dir)
        if model.config.image_size != VIT_IMAGE_SIZE[0]:
            raise ValueError(f"Pretrained model's image size {model.config.image_size} does not match "
                             f"the specified image size {VIT_IMAGE_SIZE[0]}.")
    return model
