"""FusedAdamW -- torch.optim.AdamW semantics (TIC/ViT/ntrain.py:39-41, finetune.py:314: one param group,
decoupled weight decay on EVERY parameter) as ONE HIP kernel over the engine's flat fp32 buffers
(28 B/param of HBM traffic + the bf16 GEMM-operand shadow written in the same pass).

It is a real ``torch.optim.Optimizer`` (param_groups / zero_grad / LR schedulers such as the
reference's ``get_linear_schedule_with_warmup`` keep working); ``torch.optim.AdamW(model.parameters())``
also works on a TIC model (parameters are ordinary tensors), just with more kernel launches.
"""
from __future__ import annotations

import torch

from .ViT.model import TicViTForImageClassification


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model: TicViTForImageClassification, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        if not isinstance(model, TicViTForImageClassification):
            raise TypeError("FusedAdamW steps the flat buffers of a TIC ViT; use torch.optim.AdamW for other modules")
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._m = self._v = None
        self.one_pass = True   # tic_vit_adamw (update + both bf16 operand copies in one pass) where it applies; False: tic_adamw + refresh (A/B)
        # frozen encoder (ntrain.py:35-37, full_finetune=False): only the head range is stepped
        e = model._engine
        all_trainable = all(p.requires_grad for p in model.parameters())
        self._range = (0, e.lay.n_params) if all_trainable else (e.lay.cls_w, e.lay.n_params)
        if not all_trainable and any(p.requires_grad for n, p in model.named_parameters() if not n.startswith("classifier")):
            raise ValueError("FusedAdamW supports full fine-tuning or a frozen base model (classifier only)")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        e = self.model._engine
        g = self.param_groups[0]
        if e.grads is None:
            raise RuntimeError("FusedAdamW.step() before any backward")
        if self._m is None or self._m.device != e.params.device:
            self._m = torch.zeros_like(e.params)
            self._v = torch.zeros_like(e.params)
        self._step += 1
        a, b = self._range
        if self.one_pass and (a, b) == (0, e.lay.n_params) and e.D % 64 == 0 and e.F % 64 == 0 and e.w16 is not None:
            # full fine-tuning: one pass updates the fp32 state AND writes both bf16 operand copies (w16 and the transposes the dX GEMMs
            # read) -- no cast + transpose pass over the weights before the next forward
            import ctypes
            _, st, _ = e._state(e._cur_B or 1)
            e.backend.call("tic_vit_adamw", ctypes.byref(st), self._m.data_ptr(), self._v.data_ptr(), float(g["lr"]), float(g["betas"][0]),
                           float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step, e.backend.stream())
            e.mark_weights_clean()
            return loss
        a -= a % 4
        n = b - a
        w16 = e.w16
        e.backend.call("tic_adamw", e.params[a:].data_ptr(), e.grads[a:].data_ptr(), self._m[a:].data_ptr(), self._v[a:].data_ptr(),
                       None if w16 is None else w16[a:].data_ptr(), n, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                       float(g["eps"]), float(g["weight_decay"]), self._step, e.backend.stream())
        e.mark_weights_dirty(w16_fresh=w16 is not None)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd["tic_flat"] = dict(step=self._step, m=self._m, v=self._v)
        return sd

    def load_state_dict(self, state_dict):
        flat = state_dict.pop("tic_flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self._step, self._m, self._v = flat["step"], flat["m"], flat["v"]
