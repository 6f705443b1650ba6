"""VitEngine -- owner of the HBM-resident state of one ViT replica and driver of the HIP phases.

Everything the step touches lives in a handful of large, flat, caller-owned buffers (the layout is
defined once, in C, by `tic_vit_layout`):

  params  fp32 [n_params]   master weights; every HF-named nn.Parameter is a VIEW into it
  grads   fp32 [n_params]   same layout; `param.grad` are views; one contiguous range per DP bucket
  w16     bf16 [n_params]   GEMM-operand shadow of params (autocast's per-step weight cast)
  wT16    bf16 [t_total]    transposed copies of the four big matrices per layer (dX GEMMs)
  workspace  bytes          activations saved for backward + backward temporaries, per batch size

Buckets (order of completion in backward): head, layer L-1 ... layer 0, embed.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _capi


class _HipBackend:
    """Default backend: libtic_hip.so on the current CUDA (HIP) device and stream."""

    def call(self, name, *args):
        from ._lib import call
        call(name, *args)

    def stream(self):
        from ._lib import current_stream
        return current_stream()

    def check_tensor(self, t):
        from ._lib import require_gpu
        require_gpu(t)


class VitEngine:
    def __init__(self, hidden: int, heads: int, mlp: int, layers: int, num_labels: int, image: int = 224,
                 patch: int = 16, channels: int = 3, eps: float = 1e-12, backend=None):
        self.D, self.H, self.F, self.L, self.C = hidden, heads, mlp, layers, num_labels
        self.image, self.patch, self.channels, self.eps = image, patch, channels, eps
        self.N = (image // patch) ** 2 + 1
        self.backend = backend or _HipBackend()
        self.lay = self._layout(1)
        self.device = torch.device("cpu")
        self.params = torch.zeros(self.lay.n_params, dtype=torch.float32)
        self.grads: Optional[torch.Tensor] = None
        self.w16: Optional[torch.Tensor] = None
        self.wT16: Optional[torch.Tensor] = None
        self._ws: "OrderedDict[Tuple[int, int], Tuple[torch.Tensor, _capi.TicVitState, _capi.TicVitLayout]]" = OrderedDict()
        self._weights_version = -1
        self._w16_fresh = False
        self._cur_B = None
        # extra version source: the module's Parameters keep their OWN version counters once `p.data` has been re-pointed
        # at a new flat buffer (`.to()`), so in-place updates by stock torch optimizers never show in `params._version`
        self.version_probe: Optional[Callable[[], int]] = None
        # saved activations live in `live_graphs` workspaces per batch size, used round-robin; each forward stamps its
        # workspace with a generation that backward checks (a stale stamp would mean silently wrong gradients)
        self.live_graphs = 1
        self._gen = 0
        self._slot_gen: Dict[Tuple[int, int], int] = {}
        self._slot_infer: set = set()   # (B, slot) whose last forward ran in inference mode: no backward on those
        self._slot_next: Dict[int, int] = {}

    # ---- layout ------------------------------------------------------------------------------------
    def _dims(self, B: int) -> _capi.TicVitDims:
        return _capi.TicVitDims(B, self.D, self.H, self.F, self.L, self.C, self.image, self.patch, self.channels, self.eps)

    def _layout(self, B: int) -> _capi.TicVitLayout:
        lay = _capi.TicVitLayout()
        d = self._dims(B)
        self.backend.call("tic_vit_layout", ctypes.byref(d), ctypes.byref(lay))
        return lay

    def param_table(self) -> "OrderedDict[str, Tuple[int, Tuple[int, ...]]]":
        """HF (transformers 5.x) state_dict key -> (element offset in the flat buffers, shape)."""
        y, D, F, C, N = self.lay, self.D, self.F, self.C, self.N
        t: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        t["vit.embeddings.cls_token"] = (y.cls, (1, 1, D))
        t["vit.embeddings.position_embeddings"] = (y.pos, (1, N, D))
        t["vit.embeddings.patch_embeddings.projection.weight"] = (y.patch_w, (D, self.channels, self.patch, self.patch))
        t["vit.embeddings.patch_embeddings.projection.bias"] = (y.patch_b, (D,))
        for i in range(self.L):
            b = y.layer0 + i * y.layer_stride
            p = f"vit.layers.{i}."
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                t[p + f"attention.{nm}.weight"] = (b + y.wqkv + j * D * D, (D, D))
                t[p + f"attention.{nm}.bias"] = (b + y.bqkv + j * D, (D,))
            t[p + "attention.o_proj.weight"] = (b + y.wo, (D, D))
            t[p + "attention.o_proj.bias"] = (b + y.bo, (D,))
            t[p + "layernorm_before.weight"] = (b + y.ln1_g, (D,))
            t[p + "layernorm_before.bias"] = (b + y.ln1_b, (D,))
            t[p + "layernorm_after.weight"] = (b + y.ln2_g, (D,))
            t[p + "layernorm_after.bias"] = (b + y.ln2_b, (D,))
            t[p + "mlp.fc1.weight"] = (b + y.w1, (F, D))
            t[p + "mlp.fc1.bias"] = (b + y.b1, (F,))
            t[p + "mlp.fc2.weight"] = (b + y.w2, (D, F))
            t[p + "mlp.fc2.bias"] = (b + y.b2, (D,))
        t["vit.layernorm.weight"] = (y.lnf_g, (D,))
        t["vit.layernorm.bias"] = (y.lnf_b, (D,))
        t["classifier.weight"] = (y.cls_w, (C, D))
        t["classifier.bias"] = (y.cls_b, (C,))
        return t

    def buckets(self) -> List[Tuple[str, int, int]]:
        """(name, start, end) element ranges of the flat gradient buffer, in backward completion order."""
        y = self.lay
        out = [("head", y.lnf_g, y.n_params)]
        for i in reversed(range(self.L)):
            b = y.layer0 + i * y.layer_stride
            out.append((f"layer{i}", b, b + y.layer_stride))
        out.append(("embed", 0, y.layer0))
        return out

    @staticmethod
    def view(flat: torch.Tensor, off: int, shape) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return flat[off:off + n].view(shape)

    # ---- device placement ---------------------------------------------------------------------------
    def to(self, device) -> None:
        device = torch.device(device)
        if device == self.params.device:
            return
        self.replace_params(self.params.to(device))

    def replace_params(self, flat: torch.Tensor) -> None:
        """adopt a NEW flat master-weight buffer (what `.to(device)` does): everything that holds a pointer into the old one --
        the per-batch-size state records with their workspaces, the bf16 operand copies, the gradient buffer -- is dropped or moved"""
        if flat.dtype != torch.float32 or flat.numel() != self.lay.n_params:
            raise ValueError("replace_params: need the flat fp32 buffer of this layout")
        self.params = flat
        self.grads = None if self.grads is None else self.grads.to(flat.device)
        self.w16 = self.wT16 = None
        self._ws.clear()
        self._slot_gen.clear()
        self._slot_infer.clear()
        self._weights_version = -1
        self.device = flat.device

    def _ensure_device_state(self) -> None:
        dev = self.params.device
        self.backend.check_tensor(self.params)
        if self.grads is None or self.grads.device != dev:
            self.grads = torch.zeros_like(self.params)
        if self.w16 is None or self.w16.device != dev:
            self.w16 = torch.empty(self.lay.n_params, dtype=torch.bfloat16, device=dev)
            self.wT16 = torch.empty(self.lay.t_total, dtype=torch.bfloat16, device=dev)
            self._weights_version = -1

    def _state(self, B: int, slot: int = 0):
        self._ensure_device_state()
        key = (B, slot)
        if key in self._ws:
            self._ws.move_to_end(key)
            return self._ws[key]
        lay = self._layout(B)
        # keep at most two batch sizes resident (train batch + tail / eval batch), `live_graphs` workspaces each
        while len(self._ws) >= 2 * max(1, self.live_graphs):
            old, _ = self._ws.popitem(last=False)
            self._slot_gen.pop(old, None)   # a backward that still needs it will raise (stale stamp), not read garbage
            self._slot_infer.discard(old)
        ws = torch.empty(lay.ws_bytes, dtype=torch.uint8, device=self.params.device)
        ws[lay.nt_scratch + _capi.NT_SCRATCH_BYTES - _capi.NT_FLAG_BYTES:lay.nt_scratch + _capi.NT_SCRATCH_BYTES].zero_()   # split-K flag words: zero ONCE
        st = _capi.TicVitState(self._dims(B), self.params.data_ptr(), self.grads.data_ptr(), self.w16.data_ptr(),
                               self.wT16.data_ptr(), ws.data_ptr())
        self._ws[key] = (ws, st, lay)
        return self._ws[key]

    # ---- phases --------------------------------------------------------------------------------------
    def mark_weights_dirty(self, w16_fresh: bool = False) -> None:
        """w16_fresh=True: the fused AdamW already wrote the bf16 shadow; only the transposes are stale."""
        self._weights_version = -1
        self._w16_fresh = w16_fresh

    def mark_weights_clean(self) -> None:
        """the fused AdamW (tic_vit_adamw) wrote w16 AND the transposes: nothing to refresh until the fp32 weights change again"""
        self._weights_version = self._version()
        self._w16_fresh = False

    def refresh_weights_if_needed(self) -> None:
        """bf16 operand copies follow the fp32 master weights (torch bumps `_version` on every in-place
        update of any view, e.g. optimizer.step(); our own AdamW keeps w16 fresh and calls mark_*)."""
        v = self._version()
        if v != self._weights_version:
            # any state works (dims.B is irrelevant for weights); make sure one exists
            _, st, _ = self._state(self._cur_B or 1)
            self.backend.call("tic_vit_refresh_weights", ctypes.byref(st), 1 if self._w16_fresh else 0, self.backend.stream())
            self._weights_version = self._version()
            self._w16_fresh = False

    def _version(self) -> int:
        v = self.params._version
        return v if self.version_probe is None else v + self.version_probe()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_stamped(x)[0]

    def forward_infer(self, x: torch.Tensor) -> torch.Tensor:
        """forward that NO backward will follow (torch.no_grad callers: validate_step, serve, full_judge): fc1 stores gelu(u) only, not
        the derivative the backward reads.  A backward on these activations raises."""
        return self.forward_stamped(x, infer=True)[0]

    def forward_stamped(self, x: torch.Tensor, infer: bool = False) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
        """-> (logits, stamp).  `stamp` = (B, workspace slot, generation) identifies the saved activations this forward
        left behind; pass it to `backward` to have a later overwrite (another forward at the same batch size, an
        eviction) detected instead of differentiated."""
        B = x.shape[0]
        self._cur_B = B
        slot = self._slot_next.get(B, 0) % max(1, self.live_graphs)
        self._slot_next[B] = slot + 1
        _, st, _ = self._state(B, slot)
        self.refresh_weights_if_needed()
        logits = torch.empty(B, self.C, dtype=torch.float32, device=x.device)
        self.backend.call("tic_vit_forward_infer" if infer else "tic_vit_forward", ctypes.byref(st), x.data_ptr(), logits.data_ptr(), self.backend.stream())
        self._gen += 1
        self._slot_gen[(B, slot)] = self._gen
        (self._slot_infer.add if infer else self._slot_infer.discard)((B, slot))
        return logits, (B, slot, self._gen)

    def zero_grads(self, B: int, keep_matrices: bool = False) -> None:
        """optimizer.zero_grad() for the flat gradient buffer.  keep_matrices: the per-layer weight-matrix ranges (99.7 % of ViT-L) are
        left alone and the NEXT full backward stores those gradients instead of accumulating into them -- no 1.2 GB zeroing pass and no
        read of the old values in the weight-gradient kernels' epilogues."""
        if not keep_matrices:
            self.grads.zero_()
            self._dw_overwrite = False
            return
        _, st, _ = self._state(B, (self._slot_next.get(B, 1) - 1) % max(1, self.live_graphs))
        self.backend.call("tic_vit_zero_grads", ctypes.byref(st), 1, self.backend.stream())
        self._dw_overwrite = True

    def backward(self, dlogits: torch.Tensor, bucket_hook: Optional[Callable[[str, int, int], None]] = None,
                 head_only: bool = False, stamp: Optional[Tuple[int, int, int]] = None) -> None:
        """Accumulates into self.grads (after zero_grads(keep_matrices=True): stores the weight-matrix gradients, accumulates the rest).
        bucket_hook(name, start, end) fires as each bucket's gradients are
        complete (enqueued) so a data-parallel wrapper can all-reduce it while earlier layers still run."""
        overwrite = 1 if getattr(self, "_dw_overwrite", False) else 0
        if overwrite and head_only:
            raise RuntimeError("TIC ViT backward: zero_grads(keep_matrices=True) must be followed by a FULL backward (it leaves the matrix "
                               "gradients undefined until one has stored them)")
        B = dlogits.shape[0]
        if stamp is None:   # fused step: the forward that just ran
            slot = (self._slot_next.get(B, 1) - 1) % max(1, self.live_graphs)
            if (B, slot) in self._slot_infer:
                raise RuntimeError("TIC ViT backward: the last forward at this batch size ran in inference mode (torch.no_grad): its "
                                   "activations lack what the backward reads")
        else:
            slot = stamp[1]
            if (B, slot) in self._slot_infer and self._slot_gen.get((B, slot)) == stamp[2]:
                raise RuntimeError("TIC ViT backward: this forward ran in inference mode (torch.no_grad)")
            if stamp[0] != B or self._slot_gen.get((B, slot)) != stamp[2]:
                raise RuntimeError(
                    "TIC ViT backward: the activations saved by this forward are gone -- another forward at batch size "
                    f"{stamp[0]} (or a third batch size) ran before this backward and re-used its workspace. Run each "
                    "backward before the next forward at the same batch size, or set `model._engine.live_graphs = k` to keep "
                    "k graphs per batch size alive.")
        _, st, _ = self._state(B, slot)
        s = self.backend.stream()
        bk = self.buckets()
        self.backend.call("tic_vit_backward_head", ctypes.byref(st), dlogits.data_ptr(), s)
        if bucket_hook:
            bucket_hook(*bk[0])
        if head_only:
            return
        for j, i in enumerate(reversed(range(self.L))):
            self.backend.call("tic_vit_backward_layer_ex", ctypes.byref(st), i, overwrite, s)
            if bucket_hook:
                bucket_hook(*bk[1 + j])
        self.backend.call("tic_vit_backward_embed", ctypes.byref(st), s)
        self._dw_overwrite = False
        if bucket_hook:
            bucket_hook(*bk[-1])

    def activation(self, B: int, name: str, layer: Optional[int] = None) -> Tuple[int, int]:
        """(byte offset, byte stride-to-next-layer) of a saved activation inside the workspace (tests)."""
        _, _, lay = self._state(B, (self._slot_next.get(B, 1) - 1) % max(1, self.live_graphs))
        off = getattr(lay, name)
        if layer is not None and name in ("a1", "mean1", "rstd1", "qkv", "lse", "o", "hmid", "a2", "mean2", "rstd2", "u", "g"):
            off += lay.layer_ws + layer * lay.layer_ws_stride
        return off
