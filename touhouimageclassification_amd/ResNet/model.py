"""Drop-in for the reference's ``TIC/ResNet/model.py``: ``resnet18/34/50/101/152(pretrained=False, progress=True,
num_classes=1000, **kw)`` returning an ``nn.Module`` that maps ``[B,3,H,W]`` fp32 to a raw ``[B,C]`` tensor
(TIC/ResNet/model.py:249-276, used at TIC/ResNet/train.py:52,239), with the reference's ``state_dict`` keys
(conv1.weight, bn1.{weight,bias,running_mean,running_var,num_batches_tracked}, layerN.i.convK/bnK/downsample.{0,1}, fc.*).

Forward and backward run in libtic_hip.so: NHWC bf16 activations, every convolution an implicit bf16 MFMA GEMM -- 1x1 stride-1
directly on the activation; 3x3, strided 1x1 and the 7x7 stem (image padded to 4 channels) gather their operand inside the GEMM;
stride-2 input gradients run as one small implicit GEMM per parity class of the input pixels -- no im2col / col2im buffer anywhere
on the ResNet-18 ... 152 paths (the explicit im2col route remains only for channel counts that are not multiples of 64);
train-mode BatchNorm (+ReLU, +residual) / pools as HIP kernels, fp32 master weights and fp32 gradient accumulation.
Blocks follow model.py:17-63 (BasicBlock) and :66-115
(Bottleneck, stride on the 3x3), stages / downsample follow ``_make_layer`` :185-208, init follows :168-183.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import struct

import torch
import torch.nn as nn

from ..engine import _HipBackend

_CFG = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]), "resnet50": ("bottleneck", [3, 4, 6, 3]),
        "resnet101": ("bottleneck", [3, 4, 23, 3]), "resnet152": ("bottleneck", [3, 8, 36, 3])}
_EPS, _MOMENTUM = 1e-5, 0.1
_TN_SLAB_BYTES = 64 << 20   # scratch of the few-tile weight gradients: parts x Cout x Cin x 4 B = 64 MiB when parts x tiles = 256


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')   # model.py:170
        self.weight = nn.Parameter(w)

    @property
    def stem(self) -> bool:
        """the 3-channel 7x7 / 2 stem (TIC/ResNet/model.py:148): an implicit GEMM over the image padded to 4 channels (include/tic_hip.h)"""
        return self.cin == 3 and self.k == 7 and self.stride == 2 and self.pad == 3

    @property
    def kp(self):
        return 256 if self.stem else (self.k * self.k * self.cin + 63) // 64 * 64


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = c
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _FC(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        lin = nn.Linear(cin, cout)   # default nn.Linear init, as the reference (model.py:165)
        self.weight, self.bias = lin.weight, lin.bias


class _Block(nn.Module):
    def __init__(self, kind, inplanes, planes, stride, downsample):
        super().__init__()
        self.kind = kind
        if kind == "basic":
            self.conv1, self.bn1 = _Conv(inplanes, planes, 3, stride, 1), _BN(planes)
            self.conv2, self.bn2 = _Conv(planes, planes, 3, 1, 1), _BN(planes)
            out = planes
        else:
            self.conv1, self.bn1 = _Conv(inplanes, planes, 1, 1, 0), _BN(planes)
            self.conv2, self.bn2 = _Conv(planes, planes, 3, stride, 1), _BN(planes)
            self.conv3, self.bn3 = _Conv(planes, planes * 4, 1, 1, 0), _BN(planes * 4)
            out = planes * 4
        self.downsample = nn.Sequential(_Conv(inplanes, out, 1, stride, 0), _BN(out)) if downsample else None


class _ResNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, module):
        ctx.module = module
        logits, ctx.tape = module._forward_impl(x, train=module.training, record=True)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.module._backward_impl(dlogits.contiguous().float(), ctx.tape)
        ctx.tape = None
        return None, None, None


class TicResNet(nn.Module):
    def __init__(self, arch: str, num_classes: int = 1000, zero_init_residual: bool = False, backend=None, **unsupported):
        super().__init__()
        for k in ("groups", "width_per_group", "replace_stride_with_dilation", "norm_layer"):
            if unsupported.get(k) not in (None, 1, 64, [False, False, False]):
                raise NotImplementedError(f"TIC HIP ResNet supports the reference's defaults only ({k} given)")
        kind, counts = _CFG[arch]
        self.arch, self.num_classes = arch, num_classes
        self.backend = backend or _HipBackend()
        self.conv1, self.bn1 = _Conv(3, 64, 7, 2, 3), _BN(64)
        inplanes, exp = 64, (1 if kind == "basic" else 4)
        for si, (planes, n) in enumerate(zip([64, 128, 256, 512], counts)):
            blocks = []
            for bi in range(n):
                stride = 2 if (bi == 0 and si > 0) else 1
                blocks.append(_Block(kind, inplanes, planes, stride, bi == 0 and (stride != 1 or inplanes != planes * exp)))
                inplanes = planes * exp
            setattr(self, f"layer{si + 1}", nn.Sequential(*blocks))
        self.fc = _FC(inplanes, num_classes)
        if zero_init_residual:   # model.py:178-183
            for m in self.modules():
                if isinstance(m, _Block):
                    nn.init.constant_((m.bn3 if m.kind == "bottleneck" else m.bn2).weight, 0)
        self._anchor = torch.zeros(1, requires_grad=True)
        self._bucket_hook = None

    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        self._anchor = torch.zeros(1, requires_grad=True, device=self.fc.weight.device)
        return self

    # ---- thin op wrappers over the C ABI ---------------------------------------------------------------------
    def _call(self, name, *args):
        self.backend.call(name, *args, self.backend.stream())

    def _gemm_nt(self, A, Bw, M, N, K, add_into=None):
        """A [M,K] . Bw [N,K]^T -> bf16 [M,N]; add_into (bf16 [M,N]): the product is ADDED to it in place by the GEMM epilogue
        (TIC_EPI_ADDAUX: bf16(bf16(acc) + aux), what a separate add pass over the stored product gives)"""
        if add_into is not None:
            self._call("tic_gemm_nt_bf16", A.data_ptr(), Bw.data_ptr(), M, N, K, 7, None, add_into.data_ptr(), None, None, None, add_into.data_ptr(), None, 0)
            return add_into
        out = torch.empty(M, N, dtype=torch.bfloat16, device=A.device)
        self._call("tic_gemm_nt_bf16", A.data_ptr(), Bw.data_ptr(), M, N, K, 0, None, out.data_ptr(), None, None, None, None, None, 0)
        return out

    def _pack_variants(self, conv: _Conv):
        """which bf16 operand forms of a conv weight the step reads: [Cout, Kp] (0) always; for the input gradient either the flipped
        channel-transposed filter of the implicit GEMM (2) or the plain transpose (1)"""
        if conv.stem:
            return (3,)   # the stem layout; its input needs no gradient
        if conv is self.conv1:
            return (0,)
        if self._s2_dgrad(conv):   # stride 2: one filter per parity class of the input gradient (3x3: four of them; 1x1: the flipped filter itself)
            return (0, 4, 5, 6, 7) if conv.k == 3 else (0, 2)
        return (0, 2) if (self._implicit(conv) and conv.stride == 1 and conv.cout % 64 == 0) else (0, 1)

    def _refresh_packs(self, dev):
        """re-pack EVERY conv weight in one launch when any fp32 weight changed (optimizer step, load_state_dict, .to()): persistent
        output buffers + a descriptor table in device memory, rebuilt only when a pointer moves"""
        convs = [m for m in self.modules() if isinstance(m, _Conv)]
        key = (tuple(c.weight._version for c in convs), tuple(c.weight.data_ptr() for c in convs), dev)
        st = self.__dict__.setdefault("_pack_state", {})
        if st.get("key") == key:
            return
        ptrs = key[1:]
        if st.get("ptrs") != ptrs:
            descs = []
            for c in convs:
                packed = c.__dict__["_packed"] = {}
                for t in self._pack_variants(c):
                    shape = {0: (c.cout, c.kp), 1: (c.kp, c.cout), 2: (c.cin, c.k * c.k * c.cout), 3: (c.cout, c.kp), 4: (c.cin, c.cout),
                             5: (c.cin, 2 * c.cout), 6: (c.cin, 2 * c.cout), 7: (c.cin, 4 * c.cout)}[t]
                    packed[t] = torch.empty(shape, dtype=torch.bfloat16, device=dev)
                    descs.append(struct.pack("<QQiiiiii", c.weight.data_ptr(), packed[t].data_ptr(), c.cout, c.cin, c.k, c.k, t, 0))
            st["n"] = len(descs)
            st["table"] = torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev)
            st["ptrs"] = ptrs
        self._call("tic_conv_weight_pack_many", st["table"].data_ptr(), st["n"])
        st["key"] = key

    def _pack(self, conv: _Conv, transposed):
        """bf16 GEMM operand of a conv weight ([Cout, Kp], its transpose (1), or the flipped channel-transposed filter of the
        implicit-GEMM input gradient (2)); kept fresh by _refresh_packs at the start of every forward"""
        return conv.__dict__["_packed"][int(transposed)]

    @staticmethod
    def _s2_dgrad(conv: _Conv) -> bool:
        """stride-2 convolutions whose input gradient runs as parity-class implicit GEMMs (tic_conv_igemm_dgrad_s2): the 3x3 / pad 1 of a
        stage's first block and the 1x1 downsample projection (TIC/ResNet/model.py:87, 193-197)"""
        return conv.stride == 2 and conv.cout % 64 == 0 and conv.cin % 8 == 0 and ((conv.k == 3 and conv.pad == 1) or (conv.k == 1 and conv.pad == 0))

    @staticmethod
    def _implicit(conv: _Conv) -> bool:
        # 3x3 convolutions and the strided 1x1 downsample projections (a 1x1 stride-1 conv is a plain GEMM on the activation)
        return conv.stem or ((conv.k > 1 or conv.stride > 1) and conv.cin % 64 == 0 and conv.cout % 8 == 0)

    def _conv_fwd(self, conv: _Conv, x, B, H, W):
        """x [B,H,W,Cin] bf16 -> (y [M,Cout] bf16, col or None, Ho, Wo)"""
        Ho = (H + 2 * conv.pad - conv.k) // conv.stride + 1
        Wo = (W + 2 * conv.pad - conv.k) // conv.stride + 1
        M = B * Ho * Wo
        if self._implicit(conv):   # 3x3: gather inside the GEMM, no im2col buffer (the backward gathers again from x)
            y = torch.empty(M, conv.cout, dtype=torch.bfloat16, device=x.device)
            self._call("tic_conv_igemm_fwd", x.data_ptr(), self._pack(conv, 3 if conv.stem else 0).data_ptr(), y.data_ptr(), B, H, W,
                       4 if conv.stem else conv.cin, conv.cout, conv.k, conv.k, conv.stride, conv.pad)   # stem: x is the 4-channel-padded image
            return y, x, Ho, Wo
        if conv.k == 1 and conv.stride == 1:
            col = x
        else:
            col = torch.empty(M, conv.kp, dtype=torch.bfloat16, device=x.device)
            self._call("tic_im2col_bf16", x.data_ptr(), col.data_ptr(), B, H, W, conv.cin, conv.k, conv.k, conv.stride, conv.pad)
        return self._gemm_nt(col, self._pack(conv, 0), M, conv.cout, conv.kp), col, Ho, Wo

    def _bn_scratch(self, dev):
        """partial-sum scratch of the BatchNorm column reductions: ONE buffer for every layer of the model (the kernels carry no state
        in it between calls, launches are stream-ordered).  Sized for the widest layer at the most row splits the library uses:
        tic_batchnorm_scratch_bytes <= (1 + 512) x 2 x 2048 floats = 8.4 MB; the library checks the size on every call"""
        s = self.__dict__.get("_bn_scr")
        if s is None or s.device != dev:
            s = self.__dict__["_bn_scr"] = torch.empty(513 * 2 * 2048 * 4, dtype=torch.uint8, device=dev)
        return s

    def _bn_fwd(self, bn: _BN, x, M, identity, relu, train):
        dev = x.device
        mean, rstd = torch.empty(bn.c, device=dev), torch.empty(bn.c, device=dev)
        scratch = self._bn_scratch(dev)
        y = torch.empty_like(x)
        self._call("tic_batchnorm_fwd", x.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                   bn.num_batches_tracked.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scratch.data_ptr(), scratch.numel(),
                   None if identity is None else identity.data_ptr(), y.data_ptr(), M, bn.c, _EPS, _MOMENTUM, 1 if train else 0, 1 if relu else 0)
        return y, mean, rstd

    @staticmethod
    def _grad_buf(p: nn.Parameter):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def _begin_backward(self, dev):
        """One flat fp32 buffer backs every .grad (zeroed by ONE fill when the optimizer dropped the grads) and one
        flat scratch holds the [Cout, Kp] weight-gradient GEMM outputs of all convs."""
        params = [p for p in self.parameters() if p.requires_grad]
        flat = self.__dict__.get("_flat_grad")
        total = sum(p.numel() for p in params)
        if flat is None or flat.device != dev or flat.numel() != total:
            flat = self.__dict__["_flat_grad"] = torch.empty(total, dtype=torch.float32, device=dev)
        if all(p.grad is None for p in params):
            flat.zero_()
            off = 0
            for p in params:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        else:   # kept gradients (zero_grad(set_to_none=False), accumulation): every .grad must still BE its range of the flat buffer,
            off = 0   # because the data-parallel buckets are ranges of it
            for p in params:
                want = flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    want.zero_()
                    p.grad = want
                elif p.grad.data_ptr() != want.data_ptr():
                    want.copy_(p.grad)
                    p.grad = want
                off += p.numel()
        convs = [m for m in self.modules() if isinstance(m, _Conv)]
        need = sum(c.cout * c.kp for c in convs)
        dws = self.__dict__.get("_dw_scratch")
        if dws is None or dws.device != dev or dws.numel() != need:
            dws = self.__dict__["_dw_scratch"] = torch.empty(need, dtype=torch.float32, device=dev)
            off = 0
            for c in convs:
                c.__dict__["_dw_view"] = dws[off:off + c.cout * c.kp]
                off += c.cout * c.kp
        dws.zero_()
        # slab scratch of the few-tile weight gradients (row parts stored + one reduce launch instead of stream-K atomics)
        slab = self.__dict__.get("_tn_slab")
        if slab is None or slab.device != dev:
            slab = self.__dict__["_tn_slab"] = torch.empty(_TN_SLAB_BYTES, dtype=torch.uint8, device=dev)
        self.backend.call("tic_gemm_tn_scratch", slab.data_ptr(), slab.numel())

    def _conv_bwd(self, conv: _Conv, dy, col, B, H, W, need_dx: bool, dx_accumulate_into=None):
        """dy [M,Cout]; returns dx [B*H*W, Cin] bf16 (or None) and accumulates the weight gradient"""
        M = dy.shape[0]
        dw = conv.__dict__["_dw_view"]   # zeroed by _begin_backward
        if self._implicit(conv):   # col is the NHWC input itself
            self._call("tic_conv_igemm_wgrad", dy.data_ptr(), col.data_ptr(), dw.data_ptr(), B, H, W, 4 if conv.stem else conv.cin, conv.cout,
                       conv.k, conv.k, conv.stride, conv.pad)
        else:
            self._call("tic_gemm_tn_bf16", dy.data_ptr(), col.data_ptr(), dw.data_ptr(), M, conv.cout, conv.kp)
        if not need_dx:   # (the fold of dw into the OIHW .grad happens for all convs at once, at the end of the backward)
            return None
        if self._implicit(conv) and conv.stride == 1 and conv.cout % 64 == 0 and dx_accumulate_into is None:
            # input gradient = the same gather GEMM over dY with the flipped, channel-transposed filter
            Ho = (H + 2 * conv.pad - conv.k) // conv.stride + 1
            Wo = (W + 2 * conv.pad - conv.k) // conv.stride + 1
            dx = torch.empty(B * H * W, conv.cin, dtype=torch.bfloat16, device=dy.device)
            self._call("tic_conv_igemm_fwd", dy.data_ptr(), self._pack(conv, 2).data_ptr(), dx.data_ptr(), B, Ho, Wo, conv.cout, conv.cin,
                       conv.k, conv.k, 1, conv.k - 1 - conv.pad)
            return dx
        if self._s2_dgrad(conv) and H % 2 == 0 and W % 2 == 0:
            # stride 2: every parity class of the input pixels is its own small implicit GEMM over dY, stored (or added) in place
            acc = dx_accumulate_into is not None
            # (a 1x1 / 2 projection reaches the even pixels only: a fresh buffer starts at zero; the model always accumulates there)
            dx = dx_accumulate_into if acc else (torch.empty if conv.k == 3 else torch.zeros)(B * H * W, conv.cin, dtype=torch.bfloat16, device=dy.device)
            for py, px in (((0, 0), (0, 1), (1, 0), (1, 1)) if conv.k == 3 else ((0, 0),)):
                wpk = self._pack(conv, 4 + 2 * py + px if conv.k == 3 else 2)
                self._call("tic_conv_igemm_dgrad_s2", dy.data_ptr(), wpk.data_ptr(), dx.data_ptr(), B, H, W, conv.cin, conv.cout, conv.k, py, px, 1 if acc else 0)
            return dx
        if conv.k == 1 and conv.stride == 1:   # the activation gradient IS the GEMM output; an accumulation target rides in its epilogue
            return self._gemm_nt(dy, self._pack(conv, 1), M, conv.kp, conv.cout, add_into=dx_accumulate_into)
        dcol = self._gemm_nt(dy, self._pack(conv, 1), M, conv.kp, conv.cout)
        dx = dx_accumulate_into if dx_accumulate_into is not None else torch.empty(B * H * W, conv.cin, dtype=torch.bfloat16, device=dy.device)
        self._call("tic_col2im_bf16", dcol.data_ptr(), dx.data_ptr(), B, H, W, conv.cin, conv.k, conv.k, conv.stride, conv.pad,
                   1 if dx_accumulate_into is not None else 0)
        return dx

    def _bn_bwd(self, bn: _BN, dy, y_relu, x, mean, rstd, M, dskip=None, skip_accumulate=False, relu_from_x=False):
        dx = torch.empty_like(x)
        scratch = self._bn_scratch(x.device)
        if relu_from_x:   # y = relu(bn(x)), no residual add: the mask is recomputed from x, y is not read
            self._call("tic_batchnorm_bwd_relu", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                       scratch.data_ptr(), scratch.numel(), dx.data_ptr(), self._grad_buf(bn.weight).data_ptr(), self._grad_buf(bn.bias).data_ptr(), M, bn.c)
            return dx
        self._call("tic_batchnorm_bwd", dy.data_ptr(), None if y_relu is None else y_relu.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                   bn.weight.data_ptr(), scratch.data_ptr(), scratch.numel(), dx.data_ptr(), None if dskip is None else dskip.data_ptr(),
                   1 if skip_accumulate else 0,
                   self._grad_buf(bn.weight).data_ptr(), self._grad_buf(bn.bias).data_ptr(), M, bn.c)
        return dx

    # ---- data parallelism: gradient buckets = contiguous ranges of the flat gradient buffer, in backward completion order ------
    _STAGES = ("fc", "layer4", "layer3", "layer2", "layer1", "stem")

    def _stage_modules(self, stage: str):
        return {"fc": [self.fc], "stem": [self.conv1, self.bn1]}.get(stage) or [getattr(self, stage)]

    def buckets(self):
        """[(name, start, end)] element ranges of the flat fp32 gradient buffer (parameters() order: stem, layer1..4, fc), listed in
        the order the backward completes them: fc, layer4 .. layer1, stem.  ResNet-50 (C=120): 0.25 / 15.0 / 7.1 / 1.2 / 0.2 / 0.01 M
        elements -- the reference has no distributed code; ranks exchange exactly these six ranges per step."""
        sizes = {}
        for st in self._STAGES:
            sizes[st] = sum(p.numel() for m in self._stage_modules(st) for p in m.parameters() if p.requires_grad)
        order = ("stem", "layer1", "layer2", "layer3", "layer4", "fc")   # parameters() order
        start, off = {}, 0
        for st in order:
            start[st] = off
            off += sizes[st]
        return [(st, start[st], start[st] + sizes[st]) for st in self._STAGES]

    def register_bucket_hook(self, hook):
        """hook(name, grad_slice) -- called during backward as each bucket's gradients are complete (enqueued on the stream)"""
        self._bucket_hook = hook

    def _fire(self, stage: str):
        hook = self.__dict__.get("_bucket_hook")
        if hook is not None:
            a, b = self.__dict__["_bucket_ranges"][stage]
            hook(stage, self.__dict__["_flat_grad"][a:b])

    def broadcast_state(self, src: int = 0, group=None):
        """start-up broadcast: parameters AND BatchNorm buffers (running statistics, counters), coalesced per dtype"""
        import torch.distributed as dist
        tensors = [p.data for p in self.parameters()] + [b for b in self.buffers()]
        for dt in (torch.float32, torch.int64):
            ts = [t for t in tensors if t.dtype == dt]
            if not ts:
                continue
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        # (copy_ bumped the weights' version counters: _refresh_packs re-packs the bf16 operands on the next forward)

    # ---- forward / backward -----------------------------------------------------------------------------------------
    def _blocks(self) -> List[_Block]:
        return [b for i in range(1, 5) for b in getattr(self, f"layer{i}")]

    def _block_forward(self, blk: _Block, h, B: int, Hc: int, Wc: int, train: bool):
        """one residual block (TIC/ResNet/model.py:47-63 BasicBlock, :95-115 Bottleneck) on h [B*Hc*Wc, Cin] bf16 NHWC ->
        (out, Ho, Wo, record for _block_backward)"""
        rec = {"in": h, "H": Hc, "W": Wc}
        convs = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)] + ([(blk.conv3, blk.bn3)] if blk.kind == "bottleneck" else [])
        if blk.downsample is not None:
            cd, cold, Hd, Wd = self._conv_fwd(blk.downsample[0], h, B, Hc, Wc)
            identity, md, rd = self._bn_fwd(blk.downsample[1], cd, B * Hd * Wd, None, False, train)
            rec["ds"] = (cold, cd, md, rd)
        else:
            identity = h
        t, Ht, Wt = h, Hc, Wc
        steps = []
        for i, (cv, bn) in enumerate(convs):
            last = i == len(convs) - 1
            c, col, Ho, Wo = self._conv_fwd(cv, t, B, Ht, Wt)
            y, m, r = self._bn_fwd(bn, c, B * Ho * Wo, identity if last else None, True, train)
            steps.append((col, c, y, m, r, Ht, Wt))
            t, Ht, Wt = y, Ho, Wo
        rec["steps"] = steps
        return t, Ht, Wt, rec

    def _block_backward(self, blk: _Block, rec, dh, B: int):
        """gradient of one block: dh = d(loss)/d(block output) [M, Cout] bf16 -> d(loss)/d(block input); parameter gradients accumulate"""
        convs = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)] + ([(blk.conv3, blk.bn3)] if blk.kind == "bottleneck" else [])
        steps = rec["steps"]
        dident = torch.empty_like(dh)   # gradient of the identity / downsample branch = masked block-output gradient
        d = dh
        folded = False
        for i in reversed(range(len(convs))):
            cv, bn = convs[i]
            col, c, y, m, r, Hi, Wi = steps[i]
            last = i == len(convs) - 1
            if last:
                d = self._bn_bwd(bn, d, y, c, m, r, c.shape[0], dskip=dident)
            else:
                d = self._bn_bwd(bn, d, None, c, m, r, c.shape[0], relu_from_x=True)
            # the block's first conv: when it is a 1x1 / stride-1 GEMM and the identity branch has no downsample, its input
            # gradient is added onto the identity-branch gradient by the GEMM epilogue (no separate add pass)
            fold = i == 0 and blk.downsample is None and cv.k == 1 and cv.stride == 1
            d = self._conv_bwd(cv, d, col, B, Hi, Wi, need_dx=True, dx_accumulate_into=dident if fold else None)
            folded = fold
        if blk.downsample is not None:
            cold, cd, md, rd = rec["ds"]
            dd = self._bn_bwd(blk.downsample[1], dident, None, cd, md, rd, cd.shape[0])
            self._conv_bwd(blk.downsample[0], dd, cold, B, rec["H"], rec["W"], need_dx=True, dx_accumulate_into=d)
        elif not folded:
            self._call("tic_add_bf16", d.data_ptr(), dident.data_ptr(), d.numel())
        return d

    def _forward_impl(self, x: torch.Tensor, train: bool, record: bool):
        self.backend.check_tensor(x)
        B, C, H, W = x.shape
        if C != 3:
            raise ValueError(f"expected 3 input channels, got {C}")
        x = x.to(torch.float32).contiguous()
        self._refresh_packs(x.device)
        tape: Dict = {"B": B, "blocks": []}
        if self.conv1.stem and W % 2 == 0:   # 3 channels zero-padded to 4: 8-byte pixels, the stem convolution gathers them itself (no im2col)
            xin = torch.empty(B, H, W, 4, dtype=torch.bfloat16, device=x.device)
            self._call("tic_nchw_to_nhwc_pad_bf16", x.data_ptr(), xin.data_ptr(), B, 3, 4, H, W)
        else:
            raise ValueError("TIC HIP ResNet: the stem needs an even image width")
        c0, col0, H1, W1 = self._conv_fwd(self.conv1, xin, B, H, W)
        # bn1 -> relu -> maxpool in one piece: relu(bn(c0)) (411 MB at B = 256) is never stored; the backward needs c0, the batch
        # statistics and one byte of argmax position per pooled element
        Hp, Wp = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        dev = x.device
        m0, r0 = torch.empty(64, device=dev), torch.empty(64, device=dev)
        h = torch.empty(B * Hp * Wp, 64, dtype=torch.bfloat16, device=dev)
        pidx = torch.empty(B * Hp * Wp, 64, dtype=torch.uint8, device=dev) if record else None
        bn = self.bn1
        self._call("tic_bn_relu_maxpool_fwd", c0.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                   bn.num_batches_tracked.data_ptr(), m0.data_ptr(), r0.data_ptr(), self._bn_scratch(dev).data_ptr(),
                   self._bn_scratch(dev).numel(), h.data_ptr(),
                   None if pidx is None else pidx.data_ptr(), B, H1, W1, 64, _EPS, _MOMENTUM, 1 if train else 0)
        tape["stem"] = (col0, c0, m0, r0, H, W, H1, W1, pidx)
        Hc, Wc = Hp, Wp
        for blk in self._blocks():
            h, Hc, Wc, rec = self._block_forward(blk, h, B, Hc, Wc, train)
            tape["blocks"].append(rec)
        feat = self.fc.weight.shape[1]
        z = torch.empty(B, feat, dtype=torch.bfloat16, device=x.device)
        self._call("tic_avgpool_fwd", h.data_ptr(), z.data_ptr(), B, Hc * Wc, feat)
        logits = torch.empty(B, self.num_classes, dtype=torch.float32, device=x.device)
        self._call("tic_head_fwd", z.data_ptr(), self.fc.weight.data_ptr(), self.fc.bias.data_ptr(), logits.data_ptr(), B, self.num_classes, feat)
        tape["head"] = (z, Hc, Wc, feat)
        return logits, (tape if record else None)

    def _backward_impl(self, dlogits: torch.Tensor, tape):
        try:
            self._backward_body(dlogits, tape)
        finally:   # the slab scratch registered by _begin_backward must not outlive this call (it is a raw pointer in the library)
            self.backend.call("tic_gemm_tn_scratch", None, 0)

    def _backward_body(self, dlogits: torch.Tensor, tape):
        B = tape["B"]
        z, Hc, Wc, feat = tape["head"]
        dev = dlogits.device
        self._begin_backward(dev)
        dz = torch.empty(B, feat, dtype=torch.bfloat16, device=dev)
        self._call("tic_head_bwd", dlogits.data_ptr(), z.data_ptr(), self.fc.weight.data_ptr(), dz.data_ptr(), self._grad_buf(self.fc.weight).data_ptr(),
                   self._grad_buf(self.fc.bias).data_ptr(), B, self.num_classes, feat)
        dh = torch.empty(B * Hc * Wc, feat, dtype=torch.bfloat16, device=dev)
        self._call("tic_avgpool_bwd", dz.data_ptr(), dh.data_ptr(), B, Hc * Wc, feat)
        self.__dict__["_bucket_ranges"] = {n: (a, b) for n, a, b in self.buckets()}
        self._fire("fc")
        stage_of = [(f"layer{i}", b) for i in range(1, 5) for b in getattr(self, f"layer{i}")]
        for bi, (blk, rec) in reversed(list(enumerate(zip(self._blocks(), tape["blocks"])))):
            dh = self._block_backward(blk, rec, dh, B)
            if bi == 0 or stage_of[bi - 1][0] != stage_of[bi][0]:   # first block of its stage: the stage's gradients are complete
                self._fold_weight_grads(dev, stage_of[bi][0])
                self._fire(stage_of[bi][0])
        col0, c0, m0, r0, H, W, H1, W1, pidx = tape["stem"]
        da0 = torch.empty_like(c0)   # gradient of the (never stored) relu(bn1(c0)): from the pooled gradient and the argmax positions
        self._call("tic_maxpool3x3s2_bwd_idx", pidx.data_ptr(), dh.data_ptr(), da0.data_ptr(), B, H1, W1, 64)
        dc0 = self._bn_bwd(self.bn1, da0, None, c0, m0, r0, c0.shape[0], relu_from_x=True)
        self._conv_bwd(self.conv1, dc0, col0, B, H, W, need_dx=False)
        self._fold_weight_grads(dev, "stem")
        self._fire("stem")

    def _fold_weight_grads(self, dev, stage: str):
        """grad (OIHW) += dw scratch ([Cout, Kp], tap-major) for every conv of one stage in one launch (a stage = one gradient bucket)"""
        convs = [m for top in self._stage_modules(stage) for m in top.modules() if isinstance(m, _Conv)]
        ptrs = (tuple(self._grad_buf(c.weight).data_ptr() for c in convs), tuple(c.__dict__["_dw_view"].data_ptr() for c in convs), dev)
        st = self.__dict__.setdefault("_fold_state", {}).setdefault(stage, {})
        if st.get("ptrs") != ptrs:
            descs = [struct.pack("<QQiiiiii", c.__dict__["_dw_view"].data_ptr(), self._grad_buf(c.weight).data_ptr(), c.cout, c.cin, c.k, c.k,
                                 3 if c.stem else 0, 0) for c in convs]
            st["n"] = len(descs)
            st["table"] = torch.frombuffer(bytearray(b"".join(descs)), dtype=torch.uint8).to(dev)
            st["ptrs"] = ptrs
        self._call("tic_conv_weight_grad_many", st["table"].data_ptr(), st["n"])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            return _ResNetFunction.apply(self._anchor, x, self)
        return self._forward_impl(x, train=self.training, record=False)[0]


def _resnet(arch: str, pretrained: bool, progress: bool, **kwargs) -> TicResNet:
    # the reference ignores `pretrained` (weights are loaded outside the model definition, model.py:241-245)
    return TicResNet(arch, **kwargs)


def resnet18(pretrained: bool = False, progress: bool = True, num_classes: int = 1000, **kwargs) -> TicResNet:
    return _resnet('resnet18', pretrained, progress, num_classes=num_classes, **kwargs)


def resnet34(pretrained: bool = False, progress: bool = True, num_classes: int = 1000, **kwargs) -> TicResNet:
    return _resnet('resnet34', pretrained, progress, num_classes=num_classes, **kwargs)


def resnet50(pretrained: bool = False, progress: bool = True, num_classes: int = 1000, **kwargs) -> TicResNet:
    return _resnet('resnet50', pretrained, progress, num_classes=num_classes, **kwargs)


def resnet101(pretrained: bool = False, progress: bool = True, num_classes: int = 1000, **kwargs) -> TicResNet:
    return _resnet('resnet101', pretrained, progress, num_classes=num_classes, **kwargs)


def resnet152(pretrained: bool = False, progress: bool = True, num_classes: int = 1000, **kwargs) -> TicResNet:
    return _resnet('resnet152', pretrained, progress, num_classes=num_classes, **kwargs)
