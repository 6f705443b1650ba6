"""One-layer-deep ResNet parity, shared by the simulator and the GPU tests: single modules of the reference's own
TIC/ResNet/model.py -- `conv3x3` (:6-9), `conv1x1` (:12-14), the 7x7 stem and its bn1 -> relu -> maxpool tail (:148-152), train-mode
BatchNorm2d + ReLU (+ residual add, :99-113), `Bottleneck` (:66-115, with and without the downsample branch of :193-197) and
`BasicBlock` (:17-63) -- were run in float64 by tools/gen_golden.py (tests/golden/resnet_units.npz: inputs, weights, outputs and every
gradient).  The HIP path runs the SAME code the model runs (`TicResNet._conv_fwd / _conv_bwd / _bn_fwd / _bn_bwd /
_block_forward / _block_backward`, so the route selection -- implicit GEMM, im2col, col2im, fused epilogues -- is the product's) on
stand-alone modules that carry the golden weights.  One layer deep nothing amplifies: fixed tolerances, cos >= 0.995 on every gradient.
Inputs and weights are bf16-representable, so the differences are fp32 accumulation order and the bf16 storage of results."""
import os

import numpy as np
import torch

from tests import headroom as hr

COS = 0.995         # single layers (convolutions, BatchNorm on decidable data): observed 1 - cos < 1e-5
# whole blocks: the activations between a block's layers are STORED in bf16, which moves a fraction f of the pre-activations across
# the ReLU kink; each flipped mask element is an O(|dy|) difference, so every gradient behind it sits at relative L2 ~ sqrt(2 f)
# whatever the batch size.  The reference's own blocks, run in float64 with only their stored activations rounded to bf16
# (tools/gen_golden.py prints it), land at relative L2 0.02 .. 0.096, cos 0.9954 .. 0.9998 on these cases; the HIP path reproduces
# those figures to the third digit (0.0564 vs 0.0563 on the bottleneck's dx).  Bands: twice that.
COS_BLOCK = 0.99


def _load(golden_dir):
    raw = np.load(os.path.join(golden_dir, "resnet_units.npz"))
    out = {}
    for k in raw.files:
        if k.endswith(":bf16"):
            out[k[:-5]] = torch.from_numpy(raw[k].copy()).view(torch.bfloat16).float()
        else:
            out[k] = torch.from_numpy(raw[k].copy()) if raw[k].dtype != np.int64 or raw[k].ndim else torch.tensor(int(raw[k]))
    return out


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def _nhwc(t, dev):
    """reference NCHW fp -> [B*H*W, C] bf16 on the device"""
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).to(torch.bfloat16).contiguous().to(dev)


def _nchw(t, B, H, W):
    return t.float().cpu().view(B, H, W, -1).permute(0, 3, 1, 2)


def _check(name, got, ref, rel=1e-2):
    """max |got - ref| within rel x max|ref| (a bf16 result: 2^-9 relative per element) and direction cos >= 0.995"""
    ref = ref.float()
    hr.le(f"{name} max abs / scale", float((got.float() - ref).abs().max()), rel * float(ref.abs().max()) + 1e-6)
    hr.cos_ge(f"{name} cos", _cos(got, ref), COS)


def _check_l2(name, got, ref, rel, cos=None):
    """relative L2 + direction: for gradients behind a ReLU / max-pool whose inputs the test cannot keep away from the kink (inside a
    block, or window ties of the pool): a bf16 path may put a FEW elements on the other side, each a local O(|dy|) difference"""
    ref = ref.float()
    hr.le(f"{name} rel L2", float((got.float() - ref).norm() / ref.norm()), rel)
    hr.cos_ge(f"{name} cos", _cos(got, ref), COS if cos is None else cos)


class _Harness:
    """stand-alone modules wired to a host TicResNet (any arch: only its kernels-calling methods are used)"""

    def __init__(self, backend, dev):
        from touhouimageclassification_amd.ResNet import model as rm
        self.rm, self.dev = rm, dev
        kw = {} if backend is None else {"backend": backend}
        self.host = rm.resnet18(num_classes=10, **kw).to(dev)

    def conv(self, w):
        cout, cin, k, _ = w.shape
        return cout, cin, k

    def prepare(self, mods):
        """bf16 operand packs, zeroed weight-gradient scratch and zeroed .grad for stand-alone modules (what _refresh_packs /
        _begin_backward do for the model's own)"""
        host, dev = self.host, self.dev
        for c in (m for top in mods for m in top.modules() if isinstance(m, self.rm._Conv)):
            packed = c.__dict__["_packed"] = {}
            for t in host._pack_variants(c):
                shape = {0: (c.cout, c.kp), 1: (c.kp, c.cout), 2: (c.cin, c.k * c.k * c.cout), 3: (c.cout, c.kp), 4: (c.cin, c.cout), 5: (c.cin, 2 * c.cout),
                         6: (c.cin, 2 * c.cout), 7: (c.cin, 4 * c.cout)}[t]
                packed[t] = torch.empty(shape, dtype=torch.bfloat16, device=dev)
                host._call("tic_conv_weight_pack", c.weight.data_ptr(), packed[t].data_ptr(), c.cout, c.cin, c.k, c.k, t)
            c.__dict__["_dw_view"] = torch.zeros(c.cout * c.kp, device=dev)
        for top in mods:
            for p in top.parameters():
                p.grad = torch.zeros_like(p)
        slab = torch.empty(self.rm._TN_SLAB_BYTES, dtype=torch.uint8, device=dev)
        host.backend.call("tic_gemm_tn_scratch", slab.data_ptr(), slab.numel())
        self._slab = slab

    def fold(self, mods):
        host = self.host
        for c in (m for top in mods for m in top.modules() if isinstance(m, self.rm._Conv)):
            host._call("tic_conv_weight_grad", c.__dict__["_dw_view"].data_ptr(), c.weight.grad.data_ptr(), c.cout, c.cin, c.k, c.k, 3 if c.stem else 0)
        host.backend.call("tic_gemm_tn_scratch", None, 0)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()


def check_conv_units(backend, dev, golden_dir):
    g = _load(golden_dir)
    hz = _Harness(backend, dev)
    for tag in ("conv3x3_s1", "conv3x3_s2", "conv1x1_s2", "stem7x7"):
        cin, cout, k, stride, pad = (int(v) for v in g[f"{tag}/geom"])
        conv = hz.rm._Conv(cin, cout, k, stride, pad)
        with torch.no_grad():
            conv.weight.copy_(g[f"{tag}/w"])
        conv.to(dev)
        stem = cin == 3
        if stem:   # the host's own conv1 slot: _pack_variants / the no-input-gradient rule key on identity
            hz.host.conv1 = conv
        hz.prepare([conv])
        x = g[f"{tag}/x"]
        B, _, H, W = x.shape
        xin = _nhwc(x, dev).view(B, H, W, cin)
        if stem:   # the model hands the stem its image with the 3 channels zero-padded to 4 (8-byte pixels: TicResNet._forward_impl)
            xin = torch.cat([xin, torch.zeros(B, H, W, 1, dtype=torch.bfloat16, device=dev)], -1).contiguous()
        y, col, Ho, Wo = hz.host._conv_fwd(conv, xin, B, H, W)
        _check(f"{tag} y", _nchw(y, B, Ho, Wo), g[f"{tag}/y"])
        dy = _nhwc(g[f"{tag}/dy"], dev)
        dx = hz.host._conv_bwd(conv, dy, col, B, H, W, need_dx=not stem)
        hz.fold([conv])
        _check(f"{tag} dw", conv.weight.grad.cpu(), g[f"{tag}/dw"], rel=5e-3)   # fp32 accumulation of bf16 products
        if not stem:
            _check(f"{tag} dx", _nchw(dx, B, H, W), g[f"{tag}/dx"])


def check_bn_units(backend, dev, golden_dir):
    g = _load(golden_dir)
    hz = _Harness(backend, dev)
    host = hz.host
    for tag in ("bn_relu", "bn_add_relu"):
        x = g[f"{tag}/x"]
        B, C, H, W = x.shape
        M = B * H * W
        bn = hz.rm._BN(C)
        with torch.no_grad():
            bn.weight.copy_(g[f"{tag}/gamma"])
            bn.bias.copy_(g[f"{tag}/beta"])
        bn.to(dev)
        hz.prepare([bn])
        xin = _nhwc(x, dev)
        ident = _nhwc(g[f"{tag}/ident"], dev) if tag == "bn_add_relu" else None
        y, mean, rstd = host._bn_fwd(bn, xin, M, ident, True, True)
        _check(f"{tag} y", _nchw(y, B, H, W), g[f"{tag}/y"])
        dy = _nhwc(g[f"{tag}/dy"], dev)
        if ident is None:
            dx = host._bn_bwd(bn, dy, None, xin, mean, rstd, M, relu_from_x=True)
        else:
            dskip = torch.empty_like(dy)
            dx = host._bn_bwd(bn, dy, y, xin, mean, rstd, M, dskip=dskip)
            _check(f"{tag} dident", _nchw(dskip, B, H, W), g[f"{tag}/dident"])
        hz.fold([])
        _check(f"{tag} dx", _nchw(dx, B, H, W), g[f"{tag}/dx"], rel=2e-2)
        _check(f"{tag} dgamma", bn.weight.grad.cpu(), g[f"{tag}/dgamma"], rel=5e-3)
        _check(f"{tag} dbeta", bn.bias.grad.cpu(), g[f"{tag}/dbeta"], rel=5e-3)
        hr.close(f"{tag} running_mean", bn.running_mean.cpu(), g[f"{tag}/running_mean"], atol=1e-5, rtol=1e-4)
        hr.close(f"{tag} running_var", bn.running_var.cpu(), g[f"{tag}/running_var"], atol=1e-5, rtol=1e-4)
        assert int(bn.num_batches_tracked) == 1
    # bn1 -> relu -> maxpool, the fused forward and its two-kernel backward
    tag = "stem_tail"
    x = g[f"{tag}/x"]
    B, C, H, W = x.shape
    Hp, Wp = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    bn = hz.rm._BN(C)
    with torch.no_grad():
        bn.weight.copy_(g[f"{tag}/gamma"])
        bn.bias.copy_(g[f"{tag}/beta"])
    bn.to(dev)
    hz.prepare([bn])
    xin = _nhwc(x, dev)
    m0, r0 = torch.empty(C, device=dev), torch.empty(C, device=dev)
    h = torch.empty(B * Hp * Wp, C, dtype=torch.bfloat16, device=dev)
    pidx = torch.empty(B * Hp * Wp, C, dtype=torch.uint8, device=dev)
    scr = host._bn_scratch(dev)
    host._call("tic_bn_relu_maxpool_fwd", xin.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
               bn.num_batches_tracked.data_ptr(), m0.data_ptr(), r0.data_ptr(), scr.data_ptr(), scr.numel(), h.data_ptr(), pidx.data_ptr(), B, H, W, C,
               1e-5, 0.1, 1)
    _check(f"{tag} y", _nchw(h, B, Hp, Wp), g[f"{tag}/y"])
    dy = _nhwc(g[f"{tag}/dy"], dev)
    da = torch.empty_like(xin)
    host._call("tic_maxpool3x3s2_bwd_idx", pidx.data_ptr(), dy.data_ptr(), da.data_ptr(), B, H, W, C)
    dx = host._bn_bwd(bn, da, None, xin, m0, r0, B * H * W, relu_from_x=True)
    hz.fold([])
    _check_l2(f"{tag} dx", _nchw(dx, B, H, W), g[f"{tag}/dx"], 0.12)   # near-ties of a pool window route dy to the neighbour (simulator: 0.057)
    _check_l2(f"{tag} dgamma", bn.weight.grad.cpu(), g[f"{tag}/dgamma"], 3e-2)
    _check_l2(f"{tag} dbeta", bn.bias.grad.cpu(), g[f"{tag}/dbeta"], 3e-2)
    hr.close(f"{tag} running_var", bn.running_var.cpu(), g[f"{tag}/running_var"], atol=1e-5, rtol=1e-4)


def check_block_units(backend, dev, golden_dir):
    g = _load(golden_dir)
    hz = _Harness(backend, dev)
    host = hz.host
    for tag, kind, inplanes, planes, stride, ds in (("bottleneck", "bottleneck", 256, 64, 1, False), ("bottleneck_ds", "bottleneck", 128, 64, 2, True),
                                                    ("basicblock", "basic", 64, 64, 1, False)):
        blk = hz.rm._Block(kind, inplanes, planes, stride, ds)
        sd = {k[len(tag) + 7:]: v for k, v in g.items() if k.startswith(f"{tag}/param/")}
        assert list(sd) == [k for k, _ in blk.named_parameters()], "parameter names / order differ from the reference block's"
        with torch.no_grad():
            for k, p in blk.named_parameters():
                p.copy_(sd[k])
        blk.to(dev)
        hz.prepare([blk])
        x = g[f"{tag}/x"]
        B, Cin, H, W = x.shape
        h, Ho, Wo, rec = host._block_forward(blk, _nhwc(x, dev), B, H, W, True)
        _check(f"{tag} y", _nchw(h, B, Ho, Wo), g[f"{tag}/y"], rel=1.5e-2)
        d = host._block_backward(blk, rec, _nhwc(g[f"{tag}/dy"], dev), B)
        hz.fold([blk])
        _check_l2(f"{tag} dx", _nchw(d, B, H, W), g[f"{tag}/dx"], 0.2, COS_BLOCK)
        for k, p in blk.named_parameters():
            ref = g[f"{tag}/grad/{k}"]
            _check_l2(f"{tag} grad {k}", p.grad.cpu(), ref, 0.2, COS_BLOCK)
        for k, b in blk.named_buffers():
            ref = g[f"{tag}/after/{k}"]
            if "num_batches" in k:
                assert int(b) == int(ref)
            else:
                hr.close(f"{tag} {k}", b.cpu(), ref.float(), atol=5e-3 * float(ref.abs().max()), rtol=1e-2)   # statistics of bf16-stored conv outputs
