#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ViT-L/16 224 px bf16 fine-tune step (fwd + CE + bwd + gradient
all-reduce + AdamW) on synthetic 3x224x224 batches, weak scaling over N MI355X (one process per GPU).

  python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (torch.distributed.run child)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.
  roofline      dominant kernel (grouped dW GEMM of one block): algorithmic FLOPs per launch / its average launch duration,
                measured LIVE with HIP events on its launch stream inside the timed steps; plus the whole step
                (img/s x 369.32 GFLOP per image) against the dense bf16 MFMA peak (2.5 PFLOP/s)
  parity        SURVEY 8d: the CPU oracle and the HIP model get the SAME seeded weights and the same B=4 batch in this run;
                max |dlogits|, relative loss difference, top-1 / top-5 agreement
  cpu_baseline  the CPU oracle (oracle/vit_oracle.py, a port) timed on this box's host cores, N = 1 only:
                3 warm-up + 5 timed steps, median; headline config in fp32, `also` = bf16-autocast and BASELINE config 1
  sweep         per-GPU batch {8, 16, 32, 64, 83, 128, 166} (8 / 16: what the reference's ntrain*.py launchers set; 32-128: SURVEY 8d; 83 / 166:
                the 1/4 and 1/2 of the headline batch whose 197 x B token rows are again a whole number of 256-row tiles -- what a
                data-parallel user with a global batch of 664 / 1328 on 8 GPUs should pick), outside the headline timed region
  secondary     N = 1: BASELINE config 2 (ViT-B C=10 batch 256), config 4 (ResNet-50 @224 batch 256), the reference's own ResNet run
                (ResNet-152 @256 batch 80, TIC/ResNet/train.py:213) and forward-only ViT-L at batch 1 / 64 / 256 -- each with img/s, ms/step
                and the fraction of the bound it is measured against (ResNet: max of the MFMA time and BatchNorm's 22 B/element at 8 TB/s)
  dp            N > 1: RCCL rank count actually observed, per-rank ms/step min / max, all-reduce time not hidden by backward
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODELS = {
    "large": dict(name="google/vit-large-patch16-224", hidden=1024, layers=24, heads=16, mlp=4096),
    "base": dict(name="google/vit-base-patch16-224", hidden=768, layers=12, heads=12, mlp=3072),
}
PEAK_BF16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md chip table


def train_flops_per_image(m, C):
    N, D, L = 197, m["hidden"], m["layers"]
    fwd = L * (24 * N * D * D + 4 * N * N * D) + 2 * 196 * 768 * D + 2 * D * C
    return 3.0 * fwd


def host_cores():
    """threads we may actually use: cgroup quota / affinity, not the machine's core count"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:   # noqa: BLE001
        pass
    return max(1, min(n, 16))   # a 1-GPU box owns a 16-core share


# ---- CPU side: oracle timing + the in-run parity reference -----------------------------------------------------------------------
def _oracle_spec(model_key, C):
    from oracle import vit_oracle as vo
    m = MODELS[model_key]
    return vo.ViTSpec(hidden=m["hidden"], layers=m["layers"], heads=m["heads"], mlp=m["mlp"], num_labels=C)


def _time_oracle(model_key, C, params, x, y, autocast, warm=3, timed=5, budget_s=40.0):
    """fwd + bwd + AdamW of the oracle on the host cores; median step time of `timed` steps after `warm` warm-ups (both cut
    short, never below 1 + 2, if the budget runs out: the default bench run must finish in minutes)"""
    from oracle import vit_oracle as vo
    spec = _oracle_spec(model_key, C)
    params = {k: v.clone() for k, v in params.items()}
    mom = {k: torch.zeros_like(v) for k, v in params.items()}
    var = {k: torch.zeros_like(v) for k, v in params.items()}
    first = {}

    def step(i):
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            logits, loss, grads = vo.loss_and_grads(params, x, y, spec)
        if not first:
            first.update(logits=logits.float(), loss=float(loss))
        for k in params:
            vo.adamw_step(params[k], grads[k].float(), mom[k], var[k], i, lr=1e-5, wd=0.01)

    t_start = time.perf_counter()
    i = 0
    for _ in range(warm):
        i += 1
        step(i)
        if time.perf_counter() - t_start > 0.4 * budget_s:
            break
    warm_done = i
    ts = []
    for _ in range(timed):
        i += 1
        t0 = time.perf_counter()
        step(i)
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s and len(ts) >= 2:
            break
    return statistics.median(ts), warm_done, len(ts), first


def cpu_legs(model_key, C, params, x4, y4):
    """-> (cpu_baseline object, fp32 oracle logits / loss of the parity batch at the initial weights)"""
    cores = host_cores()
    torch.set_num_threads(cores)
    B = x4.shape[0]
    print(f"[bench] cpu baseline on {cores} threads ...", file=sys.stderr, flush=True)
    med, w, n, first = _time_oracle(model_key, C, params, x4, y4, autocast=False)
    base = dict(value=round(B / med, 3), unit="images/sec", cores=cores, kind="port",
                sample=f"median of {n} timed steps after {w} warm-up: fwd+bwd+AdamW, ViT-{model_key} C={C} B={B} fp32, torch CPU ops via oracle/vit_oracle.py")
    also = []
    med, w, n, _ = _time_oracle(model_key, C, params, x4, y4, autocast=True, budget_s=30.0)
    also.append(dict(value=round(B / med, 3), unit="images/sec", cores=cores, kind="port",
                     sample=f"same, under torch.autocast('cpu', bfloat16); median of {n} after {w} warm-up"))
    # BASELINE config 1: ViT-Base/16, 10 classes, batch 4 on the CPU
    from oracle import vit_oracle as vo
    pb = vo.init_params(_oracle_spec("base", 10), seed=0)
    g = torch.Generator().manual_seed(1234)
    xb = torch.randn(4, 3, 224, 224, generator=g)
    yb = torch.randint(0, 10, (4,), generator=g)
    for ac in (False, True):
        med, w, n, _ = _time_oracle("base", 10, pb, xb, yb, autocast=ac, budget_s=15.0)
        also.append(dict(value=round(4 / med, 3), unit="images/sec", cores=cores, kind="port",
                         sample=f"BASELINE config 1: ViT-base C=10 B=4 {'bf16-autocast' if ac else 'fp32'}; median of {n} after {w} warm-up"))
    base["also"] = also
    return base, first


def parity_block(hip_logits, hip_loss, ref_logits, ref_loss, y):
    """same weights, same batch: HIP (bf16 GEMMs) vs the fp32 CPU oracle"""
    d = (hip_logits - ref_logits).abs().max().item()
    top1 = (hip_logits.argmax(-1) == ref_logits.argmax(-1)).float().mean().item()
    k = min(5, ref_logits.shape[1])
    t5h, t5r = hip_logits.topk(k, -1).indices, ref_logits.topk(k, -1).indices
    top5 = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(t5h, t5r)) / float(k * len(t5h))
    # rows whose fp32 top-1 margin is larger than the logit difference: there a disagreement would be an error, not rounding
    srt = ref_logits.sort(-1, descending=True).values
    decidable = (srt[:, 0] - srt[:, 1]) > 2 * d
    agree_dec = bool((hip_logits.argmax(-1) == ref_logits.argmax(-1))[decidable].all()) if decidable.any() else True
    return dict(batch=int(ref_logits.shape[0]), max_abs_dlogits=round(d, 6), logit_scale=round(ref_logits.abs().max().item(), 4),
                loss_hip=round(hip_loss, 6), loss_cpu=round(ref_loss, 6), loss_rel=round(abs(hip_loss - ref_loss) / max(abs(ref_loss), 1e-12), 6),
                top1_match=round(top1, 4), top5_match=round(top5, 4), top1_decidable_rows=int(decidable.sum()), top1_match_on_decidable=agree_dec,
                tolerance="logits atol 2e-2 + rtol 2e-2, loss rtol 1e-2 (tests/test_gpu_model.py)")


# ---- secondary configurations (N = 1, outside the headline timed region) ------------------------------------------------------------
def _timeit(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def resnet_work_per_image(model, image):
    """-> (train FLOPs, BatchNorm HBM bytes) per image.  FLOPs = 3 x forward conv + fc (8.174 GFLOP forward for ResNet-50
    @224).  Bytes = 22 B per BatchNorm-ed activation element over its five passes (statistics 2, apply 2 + 2, backward reduce 4 + 2 .. 6,
    backward apply 6 + 2 .. 8; DESIGN.md 4: the measured average), the term that bounds the ResNet step"""
    def conv(c, h):
        ho = (h + 2 * c.pad - c.k) // c.stride + 1
        return 2.0 * ho * ho * c.cout * c.cin * c.k * c.k, ho, ho * ho * c.cout
    fl, h, el = conv(model.conv1, image)
    h = (h + 2 - 3) // 2 + 1
    for blk in model._blocks():
        f, hm, e = conv(blk.conv1, h)
        fl, el = fl + f, el + e
        f, ho, e = conv(blk.conv2, hm)
        fl, el = fl + f, el + e
        if blk.kind != "basic":
            f, ho, e = conv(blk.conv3, ho)
            fl, el = fl + f, el + e
        if blk.downsample is not None:
            f, _, e = conv(blk.downsample[0], h)
            fl, el = fl + f, el + e
        h = ho
    fl += 2.0 * model.fc.weight.shape[1] * model.fc.weight.shape[0]
    return 3.0 * fl, 22.0 * el


def secondary_legs(dev, vit_l, x_l, steps=8, warmup=3):
    """BASELINE configs 2 and 4, the configuration the reference's ResNet harness trains (TIC/ResNet/train.py:210-255: ResNet-152 @256 px,
    batch 80) and the forward-only path (validate_step / serve), each timed on this GPU in this run.  vit_l: the headline model (its
    forward-only numbers come first; it is released before the other models are built)."""
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    from touhouimageclassification_amd.ResNet import model as rm
    out = []
    fwd_fl = train_flops_per_image(MODELS["large"], vit_l.num_labels) / 3.0
    vit_l.eval()
    for b in (1, 64, 256):
        xb = x_l[:b].contiguous()
        with torch.no_grad():
            dt = _timeit(lambda: vit_l(xb), 20 if b < 256 else steps, warmup)
        out.append(dict(config=f"ViT-L/16 C={vit_l.num_labels} forward only (validate_step / serve), batch {b}", value=round(b / dt, 1), unit="images/sec",
                        ms_per_step=round(1e3 * dt, 3), bound="mfma", frac=round(b / dt * fwd_fl / 1e12 / PEAK_BF16_TFLOPS, 4)))
    return out


def secondary_train_legs(dev, steps=8, warmup=3):
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    from touhouimageclassification_amd.ResNet import model as rm
    out = []
    g = torch.Generator().manual_seed(77)
    # BASELINE config 2: ViT-Base/16, global batch 256 on one GPU
    mb = MODELS["base"]
    torch.manual_seed(0)
    vb = ViT(10, pretrained=False, model_name=mb["name"]).to(dev)
    ob = FusedAdamW(vb, lr=1e-5, weight_decay=0.01)
    xb = torch.randn(256, 3, 224, 224, generator=g).to(dev)
    yb = torch.randint(0, 10, (256,), generator=g).to(dev)
    dt = _timeit(lambda: fused_train_step(vb, ob, xb, yb, None), steps, warmup)
    out.append(dict(config="BASELINE config 2: ViT-B/16 C=10 fine-tune step, batch 256, 1 GPU", value=round(256 / dt, 1), unit="images/sec",
                    ms_per_step=round(1e3 * dt, 3), bound="mfma", frac=round(256 / dt * train_flops_per_image(mb, 10) / 1e12 / PEAK_BF16_TFLOPS, 4)))
    del vb, ob, xb, yb
    torch.cuda.empty_cache()
    # BASELINE config 4 (ResNet-50 @224, batch 256) and the reference's own ResNet configuration (ResNet-152 @256, batch 80), SGD lr 5e-2
    for arch, image, batch, what in (("resnet50", 224, 256, "BASELINE config 4: ResNet-50 224px"), ("resnet152", 256, 80, "TIC/ResNet/train.py:210-255: ResNet-152 256px")):
        torch.manual_seed(0)
        m = getattr(rm, arch)(num_classes=120).to(dev).train()
        opt = torch.optim.SGD(m.parameters(), lr=5e-2)
        xr = torch.randn(batch, 3, image, image, generator=g).to(dev)
        yr = torch.randint(0, 120, (batch,), generator=g).to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(m(xr), yr).backward()
            opt.step()
        dt = _timeit(step, steps, warmup)
        fl, by = resnet_work_per_image(m, image)
        t_mfma, t_hbm = batch * fl / (PEAK_BF16_TFLOPS * 1e12), batch * by / 8e12
        out.append(dict(config=f"{what} C=120 train step (fwd+CE+bwd+SGD), batch {batch}, 1 GPU", value=round(batch / dt, 1), unit="images/sec",
                        ms_per_step=round(1e3 * dt, 3), bound="hbm" if t_hbm > t_mfma else "mfma",
                        frac=round(max(t_mfma, t_hbm) / dt, 4), mfma_frac=round(t_mfma / dt, 4),
                        bound_ms=dict(mfma=round(1e3 * t_mfma, 3), hbm_batchnorm_22B_per_element=round(1e3 * t_hbm, 3)),
                        flops_per_image=fl, batchnorm_bytes_per_image=by))
        del m, opt, xr, yr
        torch.cuda.empty_cache()
    return out


# ---- launching N ranks ourselves ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (a torch.distributed.run child) BEFORE this
    process makes any GPU call, relay their output, exit with their code.  Nothing here touches HIP."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    print(f"[bench] starting {n} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    rc = subprocess.run(cmd, env=env).returncode
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=332,
                    help="images per GPU per step (weak scaling). 332 x 197 tokens = 256 row tiles of 256: with 4 / 12 / 16 column tiles "
                         "every big GEMM launches a whole multiple of the 256 CUs; the `sweep` list reports 8 ... 128")
    ap.add_argument("--model", default="large", choices=list(MODELS))
    ap.add_argument("--classes", type=int, default=120)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (ViT-B B=256, ResNet-50 / -152, forward-only ViT-L)")
    ap.add_argument("--sweep", default="8,16,32,64,83,128,166", help="per-GPU batch sizes reported beside the headline")
    ap.add_argument("--autograd", action="store_true", help="drive the step through torch autograd + F.cross_entropy (plugin surface) instead of the fused step")
    ap.add_argument("--aug", action="store_true", help="BASELINE config 3: include the on-GPU augmentation (uint8 256x256 thumbnails -> crop/flip/jitter/gray/erase/normalise) and MixUp/CutMix (soft labels) in every timed step")
    ap.add_argument("--bf16-buckets", action="store_true", help="N > 1: gradient buckets cross the links as bf16 (BucketedGradSync(compress='bf16')); default fp32")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="launch rehearsal on a 1-GPU box: all N ranks share device 0 and the collectives go over gloo (RCCL refuses a "
                         "duplicate device). Exercises the self-launch and the DP code path; the number it prints is not a result")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the TIC hot path has no CPU fallback")
    if not args.rehearse_one_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPUs are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())   # the rank count the collective library really summed over
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: all_reduce of ones over {world} ranks returned {rccl_ranks}")

    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.dist import BucketedGradSync
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    from touhouimageclassification_amd._lib import call as _tic_call

    m = MODELS[args.model]
    C, B = args.classes, args.batch
    torch.manual_seed(0)
    model = ViT(C, pretrained=False, model_name=m["name"])
    model.reset_parameters(seed=0)
    init_params = {k: v.detach().clone() for k, v in model.state_dict().items()} if rank == 0 else None   # CPU copy for the oracle
    model.to(dev)
    sync = BucketedGradSync(model, compress="bf16" if args.bf16_buckets else None)
    sync.broadcast_parameters()
    opt = FusedAdamW(model, lr=1e-5, weight_decay=0.01)   # ntrain.py:256-257
    g = torch.Generator().manual_seed(1234 + rank)
    x_cpu = torch.randn(B, 3, 224, 224, generator=g)
    y_cpu = torch.randint(0, C, (B,), generator=g)
    x, y = x_cpu.to(dev), y_cpu.to(dev)
    if args.aug:
        from touhouimageclassification_amd.aug import CutMixOrMixUp, GpuAugment
        raw = torch.randint(0, 256, (B, 256, 256, 3), dtype=torch.uint8, generator=g).to(dev)   # dataset thumbnails are 256x256
        augment, mixer = GpuAugment("full", 224, seed=1 + rank), CutMixOrMixUp(C, seed=2 + rank)

    # ---- in-run parity: HIP forward + CE of the first 4 images at the initial weights (the oracle runs on the same later) ----
    hip4 = None
    if rank == 0:
        from touhouimageclassification_amd import ops
        with torch.no_grad():
            lg = model(x[:4]).logits
            ls, _ = ops.softmax_xent(lg, y[:4], want_grad=False)
        hip4 = (lg.float().cpu(), float(ls))

    def make_step(xb, yb):
        def step():
            if args.autograd:
                opt.zero_grad()
                loss = torch.nn.functional.cross_entropy(model(xb).logits, yb) * sync.grad_scale
                loss.backward()
                sync.wait()
                opt.step()
                return loss
            if args.aug:
                xa, ya = mixer(augment(raw[:xb.shape[0]]), yb)   # parameter sampling on the host, pixels on the GPU
                return fused_train_step(model, opt, xa, ya, sync)[0]
            return fused_train_step(model, opt, xb, yb, sync)[0]
        return step

    def timed(step, steps, warmup):
        """contract: W untimed steps, barrier + synchronize, EXACTLY K steps, synchronize + barrier; MAX over ranks"""
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt, mine, -mine], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt[0].item(), tt[1].item(), -tt[2].item(), loss

    step = make_step(x, y)
    if rank == 0:
        print(f"[bench] ViT-{args.model} B={B}/GPU world={world}: warm-up", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    if rank == 0:
        _tic_call("tic_kernel_timer_enable", 1)   # HIP events around every grouped-dW launch of the timed steps, on their stream
    dt, slowest, fastest, loss = timed(step, args.steps, 0)
    dw_launches, dw_total_ms = 0, 0.0
    if rank == 0:
        import ctypes as _ct
        n_, ms_ = _ct.c_int(0), _ct.c_float(0.0)
        _tic_call("tic_kernel_timer_read", _ct.byref(n_), _ct.byref(ms_))
        _tic_call("tic_kernel_timer_enable", 0)
        dw_launches, dw_total_ms = n_.value, ms_.value
    loss_v = float(loss)
    if rank == 0:
        print(f"[bench] {args.steps} steps in {dt:.3f}s = {world * B * args.steps / dt:.1f} img/s, loss {loss_v:.4f}", file=sys.stderr, flush=True)

    fl = train_flops_per_image(m, C)
    # ---- N > 1: how much of the all-reduce is NOT hidden behind backward (same steps with the bucket hook off) ------------------
    dp = None
    if world > 1:
        hook = model._bucket_hook
        model.register_bucket_hook(None)
        dt_off, _, _, _ = timed(step, max(5, args.steps // 2), 2)
        model.register_bucket_hook(hook)
        sync.broadcast_parameters()   # the unsynchronised steps let the replicas drift: re-align them
        ms_on, ms_off = 1e3 * dt / args.steps, 1e3 * dt_off / max(5, args.steps // 2)
        dp = dict(collective="gloo (one-GPU rehearsal)" if args.rehearse_one_gpu else "RCCL all_reduce(SUM) per bucket, side stream",
                  ranks_observed=rccl_ranks, ms_per_step_slowest_rank=round(1e3 * slowest / args.steps, 3),
                  ms_per_step_fastest_rank=round(1e3 * fastest / args.steps, 3), ms_per_step_no_allreduce=round(ms_off, 3),
                  allreduce_exposed_ms=round(ms_on - ms_off, 3), grad_bytes_per_step=(2 if args.bf16_buckets else 4) * model._engine.lay.n_params,
                  buckets=len(model._engine.buckets()))

    # ---- realistic per-GPU batches (outside the headline timed region) --------------------------------------------------------
    sweep = []
    if not args.no_sweep:
        for bs in [int(s) for s in args.sweep.split(",") if s]:
            if bs >= B:
                continue
            st = make_step(x[:bs].contiguous(), y[:bs].contiguous())
            k = max(5, min(args.steps, 10))
            d, _, _, _ = timed(st, k, 3)
            ips = world * bs * k / d
            sweep.append(dict(per_gpu_batch=bs, value=round(ips, 1), ms_per_step=round(1e3 * d / k, 3), steps=k,
                              step_frac=round(ips * fl / world / 1e12 / PEAK_BF16_TFLOPS, 4)))
            if rank == 0:
                print(f"[bench] sweep B={bs}: {ips:.1f} img/s", file=sys.stderr, flush=True)

    # dominant kernel of the step (22 % of it): the grouped weight-gradient GEMM of one transformer block,
    # dW_g[N,K] += dY_g^T . X_g for the 4 Linear layers, ONE launch per block.  Its duration is measured LIVE, inside the timed
    # region: the library records HIP events on the launch stream around each of its launches (tic_kernel_timer_*), so
    # ms_per_launch is the average over the L x steps launches the timed steps made (the same launches a
    # `rocprofv3 --kernel-trace --stats -- python3 bench.py` summary averages).  Algorithmic FLOPs per launch = 2 M (4 D^2 + 2 D F).
    if rank == 0:
        M, D, F = B * 197, m["hidden"], m["mlp"]
        shapes = [(D, F), (F, D), (D, D), (3 * D, D)]
        kflops = 2.0 * M * sum(n * k for n, k in shapes)
        ms = dw_total_ms / max(dw_launches, 1)
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section); only valid for
        # the kernel source it was measured on: the record carries the hash of csrc/gemm_tn256.h and goes stale (null) with any edit
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if os.path.exists(tf):
            rec = json.load(open(tf))
            src = os.path.join(ROOT, "touhouimageclassification_amd", "csrc", "gemm_tn256.h")
            if rec.get("M") == M and rec.get("hidden") == D and rec.get("src_sha256") == hashlib.sha256(open(src, "rb").read()).hexdigest():
                traffic = rec["hbm_bytes_per_launch"]
        tfl = round(kflops / (ms * 1e-3) / 1e12, 1) if dw_launches else 0.0
        ips = world * B * args.steps / dt
        achieved = ips * fl / world / 1e12
        out = {
            "metric": f"images/sec ViT-{'L' if args.model == 'large' else 'B'}/16 224px bf16 fine-tune",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"ViT-{args.model}/16 224px C={C} fine-tune step (fwd+CE+bwd+AdamW{'+grad all-reduce' if world > 1 else ''})",
                       "per_gpu_batch": B, "global_batch": B * world, "tokens": 197, "parallelism": f"dp{world}",
                       "optimizer": "AdamW lr=1e-5 wd=0.01 (fused, fp32 master weights)", "step_driver": "autograd" if args.autograd else "fused", "gpu_augmentation": bool(args.aug)},
            "loss": round(loss_v, 5),
            # dominant kernel (contract): algorithmic FLOPs per launch / its average launch duration (HIP events, live)
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tfl / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "kernel": "gemm_tn256_streamk_kernel (grouped dW of one block)", "M": M, "flops_per_launch": kflops, "ms_per_launch": round(ms, 4),
                         "launches_timed": dw_launches,
                         # whole step: img/s x 369.32 GFLOP/img / n_gpus vs the same peak
                         "step_achieved": round(achieved, 1), "step_frac": round(achieved / PEAK_BF16_TFLOPS, 4), "flops_per_image": fl},
        }
        if args.rehearse_one_gpu:
            out["rehearsal"] = "all ranks on ONE GPU over gloo: launch-path check only, not a result"
        if dp:
            out["dp"] = dp
        if sweep:
            out["sweep"] = sweep
        if world == 1 and not args.no_secondary and not args.aug and not args.autograd:
            print("[bench] secondary configurations ...", file=sys.stderr, flush=True)
            sec = secondary_legs(dev, model, x) if args.model == "large" else []
            model.train()
            model.to("cpu")          # release the headline replica's HBM (weights, gradients, workspaces) before the other models are built
            del opt
            torch.cuda.empty_cache()
            out["secondary"] = sec + secondary_train_legs(dev)
        if not args.no_cpu_baseline:
            if world == 1:
                base, first = cpu_legs(args.model, C, init_params, x_cpu[:4], y_cpu[:4])
                out["cpu_baseline"] = base
                ref_logits, ref_loss = first["logits"], first["loss"]
            else:   # N > 1: no CPU timing (contract), but the parity reference still runs (one oracle forward)
                from oracle import vit_oracle as vo
                torch.set_num_threads(max(1, host_cores() // world))
                spec = _oracle_spec(args.model, C)
                with torch.no_grad():
                    ref_logits = vo.forward(init_params, x_cpu[:4], spec)
                    ref_loss = float(vo.cross_entropy(ref_logits, y_cpu[:4]))
            out["parity"] = parity_block(hip4[0], hip4[1], ref_logits, ref_loss, y_cpu[:4])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
