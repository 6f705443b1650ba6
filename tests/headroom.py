"""Tolerance-headroom log of the parity tests (test infrastructure).

Every tolerance assertion of the `-m gpu` suite records `observed / allowed` -- how much of its band the HIP path used on
the box the suite ran on -- as one JSON line.  The log of a full run is committed under profiles/ and
tests/test_headroom_log.py (CPU) fails when any band was more than half used: a tolerance that passes at 0.9 on one box
is a failure waiting for the next box (one 15 % band at ratio 1.12 hid 86 tests).

  le(name, observed, allowed)          observed <= allowed
  cos_ge(name, observed, bound)        a cosine: the band is 1 - bound
  close(name, got, ref, atol, rtol)    elementwise |got - ref| <= atol + rtol |ref|   (torch.testing.assert_close's rule)
  install()                            wraps torch.testing.assert_close itself, so existing call sites are logged unchanged

The file is `$TIC_HEADROOM_LOG` (default gpurun_out/headroom.jsonl on a GPU box; nothing is written on the CPU).
"""
import inspect
import json
import os

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_current_test = [""]
_orig_assert_close = torch.testing.assert_close


def _path():
    p = os.environ.get("TIC_HEADROOM_LOG")
    if p:
        return p
    if torch.cuda.is_available():
        return os.path.join(_ROOT, "gpurun_out", "headroom.jsonl")
    return None


def _site():
    for fr in inspect.stack()[2:]:
        fn = os.path.abspath(fr.filename)
        if fn.startswith(os.path.join(_ROOT, "tests")) and not fn.endswith("headroom.py"):
            return f"{os.path.relpath(fn, _ROOT)}:{fr.lineno}"
    return "?"


def record(name, observed, allowed, kind="le"):
    ratio = 0.0 if observed == 0 else (float("inf") if allowed == 0 else observed / allowed)
    p = _path()
    if p:
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "a") as f:
            f.write(json.dumps({"test": _current_test[0], "site": _site(), "name": name, "kind": kind, "observed": observed,
                                "allowed": allowed, "ratio": None if ratio == float("inf") else round(ratio, 5)}) + "\n")
    return ratio


def le(name, observed, allowed, ctx=None):
    observed, allowed = float(observed), float(allowed)
    record(name, observed, allowed)
    assert observed <= allowed, (name, observed, allowed, ctx)


def cos_ge(name, observed, bound, ctx=None):
    observed, bound = float(observed), float(bound)
    record(name, 1.0 - observed, 1.0 - bound, "cos")
    assert observed >= bound, (name, observed, bound, ctx)


def close_ratio(got, ref, atol, rtol):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    d = (got - ref).abs()
    band = atol + rtol * ref.abs()
    if d.numel() == 0:
        return 0.0
    exact = band == 0          # elements that must match exactly
    if bool((d[exact] > 0).any()):
        return float("inf")
    rest = ~exact
    return float((d[rest] / band[rest]).max()) if bool(rest.any()) else 0.0


def close(name, got, ref, atol, rtol, ctx=None):
    r = close_ratio(got, ref, atol, rtol)
    record(name, r, 1.0, "close")
    assert r <= 1.0, (name, r, ctx)


def _logged_assert_close(actual, expected, *args, **kw):
    try:
        rtol, atol = kw.get("rtol"), kw.get("atol")
        a, e = torch.as_tensor(actual), torch.as_tensor(expected)
        if rtol is None and atol is None:   # torch's defaults per dtype
            rtol, atol = {torch.float32: (1.3e-6, 1e-5), torch.bfloat16: (1.6e-2, 1e-5), torch.float16: (1e-3, 1e-5),
                          torch.float64: (1e-7, 1e-7)}.get(a.dtype, (0.0, 0.0))
        if a.shape == e.shape:
            record("assert_close", close_ratio(a, e, atol or 0.0, rtol or 0.0), 1.0, "close")
    except Exception:   # logging must never change a test's verdict
        pass
    return _orig_assert_close(actual, expected, *args, **kw)


def install():
    torch.testing.assert_close = _logged_assert_close


def set_test(nodeid):
    _current_test[0] = nodeid
