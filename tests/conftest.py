import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sd_video_gen_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _fresh_env_knobs():
    """the library caches its $SVG_* knobs (svg_env_refresh): a test starts from the environment as it is now, and whatever a
    test changed through monkeypatch is dropped again before the next one looks"""
    from sd_video_gen_amd import _lib
    if os.path.exists(_lib.LIB_PATH):
        _lib.env_refresh()
    yield
    if os.path.exists(_lib.LIB_PATH):
        _lib.env_refresh()


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ---- measured parity margins -----------------------------------------------------------------------------------------
# Network-level tests report what they measured through `margin(name, err, tol)`: the value is printed (pytest -s / the
# captured-output section of a failure), asserted against its tolerance, and the whole table is written to
# gpurun_out/parity_margins.json at the end of the session (copied into profiles/ per round).
_MARGINS = []


def sd_tol(fp16, bf16):
    """tolerance of a check that runs the SD networks in SDUtils' DEFAULT storage type: fp16 (the reference's autocast arithmetic) unless
    $SVG_SD_DTYPE says bf16.  Both values are <= 3x what was measured on MI355X in that mode."""
    d = os.environ.get("SVG_SD_DTYPE", "fp16").lower()
    return bf16 if d in ("bf16", "bfloat16") else fp16


def margin(name, err, tol, unit="rel-L2"):
    err = float(err)
    _MARGINS.append({"name": name, "measured": err, "tolerance": float(tol), "unit": unit})
    print("[parity] %-62s %s %.3e   (tolerance %.1e, margin x%.1f)" % (name, unit, err, tol, tol / max(err, 1e-30)))
    assert err < tol, "%s: %s %.3e >= tolerance %.1e" % (name, unit, err, tol)
    return err


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump(_MARGINS, f, indent=1)
    except OSError:
        pass
