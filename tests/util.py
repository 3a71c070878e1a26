"""Shared synthetic inputs for the tests (seeded, no reference code involved)."""

import numpy as np


def synth_frames(n, h, w, seed=0):
    """Smooth gradient + checker + noise, float32 in [0,1], shape [n,h,w,3]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((n, h, w, 3), np.float32)
    for i in range(n):
        base = 0.5 + 0.25 * np.sin(xx * 0.11 + i * 0.3) * np.cos(yy * 0.07 - i * 0.2)
        chk = (((xx // 7).astype(np.int32) + (yy // 5).astype(np.int32) + i) % 2).astype(np.float32) * 0.2
        for c in range(3):
            noise = rng.random((h, w), dtype=np.float32) * 0.15
            out[i, ..., c] = np.clip(base * (0.8 + 0.1 * c) + chk + noise - 0.1, 0.0, 1.0)
    return out


def similarity(tx, ty, theta, scale, cx=0.0, cy=0.0):
    c, s = np.cos(theta) * scale, np.sin(theta) * scale
    m = np.array([[c, -s, tx], [s, c, ty], [0, 0, 1]], np.float64)
    t = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1]], np.float64)
    ti = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], np.float64)
    return t @ m @ ti


def test_matrices(n, w, h, kind, seed=1):
    rng = np.random.default_rng(seed)
    mats = []
    for i in range(n):
        if kind == "identity":
            m = np.eye(3)
        elif kind == "translation":
            m = similarity(rng.uniform(-9, 9), rng.uniform(-7, 7), 0.0, 1.0)
        elif kind == "similarity":
            m = similarity(rng.uniform(-12, 12), rng.uniform(-8, 8), rng.uniform(-0.06, 0.06),
                           rng.uniform(0.95, 1.06), w / 2, h / 2)
        elif kind == "perspective":
            m = similarity(rng.uniform(-6, 6), rng.uniform(-6, 6), rng.uniform(-0.04, 0.04),
                           rng.uniform(0.97, 1.03), w / 2, h / 2)
            m[2, 0] = rng.uniform(-2e-4, 2e-4)
            m[2, 1] = rng.uniform(-2e-4, 2e-4)
        elif kind == "far":
            m = similarity(3.0 * w, -2.0 * h, 0.1, 1.0)
        elif kind == "horizon":
            # the projective denominator of the inverse map changes sign inside the output: W == 0 columns, coordinates
            # beyond the INT clamp and the short saturation of cv::warpPerspective
            m = similarity(rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-0.02, 0.02), 1.0, w / 2, h / 2)
            m[2, 0] = (-1.0 if i % 2 else 1.0) * 2.0 / w
            m[2, 1] = rng.uniform(-1e-3, 1e-3)
        elif kind == "flip":
            m = similarity(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-0.03, 0.03), 1.0, w / 2, h / 2)
            m = m @ np.array([[-1.0, 0, w - 1.0], [0, 1, 0], [0, 0, 1]])
        elif kind == "minify":
            m = similarity(w * 0.45, h * 0.45, rng.uniform(-0.2, 0.2), 0.02 + 0.03 * i)
        elif kind == "magnify":
            m = similarity(rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-0.3, 0.3), 37.0 + 100.0 * i, w / 2, h / 2)
        elif kind == "quarter_turn":
            m = similarity(0.0, 0.0, np.pi / 2 + rng.uniform(-1e-3, 1e-3), 1.0, w / 2, h / 2)
        else:
            raise ValueError(kind)
        mats.append(m)
    return np.stack(mats).astype(np.float64)


def shake_path(n, w, h, kind, seed=3, amp=1.0):
    """Random-walk camera path [n,3,3] for bench.synth_clip(mats=...): translation / similarity / perspective increments
    per frame, in full-resolution px scaled to the frame size (amp 1: up to +-6 x +-4 px, +-0.004 rad, +-0.3 % scale,
    +-2e-6 / px of perspective per frame at 1920x1080)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 3, 3))
    tx = ty = th = 0.0
    sc = 1.0
    px = py = 0.0
    for i in range(n):
        if i:
            tx += rng.uniform(-6, 6) * amp * w / 1920
            ty += rng.uniform(-4, 4) * amp * h / 1080
            if kind != "translation":
                th += rng.uniform(-0.004, 0.004) * amp
                sc *= 1.0 + rng.uniform(-0.003, 0.003) * amp
            if kind == "perspective":
                px += rng.uniform(-2e-6, 2e-6) * amp * 1920 / w
                py += rng.uniform(-2e-6, 2e-6) * amp * 1080 / h
        c, s = np.cos(th) * sc, np.sin(th) * sc
        pre = np.array([[1, 0, -w / 2], [0, 1, -h / 2], [0, 0, 1.0]])
        post = np.array([[1, 0, w / 2 + tx], [0, 1, h / 2 + ty], [0, 0, 1.0]])
        out[i] = post @ np.array([[c, -s, 0], [s, c, 0], [px, py, 1.0]]) @ pre
    return out


HOOKS_PRELUDE = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import native
assert native.load_library().vstab_test_hooks() == 1, "this child needs the test build (lib/libvstab_hooks.so)"
"""


def run_with_hooks(body: str, env=None, timeout=600):
    """Run `body` (Python source) in a child process that loads the TEST build of the library (lib/libvstab_hooks.so, the
    same sources with -DVSTAB_TEST_HOOKS: the fault injectors VSTAB_DEBUG_* exist only there; the shipped libvstab.so reads
    none of them).  `env`: extra environment of the child.  Returns its stdout; a failing child fails the test with its output."""
    import os
    import subprocess
    import sys
    import textwrap
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    lib = root / "comfyui-video-stabilizer_amd" / "lib" / "libvstab_hooks.so"
    if not lib.exists():
        import __graft_entry__ as graft

        graft.build()
    child_env = dict(os.environ)
    child_env.update({"VSTAB_LIB": str(lib)}, **(env or {}))
    proc = subprocess.run([sys.executable, "-c", HOOKS_PRELUDE.format(root=str(root)) + textwrap.dedent(body)], env=child_env,
                          capture_output=True, text=True, timeout=timeout, cwd=str(root))
    assert proc.returncode == 0, proc.stdout[-3000:] + "\n" + proc.stderr[-6000:]
    return proc.stdout
