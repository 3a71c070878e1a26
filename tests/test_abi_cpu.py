"""CPU tests: libvstab.so loads and exports every symbol include/vstab.h declares; the oracle builds;
oracle known-answer checks that need no GPU; node schema (KA11)."""

import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def header_symbols():
    text = (ROOT / "include" / "vstab.h").read_text()
    return sorted(set(re.findall(r"\b(vstab_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    from vstab_amd import native

    assert native.LIB_PATH.exists(), "libvstab.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(str(native.LIB_PATH))
    declared = header_symbols()
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vstab.h but not exported"
    assert set(native.EXPORTED_SYMBOLS) == set(declared)
    lib.vstab_abi_version.restype = ctypes.c_int
    assert lib.vstab_abi_version() == 1


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md shows a reference maintainer what each C entry point replaces: none may be missing from it."""
    doc = (ROOT / "INTEGRATION.md").read_text()
    missing = [name for name in header_symbols() if name not in doc]
    assert not missing, f"INTEGRATION.md does not mention {missing}"


def header_prototypes():
    """(name, [parameter type strings]) of every `int vstab_*(...)` prototype in include/vstab.h."""
    text = re.sub(r"/\*.*?\*/", " ", (ROOT / "include" / "vstab.h").read_text(), flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = []
    for m in re.finditer(r"\bint\s+(vstab_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        params = [p.strip() for p in m.group(2).replace("\n", " ").split(",")]
        out.append((m.group(1), [] if params == ["void"] else params))
    return out


def test_null_context_is_rejected_before_any_gpu_work(pkg):
    """Every entry point that takes a context refuses a NULL one with a non-zero status and a message -- checked without a
    GPU: the argument check comes before the first HIP call (all other arguments are zeros / NULL too)."""
    from vstab_amd import native

    lib = ctypes.CDLL(str(native.LIB_PATH))
    lib.vstab_last_error.restype = ctypes.c_char_p
    checked = 0
    for name, params in header_prototypes():
        if not params or not params[0].startswith("vstab_ctx*"):
            continue
        argtypes, args = [], []
        for ptxt in params:
            if "*" in ptxt:
                argtypes.append(ctypes.c_void_p); args.append(None)
            elif ptxt.startswith(("double", "const double")):
                argtypes.append(ctypes.c_double); args.append(0.0)
            elif ptxt.startswith(("float", "const float")):
                argtypes.append(ctypes.c_float); args.append(0.0)
            elif ptxt.startswith("size_t"):
                argtypes.append(ctypes.c_size_t); args.append(0)
            else:
                argtypes.append(ctypes.c_int); args.append(0)
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
        if name == "vstab_destroy":   # destroying nothing is a no-op by contract
            assert fn(*args) == 0
            continue
        rc = fn(*args)
        msg = (lib.vstab_last_error() or b"").decode()
        assert rc != 0, f"{name}(NULL ctx, zeros) returned 0"
        assert "NULL" in msg or "null" in msg, f"{name}: message {msg!r} does not name the NULL argument"
        checked += 1
    assert checked >= 20


def test_product_fails_loudly_without_gpu(pkg):
    import torch

    from vstab_amd import native

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.VstabError):
        native.default_context()


def test_product_does_not_import_oracle(pkg):
    """The oracle is a checker only: nothing in the package (except the smoke self-check, which is
    allowed to compare against it) may import, include or load anything under oracle/."""
    pkg_dir = ROOT / "comfyui-video-stabilizer_amd"
    for path in pkg_dir.glob("*.py"):
        if path.name == "selfcheck.py":
            continue
        text = path.read_text()
        assert not re.search(r"^\s*(from|import)\s+[^\n]*oracle", text, re.M), path
        assert "libvstab_oracle" not in text and "oracle/" not in text, path
    for path in (pkg_dir / "csrc").glob("*.hip"):
        assert not re.search(r"#include[^\n]*(oracle|vo_)", path.read_text()), path


def test_node_schema_sockets(pkg):
    """KA11 (scripts/check_node_schema.py:28-64): ordered socket names of both nodes."""
    from vstab_amd import nodes

    s = nodes.VideoStabilizerFlow.define_schema()
    assert s.node_id == "video_stabilizer_flow" and s.display_name == "Video Stabilizer Flow"
    assert [i.id for i in s.inputs] == ["frames", "frame_rate", "framing_mode", "transform_mode", "camera_lock",
                                        "strength", "smooth", "keep_fov", "padding_color"]
    assert [o.id for o in s.outputs] == ["frames_stabilized", "padding_mask", "meta"]
    c = nodes.VideoStabilizerClassic.define_schema()   # classic.py:575-666: same sockets as Flow
    assert c.node_id == "video_stabilizer_classic" and c.display_name == "Video Stabilizer Classic"
    assert [i.id for i in c.inputs] == [i.id for i in s.inputs] and [o.id for o in c.outputs] == [o.id for o in s.outputs]
    s = nodes.VideoStabilizerMotionApply.define_schema()
    assert s.node_id == "video_stabilizer_motion_apply" and s.display_name == "Video Stabilizer Motion Apply"
    assert [i.id for i in s.inputs] == ["frames", "motion_meta", "framing_mode", "interpolation", "padding_color",
                                        "motion_blur", "motion_blur_quality"]
    assert [o.id for o in s.outputs] == ["frames", "padding_mask", "meta"]
    assert nodes.BLUR_QUALITY_SAMPLES == {"Draft": 5, "Standard": 9, "High": 17, "Ultra": 33}
    s = nodes.VideoStabilizerInverse.define_schema()
    assert s.node_id == "video_stabilizer_inverse" and s.is_deprecated
    assert [i.id for i in s.inputs] == ["frames", "meta", "padding_color"]
    assert [o.id for o in s.outputs] == ["frames_restored", "padding_mask", "meta"]


# ---------------------------------------------------------------- oracle known answers (no GPU)
def test_oracle_identity_warp_is_passthrough(oracle):
    """KA1: identity -> output == input, coverage all ones."""
    from tests.util import synth_frames

    f = synth_frames(2, 24, 32)
    eye = np.tile(np.eye(3, dtype=np.float32), (2, 1, 1))
    for interp in ("bilinear", "bicubic"):
        dst, mask, cnt = oracle.warp_clip(f, eye, (32, 24), interp=interp, border=(0.5, 0.5, 0.5))
        assert np.array_equal(dst, f) and mask.max() == 0 and cnt.sum() == 0


def test_oracle_integer_shift_and_tables(oracle):
    from tests.util import synth_frames

    f = synth_frames(1, 24, 32)
    m = np.eye(3, dtype=np.float32)[None].copy()
    m[0, 0, 2], m[0, 1, 2] = 3.0, 2.0
    dst, mask, cnt = oracle.warp_clip(f, m, (32, 24), border=(0.25, 0.5, 0.75))
    assert np.array_equal(dst[0, 2:, 3:], f[0, :-2, :-3])
    assert np.allclose(dst[0, 0, 0], (0.25, 0.5, 0.75)) and mask[0, :2].min() == 1 and mask[0, 2:, 3:].max() == 0
    assert cnt[0] == 24 * 32 - 22 * 29
    lin, cub = oracle.interp_tables()
    assert np.allclose(lin.sum(1), 1) and np.allclose(cub.sum(1), 1, atol=1e-6)
    assert np.allclose(cub[0], (0, 1, 0, 0)) and cub[16, 0] == pytest.approx(-0.09375)
    inv = oracle.invert3x3([[2, 0, 1], [0, 4, -2], [0, 0, 1]])
    assert np.allclose(inv @ np.array([[2, 0, 1], [0, 4, -2], [0, 0, 1.0]]), np.eye(3))
    assert np.array_equal(oracle.linspace(0.0, 0.5, 17), np.linspace(0.0, 0.5, 17))


def test_oracle_area_resize_and_gray(oracle):
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    half = oracle.resize_area_u8(g, (32, 20))
    man = ((g[0::2, 0::2].astype(int) + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(half, man)
    const = np.full((45, 73), 77, np.uint8)
    assert np.all(oracle.resize_area_u8(const, (31, 19)) == 77)
    rgb = np.full((4, 8, 3), 0.5, np.float32)
    assert np.all(oracle.rgb2gray_u8(rgb) == 127)  # truncation, not rounding (127.5 -> 127)


def test_oracle_dis_and_fit_recover_known_motion(oracle):
    from tests.test_dis_gpu import moving_clip

    gray, params = moving_clip(2, 135, 240, seed=3)
    flow = oracle.dis_flow_clip(gray)[0]
    tx, ty = params[1][0], params[1][1]
    assert abs(np.median(flow[..., 0]) - tx) < 0.1 and abs(np.median(flow[..., 1]) - ty) < 0.1
    m, mode, conf, resid = oracle.fit_from_flow(flow, 8, "similarity")
    assert mode == "similarity" and conf > 0.8 and resid < 0.5
    assert abs(m[0, 2] - tx) < 0.5 and abs(m[1, 2] - ty) < 0.5 and abs(m[0, 0] - 1) < 0.01
    assert oracle.dis_coarsest_scale(540, 960) == 5 and oracle.dis_coarsest_scale(480, 854) == 5


def test_min_eigen_map_building_blocks(oracle):
    """Oracle self-consistency of the Classic estimator restatement (oracle/vo_classic.c): pyrDown of a constant image is the
    constant, Scharr of a ramp is the analytic slope, the eigenvalue map of a flat image is zero."""
    flat = np.full((40, 60), 77, np.uint8)
    assert np.all(oracle.pyr_down(flat) == 77) and oracle.pyr_down(flat).shape == (20, 30)
    ramp = np.tile(np.arange(60, dtype=np.uint8) * 2, (40, 1))
    d = oracle.scharr_deriv(ramp)
    assert np.all(d[:, 1:-1, 0] == 2 * 2 * 16) and np.all(d[..., 1] == 0)
    assert np.all(oracle.min_eigen_val(flat) == 0)
    assert oracle.lk_levels(480, 854) == 3 and oracle.lk_levels(48, 64) == 0 and oracle.lk_levels(135, 240) == 2


def test_oracle_classic_estimator_recovers_known_motion(oracle):
    """GFTT -> LK -> fit on an analytic texture under a known similarity (no interpolation in the generator):
    the restated Classic estimator (classic.py:69-160) recovers the motion -- independent of any OpenCV."""
    from tests.test_dis_gpu import moving_clip

    h, w = 240, 426
    gray, params = moving_clip(2, h, w, seed=17)
    feats = oracle.good_features(gray[0], **oracle.GFTT)
    assert 100 < feats.shape[0] <= 400
    d = feats[:, None, :] - feats[None, :, :]
    dist2 = (d ** 2).sum(-1) + np.eye(len(feats)) * 1e9
    assert dist2.min() >= 49.0                      # minDistance = 7
    nxt, status = oracle.lk_track(gray[0], gray[1], feats, **oracle.LK)
    assert status.mean() > 0.9
    m, used, conf = oracle.classic_estimate_pair(gray[0], gray[1], "similarity")
    assert used == "similarity" and conf > 0.8

    def to_texture(pr):
        tx, ty, th, sc = pr
        c, sn = np.cos(th) / sc, np.sin(th) / sc
        lin = np.array([[c, sn, 0], [-sn, c, 0], [0, 0, 1.0]])
        return np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]]) @ lin @ np.array([[1, 0, -w / 2 - tx], [0, 1, -h / 2 - ty], [0, 0, 1.0]])

    expect = np.linalg.inv(to_texture(params[1])) @ to_texture(params[0])
    assert np.abs(m[:2, :2] - expect[:2, :2]).max() < 2e-3 and np.abs(m[:2, 2] - expect[:2, 2]).max() < 0.3


def test_host_math_is_pythons_libm(pkg):
    """vstab_host_math (no GPU involved) must return the bits of math.sqrt/atan2/log/exp/cos/sin -- the functions
    the reference's _matrix_to_params / _params_to_matrix call per frame (stabilizer_utils.py:300-358)."""
    import math

    from vstab_amd import native

    rng = np.random.default_rng(5)
    a = np.concatenate([rng.normal(0, 3, 4000), [0.0, -0.0, 1.0, 1e-300, 1e300, np.pi, -np.pi / 2]])
    b = np.concatenate([rng.normal(0, 3, 4000), [0.0, 0.0, -1.0, 1e300, 1e-300, -0.0, 0.0]])
    cases = {"sqrt": (np.abs(a),), "atan2": (a, b), "log": (np.abs(a) + 1e-12,), "exp": (np.clip(a, -700, 700),),
             "cos": (a * 1e3,), "sin": (a * 1e3,)}
    for name, args in cases.items():
        got = native.host_math(name, *args)
        want = np.array([getattr(math, name)(*[float(x[i]) for x in args]) for i in range(a.size)])
        assert np.array_equal(got, want, equal_nan=True), name


def test_shake_generator_nodes(pkg):
    """Schema of the two generator nodes (video_stabilizer_shake_generator.py:18-84, ..._manual.py:22-141) and one
    execution each on the CPU (they read only frame count / size / fps of the connected clip)."""
    import json

    import torch

    from vstab_amd import nodes, shake_generator as sg

    s = nodes.VideoStabilizerShakeGenerator.define_schema()
    assert s.node_id == "video_stabilizer_shake_generator" and s.display_name == "Video Stabilizer Shake Generator"
    assert [i.id for i in s.inputs] == ["frames_context", "frame_rate", "style", "amount", "speed", "seed"]
    assert [o.id for o in s.outputs] == ["motion_meta"]
    m = nodes.VideoStabilizerShakeGeneratorManual.define_schema()
    assert m.node_id == "video_stabilizer_shake_generator_manual"
    assert [i.id for i in m.inputs] == ["frames_context", "frame_rate", "pan", "tilt", "roll", "zoom", "drift_freq", "tremor",
                                        "tremor_freq", "jitter_rate", "step", "randomness", "virtual_fov", "amount", "speed", "seed"]
    frames = torch.zeros((12, 48, 64, 3))
    out = nodes.VideoStabilizerShakeGenerator.execute(frames, 24.0, "walking", 1.0, 1.0, 3)[0]
    want = sg.generate_shake_motion_meta(recipe=sg.STYLES["walking"], frame_count=12, width=64, height=48, fps=24.0, amount=1.0,
                                         speed=1.0, seed=3, node="shake_generator", style="walking")
    assert json.dumps(out) == json.dumps({"motion_meta": want}) and want["frame_count"] == 12 and want["generator"]["style"] == "walking"
    h = sg.STYLES["handheld"]
    out2 = nodes.VideoStabilizerShakeGeneratorManual.execute({"frames": frames, "fps": 30.0}, 16.0, *h, 1.0, 1.0, 0)[0]["motion_meta"]
    assert out2["fps"] == 30.0 and out2["generator"]["node"] == "shake_generator_manual" and out2["generator"]["style"] == "manual"
    ref = sg.generate_shake_motion_meta(recipe=h, frame_count=12, width=64, height=48, fps=30.0, amount=1.0, speed=1.0, seed=0,
                                        node="shake_generator", style="handheld")
    assert out2["per_frame"] == ref["per_frame"]      # same recipe, same seed -> same motion


def test_phase_oracle_against_float64_fft(oracle):
    """oracle/vo_phase.c (own mixed-radix float DFT) vs the same published algorithm on numpy.fft in float64, and
    exact recovery of circular shifts.  cv2.phaseCorrelate itself: parity unpinned (no OpenCV here)."""
    rng = np.random.default_rng(3)

    def np_phase(a, b):
        h, w = a.shape
        m, n = oracle.optimal_dft_size(h), oracle.optimal_dft_size(w)
        pa, pb = np.zeros((m, n)), np.zeros((m, n))
        pa[:h, :w], pb[:h, :w] = a, b
        p = np.fft.fft2(pa) * np.conj(np.fft.fft2(pb))
        c = np.real(np.fft.ifft2(p / (np.abs(p) + 1e-30))) * (m * n)
        c = np.roll(c, (m // 2, n // 2), axis=(0, 1))
        py, px = np.unravel_index(np.argmax(c), c.shape)
        r0, r1, c0, c1 = max(py - 2, 0), min(py + 2, m - 1), max(px - 2, 0), min(px + 2, n - 1)
        ys, xs = np.mgrid[r0:r1 + 1, c0:c1 + 1]
        win = c[r0:r1 + 1, c0:c1 + 1]
        return n / 2.0 - (xs * win).sum() / win.sum(), m / 2.0 - (ys * win).sum() / win.sum(), win.sum() / (m * n)

    assert [oracle.optimal_dft_size(v) for v in (1, 7, 17, 135, 161, 540, 541, 960)] == [1, 8, 18, 135, 162, 540, 576, 960]
    for h, w in [(135, 240), (100, 161), (270, 480)]:
        base = rng.integers(0, 256, (h, w), dtype=np.uint8)
        moved = np.roll(base, (4, -9), axis=(0, 1))
        clip = np.stack([base, moved, base])
        got = oracle.phase_correlate_clip(clip)
        for p in range(2):
            ref = np_phase(clip[p].astype(np.float64), clip[p + 1].astype(np.float64))
            # the four purely real bins follow OpenCV's packed-format helpers (C = P/(P^2+eps) ~ 0 instead of +-1):
            # at most 4 * 25 / (M N) in the window sum, far less in the centroid
            assert abs(got[p, 0] - ref[0]) < 5e-3 and abs(got[p, 1] - ref[1]) < 5e-3 and abs(got[p, 2] - ref[2]) < 100.0 / (h * w) + 1e-4
        if (oracle.optimal_dft_size(h), oracle.optimal_dft_size(w)) == (h, w):   # unpadded: the shift is circular
            # phasecorr.cpp measures from (cols/2.0, rows/2.0) while fftShift centres on cols/2, rows/2 (integer):
            # an odd side reports the shift + 0.5, as cv2.phaseCorrelate does
            ox, oy = w / 2.0 - w // 2, h / 2.0 - h // 2
            assert abs(got[0, 0] - (-9 + ox)) < 1e-3 and abs(got[0, 1] - (4 + oy)) < 1e-3
            assert abs(got[1, 0] - (9 + ox)) < 1e-3 and abs(got[1, 1] - (-4 + oy)) < 1e-3


def test_fallback_backend_selection(pkg, monkeypatch):
    """flow.py:90-107 restated: DIS unless the fallback is requested; the phase table walks to all-translation."""
    from vstab_amd import flow_pipeline as fp
    from vstab_amd import native

    monkeypatch.delenv("VSTAB_FLOW_BACKEND", raising=False)
    assert fp.resolve_flow_backend("flow") == "flow" and fp._backend_fields("flow") == {"flow_backend": "DIS", "flow_fallback_reason": None}
    monkeypatch.setenv("VSTAB_FLOW_BACKEND", "phase_correlate")
    assert fp.resolve_flow_backend("flow") == "flow_phase_correlate" and fp.resolve_flow_backend("classic") == "classic"
    fields = fp._backend_fields("flow_phase_correlate")
    assert fields["flow_backend"] == "phase_correlate" and "using phase correlation." in fields["flow_fallback_reason"]
    table = np.zeros((4, 3), native.FIT_DTYPE)
    table["matrix"][:] = np.eye(3, dtype=np.float32).reshape(9)
    table["computed"][:, 0] = table["accepted"][:, 0] = 1
    table["matrix"][:, 0, 2] = [1.5, -2.0, 0.25, 3.0]
    table["confidence"][:, 0] = [0.9, 0.8, 0.7, 0.6]
    for requested in ("translation", "similarity", "perspective"):
        mats, modes, confs, resids, active = fp.select_transitions(table, requested)
        assert modes == ["translation"] * 4 and active == "translation" and confs == [0.9, 0.8, 0.7, 0.6] and resids == [0.0] * 4
        assert mats[:, 0, 2].tolist() == [1.5, -2.0, 0.25, 3.0]


def test_bench_multi_gpu_launcher_spawns_children_not_exec(monkeypatch, tmp_path):
    """`python bench.py --gpus N` without a torchrun environment (how the driver called `--gpus 1` in round 1) must
    start the N ranks itself -- as a CHILD `torch.distributed.run` process, before anything touches the GPU, never an
    exec (an exec from a GPU-initialised process takes the box down) -- and hand back the child's exit code."""
    import importlib
    import subprocess

    import bench

    importlib.reload(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    assert bench.main() == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    src = (ROOT / "bench.py").read_text()
    assert "os.exec" not in src and "execv" not in src


def test_one_hip_runtime_in_the_process(pkg):
    """libvstab.so and torch must share ONE libamdhip64: torch's wheel bundles its own copy under the same SONAME, and
    whichever is loaded first wins.  If libvstab came first (e.g. build() followed by smoke() in one process), torch would
    run on /opt/rocm's runtime next to its own bundled HSA runtime and report "no ROCm-capable device" on the GPU box
    (seen in round 2).  load_library() therefore imports torch first; checked in a fresh interpreter that has not
    imported torch before."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g\n"
        "g.load_package()\n"
        "from vstab_amd import native\n"
        "assert 'torch' not in sys.modules\n"
        "native.load_library()\n"
        "libs = sorted({l.split()[-1] for l in open('/proc/self/maps').read().split('\\n') if 'libamdhip64' in l})\n"
        "print(libs)\n"
        "assert len(libs) == 1 and '/torch/lib/' in libs[0], libs\n" % str(ROOT)
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def _plain(value):
    """Enum-like members of either io implementation -> 'Enum.member' strings, as the fixture stores them."""
    if isinstance(value, (str, int, float, bool, list, type(None))):
        return value
    name = getattr(value, "name", None)
    return f"{type(value).__name__}.{name}" if name else str(value)


def test_node_schema_contents_match_reference(pkg):
    """everything the reference's define_schema bodies declare beyond the socket ids --
    type, default, min, max, step, combo options, display_name, display_mode, control_after_generate, and the schema's
    node_id / display_name / category / is_deprecated -- for all six nodes, against tests/golden/reference_schema.json
    (generated by tests/golden/make_schema_golden.py from the imported reference classes:
    video_stabilizer_flow.py:657-731, video_stabilizer_motion_apply.py:40-83, ...).  Free text (description, tooltip)
    is UI documentation and may differ (the Classic node's "CPU-friendly" is not true of this build)."""
    import json

    from vstab_amd import nodes

    golden = json.loads((ROOT / "tests" / "golden" / "reference_schema.json").read_text())
    assert len(golden) == 6 and len(nodes.NODE_CLASSES) == 6
    seen = set()
    for cls in nodes.NODE_CLASSES:
        s = cls.define_schema()
        want = golden[s.node_id]
        seen.add(s.node_id)
        assert cls.__name__ == want["class"]
        assert s.display_name == want["display_name"] and s.category == want["category"]
        assert bool(getattr(s, "is_deprecated", False)) == bool(want.get("is_deprecated", False))
        for got_list, want_list in ((s.inputs, want["inputs"]), (s.outputs, want["outputs"])):
            assert [g.id for g in got_list] == [w["id"] for w in want_list], s.node_id
            for g, w in zip(got_list, want_list):
                where = f"{s.node_id}.{w['id']}"
                assert g.kind == w["type"] and g.direction.lower() == w["direction"], where
                declared = {k: _plain(v) for k, v in g.options.items() if k != "tooltip"}
                expected = {k: v for k, v in w.items() if k not in ("type", "direction", "id", "tooltip")}
                for enum_key in ("display_mode", "control_after_generate"):   # enum members: compared by member name
                    for d in (declared, expected):
                        if enum_key in d:
                            d[enum_key] = str(d[enum_key]).split(".")[-1]
                assert declared == expected, (where, declared, expected)
    assert seen == set(golden)


@pytest.mark.parametrize("mode", ["translation", "similarity", "perspective"])
def test_clip_parameter_maps_equal_the_per_item_forms(pkg, mode):
    """vstab_transitions_to_params / vstab_params_to_matrices (host only) against the per-item restatements of
    stabilizer_utils.py:279-358 (`_rescale_transform_to_full`, `_matrix_to_params`, `_params_to_matrix`, themselves pinned
    bit for bit by the reference-run goldens of tests/test_host_golden.py) and against the vectorised NumPy forms: same
    bits, with and without the working-size rescale, incl. degenerate matrices (zero 2x2 part: the 1e-10 floor)."""
    from vstab_amd import host_math as hm
    from vstab_amd import native

    rng = np.random.default_rng(11)
    n = 300
    mats = np.tile(np.eye(3, dtype=np.float32), (n, 1, 1))
    mats[:, :2, :2] += rng.normal(0, 0.05, (n, 2, 2)).astype(np.float32)
    mats[:, :2, 2] = rng.normal(0, 20, (n, 2)).astype(np.float32)
    if mode == "perspective":
        mats[:, 2, :2] = rng.normal(0, 1e-4, (n, 2)).astype(np.float32)
    mats[0, :2, :2] = 0.0          # a*a + c*c below the floor
    mats[1, 0, 0], mats[1, 1, 0] = -1.0, 0.0   # atan2 at pi
    for size, work in (((1920, 1080), (960, 540)), ((1000, 777), (960, 746)), ((480, 270), None)):
        full, params = native.transitions_to_params(mats, mode, size, work)
        want_full = np.stack([hm._rescale_transform_to_full(m, size, work) if work else m for m in mats])
        assert full.dtype == np.float32 and np.array_equal(full, want_full)
        want_params = np.stack([hm._matrix_to_params(m, mode) for m in want_full])
        assert params.dtype == np.float64 and np.array_equal(params, want_params)
        assert np.array_equal(params, hm.matrices_to_params(hm.rescale_transforms_to_full(mats, size, work) if work else mats, mode))
    p = rng.normal(0, 1.0, (n, native.PARAM_COUNT[mode])) * (0.05 if mode != "translation" else 30.0)
    got = native.params_to_matrices(p, mode)
    want = np.stack([hm._params_to_matrix(row, mode) for row in p])
    assert got.dtype == np.float32 and np.array_equal(got, want) and np.array_equal(got, hm.params_to_matrices(p, mode))
    assert native.params_to_matrices(np.zeros((0, native.PARAM_COUNT[mode])), mode).shape == (0, 3, 3)


def test_clip_bounding_boxes_equal_numpy(pkg):
    """vstab_bounding_boxes (host only) against `_compute_bounding_boxes` (stabilizer_utils.py:1010-1034, per item: `m @
    corners` with an f32 matrix and fp64 corners) and its batched NumPy form: same bits for similarity and perspective
    matrices, for a projective denominator of zero (inf / NaN corners: NumPy's minimum / maximum propagate NaN) and for
    the frame sizes of every BASELINE config."""
    import warnings

    from vstab_amd import host_math as hm
    from vstab_amd import native

    rng = np.random.default_rng(21)
    n = 4000
    mats = np.tile(np.eye(3, dtype=np.float32), (n, 1, 1))
    mats[:, :2, :2] += rng.normal(0, 0.05, (n, 2, 2)).astype(np.float32)
    mats[:, :2, 2] = rng.normal(0, 40, (n, 2)).astype(np.float32)
    mats[n // 2:, 2, :2] = rng.normal(0, 2e-5, (n - n // 2, 2)).astype(np.float32)
    mats[0, 2] = (0.0, 0.0, 0.0)                      # W == 0 at every corner: 0/0 and x/0
    mats[1, 2] = (-1.0 / 1920, 0.0, 1.0)              # W == 0 at the right corners only
    mats[2, 0, 0] = np.nan
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        for (w, h) in ((1920, 1080), (3840, 2160), (854, 480), (73, 45)):
            mins, maxs = native.bounding_boxes(mats, w, h)
            want_mins, want_maxs = hm.bounding_boxes_batched(mats, w, h)
            assert np.array_equal(mins, want_mins, equal_nan=True) and np.array_equal(maxs, want_maxs, equal_nan=True)
            item_mins, item_maxs = hm._compute_bounding_boxes(list(mats[:300]), w, h)
            assert np.array_equal(mins[:300], item_mins, equal_nan=True) and np.array_equal(maxs[:300], item_maxs, equal_nan=True)
    assert native.bounding_boxes(np.zeros((0, 3, 3), np.float32), 10, 10)[0].shape == (0, 2)


def test_roofline_traffic_is_tied_to_the_warp_sources(tmp_path, monkeypatch):
    """bench.warp_traffic_record: the PMC traffic figure is handed out only for the workload and the kernel sources it was
    collected on (sha256 of csrc/vstab_warp.hip + vstab_internal.h in profiles/warp_traffic.json); otherwise None + reason."""
    import json

    import bench

    rec = json.loads((ROOT / "profiles" / "warp_traffic.json").read_text())
    traffic, why = bench.warp_traffic_record(256, 1920, 1080)
    if rec["kernel_source_sha256"] == bench.warp_source_sha256():
        assert traffic == rec["hbm_bytes_per_launch"] and rec["kernel_source_sha256"][:12] in why
        assert 0.99 < traffic / rec["algorithmic_bytes_per_launch"] < 1.01
    else:   # the warp sources were edited after the collection: the bench line must say so, not quote the stale figure
        assert traffic is None and "re-run tools/pmc_traffic.sh" in why
    assert bench.warp_traffic_record(128, 1920, 1080)[0] is None
    monkeypatch.setattr(bench, "warp_source_sha256", lambda: "0" * 64)
    traffic, why = bench.warp_traffic_record(256, 1920, 1080)
    assert traffic is None and "other warp kernel sources" in why


def test_shipped_library_has_no_fault_injectors(pkg):
    """the VSTAB_DEBUG_* fault injectors are compiled out of the shipped library; only the test build
    (lib/libvstab_hooks.so, -DVSTAB_TEST_HOOKS, loaded by the tests that need them in a child process) reads them."""
    import ctypes

    from vstab_amd import native

    lib_dir = native.LIB_PATH.parent
    shipped, hooks = lib_dir / "libvstab.so", lib_dir / "libvstab_hooks.so"
    assert shipped.exists() and hooks.exists(), "run __graft_entry__.build()"
    assert b"VSTAB_DEBUG_" not in shipped.read_bytes()
    blob = hooks.read_bytes()
    for knob in (b"VSTAB_DEBUG_PLAN_PERTURB", b"VSTAB_DEBUG_PIS_SPIN_LIMIT", b"VSTAB_DEBUG_XFER_SPAWN_FAIL"):
        assert knob in blob
    assert ctypes.CDLL(str(shipped)).vstab_test_hooks() == 0
    # (the hooks build is a second copy of the same exported symbols: query it in a child so that this process keeps one library)
    import subprocess
    import sys

    out = subprocess.run([sys.executable, "-c", f"import ctypes; print(ctypes.CDLL({str(hooks)!r}).vstab_test_hooks())"],
                         capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "1", out.stderr[-2000:]


def test_coded_transfer_host_loops_baseline_build():
    """The same checks on the loops' baseline (SSE2) build, which a machine with AVX2 never runs otherwise: a child process with
    VSTAB_CODEC_BASELINE=1 (the choice is made when the library is loaded)."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, VSTAB_CODEC_BASELINE="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", f"{__file__}::test_coded_transfer_host_loops"], env=env,
                         capture_output=True, text=True, cwd=str(Path(__file__).resolve().parents[1]))
    assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-800:] + out.stderr[-400:]


def test_coded_transfer_host_loops(pkg):
    """The host halves of the coded node-boundary transfers (csrc/vstab_codec.cpp, called by vstab_upload_f32_coded /
    vstab_download_mask_coded): a run of values is accepted for the byte form only if EVERY value has exactly the bits of
    float32(k) / float32(255) -- what nodes/stabilizer_utils.py:122-126 makes of an 8-bit frame -- and the bytes are those k;
    one value of any other kind anywhere in the run (vector body or scalar tail) rejects it."""
    import ctypes

    from vstab_amd import native

    lib = native.load_library()
    enc, exp = lib.vstab_host_encode_q8, lib.vstab_host_expand_mask
    enc.restype, enc.argtypes = ctypes.c_bool, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    exp.restype, exp.argtypes = None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(0)
    for n in (1, 7, 31, 32, 33, 1000, 4099):
        k = rng.integers(0, 256, n).astype(np.uint8)
        v = k.astype(np.float32) / np.float32(255.0)
        out = np.full(n, 77, np.uint8)
        assert enc(v.ctypes.data, out.ctypes.data, n) and np.array_equal(out, k)
        for special in (0.5, np.nan, -0.0, np.nextafter(np.float32(1.0), np.float32(2.0)), -1e-9, 255.0, np.inf, -np.inf,
                        np.nextafter(np.float32(3.0) / np.float32(255.0), np.float32(0.0)), 1e-40, 3e9, -3e9):
            for pos in {0, n // 2, n - 1}:
                w = v.copy()
                w[pos] = special
                assert not enc(w.ctypes.data, out.ctypes.data, n), (n, special, pos)
        back = np.empty(n, np.float32)
        exp(k.ctypes.data, back.ctypes.data, n)
        assert np.array_equal(back.view(np.uint32), np.where(k != 0, np.float32(1.0), np.float32(0.0)).view(np.uint32))
    lev = lib.vstab_host_expand_levels
    lev.restype, lev.argtypes = None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lut = rng.random(256).astype(np.float32)
    for n in (1, 31, 1000, 4099):
        c = rng.integers(0, 256, n).astype(np.uint8)
        got = np.empty(n, np.float32)
        lev(c.ctypes.data, got.ctypes.data, n, lut.ctypes.data)
        assert np.array_equal(got.view(np.uint32), lut[c].view(np.uint32))
    # every quotient, and both of its float32 neighbours (never a quotient themselves)
    q = np.arange(256, dtype=np.float32) / np.float32(255.0)
    out = np.empty(256, np.uint8)
    assert enc(q.ctypes.data, out.ctypes.data, 256) and np.array_equal(out, np.arange(256, dtype=np.uint8))
    for nb in (np.nextafter(q[1:], np.float32(0.0)), np.nextafter(q[1:], np.float32(2.0))):
        for i in range(255):
            one = nb[i:i + 1].copy()
            assert not enc(one.ctypes.data, out.ctypes.data, 1)
