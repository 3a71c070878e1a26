#!/usr/bin/env python3
"""Schema fixture from the REFERENCE's own node classes, generated in the build container.

    python tests/golden/make_schema_golden.py      (needs /root/reference; writes tests/golden/reference_schema.json)

The reference's six node classes are imported from /root/reference (under the `cv2` stand-in, which no schema touches)
with a RECORDING `comfy_api.latest.io`: every `io.<Type>.Input(id, **options)` / `.Output(...)` / `io.Schema(**fields)`
call the reference's `define_schema` bodies make (video_stabilizer_flow.py:644-731, video_stabilizer_classic.py:571-657,
video_stabilizer_motion_apply.py:32-84, video_stabilizer_inverse.py:29-60, video_stabilizer_shake_generator.py:23-75,
video_stabilizer_shake_generator_manual.py:25-130) is written down as plain data: socket type, id, default, min, max,
step, options, display_name, display_mode, tooltip, and the schema's node_id / display_name / category / is_deprecated.
The reference's own scripts/check_node_schema.py:11-94 reads the same declarations by AST but only checks ids; running
them resolves the non-literal defaults too (HANDHELD_DEFAULT.pan, list(STYLES.keys()), ...).  No reference code is
stored -- only the values.  tests/test_abi_cpu.py compares comfyui-video-stabilizer_amd/nodes.py against the fixture.
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))

from tests.golden import cv2_standin  # noqa: E402

NODE_MODULES = ["video_stabilizer_classic", "video_stabilizer_flow", "video_stabilizer_motion_apply", "video_stabilizer_inverse",
                "video_stabilizer_shake_generator", "video_stabilizer_shake_generator_manual"]


class _Enum:
    def __init__(self, prefix):
        self._prefix = prefix

    def __getattr__(self, name):
        return f"{self._prefix}.{name}"


def _socket_type(kind):
    def make(direction):
        def ctor(socket_id=None, **options):
            return {"type": kind, "direction": direction, "id": socket_id, **options}
        return staticmethod(ctor)

    return type(kind, (), {"Input": make("input"), "Output": make("output")})


class _Schema:
    def __init__(self, **fields):
        self.fields = dict(fields)
        self.inputs, self.outputs = [], []


def _install_recording_comfy():
    io = types.SimpleNamespace(
        Schema=_Schema, ComfyNode=object, NodeOutput=tuple, NumberDisplay=_Enum("NumberDisplay"),
        ControlAfterGenerate=_Enum("ControlAfterGenerate"),
        Custom=lambda kind: _socket_type(kind),
        **{k: _socket_type(k) for k in ("Image", "Mask", "Float", "Int", "Boolean", "Combo", "Color", "String")})
    latest = types.ModuleType("comfy_api.latest")
    latest.io, latest.ComfyExtension = io, type("ComfyExtension", (), {})
    api = types.ModuleType("comfy_api")
    api.latest = latest
    sys.modules["comfy_api"], sys.modules["comfy_api.latest"] = api, latest
    comfy, comfy_utils = types.ModuleType("comfy"), types.ModuleType("comfy.utils")
    comfy_utils.ProgressBar = type("ProgressBar", (), {"__init__": lambda self, total: None})
    comfy.utils = comfy_utils
    sys.modules["comfy"], sys.modules["comfy.utils"] = comfy, comfy_utils


def main() -> None:
    import importlib

    _install_recording_comfy()
    cv2_standin.install()
    sys.path.insert(0, str(REF))
    out = {}
    for name in NODE_MODULES:
        mod = importlib.import_module(f"nodes.{name}")
        classes = [c for c in vars(mod).values() if isinstance(c, type) and c.__module__ == mod.__name__ and "define_schema" in vars(c)]
        assert len(classes) == 1, (name, classes)
        schema = classes[0].define_schema()
        out[schema.fields["node_id"]] = {"class": classes[0].__name__, **schema.fields, "inputs": schema.inputs, "outputs": schema.outputs}
    path = HERE / "reference_schema.json"
    path.write_text(json.dumps(out, indent=1, sort_keys=False) + "\n")
    print(f"wrote {path} ({len(out)} nodes, {sum(len(v['inputs']) for v in out.values())} inputs)")


if __name__ == "__main__":
    main()
