#!/usr/bin/env python3
"""Run the reference's own check scripts against the reference's code under the oracle-backed cv2 stand-in
(build container only; nothing here travels).  A sanity check of the stand-in: if the reference's scripted properties
(scripts/check_motion_meta.py: identity apply, blur determinism, tick counts, crop fallback, legacy inversion ...)
hold with the stand-in's primitives, the end-to-end fixtures made with it (make_e2e_golden.py) rest on a cv2 that at
least behaves like one.  scripts/check_crop_aspect_ratio.py and check_inverse_stabilization.py draw their test clips
with cv2.rectangle / cv2.circle / cv2.warpAffine(BORDER_REFLECT); the stand-in answers those with plain NumPy (they make
inputs, they are not results under test), and serves the Classic node's goodFeaturesToTrack / calcOpticalFlowPyrLK from
the oracle, so all four functional scripts run.

    python tests/golden/run_reference_checks.py
"""
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import cv2_standin  # noqa: E402
from tests.golden.make_e2e_golden import _install_comfy_stubs  # noqa: E402

REF = Path("/root/reference")


def main() -> int:
    cv2_standin.install()
    _install_comfy_stubs()
    sys.path.insert(0, str(REF))
    rc = 0
    for script in ("check_motion_meta.py", "check_node_schema.py", "check_crop_aspect_ratio.py", "check_inverse_stabilization.py"):
        sys.argv = [str(REF / "scripts" / script)]
        try:
            runpy.run_path(str(REF / "scripts" / script), run_name="__main__")
            print(f"{script}: finished without SystemExit")
        except SystemExit as exc:
            code = exc.code if isinstance(exc.code, int) else (0 if exc.code is None else 1)
            print(f"{script}: exit code {code}")
            rc |= code
    return rc


if __name__ == "__main__":
    sys.exit(main())
