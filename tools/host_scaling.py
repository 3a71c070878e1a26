"""Host-side cost of plan + meta as the clip grows (replicated on every rank in multi-GPU runs)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as graft
graft.load_package()
from vstab_amd import flow_pipeline as fp, native
from tests.test_distributed_cpu import NumpyTrajectoryCtx, fake_records
for n in (256, 1024, 2048):
    table = native.fit_table_from_dicts(fake_records(n))
    t = time.perf_counter()
    for _ in range(5):
        plan = fp.plan_stabilization(NumpyTrajectoryCtx(), table, (1920, 1080), n, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, (127,127,127), 16.0, 16.0)
    t1 = (time.perf_counter() - t) / 5
    t = time.perf_counter()
    for _ in range(5):
        meta = fp.finish_meta(plan, np.zeros(n, np.int64))
    t2 = (time.perf_counter() - t) / 5
    print(f"n={n}: plan {t1*1e3:.2f} ms (incl. numpy trajectory stand-in), meta {t2*1e3:.2f} ms")
