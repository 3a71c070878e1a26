#!/usr/bin/env python3
"""Turns the files `tools/final_profiles.sh <tag>` left under gpurun_out/ into the round's profile notes:
profiles/<tag>_bench_c2.{log,md}, _bench_c2_kernel_stats.csv, _bench_c5.md, _c4_strong_scaling.md, _c4_single_gpu.json,
_analytic_accuracy.md.   python tools/write_round_docs.py r03"""
import csv, json, re, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = ROOT / "gpurun_out", ROOT / "profiles"


def line(name):
    return json.loads([x for x in open(G / name) if x.startswith("{")][-1])


shutil.copy(G / f"{TAG}_bench_c2.log", P / f"{TAG}_bench_c2.log")
shutil.copy(G / f"{TAG}_bench_c2_kernel_stats.csv", P / f"{TAG}_bench_c2_kernel_stats.csv")
rows = []
for r in csv.reader(open(G / f"{TAG}_bench_c2_kernel_stats.csv")):
    m = re.search(r"\(anonymous namespace\)::([A-Za-z0-9_]+(<[^>]*>)?)", r[0]) if r else None
    if m and not r[0].startswith("void at::") and "at::native" not in r[0][:30]:
        rows.append((m.group(1), int(r[1]), float(r[3]) / 1e3))
steps = max(c for n, c, a in rows if n.startswith("warp_kernel"))
ks = "| kernel | calls | avg us | us per 256-frame clip |\n|---|---|---|---|\n" + "".join(
    f"| `{n}` | {c} | {a:.1f} | {c / steps * a:.0f} |\n" for n, c, a in rows)
per_clip = {n.split("<")[0]: c / steps * a for n, c, a in rows}

j = line(f"{TAG}_bench_c2.log")
acc, par, ma = j["accuracy"]["hip"], j["parity_at_size"], j["motion_apply"]
c3, c5m = ma["c3_1080p_bicubic_blur0.5_S17"], ma["c5_4k_expand_bilinear_blur0.5_S33"]
st = j["config"]["stage_ms"]
(P / f"{TAG}_bench_c2.md").write_text(f"""# C2 bench line and kernel times, round 3 final build

Command (one MI355X box, `tools/final_profiles.sh {TAG}`): `python bench.py --steps 20 --warmup 5` -> `profiles/{TAG}_bench_c2.log`;
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-checks
--cpu-frames 0` -> `profiles/{TAG}_bench_c2_kernel_stats.csv` ({steps} steps incl. warm-up).

**{j["value"]:.0f} frames/s, {j["ms_per_step"]} ms per 256-frame 1080p clip** (round 2's driver line: 30 724 / 8.332 ms; other boxes this
round: 31.2-31.7 k / 8.07-8.20 ms -- the boxes differ by ~5 % in HBM rate).  HIP-event stage times of the timed steps: gray
{st["gray"]} / DIS {st["dis"]} / fit {st["fit"]} / warp {st["warp"]} ms (sum {sum(st.values()):.2f}; the rest: host plan between fit and warp ~0.17-0.2 ms, pyramid
preparation gaps, final sync).  Roofline of the warp kernel: 14.864 GB algorithmic / {j["roofline"]["launch_ms"]} ms = {j["roofline"]["achieved"]} GB/s =
**{j["roofline"]["frac"]} of 8 TB/s** on this box (0.69-0.73 over the boxes; rocprof average of the same kernel under tracing:
{per_clip.get("warp_kernel", 0):.0f} us); PMC traffic per launch 14.862 GB (`profiles/r03_hbm_traffic.csv`: FETCH_SIZE x 2 + WRITE_SIZE, re-collected this
round: unchanged, = algorithmic).

Self-verification on the same line (`accuracy`, `parity_at_size`, `batch_invariance`; outside the timed loop):

* analytic accuracy of the 255 reported transitions, px at working resolution: centre max {acc["centre_px"]["max"]} / mean
  {acc["centre_px"]["mean"]} / p99 {acc["centre_px"]["p99"]}; worst corner max {acc["corner_px"]["max"]}; 2x2 part max {acc["lin_2x2"]["max"]:.2e} -- identical
  for the CPU port (it produces the same bits);
* HIP run vs the oracle's run of the same 256 frames: `bit_equal: {str(par["bit_equal"]).lower()}` -- transition matrices equal, confidences equal,
  residuals within {par["residuals_max_rel_diff"]:.1e} (fp64 sum order), final matrices equal, **{par["pixels_differing"]} of 1 592 524 800 output values and
  {par["mask_pixels_differing"]} of 530 841 600 mask values differ**, padding statistics equal;
* batch invariance: pairs {{0, 127, 254}} as 2-frame clips (split DIS form) and frames {{0, 127, 255}} warped alone equal the
  whole-clip run (`fit_records_equal` {j["batch_invariance"]["fit_records_equal"]}, `frames_equal` {j["batch_invariance"]["frames_equal"]}).

CPU port on 16 host threads: {j["cpu_baseline"]["value"]:.0f} frames/s (~{j["value"] / j["cpu_baseline"]["value"]:.0f}x; a baseline, not a target).  `host_roundtrip` {j["host_roundtrip"]["ms"]} ms
({j["host_roundtrip"]["frames_per_s"]} frames/s, CPU tensor in -> CPU tensors out).  `motion_apply`: C3 {c3["ms_per_pass"]} ms per 256x1080p
({c3["frames_per_s"]} frames/s), C5 share {c5m["ms_per_pass"]} ms per 64x4K ({c5m["frames_per_s"]} frames/s).

## Kernel times (rocprofv3 --stats, library kernels only)

{ks}
DIS = `pis4_kernel` {per_clip.get("pis4_kernel", 0) / 1e3:.2f} + `level_kernel` {per_clip.get("level_kernel", 0) / 1e3:.2f} ms + preparation (pyramid, padding, Sobel, tensors; partly on a
second stream); round 2: 1.46 + 2.31.
""")

c5p, c5d = line(f"{TAG}_bench_c5_plain.log"), line(f"{TAG}_bench_c5_dist1.log")
sp, sd, hd = c5p["config"]["rank0_stage_ms"], c5d["config"]["rank0_stage_ms"], c5d["config"]["rank0_host_ms"]
shape = c5p["config"]["out_shape_rank0"]
(P / f"{TAG}_bench_c5.md").write_text(f"""# BASELINE configs[4] (C5) as a bench workload, one GPU (round 3)

`bench.py --workload c5`: one step = Flow (DIS, similarity) with expand framing over the rank's 4K frames, then Motion
Apply (expand, bilinear, motion_blur 0.5, Ultra = 33 samples) on the rank's ORIGINAL frames with the returned meta.
One MI355X holds one GPU's share of the 8-GPU config (64 of 512 frames); same box, `tools/final_profiles.sh {TAG}`:

| run | frames/s | ms per step | gray | DIS | fit | warp (Flow's own output) | blur warp | host |
|---|---|---|---|---|---|---|---|---|
| `--workload c5` (single process) | {c5p["value"]} | {c5p["ms_per_step"]} | {sp["gray"]} | {sp["dis"]} | {sp["fit"]} | {sp["warp"]} | {sp["warp_blur"]} | -- |
| `--workload c5 --gpus 1 --force-dist` (sharded code path inside a world-1 RCCL group) | {c5d["value"]} | {c5d["ms_per_step"]} | {sd["gray"]} | {sd["dis"]} | {sd["fit"]} | {sd["warp"]} | {sd["warp_blur"]} | gather_fits {hd["gather_fits"]}, plan {hd["plan"]}, meta {hd["meta"]} ms |

Output canvas {shape[2]}x{shape[1]} (expand).  The step is the blur warp ({sp["warp_blur"]} of {c5p["ms_per_step"]} ms: 64 x {shape[2]} x {shape[1]} x 33 =
{64 * shape[1] * shape[2] * 33:.3g} pixel-samples); round 2's kernel needed ~34 ms for it.  An 8-GPU run of the 512-frame clip does this per
rank plus one all-gather of 64 x 3 fit records and the replicated plan over 512 frames (~0.3 ms); the replay half has no
collective.  Every default `bench.py --gpus N` (N > 1) line carries this measurement as its `c5` object (rehearsed with 2
and 4 ranks on one GPU: `--rehearse-on-one-gpu`).
""")

c4, d1 = line(f"{TAG}_c4_single_gpu.log"), line(f"{TAG}_c4_dist1_128.log")
em = (G / f"{TAG}_emulate_c4.log").read_text().strip()
steps_ms = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"world (\d+): ([0-9.]+) ms/step", em)}
plans = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"world (\d+):.*?plan ([0-9.]+),", em)}
one = c4["ms_per_step"]
json.dump({"command": "python bench.py --gpus 1 --total-frames 1024 --steps 10 --warmup 3 --no-extras --no-checks --cpu-frames 0",
           "total_frames": 1024, "size": [1920, 1080], "frames_per_s": c4["value"], "ms_per_step": one, "stage_ms": c4["config"]["stage_ms"],
           "note": "the C4 clip (BASELINE configs[3]) on ONE MI355X, device-resident: the denominator of the strong-scaling ratio",
           "source": f"round 3 final build, tools/final_profiles.sh {TAG}"}, open(P / f"{TAG}_c4_single_gpu.json", "w"), indent=1)
table = "".join(f"| {w} | {1024 // w} | {steps_ms[w]:.2f} | {one / (steps_ms[w] + 0.4):.2f}-{one / (steps_ms[w] + 0.2):.2f} |\n" for w in (2, 4, 8))
cs, ds, dh = c4["config"]["stage_ms"], d1["config"]["stage_ms"], d1["config"]["rank0_host_ms"]
(P / f"{TAG}_c4_strong_scaling.md").write_text(f"""# C4 (1024 x 1080p, Flow similarity): the one-GPU measurements behind the multi-GPU estimate (round 3)

No multi-GPU box was available to the build in any round (the driver's SCALE run was skipped in rounds 1 and 2 for
the same reason); this is an ESTIMATE from one box (`tools/final_profiles.sh {TAG}`), not a scaling curve.

| measurement | ms per step |
|---|---|
| the whole clip on one GPU (`bench.py --gpus 1 --total-frames 1024`) | **{one}** ({c4["value"]:.0f} frames/s; gray {cs["gray"]}, DIS {cs["dis"]}, fit {cs["fit"]}, warp {cs["warp"]}) |
| one rank's share through the sharded code inside a world-1 RCCL group (`--gpus 1 --force-dist --total-frames 128`) | {d1["ms_per_step"]} (gray {ds["gray"]}, DIS {ds["dis"]} split form, fit {ds["fit"]}, warp {ds["warp"]}; host: gather_fits {dh["gather_fits"]}, plan over 128 frames {dh["plan"]}) |
| rank 0's step of an N-rank run with the gathered tables of N ranks, no collective (`tools/emulate_world.py`) | see below |

```
{em}
```

| ranks | frames per rank | rank 0's step, ms | + 0.2-0.4 ms RCCL -> speed-up vs {one} ms |
|---|---|---|---|
{table}
Unchanged conclusion: **short of the >= 6x target** (needs <= {one / 6 - 0.3:.2f} ms per rank incl. RCCL).  What the round changed: the
replicated plan over 1024 frames 0.59 -> {plans.get(8, 0):.2f} ms (clip-wide parameter maps and bounding boxes in the library), DIS for 127
pairs 2.57 -> {ds["dis"]} ms.  Where a rank's step goes (kernel trace of a 128-frame step, `gpurun_out/r03_timeline128.log`): gray
0.48 + pyramid / gradients 0.2; DIS 2.4 = patch search 1.07 (52 + 109 + 260 + 645 us over the four levels -- the finest
level takes 645 us for 127 pairs, 568 us for ONE pair and 822 us for 255: one wavefront per stripe walks 59 patches x
2 passes x up to 12 dependent descent iterations; with 127 pairs there is one wavefront per SIMD and nothing to hide
that chain behind) + refinement 1.24 (split form, throughput-bound: the same 6.3 us per pair as the fused form at 255
pairs) + preparation; fit 0.1 + D2H; all-gather; plan; warp 1.33 (HBM-bound, shards perfectly); rank 0 also builds the
meta (0.7 ms, hidden behind its warp).  gray / fit / warp and the refinement shard perfectly, the patch search does not
shard at all below ~256 pairs per GPU: its 1.07 ms are a latency floor set by OpenCV's raster dependency inside a stripe
(left and upper neighbour's result seed each patch), which bit-exactness keeps.  Splitting a rank's pairs into concurrent
halves cannot help (each half pays the same chain; measured slower in round 2).  Weak scaling (256 frames per GPU,
`--frames 256`) does not have this problem: every rank runs the N = 1 step plus the all-gather and an O(N) plan.
""")

out = ["| case | modes | centre px max / mean | corner px max / mean | 2x2 max | true motion px |", "|---|---|---|---|---|---|"]
for l in open(G / f"{TAG}_analytic.log"):
    if not l.startswith("{"):
        continue
    a = json.loads(l)
    if "centre_px" in a:
        out.append(f"| {a['case']} | {','.join(a['modes'])} | {a['centre_px']['max']:.4f} / {a['centre_px']['mean']:.4f} | {a['corner_px']['max']:.4f} / "
                   f"{a['corner_px']['mean']:.4f} | {a['lin_2x2']['max']:.2e} | {a['true_motion_px']['max']} |")
    else:
        out.append(f"| {a['case']} | safe {a['safe_fraction']} | max abs {a['max_abs']} | mean abs {a['mean_abs']} | PSNR {a['psnr_db']} dB | worst frame {a['worst_frame']} |")
doc = (P / f"{TAG}_analytic_accuracy.md").read_text()
head, tail = doc[:doc.index("| case |")], doc[doc.index("Reading:"):]
(P / f"{TAG}_analytic_accuracy.md").write_text(head + "\n".join(out) + "\n" + tail)
print("profiles written")
