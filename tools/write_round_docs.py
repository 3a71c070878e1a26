#!/usr/bin/env python3
"""Turns the files `tools/final_profiles.sh <tag>` left under gpurun_out/ into the round's measured tables:
profiles/<tag>_bench_c2.{log,md}, _bench_c2_kernel_stats.csv, _bench_chains.md (C3 / C5 as whole steps), _c4_strong_scaling.md,
_c4_single_gpu.json, _analytic_accuracy.md.   python tools/write_round_docs.py r04
(The prose of a round -- what was changed and why -- lives in the hand-written profiles/<tag>_*.md notes and DESIGN.md.)"""
import csv, json, re, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
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
TREES = (G / f"{TAG}_trees_ab.txt").read_text().strip() if (G / f"{TAG}_trees_ab.txt").exists() else "(not collected)"
tj = json.loads((P / "warp_traffic.json").read_text())
TRAFFIC = f"{tj['hbm_bytes_per_launch'] / 1e9:.4f} GB = {tj['hbm_bytes_per_launch'] / tj['algorithmic_bytes_per_launch']:.4f} of algorithmic"
acc, par, ma = j["accuracy"]["hip"], j["parity_at_size"], j["motion_apply"]
c3, c5m = ma["c3_1080p_bicubic_blur0.5_S17"], ma["c5_4k_expand_bilinear_blur0.5_S33"]
st, cb, rf = j["config"]["stage_ms"], j["cpu_baseline"], j["roofline"]
(P / f"{TAG}_bench_c2.md").write_text(f"""# C2 bench line and kernel times, {TAG} final build

Command (one MI355X box, `tools/final_profiles.sh {TAG}`): `python bench.py --steps 20 --warmup 5` -> `profiles/{TAG}_bench_c2.log`;
`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-checks
--cpu-frames 0` -> `profiles/{TAG}_bench_c2_kernel_stats.csv` ({steps} steps incl. warm-up).

**{j["value"]:.0f} frames/s, {j["ms_per_step"]} ms per 256-frame 1080p clip** (round 4's driver line: 34 499 / 7.421 ms; the boxes of the pool
differ by ~5-8 % in HBM rate, so rounds are compared on ONE box: `{TAG}_trees_ab.txt` below).  Stage times: warp {st["warp"]} ms from HIP events
inside the timed steps; gray {st["gray"]} / DIS {st["dis"]} / fit {st["fit"]} ms from three extra passes after them (an event pair costs the stream
~10 us, so the timed steps keep the warp's only); DIS by stage (one more pass, `vstab_set_timing(ctx, 2)`): {j["config"]["dis_ms"]}.
Device plan in the timed steps: {j["config"]["device_plan"]}.  Roofline of the warp kernel: 14.864 GB algorithmic / {rf["launch_ms"]} ms =
{rf["achieved"]} GB/s = **{rf["frac"]} of 8 TB/s** on this box; PMC traffic per launch `profiles/warp_traffic.json` (FETCH_SIZE x 2 + WRITE_SIZE,
re-collected on this round's final warp sources, hash-tied): {TRAFFIC}.

Round-4 tree and this tree on the same box, interleaved (`bench.py --steps 20 --warmup 5 --no-extras --no-checks --cpu-frames 0`; value, ms per
step, stage ms):

```
{TREES}
```

Self-verification on the same line (`accuracy`, `parity_at_size`, `batch_invariance`; outside the timed loop):

* analytic accuracy of the 255 reported transitions, px at working resolution: centre max {acc["centre_px"]["max"]} / mean
  {acc["centre_px"]["mean"]} / p99 {acc["centre_px"]["p99"]}; worst corner max {acc["corner_px"]["max"]}; 2x2 part max {acc["lin_2x2"]["max"]:.2e} -- identical
  for the CPU port (it produces the same bits);
* HIP run vs the oracle's run of the same 256 frames: `bit_equal: {str(par["bit_equal"]).lower()}` -- transition matrices equal, confidences equal,
  residuals within {par["residuals_max_rel_diff"]:.1e} (fp64 sum order), final matrices equal, **{par["pixels_differing"]} of 1 592 524 800 output values and
  {par["mask_pixels_differing"]} of 530 841 600 mask values differ**, padding statistics equal (the run used the device plan: the host plan's
  matrices agreed with it on every frame);
* batch invariance: pairs {{0, 127, 254}} as 2-frame clips (split DIS form) and frames {{0, 127, 255}} warped alone equal the
  whole-clip run (`fit_records_equal` {j["batch_invariance"]["fit_records_equal"]}, `frames_equal` {j["batch_invariance"]["frames_equal"]}).

`cpu_baseline`: kind "{cb["kind"]}", cv2 "{cb["cv2"]}" (no OpenCV on the box: the run-time tier decision of `bench.cv2_leg` printed it),
{cb["value"]:.0f} frames/s on {cb["cores"]} threads of {cb["cpu_model"]} ({cb["os_cpu_count"]} logical CPUs; stage seconds {cb["stage_s"]}) -- ~{j["value"] / cb["value"]:.0f}x, a
baseline, not a target; on ALL {cb.get("all_cores", {}).get("cores")} CPUs this process may use: {cb.get("all_cores", {}).get("value", 0):.0f} frames/s (stage seconds
{cb.get("all_cores", {}).get("stage_s")}) -- slower than on 16: the port's OpenMP loops are memory-bound and its fit stage is serial.  `host_roundtrip` {j["host_roundtrip"]["ms"]} ms ({j["host_roundtrip"]["frames_per_s"]} frames/s, CPU tensor in -> CPU tensors out).
`motion_apply` (shake-generator motion, affine): C3 kind {c3["ms_per_pass"]} ms per 256x1080p ({c3["frames_per_s"]} frames/s), C5 share {c5m["ms_per_pass"]} ms
per 64x4K ({c5m["frames_per_s"]} frames/s); frame 1 of each against the oracle: {c3.get("oracle_spot_check")} / {c5m.get("oracle_spot_check")}.

## Kernel times (rocprofv3 --stats, library kernels only)

{ks}
DIS = `pis4_kernel` {per_clip.get("pis4_kernel", 0) / 1e3:.2f} + `level_kernel` {per_clip.get("level_kernel", 0) / 1e3:.2f} ms + preparation (pyramid, padding, Sobel, tensors; partly on a
second stream); round 4: 1.39 + 1.81, round 3: 1.42 + 2.07, round 2: 1.46 + 2.31.
""")

c5p, c5d, c3p = line(f"{TAG}_bench_c5_plain.log"), line(f"{TAG}_bench_c5_dist1.log"), line(f"{TAG}_bench_c3_plain.log")
sp, sd, hd, s3 = c5p["config"]["rank0_stage_ms"], c5d["config"]["rank0_stage_ms"], c5d["config"]["rank0_host_ms"], c3p["config"]["rank0_stage_ms"]
shape = c5p["config"]["out_shape_rank0"]
(P / f"{TAG}_bench_chains.md").write_text(f"""# BASELINE configs[2] (C3) and configs[4] (C5) as whole bench steps, one GPU ({TAG})

`bench.py --workload c3 | c5`: one step = Flow over the clip, then Motion Apply on the ORIGINAL frames with the returned meta
(`bench.run_c5`).  Same box, `tools/final_profiles.sh {TAG}`.  C3 = 256 x 1080p, Flow perspective + crop_and_pad -> Motion Apply
(crop_and_pad, bicubic, 0.5, High = 17 samples); C5 = one GPU's share of the 8-GPU config, 64 x 4K, Flow similarity + expand ->
Motion Apply (expand, bilinear, 0.5, Ultra = 33 samples).

| run | frames/s | ms per step | gray | DIS | fit | warp (Flow's own output) | blur warp | host |
|---|---|---|---|---|---|---|---|---|
| `--workload c3` | {c3p["value"]} | {c3p["ms_per_step"]} | {s3["gray"]} | {s3["dis"]} | {s3["fit"]} | {s3["warp"]} | {s3["warp_blur"]} | -- |
| (round 4's `--workload c3`, other box) | 5469 | 46.81 | 0.99 | 3.19 | 1.52 | 2.60 | 37.1 | -- |
| (round 4's `--workload c5`, other box) | 2318 | 27.61 | 0.93 | 1.87 | 0.11 | 2.90 | 21.0 | -- |
| `--workload c5` (single process) | {c5p["value"]} | {c5p["ms_per_step"]} | {sp["gray"]} | {sp["dis"]} | {sp["fit"]} | {sp["warp"]} | {sp["warp_blur"]} | -- |
| `--workload c5 --gpus 1 --force-dist` (sharded code path inside a world-1 RCCL group) | {c5d["value"]} | {c5d["ms_per_step"]} | {sd["gray"]} | {sd["dis"]} | {sd["fit"]} | {sd["warp"]} | {sd["warp_blur"]} | gather_fits {hd["gather_fits"]}, plan {hd["plan"]}, meta {hd["meta"]} ms |

C5's output canvas: {shape[2]}x{shape[1]} (expand).  Both chains are the blur warp.  Round 5 on C3: the perspective fit (`fit` column) fell from
1.52 to ~1.1 ms (`profiles/r05_homography.md`), and tiles whose staged window lies inside the source take the interior loop for perspective
samples too (blur warp 37.1 -> ~36 ms; the per-pixel fp64 reciprocal of the denominator is the contract and stays).  Parity at these sizes:
`tests/test_configs_gpu.py` (C3: all 255 pairs + blurred frames {{0, 1, 127, 254, 255}}; C5: all 63 pairs, warped frames {{0, 31, 63}}, blurred
frames {{0, 1, 31, 62, 63}}, bit-exact against the oracle).

Device plan of the chains' Flow halves (the plan between the fits and the warp formed by `plan_kernel`, verified by the host): C3
(perspective) {c3p["config"].get("rank0_device_plan")}, C5 (similarity, expand) {c5p["config"].get("rank0_device_plan")}.  C3 with the plan on the
host / on the device, one box, alternating (`tools/r05_c3ab.sh`): 47.42 / 46.85 / 47.02 / 47.02 ms per step.

Where the GPU idles inside one step of a chain (`tools/r05_chain_timeline.sh`, kernel trace): between the Flow half's warp and Motion Apply's
first launch, while the host finishes Flow's meta and parses it again for Motion Apply.  With device-resident frames Motion Apply now launches its
range pass BEFORE it parses and validates the meta (host frames keep validation first: a bad meta must not cost an upload): idle per step C3 519 ->
237 us (the gap in front of the range pass 420 -> 149 us), C5 372 -> 294 us (185 -> 119 us; its other gap, 73-77 us, is the wait for the plan's
region that sizes the expand canvas).
""")

c4, d1 = line(f"{TAG}_c4_single_gpu.log"), line(f"{TAG}_c4_dist1_128.log")
em = (G / f"{TAG}_emulate_c4.log").read_text().strip()
dev = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"world (\d+) \(device plan\): ([0-9.]+) ms/step", em)}
hostp = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"world (\d+): ([0-9.]+) ms/step", em)}
one = c4["ms_per_step"]
json.dump({"command": "python bench.py --gpus 1 --total-frames 1024 --steps 10 --warmup 3 --no-extras --no-checks --cpu-frames 0",
           "total_frames": 1024, "size": [1920, 1080], "frames_per_s": c4["value"], "ms_per_step": one, "stage_ms": c4["config"]["stage_ms"],
           "note": "the C4 clip (BASELINE configs[3]) on ONE MI355X, device-resident: the denominator of the strong-scaling ratio",
           "source": f"{TAG} final build, tools/final_profiles.sh {TAG}"}, open(P / f"{TAG}_c4_single_gpu.json", "w"), indent=1)
table = "".join(f"| {w} | {1024 // w} | {hostp[w]:.2f} | {dev[w]:.2f} | {one / (dev[w] + 0.4):.2f}-{one / (dev[w] + 0.2):.2f} |\n" for w in (2, 4, 8))
cs, ds, dh = c4["config"]["stage_ms"], d1["config"]["stage_ms"], d1["config"]["rank0_host_ms"]
(P / f"{TAG}_c4_strong_scaling.md").write_text(f"""# C4 (1024 x 1080p, Flow similarity): the one-GPU measurements behind the multi-GPU estimate ({TAG})

No multi-GPU box was available to the build in any round (the driver's SCALE run was skipped in rounds 1-3 for the same
reason); this is an ESTIMATE from one box (`tools/final_profiles.sh {TAG}`), not a scaling curve.

| measurement | ms per step |
|---|---|
| the whole clip on one GPU (`bench.py --gpus 1 --total-frames 1024`) | **{one}** ({c4["value"]:.0f} frames/s; gray {cs["gray"]}, DIS {cs["dis"]}, fit {cs["fit"]}, warp {cs["warp"]}) -- round 4: 28.2, round 3: 29.5-30.3 |
| one rank's share through the sharded code inside a world-1 RCCL group, device plan (`--gpus 1 --force-dist --total-frames 128`) | {d1["ms_per_step"]} (gray {ds["gray"]}, DIS {ds["dis"]} split form, fit {ds["fit"]}, warp {ds["warp"]}; host: {dh}) -- round 4: 4.61, round 3: 4.9-5.0 |
| rank 0's step of an N-rank run with the gathered tables of N ranks, no collective (`tools/emulate_world.py`) | see below |

```
{em}
```

| ranks | frames per rank | rank 0's step, host plan (rounds 1-3 flow), ms | device plan (rounds 4-5), ms | + 0.2-0.4 ms RCCL -> speed-up vs {one} ms |
|---|---|---|---|---|
{table}
Still **short of the >= 6x target at 8 GPUs** (needs <= {one / 6 - 0.3:.2f} ms per rank incl. RCCL); round 4's estimate was 5.34-5.54x (4.89 ms per
rank against 28.2 ms).  What moved the per-rank step this round is what sat BETWEEN its kernels -- the copies and events on the stream, the
fill before the warp, the final sampling launch, the counts' way to the host (`profiles/r05_dis_small_steps.md`): per step those cost the same
at 128 frames as at 256, so they weigh twice as much in a rank's step.  What keeps it from {one / 8:.2f} ms (= {one} / 8) is unchanged: the
coarse-to-fine chain's per-level cost at 127 pairs (patch search 1.04 ms, refinement 1.07 ms in the split form) and the replicated plan + meta
on rank 0.  Weak scaling (256 frames per GPU, `--frames 256`) has neither problem.
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
(P / f"{TAG}_analytic_accuracy.md").write_text(
    f"# Oracle-independent accuracy on analytic clips ({TAG} final build; `tools/analytic_accuracy.py --frames 48`)\n\n"
    "Same cases and reading as `profiles/r03_analytic_accuracy.md` (the estimation arithmetic did not change this round: the numbers\n"
    "are the same bits).\n\n" + "\n".join(out) + "\n")
print("profiles written")
