#!/usr/bin/env python3
"""Static ISA histogram of the staged sample loop of warp_blur_kernel (the straight-line inner loop that every interior
tile runs): compiles vstab_warp.hip to assembly, finds the loop that starts with the scalar loads of a sample's matrix and
ends at its back edge, and counts instructions by class.    python tools/isa_loop_histogram.py > profiles/…md"""
import collections, re, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
FLAGS = "-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt".split()
asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", "-", str(ROOT / "comfyui-video-stabilizer_amd/csrc/vstab_warp.hip")],
                     capture_output=True, text=True).stdout

CLASSES = [("f32 mul", r"^v_mul_f32"), ("f32 add/sub", r"^v_(add|sub|subrev)_f32"), ("f32 fma/mac", r"^v_(fma|fmac|mac)_f32"),
           ("f64 mul", r"^v_mul_f64"), ("f64 add", r"^v_add_f64"), ("f64 fma", r"^v_fma_f64"), ("convert", r"^v_cvt_"),
           ("int shift / and / or / bfe", r"^v_(lshl|lshr|ashr|and|or|xor|bfe|lshlrev|lshrrev|ashrrev|and_or|lshl_or|lshl_add|add_lshl)"),
           ("int add / mul / mad", r"^v_(add|sub|subrev|mul|mad|add3)_(u|i|co_|nc_)?(u32|i32|u24|i24|u64)|^v_mad_u32_u24|^v_mul_u32_u24|^v_add_u32|^v_sub_u32|^v_add3_u32"),
           ("compare / select", r"^v_(cmp|cndmask)"), ("move / readlane", r"^v_(mov|readfirstlane|readlane|accvgpr)"),
           ("LDS read", r"^ds_read"), ("LDS write", r"^ds_write"), ("vector memory", r"^(global|buffer|flat|scratch)_"),
           ("scalar load", r"^s_load"), ("scalar ALU", r"^s_(add|sub|mul|and|or|xor|lshl|lshr|cmp|cselect|mov|not|bfe|ashr|addc|mulk|movk|bitcmp)"),
           ("branch", r"^s_(cbranch|branch)"), ("wait / nop", r"^s_(waitcnt|nop)")]


def kernel_body(name_part):
    start = re.search(rf"^_ZN[^\n]*warp_blur_kernel{name_part}[^\n:]*:", asm, re.M)
    end = asm.index("s_endpgm", start.end())
    return asm[start.end():end].split("\n")


def staged_loop(lines):
    # the staged loop: a loop header whose first instructions are the s_load of the sample matrix and which contains ds_read_b128
    headers = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:.*Loop Header", l)]
    for h in headers:
        label = lines[h].split(":")[0]
        end = next((j for j in range(h + 1, len(lines)) if re.search(rf"s_cbranch_\w+ {re.escape(label)}\b", lines[j])), None)
        body = lines[h:end + 1] if end else []
        text = "\n".join(body)
        if end and "s_load_dwordx8" in "\n".join(body[:6]) and "ds_read_b128" in text and "global_load" not in text:
            return body
    raise SystemExit("staged loop not found")


print("# Static ISA histogram of the staged sample loop of `warp_blur_kernel` (one iteration = one sample, TWO pixels per thread)\n")
print("`tools/isa_loop_histogram.py` on the committed source; complements the EXECUTED histograms of `r03_blur_hist_*.md`\n(which include the general-loop tiles and the per-block prologue).\n")
for title, part in (("bicubic, with mask", "ILi1ELi0ELb1"), ("bilinear, with mask", "ILi0ELi0ELb1")):
    body = [l.strip() for l in staged_loop(kernel_body(part)) if l.strip() and not l.strip().startswith((";", "."))]
    hist = collections.Counter()
    for ins in body:
        for name, rx in CLASSES:
            if re.search(rx, ins):
                hist[name] += 1
                break
        else:
            hist["other: " + ins.split()[0]] += 1
    valu = sum(v for k, v in hist.items() if k.split()[0] in ("f32", "f64", "convert", "int", "compare", "move") or k.startswith("other: v_"))
    print(f"## {title}: {len(body)} instructions per iteration, {valu} of them VALU = {valu / 2:.1f} per pixel-sample\n")
    print("| class | per iteration (2 pixels) | per pixel-sample |\n|---|---|---|")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"| {k} | {v} | {v / 2:.1f} |")
    print()
