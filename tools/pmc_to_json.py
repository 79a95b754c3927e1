#!/usr/bin/env python
"""profiles/<tag>_fetch_pmc.md + <tag>_write_pmc.md (tools/rocpd_pmc_summary.py tables) -> profiles/pmc_traffic.json.
    python tools/pmc_to_json.py FETCH.md WRITE.md <submissions | auto> [commit stamp] [attention launches per submission] [frames per submission]
Kernel families: gemm = every gemm_*kernel, the direct 3 x 3 convolution + splitk epilogues (the launches behind cofi_gemm_f32* / cofi_conv2d_nhwc),
attention, kpconv_aggregate, neighbor_maxpool, group_norm_apply.
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KB and on gfx950 FETCH_SIZE counts
128-B requests at 64 B for wide coalesced reads (MI355X_MICROARCH.md §HBM)."""
import json
import re
import sys

FAMILIES = {"gemm": ("gemm_kernel", "gemm_bf16x3_kernel", "gemm_planes_kernel", "gemm_x6_big_kernel", "gemm_f16_big_kernel", "conv3x3_direct_kernel", "splitk_epilogue"),
            "attention": ("attention_flat_kernel", "attention_x6_kernel", "attention_fwd_kernel", "attention_kv_planes_kernel"),
            "kpconv_aggregate": ("kpconv_aggregate",), "neighbor_maxpool": ("neighbor_maxpool_kernel",),
            "group_norm_apply": ("group_norm_apply",), "loftr_tail": ("loftr_tail_kernel",)}


def aux(name):
    """launches that belong to a family's bytes but are not one of its `main` launches: split-K folds, the f16x3 kernel's repair launch
    (gemm_f16_big_kernel<ANORM, CONV, ROBUST = true, ...>: exits at once unless a tile left the fp16 window), the K / V split in front of the
    attention kernel"""
    return ("splitk" in name or "attention_kv_planes_kernel" in name
            or re.search(r"gemm_f16_big_kernel<[^,]*,[^,]*,\s*(\(bool\))?\s*(1|true)\b", name) is not None)


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\| `(.*?)` \| (\w+) \| (\d+) \| ([\d.e+\-]+) \| ([\d.e+\-]+) \|", line)
        if m and m.group(2) == counter:
            out[m.group(1)] = (int(m.group(3)), float(m.group(4)))
    return out


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    if sys.argv[3] == "auto":   # 12 attention launches per submission (4 self layers x 1 joint or 2, 4 cross x 2: transformer.py)
        per = int(sys.argv[5]) if len(sys.argv) > 5 else 12
        frames = sum(v[0] for k, v in fetch.items() if "attention_flat_kernel" in k or "attention_x6_kernel" in k) // per
    else:
        frames = int(sys.argv[3])
    fps = int(sys.argv[6]) if len(sys.argv) > 6 else 1   # stack mode: frames per submission
    frames *= fps
    res = {"collected_on": sys.argv[4] if len(sys.argv) > 4 else "unknown", "frames_per_submission": fps}
    for fam, pats in FAMILIES.items():
        f = sum(v[1] for k, v in fetch.items() if any(p in k for p in pats))
        w = sum(v[1] for k, v in write.items() if any(p in k for p in pats))
        n_main = sum(v[0] for k, v in fetch.items() if any(p in k for p in pats) and not aux(k))
        if n_main == 0:
            continue
        res[fam] = {"fetch_kb_total": f, "write_kb_total": w, "main_launches": n_main, "frames": frames,
                    "traffic_bytes_per_launch": (2 * f + w) * 1024 / n_main, "traffic_bytes_per_frame": (2 * f + w) * 1024 / frames}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
