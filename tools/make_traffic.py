#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the --pmc passes of tools/profile_round.sh (gpurun_out/prof_<tag>_pmc_fetch.txt,
..._pmc_write.txt): HBM bytes per launch of the encode kernels and of the decode kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE as is;
both counters in KiB). usage: make_traffic.py <tag> [src_dir] [dst_dir]"""
import json, os, re, sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")


def counters(path, counter):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(cldn::\S.*?)\s*$", line)
        if m and "  " not in m.group(1).strip():
            cur = m.group(1).strip()
            continue
        m = re.match(r"^\s+%s\s+([0-9.]+)\s+\(dispatches: (\d+)\)" % counter, line)
        if m and cur:
            out[cur] = (float(m.group(1)), int(m.group(2)))
    return out


fetch = counters(os.path.join(src, f"prof_{tag}_pmc_fetch.txt"), "FETCH_SIZE")
write = counters(os.path.join(src, f"prof_{tag}_pmc_write.txt"), "WRITE_SIZE")
bench = json.loads(open(os.path.join(src, f"prof_{tag}_pmc_fetch_bench.json")).read())


def hbm(kernel_prefix):
    k = [n for n in fetch if n.startswith("cldn::" + kernel_prefix)]
    k.sort(key=lambda n: -fetch[n][1])  # the variant with the most dispatches is the timed one
    n = k[0]
    f, w = fetch[n][0], write[n][0]
    return n.replace("cldn::", ""), {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "dispatches": fetch[n][1],
                                     "hbm_bytes": (2.0 * f + w) * 1024.0}


piece_name, piece = hbm("k_encode_fused")
fin_name, fin = hbm("k_finish")
pal_name, pal = hbm("k_section_palette32")
dec_name, dec = hbm("k_decode_points")
pts = bench["config"]["clouds_per_gpu"] * bench["config"]["points_per_cloud"]
alg_step = pts * bench["config"]["point_step"] + bench["job_stage1_bytes"]
doc = {
    "kernel": piece_name, "workload": "c2", "clouds_per_gpu": bench["config"]["clouds_per_gpu"],
    "points_per_cloud": bench["config"]["points_per_cloud"],
    "source": f"profiles/{tag}_pmc_fetch.txt (FETCH_SIZE) and profiles/{tag}_pmc_write.txt (WRITE_SIZE): separate --pmc passes of "
              "tools/profile_round.sh, mean per dispatch",
    "FETCH_SIZE_KB": piece["FETCH_SIZE_KB"], "WRITE_SIZE_KB": piece["WRITE_SIZE_KB"],
    "corrections": "gfx950: FETCH_SIZE counts the 128-byte requests of wide coalesced reads as 64 bytes -> x2 "
                   "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE as is; KiB -> bytes x1024",
    "hbm_bytes_per_launch": piece["hbm_bytes"],
    "whole_step_framed": {piece_name: piece, fin_name: fin,
                          "hbm_bytes_per_step": piece["hbm_bytes"] + fin["hbm_bytes"],
                          "algorithmic_bytes_per_step": alg_step,
                          "ratio": round((piece["hbm_bytes"] + fin["hbm_bytes"]) / alg_step, 3)},
    "whole_step_chunk_table": {piece_name: piece, pal_name: pal,
                               "hbm_bytes_per_step": piece["hbm_bytes"] + pal["hbm_bytes"],
                               "algorithmic_bytes_per_step": alg_step,
                               "ratio": round((piece["hbm_bytes"] + pal["hbm_bytes"]) / alg_step, 3)},
    "decode": {"kernel": dec_name, **dec, "output_bytes": pts * bench["config"]["point_step"],
               "write_over_output": round(dec["WRITE_SIZE_KB"] * 1024.0 / (pts * bench["config"]["point_step"]), 3)},
}
json.dump(doc, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
