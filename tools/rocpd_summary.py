#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace and/or PMC counters) as plain text.

    python tools/rocpd_summary.py gpurun_out/prof/trace_results.db [--filter cldn] > profiles/r01_....txt
"""
import argparse
import sqlite3
import statistics


def short_name(name: str) -> str:
    """Kernel name without its argument list. Names such as `cldn::(anonymous namespace)::k_viz_insert(...)` or
    `void cldn::k_x<(anonymous namespace)::T>(...)` contain parentheses of their own: cut at the LAST top-level '('."""
    name = name.replace("void ", "")
    if not name.endswith(")"):
        return name
    depth = 0
    for i in range(len(name) - 1, -1, -1):
        if name[i] == ")":
            depth += 1
        elif name[i] == "(":
            depth -= 1
            if depth == 0:
                return name[:i]
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--filter", default="", help="substring of the kernel name")
    args = ap.parse_args()
    con = sqlite3.connect(args.db)
    cur = con.cursor()

    rows = cur.execute("select name, duration, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, sgpr_count "
                       "from kernels").fetchall()
    by = {}
    for name, dur, gx, gy, wx, lds, vg, sg in rows:
        if args.filter and args.filter not in name:
            continue
        short = short_name(name)
        by.setdefault(short, []).append((dur, gx, gy, wx, lds, vg, sg))
    if by:
        total = sum(sum(d[0] for d in v) for v in by.values())
        print("== kernel trace (durations in us) ==")
        print(f"{'kernel':46s} {'calls':>6s} {'total':>11s} {'avg':>9s} {'min':>9s} {'max':>9s} {'%':>6s}  grid x wg, lds, vgpr, sgpr")
        for k, v in sorted(by.items(), key=lambda kv: -sum(d[0] for d in kv[1])):
            ds = [d[0] / 1e3 for d in v]
            g = v[-1]
            print(f"{k[-46:]:46s} {len(ds):6d} {sum(ds):11.1f} {statistics.mean(ds):9.2f} {min(ds):9.2f} {max(ds):9.2f} "
                  f"{100 * sum(ds) * 1e3 / total:6.2f}  {g[1]}x{g[2]} x {g[3]}, {g[4]}, {g[5]}, {g[6]}")

    try:
        pm = cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        acc = {}
        for name, cname, val, disp in pm:
            if args.filter and args.filter not in name:
                continue
            short = short_name(name)
            acc.setdefault(short, {}).setdefault(cname, {}).setdefault(disp, 0.0)
            acc[short][cname][disp] += val
        print("\n== PMC counters (sum over SEs/XCDs per dispatch, then mean over dispatches) ==")
        for k, cs in acc.items():
            print(k)
            for cname, d in sorted(cs.items()):
                vals = list(d.values())
                print(f"    {cname:28s} {statistics.mean(vals):18.1f}   (dispatches: {len(vals)})")


if __name__ == "__main__":
    main()
