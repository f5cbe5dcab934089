#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of ROCm 7.x `rocprofv3 --kernel-trace`)
into a small text table: per kernel name -> calls, total / mean / min / max duration in microseconds,
optionally with PMC counter sums per dispatch.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--match sqllm] [--top 15]
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)  # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default=None)
    ap.add_argument("--top", type=int, default=15)
    ap.add_argument("--by-grid", action="store_true", help="split each kernel by grid size")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    key = "name, grid_x" if args.by_grid else "name"
    q = (f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
         f"group by {key} order by sum(duration) desc")
    rows = list(cur.execute(q))
    tot = sum(r[-4] for r in rows) or 1
    print(f"{'kernel':<92} {'grid':>7} {'calls':>7} {'total_us':>11} {'mean_us':>9} {'min_us':>8} {'max_us':>8} {'%':>6}")
    n = 0
    for r in rows:
        if args.by_grid:
            name, grid, calls, total, mean, mn, mx = r
        else:
            (name, calls, total, mean, mn, mx), grid = r, ""
        if args.match and args.match not in name:
            continue
        print(f"{short(name):<92} {str(grid):>7} {calls:>7} {total / 1e3:>11.1f} {mean / 1e3:>9.3f} {mn / 1e3:>8.3f} {mx / 1e3:>8.3f} {100 * total / tot:>6.2f}")
        n += 1
        if n >= args.top:
            break
    # PMC counters, if the run collected any (view pmc_events: one row per dispatch x counter)
    try:
        key2 = "k.name, k.grid_x, p.counter_name" if args.by_grid else "k.name, p.counter_name"
        pm = list(cur.execute(
            f"select {key2}, count(*), sum(p.counter_value), avg(k.duration) from pmc_events p "
            f"join kernels k on p.dispatch_id = k.dispatch_id group by {key2} order by k.name"))
    except sqlite3.Error as exc:
        pm = []
        print("no PMC data:", exc)
    if pm:
        print("\nPMC counters per dispatch (mean over dispatches):")
        for row in pm:
            if args.by_grid:
                name, grid, cname, cnt, val, dur = row
            else:
                (name, cname, cnt, val, dur), grid = row, ""
            if args.match and args.match not in name:
                continue
            print(f"  {short(name):<60} grid={str(grid):<8} {cname:<22} n={cnt:<6} mean={val / cnt:>16,.1f}  mean_dur_us={dur / 1e3:.3f}")


if __name__ == "__main__":
    main()
