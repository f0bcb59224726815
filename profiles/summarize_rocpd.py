#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output into the small text summaries kept under profiles/.

    python profiles/summarize_rocpd.py stats  <results.db> [kernel first count]   -> CSV on stdout (per-kernel calls/total/avg;
                                                   with a window: also the average over dispatches first .. first+count-1 of that
                                                   kernel, in launch order -- the launches bench.py times after its fast-forward)
    python profiles/summarize_rocpd.py pmc    <results.db> [...]      -> JSON on stdout (per kernel x grid: mean counter values)

rocprofv3 in this image writes a database (views `top_kernels`, `kernels`, `counters_collection`) rather than the CSV
files of older releases; gpurun_out/ is scratch, so the judged numbers are extracted with this script and committed.
"""
import json
import sqlite3
import sys


def stats(path, window=None):
    con = sqlite3.connect(path)
    print("name,calls,total_us,average_us,min_us,max_us,percent")
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, calls, total, avg, mn, mx in rows:
        print(f"\"{name}\",{calls},{total / 1e3:.3f},{avg / 1e3:.3f},{mn / 1e3:.3f},{mx / 1e3:.3f},{100.0 * total / tot:.2f}")
    if window:
        name, first, count = window[0], int(window[1]), int(window[2])
        d = [r[0] for r in con.execute("select duration from kernels where name like ? order by start", (f"%{name}%",))]
        w = d[first:first + count]
        if w:
            print(f"# window: kernel *{name}*, dispatches {first}..{first + len(w) - 1} of {len(d)} in launch order (0-based): "
                  f"average_us {sum(w) / len(w) / 1e3:.3f}, min_us {min(w) / 1e3:.3f}, max_us {max(w) / 1e3:.3f}")
    # per-grid breakdown (the dense sweep runs at two swarm sizes in the default bench)
    print("# per (kernel, grid) breakdown: name,grid,workgroup,calls,average_us,lds_bytes,scratch_bytes,vgpr,sgpr")
    try:
        for r in con.execute(
                "select name, grid_size, workgroup_size, count(*), avg(duration), max(lds_size), max(scratch_size), "
                "max(vgpr_count), max(sgpr_count) from kernels group by name, grid_size order by name, grid_size"):
            print("# \"%s\",%d,%d,%d,%.3f,%s,%s,%s,%s" % (r[0], r[1], r[2], r[3], r[4] / 1e3, r[5], r[6], r[7], r[8]))
    except sqlite3.OperationalError as e:  # column names differ between rocprofv3 builds
        print("# (per-grid view unavailable: %s)" % e)


def pmc(paths):
    out = {}
    for path in paths:
        con = sqlite3.connect(path)
        q = ("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by kernel_name, grid_size, counter_name")
        for name, grid, ctr, n, val, dur in con.execute(q):
            key = f"{name.split('(')[0].split('::')[-1].split('<')[0]}@grid{grid}"
            ent = out.setdefault(key, {"dispatches": n})
            ent[ctr] = round(val, 3)
            ent.setdefault("avg_duration_us_under_pmc", round(dur / 1e3, 3))
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("stats", "pmc"):
        sys.exit(__doc__)
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3:6] if len(sys.argv) >= 6 else None)
    else:
        pmc(sys.argv[2:])
