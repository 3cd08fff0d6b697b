#!/usr/bin/env python3
"""rocprofv3 results DB (rocpd sqlite) -> timeline of the LAST `count` kernel dispatches: start offset, duration, gap to the previous
kernel's end (us). Shows where a latency-bound sequence of launches spends its time (kernels vs gaps).
    python tools/prof_timeline.py results.db out.csv [count]"""
import csv
import sqlite3
import sys


def main(db_path, out_csv, count=24):
    db = sqlite3.connect(db_path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else None
    if view is None:
        raise SystemExit("no kernels view in %s: %s" % (db_path, tabs))
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(db.execute("select %s, start, end from %s order by start" % (name_col, view)))
    rows = rows[-count:]
    t0 = rows[0][1]
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "start_us", "dur_us", "gap_before_us"])
        prev_end = None
        for name, s, e in rows:
            short = name.split("(")[0].replace("void ", "")[:90]
            w.writerow([short, round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), "" if prev_end is None else round((s - prev_end) / 1e3, 1)])
            prev_end = e


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 24)
