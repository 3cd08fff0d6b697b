#!/usr/bin/env python3
"""rocprofv3 results DB (rocpd sqlite) -> compact per-kernel stats CSV for profiles/."""
import csv
import sqlite3
import sys


def main(db_path, out_csv, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        if note:
            f.write("# " + note + "\n")
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            if len(short) > 110:
                short = short[:107] + "..."
            w.writerow([short, calls, round(tot, 1), round(avg, 1), round(pct, 2)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
