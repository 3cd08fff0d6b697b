#!/usr/bin/env python3
"""rocprofv3 --pmc DBs (one counter per pass) -> per-kernel average HBM bytes per launch.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of
wide coalesced reads -> doubled; WRITE_SIZE taken as is; both counters are in KiB. Calibration visible in the same
run: __amd_rocclr_copyBuffer of 64 MiB reads FETCH_SIZE = 32 MiB, WRITE_SIZE = 64 MiB."""
import csv
import sqlite3
import sys


def per_kernel(db_path):
    db = sqlite3.connect(db_path)
    out = {}
    for name, cnt, avg in db.execute("select name, count(*), avg(counter_value) from pmc_events group by name"):
        short = name.split("(")[0].replace("void ", "")
        out[short] = (cnt, avg)
    return out


def main(fetch_db, write_db, out_csv, note=""):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    with open(out_csv, "w", newline="") as fh:
        if note:
            fh.write("# " + note + "\n")
        cw = csv.writer(fh)
        cw.writerow(["kernel", "launches", "FETCH_SIZE_KiB_raw_avg", "fetch_bytes_corrected_x2", "WRITE_SIZE_KiB_avg", "write_bytes", "hbm_bytes_per_launch"])
        for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] * 2 + w.get(k, (0, 0))[1])):
            if not k.startswith("csh::"):
                continue
            fc, fa = f.get(k, (0, 0.0))
            wc, wa = w.get(k, (0, 0.0))
            fb, wb = 2 * fa * 1024, wa * 1024
            cw.writerow([k[:90], fc or wc, round(fa, 1), int(fb), round(wa, 1), int(wb), int(fb + wb)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
