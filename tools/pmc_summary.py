#!/usr/bin/env python3
"""rocprofv3 --pmc DBs (FETCH_SIZE and WRITE_SIZE in separate passes) -> per-kernel average RAW counter bytes per launch.

Units: both counters are in KiB (raw * 1024 = bytes as the counter sees them). No blanket correction is applied here: the
gfx950 read-side factor depends on the access pattern (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: x2 was calibrated
for wide coalesced streaming reads only), so the factor for each kernel comes from the known-bytes launches of
tools/gpu_calib.py profiled in the same way (k_gather_calib<REC, SEQ> rows; tools/make_roofline_inputs.py applies them)."""
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
        cw.writerow(["kernel", "launches", "FETCH_SIZE_KiB_raw_avg", "fetch_bytes_raw", "WRITE_SIZE_KiB_raw_avg", "write_bytes_raw"])
        for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1])):
            if not k.startswith("csh::"):
                continue
            fc, fa = f.get(k, (0, 0.0))
            wc, wa = w.get(k, (0, 0.0))
            cw.writerow([k[:110], fc or wc, round(fa, 1), int(fa * 1024), round(wa, 1), int(wa * 1024)])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
