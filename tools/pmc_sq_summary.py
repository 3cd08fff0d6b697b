#!/usr/bin/env python3
"""rocprofv3 --pmc <SQ counters> DB -> per-kernel averages per launch and the issue / stall split of the wave cycles.

Counter meaning per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAIT_ANY = wave parked (s_waitcnt / barrier),
SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and sum to ~SQ_WAVE_CYCLES."""
import csv
import sqlite3
import sys


def main(db_path, out_csv, note=""):
    db = sqlite3.connect(db_path)
    rows = {}
    for name, counter, cnt, avg in db.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name"):
        short = name.split("(")[0].replace("void ", "")
        if short.startswith("csh::"):
            rows.setdefault(short, {"launches": cnt})[counter] = avg
    counters = sorted({c for r in rows.values() for c in r if c != "launches"})
    with open(out_csv, "w", newline="") as fh:
        if note:
            fh.write("# " + note + "\n")
        cw = csv.writer(fh)
        cw.writerow(["kernel", "launches"] + counters + ["active_frac", "issue_stall_frac", "parked_frac", "valu_frac_of_active", "lds_conflict_frac"])
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            wc = r.get("SQ_WAVE_CYCLES", 0) or float("nan")
            act = r.get("SQ_ACTIVE_INST_ANY", float("nan"))
            f = lambda x: "" if x != x else round(x, 4)
            cw.writerow([k[:90], r["launches"]] + [round(r.get(c, float("nan")), 1) for c in counters] + [
                f(act / wc), f(r.get("SQ_WAIT_INST_ANY", float("nan")) / wc), f(r.get("SQ_WAIT_ANY", float("nan")) / wc),
                f(r.get("SQ_ACTIVE_INST_VALU", float("nan")) / act if act == act and act else float("nan")),
                f(r.get("SQ_LDS_BANK_CONFLICT", float("nan")) / r["SQ_LDS_IDX_ACTIVE"] if r.get("SQ_LDS_IDX_ACTIVE") else float("nan"))])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
