#!/usr/bin/env python
"""Condense rocprofv3 `--pmc` output (one *_counter_collection.csv per pass) into the small summaries that
are committed under profiles/ and that bench.py reads for `roofline.traffic` / `roofline.valu_issue`.

    python scripts/pmc_summary.py traffic OUT.csv  FETCH_DIR WRITE_DIR      (two separate passes, as the guide asks)
    python scripts/pmc_summary.py sq      OUT.csv  SQ_DIR

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB per dispatch, summed here over the counter's
instances (XCDs) and converted to bytes.  No x2 correction is applied: that correction holds for 16-B/lane
streaming reads, this kernel's accesses are 4-8 B per lane and memory-side atomics (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict


def rows_of(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    for f in files:
        with open(f) as fh:
            yield from csv.DictReader(fh)


def revision():
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
    except OSError:
        return ""


def per_dispatch(d):
    acc = defaultdict(float)
    meta = {}
    for r in rows_of(d):
        key = (int(r["Dispatch_Id"]), r["Counter_Name"])
        acc[key] += float(r["Counter_Value"])
        meta[int(r["Dispatch_Id"])] = r
    return acc, meta


def main():
    kind, out = sys.argv[1], sys.argv[2]
    rev = os.environ.get("SIGSVGD_REVISION") or revision()
    if kind == "traffic":
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["pass", "dispatch", "kernel", "counter", "bytes_per_dispatch", "revision"])
            for d in sys.argv[3:]:
                acc, meta = per_dispatch(d)
                for (disp, name), v in sorted(acc.items()):
                    if name in ("FETCH_SIZE", "WRITE_SIZE"):
                        w.writerow([os.path.basename(os.path.normpath(d)), disp, meta[disp]["Kernel_Name"], name,
                                    f"{v * 1024.0:.0f}", rev])
    elif kind == "sq":
        acc, meta = per_dispatch(sys.argv[3])
        names = sorted({n for (_, n) in acc})
        with open(out, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                        "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"] + names + ["revision"])
            for disp in sorted(meta):
                m = meta[disp]
                w.writerow([disp, m["Kernel_Name"], m["Grid_Size"], m["Workgroup_Size"], m["LDS_Block_Size"],
                            m["VGPR_Count"], m["Accum_VGPR_Count"], m["SGPR_Count"], m["Scratch_Size"]]
                           + [f"{acc[(disp, n)]:.0f}" for n in names] + [rev])
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
