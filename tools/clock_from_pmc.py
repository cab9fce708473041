#!/usr/bin/env python
"""The effective shader clock under a kernel: GRBM_GUI_ACTIVE (cycles the GPU was busy) / the dispatch's duration, per kernel of a
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE pass (MI355X_MICROARCH.md, "DVFS give-back").  Durations from the counter file's own
Start/End timestamps where it has them, else from the kernel trace of the same pass (joined on the dispatch id).  rocprofv3 reports the counter
SUMMED over the device's 8 XCDs (each has its own GRBM): the clock is cycles / 8 / duration.  Only long dispatches give a meaningful figure:
the counter window is wider than the dispatch's timestamps by a few microseconds.
  python tools/clock_from_pmc.py <dir of the pass> [XCDs = 8]"""
import collections
import csv
import glob
import os
import sys


def main(d, xcds=8):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            try:
                dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    per = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for f in cc:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
                continue
            ns = None
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                try:
                    ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except ValueError:
                    ns = None
            if ns is None:
                ns = dur.get(r.get("Dispatch_Id"))
            if not ns or ns <= 0:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:96]
            per[k][0] += float(r["Counter_Value"]); per[k][1] += ns; per[k][2] += 1
    for k in sorted(per, key=lambda k: -per[k][1]):
        cyc, ns, n = per[k]
        print("%-96s GRBM_GUI_ACTIVE     launches=%5d cycles=%16.0f duration_ns=%16.0f effective_clock_MHz=%8.1f  (cycles summed over %d XCDs)" % (k, n, cyc, ns, cyc / xcds / ns * 1e3, xcds))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
