#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs (one pass per counter group)."""
import collections
import csv
import sys


def main(paths):
    for p in paths:
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("# " + p)
        for k in sorted(d):
            for c in sorted(d[k]):
                v = d[k][c]
                print("%-96s %-22s launches=%5d mean_per_launch=%16.1f total=%18.1f" % (k.replace("(anonymous namespace)::", "")[:96], c, len(v), sum(v) / len(v), sum(v)))
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
