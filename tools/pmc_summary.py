#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per launch."""
import collections
import csv
import glob
import sys


def main():
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for pat in sys.argv[1:]:
        for fn in glob.glob(pat):
            with open(fn) as f:
                for r in csv.DictReader(f):
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in rows.items():
        if not k.startswith("dn::"):
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")


if __name__ == "__main__":
    main()
