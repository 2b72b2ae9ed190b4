#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: MEDIAN counter value per kernel per launch (and the mean beside it).  The median, because
a run of a pipe holds partial launches -- the first launch of a group pipe carries front halves only, the flush chains only -- and a mean over a
short run dilutes the full launches with them (round 4's first pass read 24.7 MB / 2.17 M SALU per launch where a full launch moves 27 MB / 2.9 M).
    pmc_summary.py [--traffic-json out.json --kernel substring --frames-per-launch N] csv-glob ...
--traffic-json: also write profiles/pmc_traffic.json for that kernel from its FETCH_SIZE / WRITE_SIZE means (separate passes), corrected as
MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE doubled; both counters are in KB."""
import collections
import csv
import glob
import json
import sys


def main():
    args = sys.argv[1:]
    opt = {}
    while args and args[0].startswith("--"):
        opt[args[0]] = args[1]
        args = args[2:]
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for pat in args:
        for fn in glob.glob(pat):
            with open(fn) as f:
                for r in csv.DictReader(f):
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    means = {}
    for k, cs in rows.items():
        if not k.startswith("dn::"):
            continue
        print(k)
        means[k] = {}
        for c, v in sorted(cs.items()):
            w = sorted(v)
            med = w[len(w) // 2] if len(w) % 2 else 0.5 * (w[len(w) // 2 - 1] + w[len(w) // 2])
            means[k][c] = med
            print(f"   {c:28s} {med:16.1f}   (median of n={len(v)}; mean {sum(v) / len(v):.1f})")
    if not means:
        sys.exit("pmc_summary: no dn:: kernel in the counter files")
    if "--traffic-json" in opt:
        sub = opt["--kernel"]
        hit = [k for k in means if sub in k and "FETCH_SIZE" in means[k] and "WRITE_SIZE" in means[k]]
        if not hit:
            sys.exit(f"pmc_summary: no kernel matching {sub!r} with FETCH_SIZE and WRITE_SIZE")
        k = max(hit, key=lambda k: means[k]["FETCH_SIZE"] + means[k]["WRITE_SIZE"])
        fetch, write = means[k]["FETCH_SIZE"], means[k]["WRITE_SIZE"]
        frames = int(opt.get("--frames-per-launch", "256"))
        per_frame = 8872
        total = int((2 * fetch + write) * 1024)
        doc = {
            "_source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/prof_step.py (batch 256, the bench's schedule); MEDIAN per launch of {k} (a full launch: the run's first launch carries front halves only, its flush chains only); "
                       f"written by tools/collect_profiles.sh {opt.get('--tag', '')}",
            "_unit": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide streaming reads; "
                     "an upper bound for our 8-16 B/lane reads)",
            "kernel": k, "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
            "frames_per_launch": frames,
            opt.get("--key", "group_kernel_hbm_bytes_per_launch"): total,
            "algorithmic_bytes_per_launch": per_frame * frames,
            "ratio_to_algorithmic": round(total / (per_frame * frames), 2),
        }
        old = {}
        try:
            old = json.load(open(opt["--traffic-json"]))
        except Exception:          # noqa: BLE001
            pass
        for key in ("hop_kernel_hbm_bytes_per_launch", "frame_kernel_hbm_bytes_per_launch"):        # (other schedules' figures of earlier passes stay)
            if key in old and key not in doc:
                doc[key] = old[key]
        json.dump(doc, open(opt["--traffic-json"], "w"), indent=1)
        print(f"wrote {opt['--traffic-json']}: {total} bytes per launch = {doc['ratio_to_algorithmic']} x algorithmic")


if __name__ == "__main__":
    main()
