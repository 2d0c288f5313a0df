#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per (kernel, counter) over the dispatches of each kernel.

usage: pmc_summary.py <dir with *_counter_collection.csv> [substring filter ...]
Prints one row per kernel with its dispatch count, mean duration and the mean of every collected counter.
"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("gdrf::", "")[:70]


def main():
    root = sys.argv[1]
    filt = sys.argv[2:]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no *_counter_collection.csv under " + root)
    val = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    seen = set()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if filt and not any(s in k for s in filt):
                    continue
                val[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                key = (f, row["Dispatch_Id"])
                if key not in seen and "Start_Timestamp" in row:
                    seen.add(key)
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    counters = sorted({c for k in val for c in val[k]})
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "dispatches", "mean_ms"] + counters)
    for k in sorted(val, key=lambda k: -sum(dur[k])):
        n = max(len(v) for v in val[k].values())
        ms = sum(dur[k]) / max(1, len(dur[k]))
        out.writerow([k, str(n), "%.4f" % ms] + ["%.6g" % (sum(val[k][c]) / len(val[k][c])) if val[k][c] else "" for c in counters])


if __name__ == "__main__":
    main()
