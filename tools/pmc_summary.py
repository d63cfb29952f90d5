"""Fold rocprofv3 --pmc passes into profiles/<round>_pmc_summary.json.

    python tools/pmc_summary.py r02 gpurun_out/pmc_fetch gpurun_out/pmc_write [more dirs...]

Each directory holds one `*_counter_collection.csv` (one counter per pass, as gpurun requires); the
summary keeps, per kernel of the hiprec namespace, the mean counter value per dispatch (KB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "*counter_collection.csv")):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if "hiprec::" not in name:
                        continue
                    short = name.split("(")[0].replace("void ", "")
                    acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for kernel, counters in acc.items():
        out[kernel] = {}
        for cname, vals in counters.items():
            out[kernel][f"{cname}_KB_mean"] = round(sum(vals) / len(vals), 2)
            out[kernel][f"{cname}_n"] = len(vals)
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python bench.py "
                    "--steps 200 --warmup 20 --no-cpu-baseline [--optimizer ...]`; KB per dispatch, mean over "
                    "dispatches.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide "
                    "(16 B/lane) coalesced stream; the fused kernel mixes 4-B-per-lane row gathers with a "
                    "16-B-per-lane sweep, so its FETCH_SIZE is left uncorrected (a lower bound).")
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, {k: v for k, v in out.items() if "fused" in k or "grad" in k})


if __name__ == "__main__":
    main()
