"""Merge the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_workload.sh into
profiles/<round>_pmc_other_workloads.json (KB per dispatch, mean over dispatches, per hiprec kernel).

    python tools/pmc_other.py r01 pgmf t2v ngcf      # reads gpurun_out/pmc_<workload>_{FETCH,WRITE}_SIZE
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, workloads = sys.argv[1], sys.argv[2:]
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_other_workloads.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for w in workloads:
        acc = defaultdict(lambda: defaultdict(list))
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{w}_{counter}", "*counter_collection.csv")):
                for row in csv.DictReader(open(f, newline="")):
                    if "hiprec::" in row["Kernel_Name"]:
                        short = row["Kernel_Name"].split("(")[0].replace("void ", "")
                        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not acc:
            print("no counter files for", w)
            continue
        out[w] = {k: {"FETCH_SIZE": round(sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1), 1),
                      "WRITE_SIZE": round(sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1), 1),
                      "n": len(c["FETCH_SIZE"])} for k, c in acc.items()}
    json.dump(out, open(path, "w"), indent=1)
    print(path, {w: {k: v for k, v in out[w].items() if v["n"] > 3} for w in workloads if w in out})


if __name__ == "__main__":
    main()
