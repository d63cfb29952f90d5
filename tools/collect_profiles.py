"""Copy what tools/refresh_profiles.sh wrote under gpurun_out/<tag>/ into profiles/ (tracked):
bench JSON lines, rocprofv3 kernel stats (top rows), PMC summaries.   python tools/collect_profiles.py [r02]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
    if lines:
        with open(os.path.join(dst, f"{tag}_{os.path.basename(f)}"), "w") as o:
            o.write(lines[-1] + "\n")
for d in sorted(glob.glob(os.path.join(src, "prof_*"))):
    if os.path.isdir(d):
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            rows = open(f).read().splitlines()[:16]
            with open(os.path.join(dst, f"{tag}_kernel_stats_{os.path.basename(d)[5:]}.csv"), "w") as o:
                o.write("\n".join(rows) + "\n")
pmc = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[4:].rsplit("_", 2)[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "hiprec::" in row["Kernel_Name"]:
                # (kernels of an anonymous namespace: "hiprec::(anonymous namespace)::name<..>(args)")
                short = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                pmc[name][short][row["Counter_Name"]].append(float(row["Counter_Value"]))
# groups that were not re-measured in this pass keep what is committed: start from the existing summaries
def _existing(name):
    try:
        rec = json.load(open(os.path.join(dst, name)))
        return {k: v for k, v in rec.items() if not k.startswith("_")}
    except Exception:
        return {}


summary, others = _existing(f"{tag}_pmc_summary.json"), _existing(f"{tag}_pmc_other_workloads.json")
for name in pmc:
    if name not in ("adam", "sgd", "rmsprop"):
        others.pop(name, None)        # a re-measured workload replaces its old kernels wholesale
for name, kernels in pmc.items():
    for kern, c in kernels.items():
        entry = {f"{k}_KB_mean": round(sum(v) / len(v), 2) for k, v in c.items()}
        entry.update({f"{k}_n": len(v) for k, v in c.items()})
        if name in ("adam", "sgd", "rmsprop"):
            summary[kern] = entry
        else:
            others.setdefault(name, {})[kern] = {"FETCH_SIZE": entry.get("FETCH_SIZE_KB_mean", 0.0),
                                                 "WRITE_SIZE": entry.get("WRITE_SIZE_KB_mean", 0.0),
                                                 "n": entry.get("FETCH_SIZE_n", 0)}
note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of bench.py (tools/refresh_profiles.sh); KB per "
        "dispatch, mean over dispatches.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide "
        "(16 B/lane) coalesced stream; these kernels mix 4-B-per-lane row gathers with wider streams, so FETCH_SIZE is "
        "left uncorrected (a lower bound); Infinity-Cache hits are counted.")
if summary:
    summary["_note"] = note
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
if others:
    others["_note"] = note
    json.dump(others, open(os.path.join(dst, f"{tag}_pmc_other_workloads.json"), "w"), indent=1)
# the stamps: which sources each evidence group (__graft_entry__.EVIDENCE_GROUPS) was measured with -- written on the
# GPU box by refresh_profiles.sh as stamp_<group>.json (library hash + sha256 of every source file) -- and the commit
# this collection is made at.  bench.py attaches profile-sourced numbers to a line only when the files of the
# workload's group are unchanged (evidence_stamp).  Groups that were not re-measured keep their stamp.
import subprocess

sys.path.insert(0, ROOT)
import __graft_entry__ as entry

stamp_path = os.path.join(dst, f"{tag}_stamp.json")
try:
    stamp = json.load(open(stamp_path))
    assert "groups" in stamp
except Exception:
    stamp = {"round": tag, "groups": {}}
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "beta-recsys_amd/csrc", "include"],
                       capture_output=True, text=True).stdout.strip()
tree = entry.source_file_hashes()
for f in sorted(glob.glob(os.path.join(src, "stamp_*.json"))):
    group = os.path.basename(f)[6:-5]
    if group not in entry.EVIDENCE_GROUPS:
        continue
    try:
        rec = json.load(open(f))
    except Exception:
        continue
    files = {k: rec["files"].get(k) for k in entry.EVIDENCE_GROUPS[group]}
    prev = stamp["groups"].get(group)
    if prev is not None and prev.get("files") == files:
        continue        # measured earlier with the same sources: the commit it was collected at stands
    current = all(tree.get(k) == v for k, v in files.items())
    if not current:
        print(f"WARNING: group {group} under gpurun_out/{tag} was measured with sources that differ from this tree: "
              "bench.py will report its profiles as stale", file=sys.stderr)
    stamp["groups"][group] = {"commit": (head + ("+uncommitted-kernel-sources" if dirty else "")) if current else None,
                              "source_hash": rec.get("source_hash"), "files": files}
json.dump(stamp, open(stamp_path, "w"), indent=1)
for name in (f"{tag}_pmc_summary.json", f"{tag}_pmc_other_workloads.json"):
    path = os.path.join(dst, name)
    if os.path.exists(path):
        rec = json.load(open(path))
        rec["_stamp"] = {g: {k: v[k] for k in ("commit", "source_hash")} for g, v in stamp["groups"].items()}
        json.dump(rec, open(path, "w"), indent=1)
for f in glob.glob(os.path.join(src, "exp_*.txt")):
    shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(dst, f"{tag}_*"))))
