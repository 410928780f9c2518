"""Turn gpurun_out/prof/ (tools/collect_profiles.sh) into the summaries committed under profiles/."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", os.environ.get("GST_PROF_DIR", "prof"))
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counters(d):
    """per kernel: {counter: (sum over dispatches, dispatches)}"""
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = out[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
    return out


shutil.copy(glob.glob(os.path.join(SRC, "fd_stats", "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(DST, tag + "_bench_fd_kernel_stats.csv"))
shutil.copy(glob.glob(os.path.join(SRC, "an_stats", "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(DST, tag + "_bench_analytic_kernel_stats.csv"))
hbm = {}
for mode in ("fd", "analytic"):
    hbm[mode] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, cs in counters("pmc_%s_%s" % (mode, c)).items():
            if c in cs:
                e = hbm[mode].setdefault(k, {})
                e[c + "_KB_per_launch"] = cs[c][0] / cs[c][1]
                e["launches"] = cs[c][1]
json.dump(hbm, open(os.path.join(DST, tag + "_hbm_counters.json"), "w"), indent=1)
sq = {}
for name, d in (("fd", "pmc_fd_sq"), ("analytic", "pmc_an_sq")):
    sq[name] = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in counters(d).items()}
json.dump(sq, open(os.path.join(DST, tag + "_bench_pmc_sq_current.json"), "w"), indent=1)
print(json.dumps(hbm, indent=1)); print(json.dumps(sq, indent=1))
