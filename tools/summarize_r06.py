"""Turn gpurun_out/r06f/ (tools/collect_r06.sh) into the summaries committed under profiles/r06_*."""
import csv, glob, json, os, re, shutil
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06f")
DST = os.path.join(ROOT, "profiles")
TAG = "r06"


def counters(d):
    """per kernel: {counter: (sum over dispatches, dispatches)}"""
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            c = out[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
    return out


def last_json(path):
    line = None
    try:
        for ln in open(path):
            ln = ln.strip()
            if ln.startswith("{"):
                line = ln
    except OSError:
        return None
    return json.loads(line) if line else None


def dst(name):
    return os.path.join(DST, TAG + "_" + name)


for d, name in (("fd_stats", "bench_fd"), ("an_stats", "bench_analytic"), ("lv_stats", "level_passes"), ("cfg_stats", "other_configs"), ("lb_stats", "cptplnd_exact")):
    g = glob.glob(os.path.join(SRC, d, "**", "*kernel_stats.csv"), recursive=True)
    if g:
        shutil.copy(g[0], dst(name + "_kernel_stats.csv"))

hbm = {}
for mode in ("fd", "analytic", "cfg"):
    hbm[mode] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, cs in counters("pmc_%s_%s" % (mode, c)).items():
            if c in cs:
                e = hbm[mode].setdefault(k, {})
                e[c + "_KB_per_launch"] = cs[c][0] / cs[c][1]
                e["launches"] = cs[c][1]
hbm["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate counter-only passes; KB per launch; HBM bytes = (2 x FETCH_SIZE + "
               "WRITE_SIZE) x 1024 (the gfx950 correction of MI355X_MICROARCH.md); 'cfg' = tools/bench_configs.py (1Q L<=128 and 3Q D=64 legs)")
json.dump(hbm, open(dst("hbm_counters.json"), "w"), indent=1)

sq = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in counters("pmc_fd_sq").items()}
json.dump({"fd": sq}, open(dst("bench_pmc_sq_current.json"), "w"), indent=1)

for f, name in (("bench.json", "bench.json"), ("bench_jtj.json", "bench_jtj.json"), ("bench_analytic.json", "bench_analytic.json"),
                ("bench_analytic_tiles.json", "bench_analytic_tiles.json"), ("two_ranks_one_gpu.json", "two_ranks_one_gpu.json")):
    b = last_json(os.path.join(SRC, f))
    if b:
        json.dump(b, open(dst(name), "w"), indent=1)

ranks = []
try:
    for ln in open(os.path.join(SRC, "emulate_all_ranks.txt")):
        m = re.match(r"rank (\d+) of (\d+): step ([\d.]+) ms kernel ([\d.]+)\s+tasks (\d+)\s+applies/pass (\d+)\s+nE (\d+)", ln)
        if m:
            ranks.append({"rank": int(m.group(1)), "of": int(m.group(2)), "step_ms": float(m.group(3)), "kernel_ms": float(m.group(4)),
                          "tasks": int(m.group(5)), "applies_per_pass": int(m.group(6)), "n_elements": int(m.group(7))})
except OSError:
    pass
head = last_json(os.path.join(SRC, "bench.json"))
if ranks and head:
    slow = max(r["step_ms"] for r in ranks)
    json.dump({"ranks": ranks, "full_design_ms_per_step": head["ms_per_step"], "slowest_rank_ms": slow,
               "projected_speedup_at_%d_gpus" % ranks[0]["of"]: head["ms_per_step"] / slow,
               "note": "one GPU runs each rank's 1/N atom in turn (bench.py --emulate-ranks N --emulate-rank r); the N-GPU fill step is the "
                       "slowest rank's; NOT a multi-GPU measurement"}, open(dst("emulate_all_ranks.json"), "w"), indent=1)
for f in ("pytest.txt", "soak.txt", "smoke.txt"):
    try:
        shutil.copy(os.path.join(SRC, f), dst(f))
    except OSError:
        pass
print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(DST, TAG + "_*"))))
