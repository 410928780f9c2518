"""Turn gpurun_out/r03/ (tools/collect_r03.sh) into the summaries committed under profiles/r03_*."""
import csv, glob, json, os, shutil, statistics
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r03")
DST = os.path.join(ROOT, "profiles")
TAG = "r03"


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
    """the last JSON line of a bench output"""
    line = None
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith("{"):
            line = ln
    return json.loads(line) if line else None


def dst(name):
    return os.path.join(DST, TAG + "_" + name)


for d, name in (("fd_stats", "bench_fd"), ("an_stats", "bench_analytic"), ("jtj_stats", "jtj"), ("lb_stats", "lindblad"), ("emu8_stats", "emulate8")):
    shutil.copy(glob.glob(os.path.join(SRC, d, "**", "*kernel_stats.csv"), recursive=True)[0], dst(name + "_kernel_stats.csv"))

hbm = {}
for mode in ("fd", "analytic"):
    hbm[mode] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, cs in counters("pmc_%s_%s" % (mode, c)).items():
            if c in cs:
                e = hbm[mode].setdefault(k, {})
                e[c + "_KB_per_launch"] = cs[c][0] / cs[c][1]
                e["launches"] = cs[c][1]
json.dump(hbm, open(dst("hbm_counters.json"), "w"), indent=1)

sq = {}
for name, d in (("fd", "pmc_fd_sq"), ("lindblad", "pmc_lb_sq"), ("jtj_block_sparse", "pmc_jtj"), ("jtj_dense", "pmc_jtj_dense")):
    sq[name] = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in counters(d).items()}
json.dump(sq, open(dst("bench_pmc_sq_current.json"), "w"), indent=1)

# normal equations: block-sparse against dense, with the MFMA pipe's busy fraction from the counters
ne = {}
for name, f in (("block_sparse", "bench_jtj.json"), ("dense", "bench_jtj_dense.json")):
    b = last_json(os.path.join(SRC, f))
    ne[name] = b.get("normal_equations")
for name, key in (("block_sparse", "jtj_block_sparse"), ("dense", "jtj_dense")):
    for k, cs in sq[key].items():
        if "jtj_mfma" in k and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            ne[name]["pmc"] = dict(kernel=k, **cs)
            if cs.get("GRBM_GUI_ACTIVE"):
                # one f64 16x16x4 MFMA holds its matrix pipe 64 cycles; 1,024 pipes; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                ne[name]["pmc"]["matrix_pipe_busy_frac"] = cs["SQ_INSTS_MFMA"] * 64.0 / 1024.0 / (cs["GRBM_GUI_ACTIVE"] / 8.0)
json.dump(ne, open(dst("normal_equations.json"), "w"), indent=1)

emu = {}
for E in ("2", "4", "8", "8_nooverlap"):
    runs = [last_json(f) for f in sorted(glob.glob(os.path.join(SRC, "emu%s_rep*.json" % E)))]
    runs = [r for r in runs if r]
    emu[E] = {
        "ms_per_step": [r["ms_per_step"] for r in runs],
        "dominant_kernel_ms": [r["roofline"].get("kernel_ms") for r in runs],
        "median_ms_per_step": statistics.median(r["ms_per_step"] for r in runs),
        "emulate_ranks": runs[0]["config"].get("emulate_ranks"),
    }
head = last_json(os.path.join(SRC, "bench.json"))
emu["full_design_ms_per_step"] = head["ms_per_step"]
for E in ("2", "4", "8", "8_nooverlap"):
    emu[E]["projected_speedup"] = head["ms_per_step"] / emu[E]["median_ms_per_step"]
json.dump(emu, open(dst("emulate_ranks.json"), "w"), indent=1)

for f, name in (("bench.json", "bench.json"), ("bench_jtj.json", "bench_jtj.json"), ("bench_analytic.json", "bench_analytic.json"),
                ("bench_analytic_keepzeros.json", "bench_analytic_keepzeros.json"), ("two_ranks_one_gpu.json", "two_ranks_one_gpu.json")):
    json.dump(last_json(os.path.join(SRC, f)), open(dst(name), "w"), indent=1)
for f in ("fd_small_atom_trace.txt", "lindblad_timing.txt", "lindblad_timing_lite.txt", "lindblad_timing_noshare_lite.txt"):
    shutil.copy(os.path.join(SRC, f), dst(f))
print(json.dumps(emu, indent=1)); print(json.dumps(ne, indent=1)); print(json.dumps(hbm, indent=1))
