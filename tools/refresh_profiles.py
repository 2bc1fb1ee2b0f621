"""Copies the outputs of tools/collect_profiles.sh (gpurun_out/final) into profiles/ under the
round's names and rebuilds the PMC traffic summary.  usage: python tools/refresh_profiles.py r01"""
import collections, csv, glob, json, os, re, shutil, sys

tagr = sys.argv[1] if len(sys.argv) > 1 else "r01"
O = "gpurun_out/final"
names = {"bench_c2": "c2_bench", "bench_c2_lanes1": "c2_bench_lanes1", "bench_c2_free": "c2_bench_lanes2_free",
         "bench_c3": "c3_bench", "bench_c3_auto": "c3_bench_upsample_auto", "bench_c3_type1": "c3_bench_type1", "bench_c5": "c5_bench",
         "bench_c5_auto": "c5_bench_upsample_auto"}
for src, dst in names.items():
    if os.path.exists(f"{O}/{src}.json"):
        shutil.copy(f"{O}/{src}.json", f"profiles/{tagr}_{dst}.json")
for tag in ("c2", "c3"):
    f = sorted(glob.glob(f"{O}/prof_{tag}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if f:  # merged gpurun_out directories keep earlier runs' files: newest wins
        shutil.copy(f[-1], f"profiles/{tagr}_{tag}_kernel_stats.csv")


def agg(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "").replace("fv::", "")
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


path = f"profiles/{tagr}_hbm_traffic_pmc.json"
out = json.load(open(path)) if os.path.exists(path) else {"note": "", "counters": {}}
for tag in ("c2", "c3"):
    F, W = agg(f"{O}/pmc_fetch_{tag}"), agg(f"{O}/pmc_write_{tag}")
    if not F:
        continue
    out["counters"][tag] = {}
    for k in ("k_strengths", "k_spread2d", "k_spread2d_cg", "k_rowfft_st", "k_transpose", "k_interp"):
        if k in F and k in W:
            out["counters"][tag][k] = {"FETCH_SIZE_KB_avg_per_launch": F[k][0] / F[k][1], "launches": F[k][1],
                                       "WRITE_SIZE_KB_avg_per_launch": W[k][0] / W[k][1]}
json.dump(out, open(path, "w"), indent=1)
# the C2 bench line was taken before the PMC passes of the same collection: fill its traffic field from them
p2 = f"profiles/{tagr}_c2_bench.json"
if os.path.exists(p2) and "c2" in out["counters"]:
    d = json.load(open(p2))
    k = out["counters"]["c2"].get(d["roofline"]["kernel"])
    if k and d["roofline"].get("traffic") is None:
        d["roofline"]["traffic"] = (2 * k["FETCH_SIZE_KB_avg_per_launch"] + k["WRITE_SIZE_KB_avg_per_launch"]) * 1024
        d["roofline"]["traffic_source"] = path + " (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE; filled in by tools/refresh_profiles.py)"
        json.dump(d, open(p2, "w"))
for dst in names.values():
    p = f"profiles/{tagr}_{dst}.json"
    if os.path.exists(p):
        d = json.load(open(p))
        print(dst, f"{d['value']:.4g}", round(d["ms_per_step"], 3), d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
              round(d["roofline"]["avg_launch_ms"], 4), "fft", round(d["roofline_fft"]["frac"], 3))
