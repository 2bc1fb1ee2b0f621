"""Copies the outputs of tools/collect_profiles.sh (gpurun_out/final) into profiles/ under the round's names,
rebuilds the PMC traffic summary and prints the numbers DESIGN.md quotes.  usage: python tools/refresh_profiles.py r04"""
import collections, csv, glob, json, os, re, shutil, sys

tagr = sys.argv[1] if len(sys.argv) > 1 else "r04"
O = "gpurun_out/final"
names = {"bench_c3": "c3_bench", "bench_c3_auto": "c3_bench_upsample_auto", "bench_c3_type1": "c3_bench_type1",
         "bench_c3_four_transforms": "c3_bench_four_transforms", "bench_c2": "c2_bench", "bench_c5": "c5_bench",
         "bench_c4slice": "c4slice_bench", "bench_c4_full": "c4_full_one_gpu_bench", "bench_c5_full": "c5_full_one_gpu_bench",
         "bench_c3_two_ranks_one_gpu": "c3_two_ranks_on_one_gpu_rehearsal_bench",
         "bench_c3_one_lane": "c3_bench_one_lane", "bench_c3_scattered": "c3_scattered_bench", "bench_c3z": "c3z_bench",
         "bench_c3z_grid": "c3z_bench_grid_path", "bench_c3z_grid_three_pass": "c3z_bench_grid_path_three_pass",
         "bench_c3z_1m": "c3z_1m_scatter_bench"}
for w in ("C3", "C4"):
    for r in (0, 1):
        names[f"bench_{w}_rank{r}of8"] = f"{w.lower()}_rank{r}_of_8_block_bench"
for src, dst in names.items():
    if os.path.exists(f"{O}/{src}.json") and os.path.getsize(f"{O}/{src}.json") > 10:
        shutil.copy(f"{O}/{src}.json", f"profiles/{tagr}_{dst}.json")
TAGS = ("c3", "c2", "c4slice", "c3type1", "c3scattered", "c3z", "c3onelane", "c3zgrid")
for tag in TAGS:
    f = sorted(glob.glob(f"{O}/prof_{tag}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if f:  # merged gpurun_out directories keep earlier runs' files: newest wins
        shutil.copy(f[-1], f"profiles/{tagr}_{tag}_kernel_stats.csv")


def short(name):
    return re.sub(r"[<(].*", "", name).replace("void ", "").replace("fv::", "")


def agg(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


def logical_launches(stats_csv):
    """k_strengths (type 3) / k_t1_strengths (type 1) runs once per (time, frequency group, beam pair): the unit
    bench.py calls a launch (a spread 'launch' of 24 transforms is a 16- and an 8-transform kernel launch)."""
    n, spread_ns = 0, 0.0
    for r in csv.DictReader(open(stats_csv)):
        k = short(r["Name"])
        if k in ("k_strengths", "k_t1_strengths"):
            n += int(r["Calls"])
        if k.startswith("k_spread") or k == "k_t1_spread":
            spread_ns += float(r["TotalDurationNs"])
    return n, spread_ns


path = f"profiles/{tagr}_hbm_traffic_pmc.json"
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on MI355X, KB per "
               "LOGICAL launch (bench.py's unit: one (time, frequency group, beam pair); a spread of 24 transforms is two "
               "kernel launches) = counter sum over the kernel family / k_strengths calls.  Per /opt/skills/guides/"
               "MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950: "
               "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024.  c3 = bench.py --ntimes 1 (full-size launches of the "
               "default workload), c2 = --workload C2, c4slice = --workload C4 --nfreq 32 --ntimes 1, c3type1 = --path type1 "
               "--ntimes 1, c3zgrid = --workload C3z --nfreq 4 --ntimes 1 with FFTVIS_HIP_NO_WTERM=1 (the 3-D transform).  All "
               "passes with FFTVIS_HIP_LANES=1: one stream, every kernel with the chip to itself.  k_rowfft_st sums every row / "
               "column pass.",
       "counters": {}}
for tag in TAGS:
    F, W = agg(f"{O}/pmc_FETCH_SIZE_{tag}"), agg(f"{O}/pmc_WRITE_SIZE_{tag}")
    if not F or not W:
        continue
    nlog = sum(F[k][1] for k in ("k_strengths", "k_t1_strengths") if k in F) or 1
    out["counters"][tag] = {"logical_launches": nlog}
    fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k in F:
        base = k
        if k in W:
            fam[base][0] += F[k][0]
            fam[base][1] += W[k][0]
            fam[base][2] += F[k][1]
    for k, (f_kb, w_kb, n) in fam.items():
        if k.startswith(("k_spread", "k_rowfft", "k_transpose", "k_interp", "k_strengths", "k_t1_")):
            out["counters"][tag][k] = {"FETCH_SIZE_KB_avg_per_launch": f_kb / nlog, "WRITE_SIZE_KB_avg_per_launch": w_kb / nlog,
                                       "kernel_launches": n}
json.dump(out, open(path, "w"), indent=1)
# bench lines taken before the PMC passes of the same collection: fill their traffic field from them
for tag, dst in (("c3", "c3_bench"), ("c2", "c2_bench")):
    p2 = f"profiles/{tagr}_{dst}.json"
    if os.path.exists(p2) and tag in out["counters"]:
        d = json.load(open(p2))
        k = out["counters"][tag].get(d["roofline"]["kernel"])
        if k and "four" not in dst:
            d["roofline"]["traffic"] = (2 * k["FETCH_SIZE_KB_avg_per_launch"] + k["WRITE_SIZE_KB_avg_per_launch"]) * 1024
            d["roofline"]["traffic_source"] = path + " (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE; filled in by tools/refresh_profiles.py)"
            json.dump(d, open(p2, "w"))
for dst in names.values():
    p = f"profiles/{tagr}_{dst}.json"
    if os.path.exists(p):
        d = json.load(open(p))
        ff = d.get("roofline_fft") or {}
        print(dst, f"{d['value']:.4g}", round(d["ms_per_step"], 3), d["roofline"]["kernel"], "frac", round(d["roofline"]["frac"], 3),
              "avg_ms", round(d["roofline"]["avg_launch_ms"], 4), "fft", round(ff.get("frac", 0), 3), "traffic", d["roofline"].get("traffic"))
for tag in TAGS:
    p = f"profiles/{tagr}_{tag}_kernel_stats.csv"
    if os.path.exists(p):
        n, ns = logical_launches(p)
        if n:
            print(f"{tag}: rocprofv3 spread family {ns / 1e6:.2f} ms over {n} logical launches = {ns / n / 1e6:.4f} ms per launch")
