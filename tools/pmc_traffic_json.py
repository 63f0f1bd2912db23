"""Builds profiles/rNN_hbm_traffic_pmc.json from two rocprofv3 passes of the same bench.py command:
    rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d <dir>/fetch -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d <dir>/write -o t -- python3 bench.py ...   (same)
usage: pmc_traffic_json.py <dir> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled
(MI355X_MICROARCH.md, section HBM).  A kernel launched with several grids (the attention kernels: 5 layer launches + 1
pooling launch per step) is reported for its LARGEST grid only, so the bytes are per layer launch."""
import collections, csv, glob, json, sys
def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    biggest = {}
    for (name, grid) in tot:
        if name not in biggest or grid > biggest[name]: biggest[name] = grid
    return {n: tot[(n, g)] for n, g in biggest.items()}, {n: cnt[(n, g)] for n, g in biggest.items()}
ft, fc = per_kernel(sys.argv[1] + "/fetch", "FETCH_SIZE")
wt, wc = per_kernel(sys.argv[1] + "/write", "WRITE_SIZE")
out = {}
for k in sorted(ft, key=lambda k: -ft[k]):
    if fc[k] == 0 or ft[k] / fc[k] < 1024: continue          # < 1 MiB per launch: not interesting
    raw = ft[k] / fc[k] * 1024 / 1e6
    out[k] = {"launches": fc[k], "fetch_MB_per_launch_raw": round(raw, 1), "fetch_MB_x2_gfx950": round(2 * raw, 1),
              "write_MB_per_launch": round(wt.get(k, 0.0) / max(wc.get(k, 1), 1) * 1024 / 1e6, 1)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
