"""Effective shader clock per kernel from a rocprofv3 pass with --pmc GRBM_GUI_ACTIVE and --kernel-trace of the same command:
clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md, DVFS give-back: reads high on dispatches shorter than
about 0.3 ms).  usage: kernel_clocks.py <dir>   (the -d directory of:  rocprofv3 --kernel-trace --output-format csv --pmc
GRBM_GUI_ACTIVE -d <dir> -o t -- python3 bench.py --launch eager --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing)"""
import collections, csv, glob, sys
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
        name, ns = dur[r["Dispatch_Id"]]
        if ns < 100000: continue          # shorter than 0.1 ms: the quotient is not a clock
        a = acc[name]; a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
print(f"{'kernel (dispatches of at least 0.1 ms)':70s} {'launches':>8s} {'avg us':>9s} {'GHz':>6s}")
for name, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:70]:70s} {n:8d} {ns / n / 1e3:9.1f} {cyc / 8 / ns:6.2f}")
