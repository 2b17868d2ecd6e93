"""print the per-kernel table of a rocprofv3 --stats csv directory:  python tools/kstats.py <dir>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} min {float(r["MinNs"]) / 1e3:9.1f} max {float(r["MaxNs"]) / 1e3:9.1f}')
