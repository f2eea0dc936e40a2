"""avg duration of kernels from a rocprofv3 --stats directory.   python tools/kstats.py DIR [name-substring ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) < 3 or any(s in n for s in sys.argv[2:]):
        print(f"  {n[:44]:44s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.3f} ms")
