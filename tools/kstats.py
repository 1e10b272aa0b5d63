"""Short table of a rocprofv3 --kernel-trace --stats directory: kernel (template arguments cut), calls, average us, share.
    python tools/kstats.py gpurun_out/<dir> [min_calls]"""
import csv, glob, os, re, sys
d = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    print(f)
    for row in csv.DictReader(open(f)):
        if int(row["Calls"]) < min_calls:
            continue
        name = row["Name"]
        m = re.match(r"(?:void )?([\w:]+)(<[^(]{0,60})?", name)
        short = (m.group(1) + (m.group(2) or "")) if m else name[:80]
        print("  %-90s calls %6s  avg %9.2f us  %6.2f %%  (min %.1f max %.1f)" % (short[:90], row["Calls"], float(row["AverageNs"]) / 1e3,
              float(row["Percentage"]), float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
