"""Kernel intervals of the last steps of a rocprofv3 --kernel-trace CSV (tools/overlap_trace.py): start and end of every kernel
relative to the first one shown, and the overlap of consecutive kernels."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
last = rows[-n:]
t0 = last[0][0]
prev_end = None
for s, e, name, qid in last:
    short = name.split("(")[0].split("<")[0][-44:]
    gap = "" if prev_end is None else ("  gap %+.1f us" % ((s - prev_end) / 1e3))
    print("%8.1f -> %8.1f us  (%.1f us)  queue %s  %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qid, short, gap))
    prev_end = e if prev_end is None else max(prev_end, e)
