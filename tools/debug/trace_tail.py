#!/usr/bin/env python3
"""Last N kernel dispatches of a rocprofv3 --kernel-trace directory: start, end (ms), queue, name."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} {(int(r["End_Timestamp"]) - t0) / 1e6:10.3f} ms  q={r.get("Queue_Id")} {r["Kernel_Name"][:50]}')
