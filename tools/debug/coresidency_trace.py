#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of tools/debug/coresidency.py: do the elementwise kernels' execution intervals
lie inside the GEMM's?"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} {(int(r["End_Timestamp"]) - t0) / 1e6:10.3f} ms  q={r.get("Queue_Id")} {r["Kernel_Name"][:60]}')
