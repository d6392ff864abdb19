"""One frame's kernels (sequential mode) from a rocprofv3 --kernel-trace CSV of `bench.py --pipeline 0`: name, start offset, duration (us)."""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
begin = [i for i, r in enumerate(rows) if "k_loop_init" in r[2]]
a, b = begin[len(begin) // 2], begin[len(begin) // 2 + 1]
t0 = rows[a][0]
prev = None
for s, e, n in rows[a - 1:b]:
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, name))
    prev = e
print("frame us", (rows[b][0] - t0) / 1e3)
