"""One training step's device timeline from a rocprofv3 --kernel-trace CSV of `bench.py --mode train`: kernels between two successive
k_train_encode launches in the middle of the run, with queue, start offset and duration (us)."""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
enc = [i for i, r in enumerate(rows) if "k_train_encode" in r[2]]
mid = enc[len(enc) // 2]
nxt = enc[len(enc) // 2 + 1]
t0 = rows[mid][0]
print("step length us", (rows[nxt][0] - t0) / 1e3)
last_end = {}
for s, e, n, q in rows[mid - 8:nxt + 1]:
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("q%-3s %9.1f  dur %7.1f  gap_in_queue %6.1f  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, name))
