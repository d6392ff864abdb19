"""Timeline of ONE frame rendered alone, from a rocprofv3 --kernel-trace CSV of
    python3 bench.py --steps 20 --warmup 3 --pipeline 0 --group-frames 1 --no-cpu-baseline --no-secondary --min-timed-s 0
Prints every launch of the last complete frame -- kernel[workgroups] +gap-before duration, microseconds -- the sums per kernel family and
the gaps between consecutive launches aggregated by (previous kernel, kernel) over all frames of the trace.

    python tools/lone_frame_timeline.py gpurun_out/xyz/p_kernel_trace.csv
"""
import collections
import csv
import re
import sys


def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1)[2:] if m else n[:24]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_near_far" in r["Kernel_Name"]]
    a, b = starts[-3], starts[-2]
    frame = rows[a:b]
    t0 = int(frame[0]["Start_Timestamp"])
    print("launches %d, first start -> last end %.1f us" % (len(frame), (int(frame[-1]["End_Timestamp"]) - t0) / 1e3))
    prev, out, fam, gaps = t0, [], collections.Counter(), 0.0
    for r in frame:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = short(r["Kernel_Name"])
        out.append("%s[%d] +%.1f %.1f" % (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), (s - prev) / 1e3, (e - s) / 1e3))
        fam[name] += (e - s) / 1e3
        gaps += max(0.0, (s - prev) / 1e3)
        prev = e
    print(" | ".join(out))
    print("kernel time per family (us):", {k: round(v, 1) for k, v in fam.most_common()}, " gaps: %.1f" % gaps)
    agg = collections.defaultdict(list)
    for i in range(starts[3], len(rows) - 1):
        g = (int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3
        agg[(short(rows[i]["Kernel_Name"]), short(rows[i + 1]["Kernel_Name"]))].append(g)
    frames = max(1, len(starts) - 4)
    print("gaps between consecutive launches, by pair (inside frames):")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if k[0] == "loop_finish" or len(v) < frames // 2:
            continue
        v2 = sorted(v)
        if sum(v) / frames < 1.0:
            continue
        print("   %-22s -> %-22s n %4d  median %5.1f  mean %5.1f  per frame %6.1f us" % (k[0], k[1], len(v), v2[len(v2) // 2], sum(v) / len(v), sum(v) / frames))


if __name__ == "__main__":
    main()
