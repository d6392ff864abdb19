"""Timeline statistics of a rocprofv3 --kernel-trace CSV (`*_kernel_trace.csv`): for the pipelined stream of frames, how much of the
wall time has a field kernel running, only marchers / small kernels running, or nothing at all -- and the concurrency of the field
kernel with itself.

    python tools/trace_overlap.py gpurun_out/xyz/p_kernel_trace.csv [--skip-first-ms 200]
"""
import argparse
import csv
import json


def union(iv):
    iv = sorted(iv)
    out, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                out += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        out += cur_e - cur_s
    return out


def kind(n):
    return "F" if "k_field" in n else ("M" if ("k_composite_march" in n or "k_march_rays" in n) else "o")


def per_queue(rows, t0):
    """Per hardware queue (= loop context): time inside field kernels, marchers, other kernels, and the gaps between the end of one
    kernel and the start of the next, split by what the next kernel is; plus the histogram of how many field kernels run at once."""
    qs = {}
    for s, e, n, q in rows:
        if e > t0:
            qs.setdefault(q, []).append((s, e, kind(n)))
    out = {}
    for q, lst in qs.items():
        lst.sort()
        if len(lst) < 50:
            continue
        tot = {"F": 0, "M": 0, "o": 0, "gap_before_F": 0, "gap_before_M": 0, "gap_before_o": 0}
        gaps = {"F": [], "M": [], "o": []}
        for i, (s, e, k) in enumerate(lst):
            tot[k] += e - s
            if i:
                g = max(0, s - lst[i - 1][1])
                tot["gap_before_" + k] += g
                gaps[k].append(g)
        span = lst[-1][1] - lst[0][0]
        med = {k: (sorted(v)[len(v) // 2] / 1e3 if v else None) for k, v in gaps.items()}
        p90 = {k: (sorted(v)[int(len(v) * 0.9)] / 1e3 if v else None) for k, v in gaps.items()}
        out[q] = {"kernels": len(lst), "span_ms": span / 1e6, **{k: round(v / span, 4) for k, v in tot.items()}, "gap_us_median": med, "gap_us_p90": p90}
    ev = []
    for s, e, n, q in rows:
        if e > t0 and kind(n) == "F":
            ev += [(max(s, t0), 1), (e, -1)]
    ev.sort()
    hist, depth, last = {}, 0, t0
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth, last = depth + d, t
    tot = sum(hist.values()) or 1
    print(json.dumps({"per_queue": out, "field_kernels_running_at_once": {str(k): round(v / tot, 4) for k, v in sorted(hist.items())}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--window-ms", type=float, default=150.0, help="analyse the LAST this-many ms of the trace (the timed stream)")
    ap.add_argument("--densest", action="store_true", help="analyse the window with the most launches instead of the last one")
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(args.window_ms * 1e6)
    if args.densest:   # the window of that length holding the most kernel launches: the pipelined stream, wherever it sits in the trace
        starts = sorted(r[0] for r in rows)
        w, best, j = int(args.window_ms * 1e6), (0, starts[0]), 0
        for i, s in enumerate(starts):
            while starts[j] < s - w:
                j += 1
            if i - j + 1 > best[0]:
                best = (i - j + 1, s)
        t_end = best[1]
        t0 = t_end - w
        rows = [r for r in rows if r[0] < t_end]
        rows = [(s, min(e, t_end), n, q) for s, e, n, q in rows]
    per_queue(rows, t0)
    rows = [(max(s, t0), e, n) for s, e, n, _ in rows if e > t0]
    wall = t_end - t0
    field = [(s, e) for s, e, n in rows if "k_field" in n]
    march = [(s, e) for s, e, n in rows if "k_composite_march" in n or "k_march_rays" in n]
    allk = [(s, e) for s, e, _ in rows]
    u_all, u_field, u_march = union(allk), union(field), union(march)
    # field concurrency: sum of field durations / union
    print(json.dumps({"window_ms": wall / 1e6, "kernels": len(rows), "busy_frac": u_all / wall, "field_running_frac": u_field / wall,
                      "marcher_running_frac": u_march / wall, "idle_frac": 1 - u_all / wall,
                      "field_sum_over_union": sum(e - s for s, e in field) / max(u_field, 1),
                      "marcher_sum_over_union": sum(e - s for s, e in march) / max(u_march, 1),
                      "field_sum_ms": sum(e - s for s, e in field) / 1e6, "marcher_sum_ms": sum(e - s for s, e in march) / 1e6}))


if __name__ == "__main__":
    main()
