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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--window-ms", type=float, default=150.0, help="analyse the LAST this-many ms of the trace (the timed stream)")
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    t_end = max(e for _, e, _ in rows)
    t0 = t_end - int(args.window_ms * 1e6)
    rows = [(max(s, t0), e, n) for s, e, n in rows if e > t0]
    wall = t_end - t0
    field = [(s, e) for s, e, n in rows if "k_field_f16" in n]
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
