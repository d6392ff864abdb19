"""Per-kernel sums of rocprofv3 --pmc passes (one counter set per pass, `--output-format csv`), for the kernels whose name
contains any of the given patterns.

    python tools/pmc_kernel_summary.py OUT.json --pass gpurun_out/pmc_fetch --pass gpurun_out/pmc_write [...] \
        --kernel k_field_f16 --kernel k_grid_fwd [--units N --unit-name points]

Each pass directory holds one `*counter_collection.csv`.  For every (kernel pattern, counter) the script reports the number of
dispatches, the counter's sum and its mean per dispatch.  FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 reports them; the summary
adds bytes, and -- per MI355X_MICROARCH.md "HBM" -- both the raw FETCH_SIZE figure and its x2 bound (on gfx950 FETCH_SIZE tallies a
wide coalesced streaming read at half its bytes; narrow gathers are uncalibrated, so the truth lies between the two).
--units: number of work units (e.g. sampled points) the summed dispatches processed, to express bytes per unit.
"""
import argparse
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--pass", dest="passes", action="append", required=True)
    ap.add_argument("--kernel", action="append", required=True)
    ap.add_argument("--units", type=float, default=0)
    ap.add_argument("--unit-name", default="unit")
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    res = {"note": args.note, "passes": args.passes, "kernels": {}}
    for pat in args.kernel:
        entry = {}
        for folder in args.passes:
            files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                continue
            for r in csv.DictReader(open(files[0])):
                if pat not in r["Kernel_Name"]:
                    continue
                c = entry.setdefault(r["Counter_Name"], {"dispatches": 0, "sum": 0.0})
                c["dispatches"] += 1
                c["sum"] += float(r["Counter_Value"])
        for name, c in entry.items():
            c["mean_per_dispatch"] = c["sum"] / max(c["dispatches"], 1)
        if "FETCH_SIZE" in entry and "WRITE_SIZE" in entry:
            fb, wb = entry["FETCH_SIZE"]["sum"] * 1024, entry["WRITE_SIZE"]["sum"] * 1024
            hb = {"fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_raw": fb + wb, "hbm_bytes_fetch_x2": 2 * fb + wb}
            if args.units:
                hb[f"hbm_bytes_per_{args.unit_name}_raw"] = (fb + wb) / args.units
                hb[f"hbm_bytes_per_{args.unit_name}_fetch_x2"] = (2 * fb + wb) / args.units
                hb[args.unit_name + "s"] = args.units
            entry["hbm"] = hb
        res["kernels"][pat] = entry
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps({k: {n: (v if n == "hbm" else round(v["sum"], 1)) for n, v in e.items()} for k, e in res["kernels"].items()}))


if __name__ == "__main__":
    main()
