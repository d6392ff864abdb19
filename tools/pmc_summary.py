"""Builds profiles/<name>_field_pmc_summary.json from two rocprofv3 --pmc passes over the default bench command:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_field_pmc_summary.json

One counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md).  Values are KiB as rocprofv3
reports them.  gfx950 correction (same guide): FETCH_SIZE tallies a wide coalesced streaming read at half its bytes, other
access widths are uncalibrated -- the raw figure and the x2 bound are both recorded; WRITE_SIZE is exact for streaming stores.
"""
import csv
import glob
import json
import os
import sys


def per_dispatch(folder, counter):
    f = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "k_field_f16" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = per_dispatch(fetch_dir, "FETCH_SIZE"), per_dispatch(write_dir, "WRITE_SIZE")
    frames = 4  # 1 counting render + 1 warm-up + 2 timed steps, every one the same frame
    assert len(fetch) == len(write) and len(fetch) % frames == 0, (len(fetch), len(write))
    per_frame = len(fetch) // frames
    fetch, write = fetch[-per_frame:], write[-per_frame:]          # the last frame
    points = 1395285
    fb, wb = sum(fetch) * 1024, sum(write) * 1024
    res = {"note": __doc__.strip().split("\n\n")[1].replace("\n", " "),
           "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (last frame of the run)",
           "field_forward_f16": {"launches_per_frame": per_frame, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
                                 "fetch_bytes_per_frame_raw": fb, "write_bytes_per_frame": wb,
                                 "hbm_bytes_per_point_raw": (fb + wb) / points, "hbm_bytes_per_point_fetch_x2": (2 * fb + wb) / points,
                                 "points_per_frame": points,
                                 "algorithmic_bytes_per_point": {"table_gathers_f16": 512, "inputs": 24, "outputs": 16, "live_index": 4,
                                                                 "weights_per_point_at_256_points_per_workgroup": 960}}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res["field_forward_f16"].items() if not k.endswith("per_launch")}))


if __name__ == "__main__":
    main()
