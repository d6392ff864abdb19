"""Runs one program under several rocprofv3 --pmc passes (one counter set per pass, `--kernel-trace` only) and writes a per-kernel
table of counter sums.

    python tools/pmc_passes.py OUTDIR SUMMARY.json --set A=SQ_WAVES,SQ_WAVE_CYCLES,... --set B=... [--kernel PATTERN ...] -- python3 bench.py ...

* Counter names the installed rocprofv3 does not list (`rocprofv3 -L`) are dropped from a set (and reported), so a pass never fails
  on a name.
* The profiled program comes right after `--` in the rocprofv3 command (no shell, no env wrapper): this script itself never
  touches the GPU, it only starts rocprofv3 as a child process.
* Per kernel pattern the summary holds, per counter: dispatches, sum, mean per dispatch -- plus derived figures when their inputs
  were collected: wave-cycle split, matrix-pipe busy fraction, VALU lane utilisation, instructions per wave.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys


def available():
    try:
        out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=300).stdout
    except Exception as e:  # noqa: BLE001
        print("rocprofv3 -L failed:", e, file=sys.stderr)
        return None
    names = set()
    for line in out.splitlines():
        line = line.strip()
        if line.startswith("Counter_Name") or line.startswith("Name"):
            names.add(line.split(":", 1)[1].strip())
    return names


def derive(e):
    def s(n):
        return e[n]["sum"] if n in e else None
    d = {}
    wc = s("SQ_WAVE_CYCLES")
    if wc:
        for n, label in (("SQ_ACTIVE_INST_ANY", "executing"), ("SQ_WAIT_INST_ANY", "issue_stalled"), ("SQ_WAIT_ANY", "waiting")):
            if s(n) is not None:
                d["wave_cycles_" + label] = s(n) / wc
    if s("SQ_VALU_MFMA_BUSY_CYCLES") is not None and s("SQ_BUSY_CU_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (exactly 32 per v_mfma_f32_32x32x16: see mfma_busy_cycles_per_mfma_inst) summed over
        # the SIMDs.  SQ_BUSY_CU_CYCLES, calibrated on this kernel against GRBM_GUI_ACTIVE / 8 (the launch's wall cycles): its sum over
        # a launch is <= 256 CUs x wall cycles and ~0.57 of it for launches that leave CUs idle at the tail -- i.e. it counts CYCLES per
        # busy CU (not quad-cycles).  Matrix-pipe busy fraction while the CU is busy = MFMA busy cycles / (4 SIMDs x busy-CU cycles).
        d["mfma_busy_over_busy_cu_raw"] = s("SQ_VALU_MFMA_BUSY_CYCLES") / s("SQ_BUSY_CU_CYCLES")
        d["mfma_pipe_busy_frac_of_busy_cu_cycles"] = s("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * s("SQ_BUSY_CU_CYCLES"))
        if s("GRBM_GUI_ACTIVE"):
            # over the launches' wall time and the whole chip: 1024 SIMDs x (GRBM_GUI_ACTIVE / 8 XCDs) cycles
            d["mfma_pipe_busy_frac_of_chip_wall_cycles"] = s("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * s("GRBM_GUI_ACTIVE") / 8.0)
    if s("SQ_VALU_MFMA_BUSY_CYCLES") is not None and s("SQ_INSTS_MFMA"):
        d["mfma_busy_cycles_per_mfma_inst"] = s("SQ_VALU_MFMA_BUSY_CYCLES") / s("SQ_INSTS_MFMA")
    if s("SQ_THREAD_CYCLES_VALU") is not None and s("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilisation"] = s("SQ_THREAD_CYCLES_VALU") / (64.0 * s("SQ_ACTIVE_INST_VALU"))
    # (SQ_WAVES may be collected in several passes: compare per-dispatch means, not sums)
    w = e["SQ_WAVES"]["sum"] / max(e["SQ_WAVES"]["dispatches"], 1) if "SQ_WAVES" in e else None
    if w:
        def per_dispatch(n):
            return e[n]["sum"] / max(e[n]["dispatches"], 1)
        d["waves_per_dispatch"] = w
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if s(n) is not None:
                d[n.lower().replace("sq_insts_", "") + "_per_wave"] = per_dispatch(n) / w
        if wc:
            d["wave_quad_cycles_per_wave"] = per_dispatch("SQ_WAVE_CYCLES") / w
    if s("SQ_LDS_BANK_CONFLICT") is not None and s("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = s("SQ_LDS_BANK_CONFLICT") / s("SQ_LDS_IDX_ACTIVE")
    if s("GRBM_GUI_ACTIVE") is not None and e["GRBM_GUI_ACTIVE"]["dispatches"]:
        d["grbm_gui_active_per_dispatch"] = s("GRBM_GUI_ACTIVE") / e["GRBM_GUI_ACTIVE"]["dispatches"]
    return d


def main():
    argv = sys.argv[1:]
    if "--" not in argv:
        raise SystemExit(__doc__)
    cut = argv.index("--")
    prog = argv[cut + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("summary")
    ap.add_argument("--set", dest="sets", action="append", required=True)
    ap.add_argument("--kernel", action="append", default=[])
    ap.add_argument("--note", default="")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--summarize-only", action="store_true", help="no GPU step: build the summary from the pass folders OUTDIR/pmc_* already there "
                    "(a collection that ran out of time leaves its finished passes behind)")
    args = ap.parse_args(argv[:cut])
    os.makedirs(args.outdir, exist_ok=True)
    names = None if args.summarize_only else available()
    env = dict(os.environ, TMPDIR="/tmp")
    passes, dropped = [], []
    for spec in args.sets:
        tag, lst = spec.split("=", 1)
        want = [c for c in lst.split(",") if c]
        have = [c for c in want if names is None or c in names]
        dropped += [c for c in want if c not in have]
        if not have:
            continue
        folder = os.path.join(args.outdir, "pmc_" + tag)
        if args.summarize_only:
            if glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
                passes.append((tag, folder, have))
            continue
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *have, "--output-format", "csv", "-d", folder, "-o", "p", "--", *prog]
        print("[pmc_passes]", " ".join(cmd), flush=True)
        with open(os.path.join(args.outdir, f"pmc_{tag}.out"), "w") as so, open(os.path.join(args.outdir, f"pmc_{tag}.err"), "w") as se:
            rc = subprocess.run(cmd, stdout=so, stderr=se, env=env, timeout=args.timeout).returncode
        print(f"[pmc_passes] pass {tag} rc {rc}", flush=True)
        if rc != 0:
            raise SystemExit(rc)   # a failed GPU step: start no further one
        passes.append((tag, folder, have))
    kernels = {}
    for tag, folder, have in passes:
        files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            name = r["Kernel_Name"]
            key = next((p for p in args.kernel if p in name), None) if args.kernel else name.split("(")[0][:80]
            if key is None:
                continue
            c = kernels.setdefault(key, {}).setdefault(r["Counter_Name"], {"dispatches": 0, "sum": 0.0})
            c["dispatches"] += 1
            c["sum"] += float(r["Counter_Value"])
    out = {"note": args.note, "command": " ".join(prog), "passes": {t: h for t, _, h in passes}, "counters_not_available": dropped, "kernels": {}}
    for k, e in kernels.items():
        for c in e.values():
            c["mean_per_dispatch"] = c["sum"] / max(c["dispatches"], 1)
        out["kernels"][k] = {"counters": e, "derived": derive(e)}
    json.dump(out, open(args.summary, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v["derived"].items()}))
    # drop the bulky per-dispatch traces, keep the counter tables
    for folder in (p[1] for p in passes):
        for f in glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True) + glob.glob(os.path.join(folder, "**", "*agent_info.csv"), recursive=True):
            os.remove(f)


if __name__ == "__main__":
    main()
