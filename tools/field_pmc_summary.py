"""profiles/<round>_field_pmc_summary.json from a tools/pmc_passes.py summary over a sequential static-frame bench run:

    python tools/field_pmc_summary.py PMC_SUMMARY.json BENCH.json LOOPS OUT.json

BENCH.json is the bench line of the same command (sampled points per frame, frames per loop); LOOPS the number of loops the command
rendered (count + warm-up + timed + latency).  FETCH_SIZE on gfx950 tallies a wide coalesced read at half its bytes and is
uncalibrated for the 8-byte gathers of this kernel (MI355X_MICROARCH.md, HBM): raw and x2 figures are both given.
"""
import json
import sys


def main():
    pmc, bench, loops, out = json.load(open(sys.argv[1])), json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]), int(sys.argv[3]), sys.argv[4]
    k = pmc["kernels"].get("k_field") or pmc["kernels"]["k_field_f16"]      # "k_field": k_field_f16 and k_field_pp_f16 dispatches together
    c, d = k["counters"], k["derived"]
    frames_per_loop = int(bench["config"]["frames_per_loop"])
    points = bench["config"]["sampled_points_per_frame"] * frames_per_loop * loops
    fetch, write = c["FETCH_SIZE"]["sum"] * 1024.0, c["WRITE_SIZE"]["sum"] * 1024.0
    waves = c["SQ_WAVES"]["sum"] / c["SQ_WAVES"]["dispatches"] * c["SQ_INSTS_MFMA"]["dispatches"]      # waves over the MFMA pass's dispatches
    busy_waves = c["SQ_INSTS_MFMA"]["sum"] / 240.0                                                        # waves that did a tile (240 MFMAs each)
    res = {"note": ("rocprofv3 --pmc passes (SQ sets, GRBM, FETCH_SIZE, WRITE_SIZE: each its own pass with --kernel-trace, tools/pmc_passes.py) over `"
                    + pmc["command"] + f"`: {loops} loops of {frames_per_loop} copies of the static 800x800 frame, one loop at a time; every fused-field dispatch (k_field_f16 / k_field_pp_f16) "
                    "summed.  The kernel's algorithmic bytes are 552 B per point (512 B of table gathers + 24 in + 16 out): either traffic figure is BELOW them -- "
                    "table and weights are served by L2 / Infinity Cache."),
           "field_forward_f16": {
               "points": points, "dispatches": c["FETCH_SIZE"]["dispatches"],
               "hbm_bytes_per_point": (fetch + write) / points, "hbm_bytes_per_point_fetch_x2": (2 * fetch + write) / points, "algorithmic_bytes_per_point": 552,
               "matrix_pipe": {
                   "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"], "SQ_BUSY_CU_CYCLES": c["SQ_BUSY_CU_CYCLES"]["sum"],
                   "SQ_INSTS_MFMA": c["SQ_INSTS_MFMA"]["sum"], "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE", {}).get("sum"),
                   "busy_cycles_per_mfma": d.get("mfma_busy_cycles_per_mfma_inst"),
                   "mfma_busy_over_busy_cu": d.get("mfma_busy_over_busy_cu_raw"),
                   "matrix_pipe_busy_frac_while_cu_busy": d.get("mfma_pipe_busy_frac_of_busy_cu_cycles"),
                   "matrix_pipe_busy_frac_of_chip_wall": d.get("mfma_pipe_busy_frac_of_chip_wall_cycles"),
                   "reading": "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES per k_field_f16 dispatch, summed; both count cycles (32 per v_mfma_f32_32x32x16, per busy CU), "
                              "4 matrix pipes per CU: busy fraction while the CU is busy = ratio / 4; over the launches' wall time and all 1024 SIMDs = "
                              "MFMA busy cycles / (1024 x GRBM_GUI_ACTIVE / 8)"},
               "valu_instructions_per_tile_wave": c["SQ_INSTS_VALU"]["sum"] / busy_waves, "mfma_instructions_per_tile_wave": 240,
               "valu_per_mfma": c["SQ_INSTS_VALU"]["sum"] / c["SQ_INSTS_MFMA"]["sum"],
               "lds_instructions_per_tile_wave": c["SQ_INSTS_LDS"]["sum"] / busy_waves,
               "wave_cycle_split": {"executing (SQ_ACTIVE_INST_ANY)": d.get("wave_cycles_executing"), "issue-stalled (SQ_WAIT_INST_ANY)": d.get("wave_cycles_issue_stalled"),
                                    "waiting (SQ_WAIT_ANY)": d.get("wave_cycles_waiting")},
               "valu_mfma_coexec_over_mfma_busy": (c["SQ_VALU_MFMA_COEXEC_CYCLES"]["sum"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"]) if "SQ_VALU_MFMA_COEXEC_CYCLES" in c else None,
               "lds_bank_conflict_frac": d.get("lds_bank_conflict_frac"), "valu_lane_utilisation": d.get("valu_lane_utilisation")},
           "raw_counters": {n: v["sum"] for n, v in c.items()}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["field_forward_f16"]["matrix_pipe"]))
    print("hbm B/point", res["field_forward_f16"]["hbm_bytes_per_point"], res["field_forward_f16"]["hbm_bytes_per_point_fetch_x2"])


if __name__ == "__main__":
    main()
