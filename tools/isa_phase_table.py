"""Static attribution of the fused field kernel's instructions to its phases, from the compiler's ISA listing.

    hipcc ... -S --cuda-device-only -o field.s seald-nerf_amd/csrc/field.hip
    python tools/isa_phase_table.py field.s [kernel-substring]

Phases are cut at the markers the source plants (s_setprio, s_barrier, the hidden-layer loop label) and, behind the last
s_setprio, at runs of MFMA instructions (D7 | grid | sigma net | colour net | epilogue).  Loop bodies are weighted by their trip
count (the hidden-layer loop runs 3 times, two layers per trip).  Output: one row per phase with VALU (non-MFMA v_*), transcendental
VALU, MFMA, LDS, vector-memory and scalar instruction counts per wave, and the 4-cycle issue slots they take (transcendentals 8,
an MFMA holds the vector issue port for 8: MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')."""
import json
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "trans" if op.startswith(TRANS) else "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_field_f16ILi4ELi2ELb0ELb1E"
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = []
    for i in range(start + 1, end):
        l = lines[i].split(";")[0].strip()
        if not l or l.startswith("."):
            if re.match(r"\.LBB\d+_\d+:", l):
                body.append(("label", l[:-1]))
            continue
        body.append(("inst", l.split()[0], l))
    # phase boundaries
    phases, cur, name = [], [], "prologue + freq features + bias"
    state = {"setprio": 0, "barrier": 0, "loop": False}
    loop_label = None
    for k, it in enumerate(body):
        if it[0] == "label":
            # the hidden-layer loop: the label the only backward s_cbranch_scc1 behind the first s_setprio targets
            continue
        op, text = it[1], it[2]
        cur.append(op)
        if op == "s_setprio":
            state["setprio"] += 1
            phases.append((name, cur, 1))
            cur = []
            name = "layer D0 (wait + 16 MFMA)" if state["setprio"] == 1 else "TAIL"
        elif op == "s_endpgm":
            phases.append((name, cur, 1))
            cur = []
            name = "exit paths"
    if cur:
        phases.append((name, cur, 1))
    # split the region between the two s_setprio at the loop: find the backward branch
    out = []
    for name, ops, w in phases:
        if name.startswith("layer D0"):
            # D0 | loop body x3 | post-loop conversion + D7
            idx_b = [i for i, o in enumerate(ops) if o == "s_barrier"]
            br = max(i for i, o in enumerate(ops) if o.startswith("s_cbranch_scc"))
            # loop header: first barrier inside the loop is idx_b[1]; the loop starts a few scalar instructions before it -- take the
            # conversion block in front of it (the instructions after D0's last MFMA) as part of the loop body
            first_mfma_run_end = 0
            for i, o in enumerate(ops[:idx_b[1]]):
                if o.startswith("v_mfma"):
                    first_mfma_run_end = i + 1
            out.append(("layer D0: stage wait + 16 MFMA", ops[:first_mfma_run_end], 1))
            out.append(("hidden layers D1..D6: 2 layers per trip (acc->fp16 + ReLU, barrier, refill, 32 MFMA each) x3", ops[first_mfma_run_end:br + 1], 3))
            out.append(("after the loop: last conversion, tail-stage wait, layer D7 (8 MFMA), deformation", ops[br + 1:], 1))
        elif name == "TAIL":
            # cut at MFMA runs: grid phase = up to the first MFMA; then sigma net (8 MFMA) ; SH + colour net (16 MFMA) ; epilogue
            m = [i for i, o in enumerate(ops) if o.startswith("v_mfma")]
            g_end = m[0]
            out.append(("grid encode: 8 levels per lane-half (index math, 32 gathers, interpolation)", ops[:g_end], 1))
            s_end = m[7] + 1
            out.append(("sigma net: 8 MFMA + conversions", ops[g_end:s_end], 1))
            c_end = m[-1] + 1
            out.append(("exp + SH + colour net: 16 MFMA + conversions", ops[s_end:c_end], 1))
            out.append(("epilogue: sigmoid, stores", ops[c_end:], 1))
        else:
            out.append((name, ops, w))
    rows, tot = [], {"valu": 0, "trans": 0, "mfma": 0, "lds": 0, "vmem": 0, "salu": 0}
    for name, ops, w in out:
        if name == "exit paths":
            continue
        c = {"valu": 0, "trans": 0, "mfma": 0, "lds": 0, "vmem": 0, "salu": 0}
        top = {}
        for o in ops:
            k = classify(o)
            if k:
                c[k] += w
                if k in ("valu", "trans"):
                    top[o] = top.get(o, 0) + w
        for k in c:
            tot[k] += c[k]
        c["issue_cycles"] = 4 * c["valu"] + 8 * c["trans"] + 8 * c["mfma"]
        c["top_valu"] = dict(sorted(top.items(), key=lambda kv: -kv[1])[:8])
        rows.append({"phase": name, **c})
    tot["issue_cycles"] = 4 * tot["valu"] + 8 * tot["trans"] + 8 * tot["mfma"]
    tot["valu_incl_mfma_per_mfma"] = (tot["valu"] + tot["trans"] + tot["mfma"]) / max(1, tot["mfma"])
    print(json.dumps({"kernel": want, "per_wave_tile_of_32_points": rows, "total": tot}, indent=1))


if __name__ == "__main__":
    main()
