"""profiles/r03_gemm_pmc.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_step.sh: HBM bytes per launch of
the gene-width GEMM kernels (wide::gemm_wide_kernel<*,*,8,*> and wide::gemm_wide16_kernel<*,*,*>) in the real
training step.
HBM bytes = FETCH_SIZE [KB] x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE [KB] x 1024.
usage: pmc_step_json.py <dir with fetch/ and write/> <steps the profiled command ran (warm-up + timed)>"""
import csv, glob, json, os, sys
from collections import defaultdict


def read(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            # the gene-width launches: the eight-wave 256x256 kernel and its sixteen-wave form (the (XC, RC) weight gradient)
            wide = ("gemm_wide_kernel" in n and ", 8" in n) or "gemm_wide16_kernel" in n
            if not wide or r["Counter_Name"] != counter:
                continue
            acc[(n.split("(")[0].replace("void c2::wide::", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


# The step's HBM-bound kernels in the same two passes: (kernel-name fragment, label).  Their algorithmic bytes are quoted in
# DESIGN.md section 5 beside these measurements.
OTHERS = [("nb_rowreg_kernel", "NB loss"), ("adam_dev_kernel", "fused Adam"), ("adam_kernel", "fused Adam"),
          ("mmd_fast_kernel", "MMD"),
          ("epilogue_fwd_vec_kernel<float, 4>", "encoder L0 tail, forward"),
          ("epilogue_bwd_vec_kernel<float, 4>", "encoder L0 tail, backward"),
          ("nb_dtheta_stage1", "dispersion / bias partial reduction, stage 1"), ("gemv_bias_kernel", "W c GEMV")]


def read_others(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for frag, label in OTHERS:
                if frag in r["Kernel_Name"]:
                    acc[label].append(float(r["Counter_Value"]))
    return acc


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    fe, wr = read(os.path.join(out, "fetch"), "FETCH_SIZE"), read(os.path.join(out, "write"), "WRITE_SIZE")
    ofe, owr = read_others(os.path.join(out, "fetch"), "FETCH_SIZE"), read_others(os.path.join(out, "write"), "WRITE_SIZE")
    others = {}
    for label in ofe:
        f_kb = sum(ofe[label]) / len(ofe[label])
        w_kb = sum(owr[label]) / max(len(owr[label]), 1)
        others[label] = dict(launches_per_step=len(ofe[label]) / steps, fetch_kb=f_kb, write_kb=w_kb,
                             hbm_bytes_per_launch=f_kb * 1024 * 2 + w_kb * 1024)
    shapes, total, launches = {}, 0.0, 0.0
    for k in sorted(fe):
        f_kb = sum(fe[k]) / len(fe[k])
        w_kb = sum(wr[k]) / len(wr[k])
        per_step = len(fe[k]) / steps  # two launches of one step can share kernel instance and grid
        hbm = f_kb * 1024 * 2 + w_kb * 1024
        shapes[f"{k[0]} grid={k[1]}"] = dict(dispatches_seen=len(fe[k]), launches_per_step=per_step, fetch_kb=f_kb,
                                             write_kb=w_kb, hbm_bytes_per_launch=hbm)
        total += hbm * per_step
        launches += per_step
    algorithmic = 1365655552  # operands read once + outputs written once, 5 GEMMs of the C4 step
    print(json.dumps({
        "source": "tools/pmc_step.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --profile-pass --graph 0`, "
                  "separate passes)",
        "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, mean per launch over the step's gene-width GEMM launches "
                "(5 GEMMs as 7 launches: each wgrad = full rounds + split strip; the encoder's forward and weight gradient "
                "read the batch in place from the resident matrix)",
        "steps_profiled": steps, "shapes": shapes, "hbm_bytes_per_step": total,
        "algorithmic_bytes_per_step": algorithmic, "launches_per_step": launches,
        "traffic_bytes_per_launch": total / max(launches, 1e-9),
        "other_kernels_note": "the step's HBM-bound kernels in the same two passes (same correction); not part of any field above",
        "other_kernels": others}, indent=1))


if __name__ == "__main__":
    main()
