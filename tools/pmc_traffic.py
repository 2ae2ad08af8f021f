#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of bench.py (FETCH_SIZE in one, WRITE_SIZE in the other; --output-format csv) into
profiles/pmc_traffic.json: HBM bytes per stage per step (= per launch of the stage), which bench.py reports as
roofline.traffic.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 > profiles/pmc_traffic.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB;
FETCH_SIZE tallies 128-byte requests as 64 bytes, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import sys

STAGES = {                      # stage name of bench.py -> substring of the kernel name
    "inter_job_build": "inter_build_kernel",
    "inter_pred_luma_dmvr_bdof": "bipred_kernel",    # all launches of the kernel in a step: the regular sub-blocks and the CIIP units' inter part
    "inter_pred_chroma": "bipred_chroma_pair_kernel",
    "inter_pred_gpm": "gpm_kernel",
    "inter_pred_affine_prof": "affine_kernel",
    "dequant_itx_add_residual": "itx_shape_kernel",
    "intra_tb_dequant_lfnst_itx": "itx_kernel",        # + lfnst_batch_kernel, listed on its own below
    "intra_tb_lfnst": "lfnst_batch_kernel",
    "intra_recon_wavefront": "recon_wavefront_kernel",
    "lmcs_inverse_luma": "lmcs_kernel",
    "deblock_bs": "deblock_bs_kernel",
    "deblock_vertical": "deblock_frame_kernel", # first deblock launch of a step
    "deblock_horizontal": "deblock_frame_kernel",   # second one
    "sao": "sao_frame_kernel",
    "alf": "alf_ctb_kernel",                    # the CTB kernel (luma + chroma + CC-ALF)
    "alf_job_build": "alf_build_kernel",
    "lmcs_chroma_residual_scale": "lmcs_chroma_resid_kernel",
    "lmcs_vpdu_scale_table": "lmcs_vpdu_scale_kernel",
    "side_tables_fill": "tabfill_kernel",       # both launches of a step: the side tables and the inter stage's MvField table
    "itx_job_build": "itx_build_kernel",
}


def per_stage(directory, counter, n_passes):
    """Mean counter value per dispatch of every kernel a stage launches (each kernel runs once per step), summed per stage.
    Means per dispatch, not totals / steps: bench.py launches the bS kernel once more while it builds the chain."""
    path = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = {}
    for stage, pat in STAGES.items():
        sel = [r for r in rows if pat in r["Kernel_Name"]]
        if pat == "deblock_frame_kernel":
            sel = sel[0::2] if stage == "deblock_vertical" else sel[1::2]
        by_kernel = {}
        # a kernel launched several times per step with different sizes (bipred_kernel: the regular sub-blocks, then the few CIIP
        # units) is represented by its largest launch: the one the stage's time and algorithmic bytes refer to
        largest = {}
        for r in sel:
            largest[r["Kernel_Name"]] = max(largest.get(r["Kernel_Name"], 0), int(r["Grid_Size"]))
        for r in sel:
            if int(r["Grid_Size"]) == largest[r["Kernel_Name"]] or stage == "side_tables_fill":
                by_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        out[stage] = sum(sum(v) / len(v) for v in by_kernel.values())
    return out


def main():
    fetch_dir, write_dir, n_passes = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, write = per_stage(fetch_dir, "FETCH_SIZE", n_passes), per_stage(write_dir, "WRITE_SIZE", n_passes)
    doc = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of bench.py on the 8K 10-bit frame; KiB per "
                    "stage per step; read bytes = 2 x FETCH_SIZE (gfx950: 128-B requests tallied as 64 B), WRITE_SIZE exact "
                    "(MI355X_MICROARCH.md, HBM section).  Made by tools/pmc_traffic.py."}
    for stage in STAGES:
        rd, wr = int(2 * fetch[stage] * 1024), int(write[stage] * 1024)
        doc[stage] = {"kernel": STAGES[stage], "FETCH_SIZE_KiB": round(fetch[stage], 1), "WRITE_SIZE_KiB": round(write[stage], 1),
                      "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
