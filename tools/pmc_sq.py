"""Mean SQ counters per dispatch and kernel from a rocprofv3 PMC pass (rocpd database):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY \
              SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmcS -o r -- python3 bench.py --steps 3 --warmup 1
    python tools/pmc_sq.py gpurun_out/pmcS/r_results.db > profiles/r02_pmc_sq_summary.json

mfma_pipe_busy_frac = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
"""
import json
import sqlite3
import sys

KEEP = ("loss_fwd_dense_kernel", "loss_bwd_dense_kernel", "select_kernel<", "select_seed_kernel<", "gather_rows_kernel",
        "update_rows_kernel", "update_fused_kernel", "mask_sweep_kernel", "hits_kernel", "prep_kernel", "finish_kernel",
        "gt_insert_kernel", "sum_parts_kernel", "bf3_scan_kernel", "bf3_bound_kernel", "bf3_final_kernel", "bf3_prep_kernel", "mine_")
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                  "group by kernel_name, counter_name").fetchall()
out = {}
for name, counter, total, n in rows:
    if any(k in name for k in KEEP):
        out.setdefault(name[:90], {})[counter] = round(total / n, 1)
for name, c in out.items():
    if c.get("GRBM_GUI_ACTIVE"):
        c["mfma_pipe_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4)
print(json.dumps({"note": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY "
                          "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE, mean per dispatch (bench.py --steps 3 --warmup 1); "
                          "mfma_pipe_busy_frac = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); in-kernel clock measured "
                          "with s_memtime / s_memrealtime: 2.27 GHz; produced by tools/pmc_sq.py",
                  "kernels": out}, indent=1))
