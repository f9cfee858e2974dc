"""HBM traffic per launch from two rocprofv3 PMC passes (rocpd databases):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcF -o r -- python3 bench.py --steps 3 --warmup 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmcW -o r -- python3 bench.py --steps 3 --warmup 1
    python tools/pmc_traffic.py gpurun_out/pmcF/r_results.db gpurun_out/pmcW/r_results.db > profiles/r02_traffic.json

FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests of a wide
coalesced stream are tallied at 64 B); both counters are in KiB.  Kernels are grouped under the names
bench.py's roofline objects use; a group with several kernels per launch (topk_select = seeding pass +
bound + main pass; topk_bf3 = two scans + bound + final) sums them.  The output carries the fingerprint of the kernel
sources (bench.csrc_sha): bench.py quotes a figure only while the sources are the ones it was measured on.
"""
import json
import pathlib
import sqlite3
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

GROUPS = {
    "loss_fwd_dense": ["loss_fwd_dense_kernel"],
    "loss_bwd_du": ["loss_bwd_dense_kernel<128, true"],       # (the headline width: --no-extras runs launch no other)
    "loss_bwd_dv": ["loss_bwd_dense_kernel<128, false"],
    "topk_select": ["select_kernel<", "select_seed_kernel<", "select_bound_kernel<"],
    "gather_rows": ["gather_rows_kernel"],
    "topk_bf3": ["bf3_scan_kernel<", "bf3_bound_kernel", "bf3_final_kernel<", "bf3_prep_kernel"],
    "update_rows": ["update_fused_kernel", "update_rows_kernel"],
    "mask_sweep": ["mask_sweep_kernel"],
    "sum_parts": ["sum_parts_kernel"],
}
MULTI = {"topk_select": 1, "topk_bf3": 1}      # several kernels (or several launches of one) per call: per-call sums


def per_launch(db_path: str, counter: str):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for group, pats in GROUPS.items():
        tot, launches = 0.0, 0
        for name, val, n in rows:
            if any(p in name for p in pats):
                tot += val
                launches = max(launches, n) if group in MULTI else launches + n
        if group == "topk_bf3" and launches:
            launches = [n for name, val, n in rows if "bf3_final_kernel" in name][0]     # one final kernel per call
        if launches:
            out[group] = tot / launches
    return out


fetch = per_launch(sys.argv[1], "FETCH_SIZE")
write = per_launch(sys.argv[2], "WRITE_SIZE")
import bench  # noqa: E402

res = {"csrc_sha": bench.csrc_sha(),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1`; FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request of a wide coalesced stream); KiB units; "
               "produced by tools/pmc_traffic.py",
       "kernels": {}}
for k in GROUPS:
    if k in fetch or k in write:
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res["kernels"][k] = {"fetch_size_kib_raw": f, "write_size_kib": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
print(json.dumps(res, indent=1))
