"""Kernel timeline of one steady training step from a rocprofv3 rocpd database (kernel trace of bench.py).

    python tools/step_timeline.py gpurun_out/prof/bench_results.db [which_fwd_call]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = db.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "loss_fwd_dense" in r[0]]
a, b = idx[which], idx[which + 1]
prev_end = None
tail = 0.0
for r in rows[a:b]:
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0.0
    dur = (r[2] - r[1]) / 1e3
    if "loss_fwd_dense" not in r[0] and "loss_bwd_dense" not in r[0]:
        tail += dur + max(gap, 0.0)
    print(f"{r[0][:64]:64s} dur {dur:7.1f} gap {gap:6.1f}")
    prev_end = r[2]
print(f"step {(rows[b][1] - rows[a][1]) / 1e3:.1f} us, outside the three sweeps {tail:.1f} us")
