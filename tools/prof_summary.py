"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel calls / average / share, and the
busy time of the GPU (union of kernel intervals) over the traced span.

    python tools/prof_summary.py gpurun_out/prof/x_results.db [top_n] [name_filter_for_window]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"{'kernel':72s} calls   avg_us  share%")
for r in rows[:top]:
    print(f"{r[0][:72]:72s} {r[1]:5d} {r[2] / 1e3:8.1f} {100 * r[3] / tot:6.2f}")
iv = db.execute("select start, end, stream_id from kernels order by start").fetchall()
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels: {len(iv)}  sum of durations {tot / 1e6:.3f} ms  union (GPU busy) {busy / 1e6:.3f} ms  span {(iv[-1][1] - iv[0][0]) / 1e6:.3f} ms"
      f"  streams {len({x[2] for x in iv})}")
