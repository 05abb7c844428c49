"""Summarise a rocprofv3 --kernel-trace --stats SQLite output (top_kernels view) as text for profiles/."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"{'kernel':84s} {'calls':>7s} {'total_s':>10s} {'avg_us':>10s} {'pct':>6s}")
for name, calls, tot, avg, pct in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    n = re.sub(r"\(anonymous namespace\)::", "", str(name))
    n = re.sub(r"\(.*", "", n)[:84]
    print(f"{n:84s} {calls:7d} {tot / 1e6:10.4f} {avg:10.1f} {pct:6.2f}")
