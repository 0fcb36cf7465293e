"""Per message size: kernel durations of the all-reduce calls in a rocprofv3 trace of
test_allreduce.py (groups of `w` consecutive dispatches = one call on w virtual ranks)."""
import sqlite3, statistics as st, sys
db = sqlite3.connect(sys.argv[1]); w = int(sys.argv[2]); per_size = int(sys.argv[3]); skip = int(sys.argv[4])
rows = db.execute("select name,start,end,grid_x from kernels where name like '%allreduce%' order by start").fetchall()
calls = [rows[i:i + w] for i in range(0, len(rows), w)]
for ci in range(0, len(calls), per_size):
    grp = calls[ci + skip:ci + per_size]
    mins = [min((e - s) / 1e3 for (_, s, e, _) in c) for c in grp]
    maxs = [max((e - s) / 1e3 for (_, s, e, _) in c) for c in grp]
    kind = "one_shot" if "one_shot" in grp[0][0][0] else "two_shot"
    print(f"{kind} workgroups {grp[0][0][3] // 512:3d}: last-started rank {st.median(mins):7.1f} us   first-started rank {st.median(maxs):7.1f} us")
