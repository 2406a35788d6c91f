#!/usr/bin/env python3
"""Turn rocprofv3's rocpd (SQLite) outputs into the small text summaries kept under profiles/.

  rocpd_summary.py stats  <kt_results.db>                 -> CSV: name, calls, total us, average us, %, min, max  (--kernel-trace --stats run)
  rocpd_summary.py pmc    <results.db> <COUNTER> [kernel-substring] [last_n]
                                                          -> per-dispatch counter values of the kernels whose name contains the
                                                             substring: all-mean and the mean of the last `last_n` dispatches
                                                             (steady state: bench.py's post-burn-in launches)
  rocpd_summary.py timeline <kt_results.db> [last_n]      -> the last `last_n` dispatches in start order: start offset (us), duration,
                                                             gap since the previous dispatch ended, kernel name (what a short timed
                                                             window really contains)
"""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                      "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print('"Name","Calls","TotalDurationUs","AverageUs","Percentage","MinUs","MaxUs"')
    for name, n, s, a, lo, hi in rows:
        if len(name) > 160:
            name = name[:157] + "..."
        print('"%s",%d,%.3f,%.3f,%.2f,%.3f,%.3f' % (name, n, s / 1e3, a / 1e3, 100.0 * s / tot, lo / 1e3, hi / 1e3))


def pmc(path, counter, sub="phase_fused_kernel", last_n=40):
    db = sqlite3.connect(path)
    vals = [r[0] for r in db.execute("select value from counters_collection where counter_name = ? and kernel_name like ? order by dispatch_id",
                                     (counter, "%" + sub + "%"))]
    if not vals:
        print("no %s samples for kernels matching %r" % (counter, sub))
        return
    tail = vals[-last_n:]
    print("%s per %s dispatch: n=%d all-mean=%.1f steady(last %d dispatches)-mean=%.1f min=%.1f max=%.1f"
          % (counter, sub, len(vals), sum(vals) / len(vals), len(tail), sum(tail) / len(tail), min(tail), max(tail)))


def timeline(path, last_n=80):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()[-last_n:]
    t0, prev_end = rows[0][1], None
    print("start_us  dur_us  gap_us  kernel")
    for name, st, en in rows:
        short = name.split("(")[0][-70:]
        print("%9.2f %7.2f %7.2f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, 0.0 if prev_end is None else (st - prev_end) / 1e3, short))
        prev_end = en


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "timeline":
        timeline(sys.argv[2], *([int(sys.argv[3])] if len(sys.argv) > 3 else []))
    elif len(sys.argv) >= 4 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3], *(sys.argv[4:5] or ["phase_fused_kernel"]), *([int(sys.argv[5])] if len(sys.argv) > 5 else []))
    else:
        sys.exit(__doc__)
