"""Summarise a rocprofv3 --kernel-trace (rocpd sqlite .db) into a per-kernel stats table (avg/min/max, registers, LDS).
usage: python tools/rocprof_summary.py <results.db> [substring filter, default 'nrdhip']"""
import sqlite3
import sys


def main():
    db, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nrdhip")
    c = sqlite3.connect(db)
    q = ("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
         "max(lds_size), max(scratch_size), max(grid_x), max(grid_y), max(workgroup_x) from kernels where name like ? group by name order by 6 desc")
    rows = list(c.execute(q, ("%" + flt + "%",)))
    total = sum(r[5] for r in rows)
    print("%-96s %6s %10s %10s %10s %7s %5s %5s %6s %7s %14s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "scratch", "grid(threads)"))
    for r in rows:
        print("%-96s %6d %10.1f %10.1f %10.1f %6.1f%% %5d %5d %6d %7d %14s" % (r[0].replace("void ", "")[:96], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / total, r[6], r[8], r[9], r[10],
                                                                            "%dx%d/%d" % (r[11], r[12], r[13])))


if __name__ == "__main__":
    main()
