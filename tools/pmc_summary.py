"""Per-kernel averages of the hardware counters in a rocprofv3 --pmc run (rocpd sqlite .db).
usage: python tools/pmc_summary.py <results.db> [kernel-name substring, default 'nrdhip']
Counter values are summed over the instances of a counter (XCDs / channels) per dispatch, then averaged over dispatches."""
import collections
import sqlite3
import sys


def main():
    db, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nrdhip")
    c = sqlite3.connect(db)
    rows = c.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events where name like ? group by name, dispatch_id, counter_name", ("%" + flt + "%",))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for name, disp, counter, value, duration in rows:
        acc[name][counter].append(value)
        dur[name].append(duration)
    counters = sorted({k for v in acc.values() for k in v})
    print("%-84s %6s %10s " % ("kernel", "calls", "avg_us") + " ".join("%16s" % k[:16] for k in counters))
    for name in sorted(acc, key=lambda n: -sum(dur[n])):
        n = max(len(v) for v in acc[name].values())
        print("%-84s %6d %10.1f " % (name.replace("void ", "").replace("nrdhip::", "").replace("(anonymous namespace)::", "")[:84], n, sum(dur[name]) / len(dur[name]) / 1e3) +
              " ".join("%16.4g" % (sum(acc[name][k]) / len(acc[name][k])) if k in acc[name] else "%16s" % "-" for k in counters))


if __name__ == "__main__":
    main()
