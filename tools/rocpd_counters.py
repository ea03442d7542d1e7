"""Every counter of rocprofv3 rocpd .db files as a per-kernel table: average value per dispatch over launches 2.. (a first-touch launch
would skew it).  Usage: python tools/rocpd_counters.py db1 db2 ..."""
import sqlite3
import sys
from collections import defaultdict


def main(dbs):
    data = defaultdict(dict)
    order = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = defaultdict(list)
        for name, ctr, val, dur in cur.execute("select kernel_name, counter_name, value, duration from counters_collection order by dispatch_id"):
            name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            rows[(name, ctr)].append((val, dur))
            if name not in order:
                order.append(name)
        for (name, ctr), v in rows.items():
            tail = v[1:] if len(v) > 1 else v
            data[name][ctr] = (len(v), sum(x[0] for x in tail) / len(tail), sum(x[1] for x in tail) / len(tail) / 1e3)
    ctrs = sorted({c for d in data.values() for c in d})
    print('%-34s %5s %9s ' % ('kernel', 'calls', 'avg_us') + ' '.join('%22s' % c for c in ctrs))
    for name in order:
        d = data[name]
        any_ = next(iter(d.values()))
        print('%-34s %5d %9.1f ' % (name[:34], any_[0], any_[2]) + ' '.join('%22.1f' % d[c][1] if c in d else '%22s' % '-' for c in ctrs))


if __name__ == '__main__':
    main(sys.argv[1:])
