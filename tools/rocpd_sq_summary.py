"""Dump every PMC counter per kernel from rocprofv3 rocpd .db files, normalised per wave-cycle where that makes sense.
Usage: python tools/rocpd_sq_summary.py db1 [db2 ...] [--filter substr]"""
import sqlite3
import sys
from collections import defaultdict


def main(argv):
    flt = None
    if '--filter' in argv:
        i = argv.index('--filter')
        flt = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    data = defaultdict(dict)
    for db in argv:
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by kernel_name, counter_name")
        for name, ctr, n, val, dur in cur.execute(q):
            name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            data[name][ctr] = val
            data[name]['_n'] = n
            data[name]['_dur_us'] = dur / 1e3
    for name in sorted(data, key=lambda k: -data[k]['_n'] * data[k]['_dur_us']):
        if flt and flt not in name:
            continue
        d = data[name]
        if d['_dur_us'] * d['_n'] < 500:
            continue
        print('%s  calls=%d avg_us=%.1f' % (name[:90], d['_n'], d['_dur_us']))
        gui = d.get('GRBM_GUI_ACTIVE', 0) / 8.0
        wc = d.get('SQ_WAVE_CYCLES')
        for c in sorted(d):
            if c.startswith('_'):
                continue
            extra = ''
            if gui and c.startswith('SQ_'):
                extra += '  per_simd_cycle=%.4f' % (d[c] / (gui * 1024))
            if wc and c.startswith('SQ_') and c != 'SQ_WAVE_CYCLES':
                extra += '  per_wave_cycle=%.4f' % (d[c] / wc)
            print('    %-34s %16.1f%s' % (c, d[c], extra))


if __name__ == '__main__':
    main(sys.argv[1:])
