"""Timeline of the LAST forward in a rocprofv3 kernel trace (rocpd .db): per launch start offset, duration, gap to the previous
kernel's end, grid size in workgroups - for small-batch latency work (where does a B = 1 forward spend its time?).
usage: rocpd_timeline.py trace.db first_kernel_substring [out.txt]"""
import sqlite3
import sys


def main(db, first, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if len(starts) < 2:
        raise SystemExit('need at least two forwards in the trace')
    a, b = starts[-2], starts[-1]                    # the last COMPLETE forward: from its first kernel to the next forward's
    fw = rows[a:b]
    t0 = fw[0][1]
    span = (rows[b][1] - t0) / 1e3
    busy = sum(r[2] - r[1] for r in fw) / 1e3
    lines = [f'# forward of {len(fw)} launches: span {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us']
    lines.append('%9s %9s %7s %7s  %s' % ('start_us', 'dur_us', 'gap_us', 'blocks', 'kernel'))
    prev_end = None
    agg = {}
    for r in fw:
        gap = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
        name = r[0].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        lines.append('%9.1f %9.1f %7.1f %7d  %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3] // max(1, r[4]), name))
        k = agg.setdefault(name, [0, 0.0, 0.0])
        k[0] += 1; k[1] += (r[2] - r[1]) / 1e3; k[2] += max(0.0, gap)
        prev_end = max(prev_end or 0, r[2])
    lines.append('# by kernel: calls, total us, gap-before us')
    for n, k in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('%5d %9.1f %8.1f  %s' % (k[0], k[1], k[2], n))
    txt = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
