"""Summarise a rocprofv3 rocpd .db (kernel trace) into the per-kernel stats table kept under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = [f'# rocprofv3 --kernel-trace --stats summary of {db}',
             f'# total kernel time {total:.2f} ms over {sum(r[1] for r in rows)} dispatches',
             '%-100s %6s %10s %6s %10s %10s %10s %9s %5s %7s %5s %5s' % (
                 'kernel', 'calls', 'total_ms', 'pct', 'avg_us', 'min_us', 'max_us', 'grid_x', 'wg', 'lds', 'vgpr', 'sgpr')]
    for r in rows:
        lines.append('%-100s %6d %10.3f %6.2f %10.1f %10.1f %10.1f %9d %5d %7d %5d %5d' % (
            r[0][:100], r[1], r[2], 100 * r[2] / total, r[3], r[4], r[5], r[6], r[7], r[8], r[9] + r[10], r[11]))
    txt = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
