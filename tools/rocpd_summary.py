#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace rocpd database (SQLite) as the per-kernel stats table
rocprofv3 --stats would print: calls, total / average / min / max duration, share of GPU time.

    python tools/rocpd_summary.py gpurun_out/prof_celeba/celeba_results.db > profiles/r01_celeba_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name[:150]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    rows = c.execute('select name, start, end from kernels').fetchall() if 'name' in cols else []
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print('# source: %s   (rocprofv3 --kernel-trace --stats, durations in microseconds)' % path)
    print('%-8s %12s %10s %10s %10s %7s  %s' % ('calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'kernel'))
    for name, (n, tot, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-8d %12.1f %10.2f %10.2f %10.2f %6.2f%%  %s' % (n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3,
                                                                 100.0 * tot / total, name))


def timeline(path, tail=0.5):
    """Busy / idle analysis of the last ``tail`` fraction of the trace (the timed, graph-replayed
    steps): span, union of kernel intervals (GPU busy), sum of durations (> union when streams
    overlap), and the idle gaps between kernels."""
    c = sqlite3.connect(path)
    rows = sorted(c.execute('select start, end, name from kernels').fetchall())
    rows = rows[int(len(rows) * (1.0 - tail)):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _ in rows)
    gaps.sort()
    print('# timeline of the last %d kernels of %s' % (len(rows), path))
    print('span_us %.1f  busy_us %.1f (%.1f%%)  sum_of_durations_us %.1f (overlap x%.2f)  kernels %d' % (
        (t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0), tot / 1e3, tot / max(busy, 1), len(rows)))
    if gaps:
        print('idle gaps: n %d  total_us %.1f  median_us %.2f  p90_us %.2f  max_us %.1f' % (
            len(gaps), sum(gaps) / 1e3, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))


def one_step(path, marker='ingest_kernel'):
    """One graph-replayed step, kernel by kernel: the shortest interval between two consecutive ``marker`` launches
    (once per replay) -- start offset, duration, queue / stream column if the database has one, name."""
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    lane_col = next((k for k in ('stream_id', 'queue_id', 'stream', 'queue') if k in cols), None)
    q = 'select start, end, name%s from kernels order by start' % (', ' + lane_col if lane_col else '')
    rows = c.execute(q).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        print('# no %s launches: columns %s' % (marker, cols)); return
    a, b = min(zip(marks[:-1], marks[1:]), key=lambda ab: rows[ab[1]][0] - rows[ab[0]][0])
    t0 = rows[a][0]
    print('# one step (%d kernels, %.1f us between two %s launches); columns: offset_us dur_us %s name' % (
        b - a, (rows[b][0] - t0) / 1e3, marker, lane_col or '-'))
    for r in rows[a:b]:
        print('%8.1f %7.1f %6s  %s' % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3] if lane_col else '-', short(r[2])[:110]))


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == '--step':
    one_step(sys.argv[1], *(sys.argv[3:4]))
elif __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == '--timeline':
    timeline(sys.argv[1])
elif __name__ == '__main__' and not (len(sys.argv) > 2 and sys.argv[2] == '--pmc'):
    main(sys.argv[1])


def pmc_summary(path):
    """Per-kernel average of a --pmc counter pass (counters_collection view), values as reported
    by rocprofv3 (FETCH_SIZE / WRITE_SIZE are kilobytes)."""
    c = sqlite3.connect(path)
    rows = c.execute('select kernel_name, counter_name, value from counters_collection').fetchall()
    agg = {}
    for name, counter, value in rows:
        a = agg.setdefault((short(name), counter), [0, 0.0])
        a[0] += 1; a[1] += value
    print('# source: %s   (rocprofv3 --pmc, per-dispatch averages)' % path)
    print('%-8s %-12s %14s %14s  %s' % ('calls', 'counter', 'avg', 'total', 'kernel'))
    for (name, counter), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-8d %-12s %14.1f %14.1f  %s' % (n, counter, tot / n, tot, name))


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == '--pmc':
    pmc_summary(sys.argv[1])
