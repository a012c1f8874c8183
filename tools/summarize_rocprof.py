#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace / stats / PMC) into a small text summary for profiles/."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name[:110]


def main(root, out):
    lines = []
    traces = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
    agg = defaultdict(lambda: [0, 0])
    by_grid = defaultdict(lambda: [0, 0])
    for f in traces:
        for r in csv.DictReader(open(f)):
            k = short(r.get('Kernel_Name', '?'))
            dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            agg[k][0] += 1
            agg[k][1] += dur
            g = tuple(r.get('Grid_Size_' + a, '?') for a in 'XYZ')
            by_grid[(k[:60], g)][0] += 1
            by_grid[(k[:60], g)][1] += dur
    if agg:
        tot = sum(v[1] for v in agg.values())
        lines.append('# rocprofv3 --kernel-trace: per kernel (all dispatches of the run)')
        lines.append('%-112s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append('%-112s %8d %12.1f %10.2f %6.2f' % (k, n, t / 1e3, t / 1e3 / n, 100.0 * t / tot))
    if by_grid and os.environ.get('ROCPROF_BY_GRID'):
        lines.append('')
        lines.append('# the same dispatches grouped by (kernel, grid size in work-items): top %s' % os.environ['ROCPROF_BY_GRID'])
        lines.append('%-62s %-24s %8s %12s %10s' % ('kernel', 'grid', 'calls', 'total_us', 'avg_us'))
        for (k, g), (n, t) in sorted(by_grid.items(), key=lambda kv: -kv[1][1])[:int(os.environ['ROCPROF_BY_GRID'])]:
            lines.append('%-62s %-24s %8d %12.1f %10.2f' % (k, 'x'.join(g), n, t / 1e3, t / 1e3 / n))
    stats = glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True)
    for f in stats:
        lines.append('')
        lines.append('# rocprofv3 --stats (%s)' % os.path.basename(f))
        rows = list(csv.reader(open(f)))
        for r in rows[:40]:
            lines.append(' | '.join(c[:100] for c in r))
    pmc = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
    cagg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in pmc:
        for r in csv.DictReader(open(f)):
            k = short(r.get('Kernel_Name', '?'))
            c = r.get('Counter_Name', '?')
            cagg[c][k][0] += 1
            cagg[c][k][1] += float(r.get('Counter_Value', 0))
    for c, per in cagg.items():
        lines.append('')
        lines.append('# PMC %s: per kernel sum over dispatches (raw counter units; FETCH_SIZE/WRITE_SIZE are KiB;'
                     ' gfx950: double FETCH_SIZE for wide coalesced reads - MI355X_MICROARCH.md)' % c)
        lines.append('%-112s %8s %16s %16s' % ('kernel', 'calls', 'sum', 'per_call'))
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            lines.append('%-112s %8d %16.1f %16.1f' % (k, n, v, v / n))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('wrote', out, len(lines), 'lines')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
