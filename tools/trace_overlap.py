"""Concurrency in a rocprofv3 --kernel-trace CSV: sum of kernel durations against the length of their union (wall time with at least
one kernel running) and the time with >= 2 kernels running.  usage: python3 tools/trace_overlap.py <dir with *kernel_trace.csv> [last_ms]"""
import csv, glob, os, sys

root = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
iv = []
for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
iv.sort()
if last_ms:
    t_end = max(e for _, e in iv)
    iv = [(s, e) for s, e in iv if s >= t_end - last_ms * 1e6]
total = sum(e - s for s, e in iv)
ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
depth, prev, busy1, busy2, peak = 0, ev[0][0], 0, 0, 0
for t, d in ev:
    if depth >= 1:
        busy1 += t - prev
    if depth >= 2:
        busy2 += t - prev
    depth += d
    peak = max(peak, depth)
    prev = t
span = iv[-1][1] - iv[0][0] if iv else 0
print('dispatches %d  span %.3f ms  sum of durations %.3f ms  union %.3f ms  >=2 kernels running %.3f ms (%.1f %% of union)  peak concurrency %d'
      % (len(iv), span / 1e6, total / 1e6, busy1 / 1e6, busy2 / 1e6, 100.0 * busy2 / max(busy1, 1), peak))
