"""Per-kernel utilisation table from a `tools/pmc_gpu.sh` run (rocprofv3 --pmc with SQ counters).

usage: python tools/pmc_sq_table.py gpurun_out/rocprof_pmc_sq2_summary.txt <forwards> > profiles/<name>.txt

Columns: time per forward, vector-ALU wave-instructions per forward, SIMD cycles per VALU instruction, VALU busy
(SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles), MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles), LDS active
(SQ_LDS_IDX_ACTIVE / CU-cycles), share of wave cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES).
Clock assumed 2.4 GHz (profiled passes run lower, so the percentages are slightly underestimated)."""
import sys


def main():
    path, fw = sys.argv[1], float(sys.argv[2])
    txt = open(path).read()
    dur, data = {}, {}
    for sec in txt.split('\n# '):
        lines = sec.strip().split('\n')
        head = lines[0]
        if 'kernel-trace' in head:
            for l in lines[2:]:
                f = l.rsplit(None, 4)
                if len(f) == 5:
                    try:
                        dur[f[0].strip()] = (int(f[1]), float(f[2]))
                    except ValueError:
                        pass
        elif head.startswith('PMC'):
            name = head.split()[1].rstrip(':')
            for l in lines[2:]:
                f = l.rsplit(None, 3)
                if len(f) == 4:
                    try:
                        data.setdefault(f[0].strip(), {})[name] = float(f[2])
                    except ValueError:
                        pass
    print('%-56s %6s %9s %10s %7s %7s %7s %7s %7s' % ('kernel', 'calls', 'us/fwd', 'valu/fwd', 'cyc/i', 'VALU%', 'MFMA%', 'LDS%', 'wait%'))
    tot = 0.0
    for k, (calls, total_us) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        d = data.get(k)
        if not d or 'at::' in k or 'rocclr' in k:
            continue
        simd = total_us * 2400.0 * 1024.0
        v = d.get('SQ_INSTS_VALU', 0.0)
        tot += v
        print('%-56s %6.1f %9.1f %10.3g %7.2f %7.1f %7.1f %7.1f %7.1f' % (
            k[-56:], calls / fw, total_us / fw, v / fw, simd / max(v, 1.0),
            100 * 4 * d.get('SQ_ACTIVE_INST_VALU', 0.0) / simd, 100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / simd,
            100 * d.get('SQ_LDS_IDX_ACTIVE', 0.0) / (total_us * 2400.0 * 256.0),
            100 * d.get('SQ_WAIT_ANY', 0.0) / max(d.get('SQ_WAVE_CYCLES', 1.0), 1.0)))
    print('# vector-ALU wave-instructions per forward: %.3g' % (tot / fw))


if __name__ == '__main__':
    main()
