"""Per-kernel-family HBM traffic from the rocprofv3 FETCH_SIZE / WRITE_SIZE summaries (tools/profile_gpu.sh).

usage: python tools/pmc_traffic.py <fetch_summary.txt> <write_summary.txt> <forwards | 0 = from the stem kernel's call count> <out.json> [workload]

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM/rocprofv3 section), so reads are doubled.  `forwards` = number of forward passes
the profiled command dispatched (bench.py --steps 5 --warmup 2 --no-graph: 7 + 1 build pass + 8 in the
per-launch profiling pass = 16 dispatches per launch site)."""
import json, sys

FAMILIES = {'mbconv': ('mbconv_front_kernel', 'mbconv_deep_kernel', 'mbconv_roll_kernel', 'mbconv_wide_kernel'), 'sepconv': ('sepconv_kernel',),
            'pw_gemm': ('pw_gemm_kernel',), 'stem_dw': ('stem_dw_kernel', 'stem_roll_kernel'), 'se_gate': ('se_gate_kernel',),
            'topk': ('topk_', 'anchor_collect', 'pair_finish', 'row_max'), 'nms': ('nms_', 'decode_threshold', 'gather_ood')}


def pmc_section(path, counter):
    out, on = {}, False
    for line in open(path):
        if line.startswith('# PMC'):
            on = counter in line
            continue
        if line.startswith('#'):
            on = False
        if on and line.strip() and not line.startswith('kernel'):
            f = line.rsplit(None, 3)
            if len(f) == 4:
                try:
                    out[f[0].strip()] = float(f[2])
                except ValueError:
                    pass
    return out


def calls_of(path, pattern):
    """dispatch count of the first kernel whose name contains `pattern` in a PMC section (one stem launch per forward)"""
    on = False
    for line in open(path):
        if line.startswith('# PMC'):
            on = True
            continue
        if on and pattern in line:
            f = line.rsplit(None, 3)
            if len(f) == 4:
                return int(f[1])
    return 0


def main():
    import os
    fetch, write, forwards, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    workload = sys.argv[5] if len(sys.argv) > 5 else ''
    if forwards <= 0:
        forwards = calls_of(fetch, 'stem_roll_kernel') or calls_of(fetch, 'stem_dw_kernel') or calls_of(fetch, 'stem_conv')
    rd, wr = pmc_section(fetch, 'FETCH_SIZE'), pmc_section(write, 'WRITE_SIZE')
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res = {'forwards': forwards, 'workload': workload, 'kernels_sha16': bench.kernels_sha16(), 'note': 'bytes per forward pass; reads = 2 x FETCH_SIZE KiB (gfx950), writes = WRITE_SIZE KiB', 'families': {}}
    for fam, pats in FAMILIES.items():
        r = sum(v for k, v in rd.items() if any(p in k for p in pats)) * 1024 * 2 / forwards
        w = sum(v for k, v in wr.items() if any(p in k for p in pats)) * 1024 / forwards
        res['families'][fam] = {'read_bytes': int(r), 'write_bytes': int(w), 'bytes': int(r + w)}
    json.dump(res, open(dst, 'w'), indent=1)
    print(json.dumps(res['families']))


if __name__ == '__main__':
    main()
