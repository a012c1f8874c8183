"""Register / LDS / spill table of one HIP translation unit (compile-time, no GPU):
    python tools/kernel_resources.py pw_gemm [filter-substring]
Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
import os
import re
import subprocess
import sys

unit = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ood_object_detection_amd', 'csrc')
extra = sys.argv[3:] 
r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-fno-slp-vectorize', '-Rpass-analysis=kernel-resource-usage',
                    '-c', unit + '.hip', '-o', '/dev/null'] + extra, cwd=csrc, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r'remark: (?:Function Name: (\S+)|\s+([A-Za-z \[\]/]+): (\d+))', line)
    if not m:
        continue
    if m.group(1):
        cur = {'name': m.group(1)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(2).strip()] = int(m.group(3))
for c in rows:
    name = subprocess.run(['c++filt', c['name']], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print('%-110s VGPR %3d AGPR %3d spill %3d occ %d LDS %6d scratch %d' % (
        name[:110], c.get('VGPRs', -1), c.get('AGPRs', -1), c.get('VGPRs Spill', -1), c.get('Occupancy [waves/SIMD]', -1),
        c.get('LDS Size [bytes/block]', -1), c.get('ScratchSize [bytes/lane]', -1)))
if r.returncode:
    print(r.stderr[-3000:])
