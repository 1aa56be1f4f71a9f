"""Developer aid: VGPRs / SGPRs / LDS / scratch / occupancy of every kernel of csrc/ (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/resource_usage.py [substring ...]"""
import os
import re
import subprocess
import sys

CS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fletcherpenaltysolver.jl_amd", "csrc")
out = subprocess.run(["make", "-s", "-C", CS, "resource-usage"], capture_output=True, text=True).stdout
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur:
            rows[cur].setdefault(key, int(m.group(1)))
for k, v in rows.items():
    if len(sys.argv) > 1 and not any(a in k for a in sys.argv[1:]):
        continue
    print(f"{k[:90]:90s} vgpr {v.get('vgpr')} sgpr {v.get('sgpr')} occ {v.get('occ')} lds {v.get('lds')} scratch {v.get('scratch')}")
