#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output written by build.py --asm."""
import glob, os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for path in sorted(glob.glob(os.path.join(here, "build", "*.resources.txt"))):
    cur = None
    rows = []
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("nvh::", "")}
            rows.append(cur)
            continue
        for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"):
            m = re.search(r" " + re.escape(key) + r": (\d+)", line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    print(os.path.basename(path))
    for r in rows:
        if len(sys.argv) > 1 and sys.argv[1] not in r["name"]:
            continue
        print(f"  {r['name']:<56} vgpr {r.get('VGPRs','?'):>3} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('SGPRs','?'):>3} "
              f"scratch {r.get('ScratchSize [bytes/lane]','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
