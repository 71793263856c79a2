#!/usr/bin/env python
"""Register / LDS / scratch figures of every kernel in a hipcc -S --cuda-device-only listing (the amdhsa metadata block).
    python tools/isa_regs.py /tmp/h3d.s [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in txt.split("- .agpr_count:")[1:]:
    ag = int(b.split("\n")[0])
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1))
    sg = int(re.search(r"\.sgpr_count:\s+(\d+)", b).group(1))
    sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1))
    scr = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void ", "").split("(")[0]
    if flt in dn:
        print(f"{dn:64s} vgpr {vg:3d} agpr {ag:3d} sgpr {sg:3d} spill {sp} scratch {scr}")
