#!/usr/bin/env python
"""Resource usage per kernel / function of a gfx950 .s dump (tools/lab/isa.sh): code size, SGPRs, VGPRs, scratch, LDS, occupancy."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.size\t(\S+), \.Lfunc_end\d+-\S+\n\s*; -- End function\n.*?; codeLenInByte = (\d+)\n; TotalNumSgprs: (\d+)\n; NumVgprs: (\d+)\n.*?; ScratchSize: (\d+)\n.*?; LDSByteSize: (\d+).*?; Occupancy: (\d+)", txt, re.S):
    dn = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-62s code %6s sgpr %3s vgpr %3s scratch %4s lds %6s occ %s" % ((dn[:62],) + m.groups()[1:]))
