#!/usr/bin/env python3
"""Register / LDS use of the kernels in a gfx950 code object whose name contains a pattern:
   llvm-objdump --offloading tts_amd/csrc/t2_decoder.o   (extracts t2_decoder.o.0.hipv4-...gfx950)
   llvm-readelf --notes <that file> > notes.txt ; tools/kernel_regs.py notes.txt PartOp"""
import re
import sys

t = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n\s+- \.agpr_count:", t)[1:]:
    blk = ".agpr_count:" + blk
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m or pat not in m.group(1):
        continue
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print(m.group(1)[:120], "vgpr", g("vgpr_count"), "agpr", g("agpr_count"), "spill", g("vgpr_spill_count"),
          "scratch", g("private_segment_fixed_size"), "lds", g("group_segment_fixed_size"))
