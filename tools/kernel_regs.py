"""Parse `llvm-readelf --notes` of a gfx950 code object: per-kernel VGPR / AGPR / SGPR / spill / LDS / scratch table."""
import re
import sys

txt = sys.stdin.read()
pat = sys.argv[1] if len(sys.argv) > 1 else "."
for blk in txt.split("- .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name, agpr = g("name"), blk.split()[0]
    line = "%-100s vgpr %4s agpr %4s sgpr %4s spill %3s lds %7s scratch %5s" % (name[:100], g("vgpr_count"), agpr, g("sgpr_count"), g("vgpr_spill_count"),
                                                                                g("group_segment_fixed_size"), g("private_segment_fixed_size"))
    if re.search(pat, line):
        print(line)
