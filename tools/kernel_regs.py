"""Print register / LDS / spill figures of every kernel in a hipcc -save-temps gfx950 assembly file."""
import re
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                     r"\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    ag, lds, name, scratch, sg, sgs, vg, vgs = m.groups()
    print(f"{name:70s} vgpr {vg:>4s} (agpr {ag:>3s}) sgpr {sg:>3s} sgpr_spill {sgs:>3s} vgpr_spill {vgs:>3s} scratch {scratch:>5s} lds {lds}")
