"""Instruction histogram of one kernel in a hipcc -S listing: python tools/isa_count.py file.s <substring of the mangled name> [--loop]
--loop: only the largest loop body (from the first 'Loop Header' label to the last branch back to it)."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:\s*(;.*)?$", l) and pat in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
if "--loop" in sys.argv:      # the instructions of every block that belongs to the first depth-1 loop (blocks are tagged by the compiler's comments)
    hdr = next(l for l in body if "Loop Header: Depth=1" in l).split(":")[0].lstrip(".L")
    keep, inside = [], False
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb\.\d+:", l):
            inside = ("Header=" + hdr + " ") in l + " " or l.startswith(".L" + hdr + ":")
        elif inside: keep.append(l)
    body = keep
ops = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
c = collections.Counter(ops)
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print(f"total {len(ops)}  VALU {valu}  s_nop {c['s_nop']}  v_mov {c['v_mov_b32_e32'] + c['v_mov_b64_e32']}  exp {c['v_exp_f32_e32']}  lds {sum(v for k, v in c.items() if k.startswith('ds_'))}  waitcnt {c['s_waitcnt']}")
if "-v" in sys.argv:
    for k, v in c.most_common(40): print(f"  {v:5d} {k}")
for l in lines[end:end + 80]:
    if re.search(r"NumVgprs|ScratchSize|Occupancy|LDSByteSize", l): print(l.strip())
