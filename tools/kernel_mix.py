"""Instruction mix of the main loop of one kernel in a gfx950 assembly file: python kernel_mix.py file.s <mangled-substring>"""
import collections, re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):", txt, re.M)
a = m.start(); b = txt.index(".Lfunc_end", a)
body = txt[a:b].split("\n")
start = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
end = max(i for i, l in enumerate(body) if "in Loop: Header" in l or "Parent Loop" in l)
# extend to the backward branch of the loop
loop = body[start:]
cnt = collections.Counter()
n = 0
for l in loop:
    mm = re.match(r"\s+([a-z_0-9]+)", l)
    if mm:
        cnt[mm.group(1)] += 1; n += 1
    if "s_endpgm" in l: break
valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
print(f"{m.group(1)}: from loop header to s_endpgm: {n} instrs, VALU {valu}, SALU {sum(v for k, v in cnt.items() if k.startswith('s_'))}, "
      f"LDS {sum(v for k, v in cnt.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in cnt.items() if k.startswith(('global_', 'scratch_', 'buffer_')))}")
print(", ".join(f"{k} {v}" for k, v in cnt.most_common(28)))
