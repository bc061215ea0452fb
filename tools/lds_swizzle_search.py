import itertools
G = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G += [[l+32 for l in g] for g in G]
def read_cycles(addr_units):  # addr_units[lane] = 16B unit index; returns LDS cycles (sum over groups of max bank-quad multiplicity), 64 banks = 16 quads
    cyc = 0
    for g in G:
        cnt = {}
        for l in g:
            q = addr_units[l] % 16
            cnt.setdefault(q, set()).add(addr_units[l])
        cyc += max(len(v) for v in cnt.values())
    return cyc
def write_cycles(addr_units):  # 8 groups of 8 contiguous lanes, 32 banks = 8 quads
    cyc = 0
    for g in range(8):
        cnt = {}
        for l in range(g*8, g*8+8):
            q = addr_units[l] % 8
            cnt.setdefault(q, set()).add(addr_units[l])
        cyc += max(len(v) for v in cnt.values())
    return cyc
bases_A = sorted({r*18+dx for r in range(10) for dx in range(3)})
bases_B = [16*i for i in range(27)]
def evaluate(P, f):
    unit = lambda row, pc: row*P + (pc ^ f(row))
    ra = sum(read_cycles([unit(b + (l & 15), l >> 4) for l in range(64)]) for b in bases_A) / len(bases_A)
    rb = sum(read_cycles([unit(b + (l & 15), l >> 4) for l in range(64)]) for b in bases_B) / len(bases_B)
    wr = sum(write_cycles([unit(b + (l & 15), l >> 4) for l in range(64)]) for b in bases_B[:12]) / 12
    return ra, rb, wr
print("pitch 5 no swizzle", evaluate(5, lambda r: 0))
best = []
for P in (4, 5, 6, 8):
    for s in (0, 1, 2, 3):
        for g in itertools.product(range(4), repeat=4):
            f = lambda r, g=g, s=s: g[(r >> s) & 3]
            ra, rb, wr = evaluate(P, f)
            best.append((ra + rb * 1.5 + wr * 0.3, ra, rb, wr, P, s, g))
    for name, f in (("r>>2 ^ r>>4", lambda r: ((r >> 2) ^ (r >> 4)) & 3), ("(r>>1)&3", lambda r: (r >> 1) & 3), ("r>>2+r>>3", lambda r: ((r >> 2) + (r >> 3)) & 3)):
        ra, rb, wr = evaluate(P, f); best.append((ra + rb*1.5 + wr*0.3, ra, rb, wr, P, name, None))
best.sort(key=lambda t: t[0])
for b in best[:12]: print(b)
print("--- gemm plain tile, pitch 9 units, 8 pieces/row")
for ks in (0, 1):
    print(ks, read_cycles([( (l & 15)) * 9 + ks * 4 + (l >> 4) for l in range(64)]))
# candidates for gemm: pitch 8 (128 B rows, unpadded) with xor swizzle on the 8 pieces
for s in (0,1,2):
  for mask in (3, 7):
    f = lambda r: (r >> s) & mask
    rd = [read_cycles([((l & 15)) * 8 + ((ks * 4 + (l >> 4)) ^ f(l & 15)) for l in range(64)]) for ks in (0,1)]
    wr = write_cycles([ (l >> 3) * 8 + ((l & 7) ^ f(l >> 3)) for l in range(64)])   # write: id -> row = id/8, kc = id%8
    print("P=8 s", s, "mask", mask, rd, "write", wr)
for P in (9, 10, 12):
    wr = write_cycles([ (l >> 3) * P + (l & 7) for l in range(64)])
    print("P", P, [read_cycles([(l & 15) * P + ks*4 + (l >> 4) for l in range(64)]) for ks in (0,1)], "write", wr)
