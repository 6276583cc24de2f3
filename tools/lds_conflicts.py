import itertools
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[l+32 for l in g] for g in G128]
G2x32 = [list(range(32)), list(range(32,64))]
G4x16 = [list(range(16*i,16*i+16)) for i in range(4)]
lanes = range(64)
def cyc(addrs, width, groups, nbanks):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            for d in range(width // 4):
                banks.setdefault(((a // 4) + d) % nbanks, set()).add((a // 4) + d)
        tot += max(len(v) for v in banks.values())
    return tot
def addr(row, col_u16, stride_b, f):
    """byte address of element (row, col) with 16-byte piece swizzle f(row)"""
    piece, within = divmod(col_u16 * 2, 16)
    if piece < 8: piece ^= f(row)
    return row * stride_b + piece * 16 + within
best = []
for stride in (128, 144):
  for name, f in [("none", lambda r: 0)] + [(f"(r>>{a})&{m}", (lambda a, m: (lambda r: (r >> a) & m))(a, m)) for a in range(0, 4) for m in (1, 3, 7)] + \
                 [(f"((r>>{a})^(r>>{b}))&7", (lambda a, b: (lambda r: ((r >> a) ^ (r >> b)) & 7))(a, b)) for a in range(0,3) for b in range(a+1,5)] + \
                 [(f"(r+(r>>3))&7", lambda r: (r + (r >> 3)) & 7), ("(r*3)&7", lambda r: (r*3) & 7), ("((r&7)*1 ^ (r>>3)&1... )", lambda r: (r & 7) ^ (((r >> 3) & 1) * 4))]:
    res = {}
    # operand reads, worst over tn, ks
    res['b128'] = max(cyc([addr(tn*16 + (l&15), ks*32 + (l>>4)*8, stride, f) for l in lanes], 16, G128, 64) for tn in range(4) for ks in range(2))
    res['tr'] = max(cyc([addr(ks*32 + (l>>4)*8 + ((l&15)>>2) + h*4, tn*16 + (l&3)*4, stride, f) for l in lanes], 8, G2x32, 64) for tn in range(4) for ks in range(2) for h in range(2))
    res['stw'] = max(cyc([addr(st*16 + (l&15), tn*16 + (l>>4)*4, stride, f) for l in lanes], 8, G4x16, 32) for st in range(4) for tn in range(4))
    res['str'] = max(cyc([addr(p*8 + (l>>3), (l&7)*8, stride, f) for l in lanes], 16, G128, 64) for p in range(8))
    tot = res['b128']/4 + res['tr']/2 + res['stw']/4 + res['str']/4
    best.append((tot, stride, name, res))
best.sort(key=lambda x: x[0])
for b in best[:12]: print(b)
