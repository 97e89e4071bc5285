"""Which plane row stride / 16-byte-unit XOR swizzle makes the LDS accesses of csrc/chain2.hip conflict-free?
Brute force over (stride, swizzle): worst conflict degree of the ds_read_b128 fragment reads (b128 lane groups of
MI355X_MICROARCH.md, 64 banks) and of the ds_write_b64 epilogue writes (16 contiguous lanes, 32 banks).
Result: reads 1-way / writes 2-way is the optimum; stride 256 B with unit ^= row & 15 reaches it without padding."""
R128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
R128 = R128 + [[l + 32 for l in g] for g in R128]
W64 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]


def degree(addrs, groups, width, nbanks):
    worst = 1
    for g in groups:
        cnt = {}
        for l in g:
            for b in range(addrs[l] // 4, (addrs[l] + width) // 4):
                cnt.setdefault(b % nbanks, set()).add(addrs[l])
        worst = max(worst, max(len(v) for v in cnt.values()))
    return worst


SW = {"none": lambda m: 0, "m>>2": lambda m: (m >> 2) & 3, "m>>1": lambda m: (m >> 1) & 3, "m&3": lambda m: m & 3,
      "(m>>1)&7": lambda m: (m >> 1) & 7, "m&7": lambda m: m & 7, "m&15": lambda m: m & 15}
rows = []
for stride in range(256, 400, 16):
    for name, sw in SW.items():
        rd = wr = 1
        for c in range(4):
            rd = max(rd, degree({l: (l & 15) * stride + ((4 * c + (l >> 4)) ^ sw(l & 15)) * 16 for l in range(64)}, R128, 16, 64))
        for w in range(8):
            a = {}
            for l in range(64):
                n0 = 16 * w + 4 * (l >> 4)
                a[l] = (l & 15) * stride + ((n0 // 8) ^ sw(l & 15)) * 16 + (n0 % 8) * 2
            wr = max(wr, degree(a, W64, 8, 32))
        rows.append((rd, wr, stride, name))
for r in sorted(rows)[:10]:
    print("read %d-way  write %d-way  stride %d B  swizzle %s" % r)
