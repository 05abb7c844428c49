"""Bank-conflict model of the attention kernels' LDS tiles (64 rows x 128 B, eight 16-byte chunks per row XOR-swizzled by a key of the
row): 64 banks x 4 B; ds_read_b64_tr_b16 is serviced in 2 groups of 32 lanes, ds_read_b128 in 4 groups of 16 lanes
(MI355X_MICROARCH.md, LDS table).  Prints the worst and mean number of distinct addresses per bank for the transposing V^T / K^T / Q^T
reads and the fragment reads, for the key used through round 4 and the round-5 key (csrc/attention.hip, swz_key)."""
from collections import defaultdict


def swz_old(row, chunk): return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
def swz_new(row, chunk): return row * 128 + ((chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4)


def tr_addr(swz, row_base, col_base16, lane):
    i = lane & 15
    row, col = row_base + (i >> 2), col_base16 + 4 * (i & 3)
    return swz(row, col >> 3) + ((col & 7) << 1)


def ways(addrs, nbytes):
    b = defaultdict(set)
    for a in addrs:
        for k in range(0, nbytes, 4):
            b[((a + k) // 4) % 64].add((a + k) // 4)
    return max(len(v) for v in b.values())


for name, swz in (("key (row >> 1) & 7 (rounds 1-4)", swz_old), ("key of round 5", swz_new)):
    w = []
    for kt in range(2):
        for h2 in range(2):
            for j in range(2):
                for dt in range(2):
                    for grp in range(2):
                        w.append(ways([tr_addr(swz, 32 * kt + 16 * h2 + 4 * (lane >> 5) + 8 * j, 32 * dt + 16 * ((lane >> 4) & 1), lane)
                                       for lane in range(32 * grp, 32 * grp + 32)], 8))
    g0 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups = g0 + [[l + 32 for l in g] for g in g0]
    f = [ways([swz(32 * kt + (l & 31), 2 * s + (l >> 5)) for l in g], 16) for kt in range(2) for s in range(4) for g in groups]
    print(f"{name}: transposing b64 reads worst {max(w)}-way, mean {sum(w) / len(w):.2f}; b128 fragment reads worst {max(f)}-way")
