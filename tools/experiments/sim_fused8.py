#!/usr/bin/env python3
"""CPU model of the index algebra of the radix-8 / 512-lane fused commitment tile kernel (lsr_commit_fused.hpp):
lane maps, LDS slot weights, twiddle indices.  Checks (1) the network equals the plain stage loop of the tile kernel's
definition (stage of index bit b uses table[2^(L-1-b) + (pos >> (b+1))]), forward and inverse, and (2) every LDS access
pattern is bank-conflict free (ds_read_b64: 32 lanes over 32 8-byte banks; ds_write_b64: 16 lanes over 16)."""
import random

L = 16
TILE = 4096
T = 512
W = [1, 2, 4, 8, 16, 32, 72, 144, 289, 578, 1156, 2304]
LO = [9, 6, 3, 0]


def slot(idx):
    return sum(W[j] for j in range(12) if (idx >> j) & 1)


def base(r, t):
    if r == 0:
        return t
    if r == 1:
        return (t & 63) | ((t >> 6) << 9)
    if r == 2:
        return (t & 7) | ((t >> 3) << 6)
    return (((t >> 3) & 31) << 3) | ((t & 7) << 8) | ((t >> 8) << 11)


def check_maps():
    slots = [slot(i) for i in range(TILE)]
    assert len(set(slots)) == TILE and max(slots) < 4608
    for r in range(4):
        seen = set()
        for t in range(T):
            for k in range(8):
                seen.add(base(r, t) | (k << LO[r]))
        assert len(seen) == TILE, r
        for k in range(8):
            for g in range(0, T, 32):   # reads: 32-lane groups over 32 banks
                banks = [(slot(base(r, t) | (k << LO[r]))) % 32 for t in range(g, g + 32)]
                assert len(set(banks)) == 32, ("read", r, k, g)
            for g in range(0, T, 16):   # writes: 16-lane groups over 16 banks
                banks = [(slot(base(r, t) | (k << LO[r]))) % 16 for t in range(g, g + 16)]
                assert len(set(banks)) == 16, ("write", r, k, g)
            # additivity: slot(base | koff) = slot(base) + slot(koff)
            for t in (0, 1, 77, 300, 511):
                assert slot(base(r, t) | (k << LO[r])) == slot(base(r, t)) + slot(k << LO[r])
    print("lane maps, slots: ok")


def plain_forward(a, table, tile_pos, q):
    a = list(a)
    for b in range(11, -1, -1):
        for x in range(TILE):
            if (x >> b) & 1:
                continue
            y = x | (1 << b)
            w = table[(1 << (L - 1 - b)) + ((tile_pos + x) >> (b + 1))]
            u, v = a[x], a[y] * w % q
            a[x], a[y] = (u + v) % q, (u - v) % q
    return a


def plain_inverse(a, table, tile_pos, q):
    a = list(a)
    for b in range(0, 12):
        for x in range(TILE):
            if (x >> b) & 1:
                continue
            y = x | (1 << b)
            w = table[(1 << (L - 1 - b)) + ((tile_pos + x) >> (b + 1))]
            u, v = a[x], a[y]
            a[x], a[y] = (u + v) % q, (u - v) * w % q
    return a


def round_twiddle_indices(r, t, tile_pos):
    """7 table indices of lane t in round r in the order [j=2][j=1: u=0,1][j=0: u=0..3] (forward execution order)."""
    pos0 = tile_pos + base(r, t)
    out = []
    for j in (2, 1, 0):
        b = LO[r] + j
        first = (1 << (L - 1 - b)) + (pos0 >> (b + 1))
        out += [first + u for u in range(1 << (2 - j))]
    return out


def fused_forward(a, table, tile_pos, q):
    lds = {}
    regs = [[a[base(0, t) | (k << 9)] for k in range(8)] for t in range(T)]
    for r in range(4):
        if r:
            regs = [[lds[slot(base(r, t) | (k << LO[r]))] for k in range(8)] for t in range(T)]
        for t in range(T):
            tw = [table[i] for i in round_twiddle_indices(r, t, tile_pos)]
            v = regs[t]
            s = 0
            for j in (2, 1, 0):
                half = 1 << j
                for u in range(1 << (2 - j)):
                    w = tw[s]; s += 1
                    for l in range(half):
                        kx = (u << (j + 1)) | l
                        x, y = v[kx], v[kx + half] * w % q
                        v[kx], v[kx + half] = (x + y) % q, (x - y) % q
        if r < 3:
            lds = {}
            for t in range(T):
                for k in range(8):
                    lds[slot(base(r, t) | (k << LO[r]))] = regs[t][k]
    out = [0] * TILE
    for t in range(T):
        for k in range(8):
            out[base(3, t) | k] = regs[t][k]
    return out


def fused_inverse(a, table, tile_pos, q):
    regs = [[a[base(3, t) | k] for k in range(8)] for t in range(T)]
    for r in (3, 2, 1, 0):
        for t in range(T):
            idx = round_twiddle_indices(r, t, tile_pos)
            # inverse execution order: j = 0 (4 twiddles), j = 1 (2), j = 2 (1)
            tw = {2: [table[idx[0]]], 1: [table[i] for i in idx[1:3]], 0: [table[i] for i in idx[3:7]]}
            v = regs[t]
            for j in (0, 1, 2):
                half = 1 << j
                for u in range(1 << (2 - j)):
                    w = tw[j][u]
                    for l in range(half):
                        kx = (u << (j + 1)) | l
                        x, y = v[kx], v[kx + half]
                        v[kx], v[kx + half] = (x + y) % q, (x - y) * w % q
        if r > 0:
            lds = {}
            for t in range(T):
                for k in range(8):
                    lds[slot(base(r, t) | (k << LO[r]))] = regs[t][k]
            regs = [[lds[slot(base(r - 1, t) | (k << LO[r - 1]))] for k in range(8)] for t in range(T)]
    out = [0] * TILE
    for t in range(T):
        for k in range(8):
            out[base(0, t) | (k << 9)] = regs[t][k]
    return out


def check_twiddle_image():
    """R2 sub-tables in natural order, R3 thread-major: reads conflict free; wave-uniformity of R0/R1 twiddles."""
    tile_pos = 5 << 12
    for t in range(T):
        i0 = round_twiddle_indices(0, t, tile_pos)
        assert i0 == round_twiddle_indices(0, 0, tile_pos)                      # workgroup-uniform
        assert round_twiddle_indices(1, t, tile_pos) == round_twiddle_indices(1, t & ~63, tile_pos)   # wave-uniform
    # R2: sub-table of bit b holds entries e = idx >> (b+1), natural order at offsets 0 / 64 / 192
    off = {5: 0, 4: 64, 3: 192}
    for g in range(0, T, 32):
        for jj, (b, cnt) in enumerate(((5, 1), (4, 2), (3, 4))):
            for u in range(cnt):
                addrs = [off[b] + ((base(2, t) >> (b + 1)) + u) for t in range(g, g + 32)]
                per_bank = {}
                for a in set(addrs):
                    per_bank.setdefault(a % 32, set()).add(a)
                assert all(len(v) == 1 for v in per_bank.values()), ("R2 tw", b, u, g)
    print("twiddle uniformity / image: ok")


def main():
    check_maps()
    check_twiddle_image()
    q = 12289
    rnd = random.Random(5)
    table = [rnd.randrange(1, q) for _ in range(1 << L)]
    for tile in (0, 5, 15):
        a = [rnd.randrange(q) for _ in range(TILE)]
        tile_pos = tile << 12
        assert fused_forward(a, table, tile_pos, q) == plain_forward(a, table, tile_pos, q), tile
        assert fused_inverse(a, table, tile_pos, q) == plain_inverse(a, table, tile_pos, q), tile
    print("fused network == plain stage loop (forward, inverse): ok")


if __name__ == "__main__":
    main()
