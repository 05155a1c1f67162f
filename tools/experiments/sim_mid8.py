#!/usr/bin/env python3
"""CPU model of the index algebra of the barrier-free 8-stage commitment tile kernel (mlwe_mid8 in lsr_commit_fused.hpp):
register/lane maps of the three rounds, the two in-wavefront transposes, the twiddle-image addressing and the final
(matrix-product) layout.  Checked against the plain stage loop: stage of polynomial index bit b uses
table[2^(L-1-b) + (pos >> (b+1))], forward b = 7..0 and inverse b = 0..7 on a 4096-residue tile of an n = 2^16 polynomial."""
import random

L = 16
TILE = 4096
T = 512


def lane_bits(t):
    return [(t >> i) & 1 for i in range(6)], t >> 6      # l0..l5, wave


# index of register k of thread t in the layout of round 'a' (regs p7 p6 p5), 'b' (p4 p3 p2), 'c' (p4 p1 p0)
def idx_a(t, k):
    l, w = lane_bits(t)
    p = {8: l[0], 0: l[1], 1: l[2], 2: l[3], 3: l[4], 4: l[5], 5: k & 1, 6: (k >> 1) & 1, 7: (k >> 2) & 1}
    return sum(v << b for b, v in p.items()) | (w << 9)


def idx_b(t, k):
    l, w = lane_bits(t)
    p = {8: l[0], 0: l[1], 1: l[2], 5: l[3], 6: l[4], 7: l[5], 2: k & 1, 3: (k >> 1) & 1, 4: (k >> 2) & 1}
    return sum(v << b for b, v in p.items()) | (w << 9)


def idx_c(t, k):
    l, w = lane_bits(t)
    p = {8: l[0], 2: l[1], 3: l[2], 5: l[3], 6: l[4], 7: l[5], 0: k & 1, 1: (k >> 1) & 1, 4: (k >> 2) & 1}
    return sum(v << b for b, v in p.items()) | (w << 9)


def transpose(regs, pairs):
    """pairs: list of (register bit, lane bit): element (lane bit = x, reg bit = y) moves to (lane bit = y, reg bit = x)."""
    for rb, lb in pairs:
        new = [row[:] for row in regs]
        for t in range(T):
            up = (t >> lb) & 1
            partner = t ^ (1 << lb)
            for k in range(8):
                if (k >> rb) & 1:
                    continue
                hi = k | (1 << rb)
                if up:
                    new[t][k] = regs[partner][hi]          # receives the partner's upper register into its lower one
                else:
                    new[t][hi] = regs[partner][k]
        regs = new
    return regs


def twiddle_index(b, pos):
    return (1 << (L - 1 - b)) + (pos >> (b + 1))


# LDS twiddle image: per stage b the entries a workgroup needs, addressed by (compacted relevant thread bits) | (u << bits)
IMAGE_OFFSET = {7: 0, 6: 16, 5: 48, 4: 112, 3: 240, 2: 496, 1: 1008, 0: 2032}


def image_address(b, t, u):
    if b >= 5:      # round a: relevant thread bits l0 (p8) and the wave; u = register bits above b
        c = (t & 1) | ((t >> 6) << 1)
        return IMAGE_OFFSET[b] + (c | (u << 4))
    if b >= 2:      # round b: l0, l3, l4, l5, wave
        c = (t & 1) | ((t >> 3) << 1)
        return IMAGE_OFFSET[b] + (c | (u << 7))
    return IMAGE_OFFSET[b] + (t | (u << 9))      # round c: every thread bit


def butterflies(regs, stage_bits, layout, table, tile_pos, q, inverse, image):
    for b, rb in stage_bits:
        for t in range(T):
            v = regs[t]
            for k in range(8):
                if (k >> rb) & 1:
                    continue
                hi = k | (1 << rb)
                pos = tile_pos + layout(t, k)
                # u = the register bits of this round that lie above bit b, as the kernel counts them
                want = table[twiddle_index(b, pos)]
                u = u_of(b, k)
                addr = image_address(b, t, u)
                assert image.setdefault(addr, want) == want, (b, t, k)
                x, y = v[k], v[hi]
                if not inverse:
                    y = y * want % q
                    v[k], v[hi] = (x + y) % q, (x - y) % q
                else:
                    v[k], v[hi] = (x + y) % q, (x - y) * want % q
    return regs


def u_of(b, k):
    if b >= 5:
        return k >> (b - 5 + 1)                  # round a: register bits (p5, p6, p7) = (0, 1, 2)
    if b >= 2:
        return k >> (b - 2 + 1)                  # round b: (p2, p3, p4)
    if b == 1:
        return k >> 2                            # round c: registers (p0, p1, p4): above p1 lies p4 = bit 2
    return k >> 1                                # b = 0: p1 and p4


def plain(a, table, tile_pos, q, inverse):
    a = list(a)
    order = range(0, 8) if inverse else range(7, -1, -1)
    for b in order:
        for x in range(TILE):
            if (x >> b) & 1:
                continue
            y = x | (1 << b)
            w = table[twiddle_index(b, tile_pos + x)]
            if not inverse:
                u, v = a[x], a[y] * w % q
                a[x], a[y] = (u + v) % q, (u - v) % q
            else:
                u, v = a[x], a[y]
                a[x], a[y] = (u + v) % q, (u - v) * w % q
    return a


T1 = [(2, 5), (1, 4), (0, 3)]      # register bit <-> lane bit: p7<->p4(l5), p6<->p3(l4), p5<->p2(l3)
T2 = [(1, 2), (0, 1)]              # p3<->p1(l2), p2<->p0(l1)


def fused_forward(a, table, tile_pos, q):
    image = {}
    regs = [[a[idx_a(t, k)] for k in range(8)] for t in range(T)]
    regs = butterflies(regs, [(7, 2), (6, 1), (5, 0)], idx_a, table, tile_pos, q, False, image)
    regs = transpose(regs, T1)
    regs = butterflies(regs, [(4, 2), (3, 1), (2, 0)], idx_b, table, tile_pos, q, False, image)
    regs = transpose(regs, T2)
    regs = butterflies(regs, [(1, 1), (0, 0)], idx_c, table, tile_pos, q, False, image)
    out = [0] * TILE
    for t in range(T):
        for k in range(8):
            out[idx_c(t, k)] = regs[t][k]
    assert len(image) <= 4080 and max(image) < 4080
    return out


def fused_inverse(a, table, tile_pos, q):
    image = {}
    regs = [[a[idx_c(t, k)] for k in range(8)] for t in range(T)]
    regs = butterflies(regs, [(0, 0), (1, 1)], idx_c, table, tile_pos, q, True, image)
    regs = transpose(regs, T2[::-1])
    regs = butterflies(regs, [(2, 0), (3, 1), (4, 2)], idx_b, table, tile_pos, q, True, image)
    regs = transpose(regs, T1[::-1])
    regs = butterflies(regs, [(5, 0), (6, 1), (7, 2)], idx_a, table, tile_pos, q, True, image)
    out = [0] * TILE
    for t in range(T):
        for k in range(8):
            out[idx_a(t, k)] = regs[t][k]
    return out


def main():
    for name, f in (("a", idx_a), ("b", idx_b), ("c", idx_c)):
        assert len({f(t, k) for t in range(T) for k in range(8)}) == TILE, name
    # transposes map layout a -> b -> c
    regs = [[idx_a(t, k) for k in range(8)] for t in range(T)]
    regs = transpose(regs, T1)
    assert all(regs[t][k] == idx_b(t, k) for t in range(T) for k in range(8))
    regs = transpose(regs, T2)
    assert all(regs[t][k] == idx_c(t, k) for t in range(T) for k in range(8))
    # a wave-instruction of the round-a layout touches 2 runs of 32 consecutive residues (coalesced 256-byte segments)
    for k in range(8):
        got = sorted(idx_a(t, k) for t in range(64))
        runs = sum(1 for i, x in enumerate(got) if i == 0 or x != got[i - 1] + 1)
        assert runs == 2
    q = 12289
    rnd = random.Random(11)
    table = [rnd.randrange(1, q) for _ in range(1 << L)]
    for tile in (0, 7, 15):
        a = [rnd.randrange(q) for _ in range(TILE)]
        assert fused_forward(a, table, tile << 12, q) == plain(a, table, tile << 12, q, False), tile
        assert fused_inverse(a, table, tile << 12, q) == plain(a, table, tile << 12, q, True), tile
    print("mid8 lane maps, transposes, twiddle image, butterfly network: ok")


if __name__ == "__main__":
    main()
