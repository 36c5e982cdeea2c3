#!/usr/bin/env python3
"""LDS image of the attention kernels, simulated lane by lane (csrc/attn_helpers.h): bank conflicts of the row reads and the
transposed reads for the round-1/2 padded rows and the round-3 subtile image, the LDS-DMA fill map, and the per-lane parts
of the transposed-read addresses.  MI355X_MICROARCH.md LDS table: ds_read_b128 is serviced in 4 groups of 16 lanes,
ds_read_b64_tr_b16 in 2 halves of 32; bank = (byte address / 4) mod 64; distinct dwords on one bank within a group
serialise.  Run: python tools/lds_image_check.py"""

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in g] for g in B128_GROUPS]
TR_GROUPS = [list(range(32)), list(range(32, 64))]


def off_new(DH, row, ch):
    NC = DH // 32
    return NC * 512 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3))


def off_old(DH, row, ch):
    return row * (DH + 8) * 2 + ch * 16


def cycles(addrs, nbytes, groups):
    cyc = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            for b in range(a // 4, (a + nbytes) // 4):
                banks.setdefault(b % 64, set()).add(b)
        cyc += max(len(v) for v in banks.values())
    return cyc


def tr_lane(lane, hi):
    h, g, q, p = lane >> 5, (lane >> 4) & 1, (lane >> 2) & 3, lane & 3
    return 64 * (4 * h + q) + 16 * ((2 * g + (p >> 1)) ^ (h ^ (2 if hi else 0))) + 8 * (p & 1)


def read_cycles(DH, off):
    KS, DT = DH // 16, DH // 32
    tot = n = 0
    for r0 in (0, 32):
        for ks in range(KS):
            tot += cycles([off(DH, r0 + (l & 31), 2 * ks + (l >> 5)) for l in range(64)], 16, B128_GROUPS)
            n += 1
    row = tot / n
    tot = n = 0
    for r0 in (0, 16, 32, 48):
        for dt in range(DT):
            for hi in (0, 8):
                addrs = []
                for l in range(64):
                    h, g, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
                    addrs.append(off(DH, r0 + 4 * h + q + hi, 4 * dt + 2 * g + (p >> 1)) + 8 * (p & 1))
                tot += cycles(addrs, 8, TR_GROUPS)
                n += 1
    return row, tot / n


def dma_map_ok(DH):
    """TileDma::init: piece p, lane L -> (row, chunk); the piece's LDS position 1024 p + 16 L must be img_off(row, chunk)."""
    NC, PIECES = DH // 32, DH // 8
    seen = set()
    for p in range(PIECES):
        for L in range(64):
            st = 2 * p + (L >> 5)
            rg, cc = st // NC, st % NC
            w = L & 31
            row = 8 * rg + (w >> 2)
            ch = 4 * cc + ((w & 3) ^ ((row >> 2) & 3))
            if 1024 * p + 16 * L != off_new(DH, row, ch):
                return False
            seen.add((row, ch))
    return len(seen) == 64 * DH // 8


def tr_parts_ok(DH):
    """tr_lane_lo / tr_lane_hi + img_tr_imm<G, D> (+ (DH/32)*512 for the second read) == img_off of the mechanism address."""
    NC, DT = DH // 32, DH // 32
    for G in range(4):
        for D in range(DT):
            imm = NC * 512 * (2 * G) + 512 * D
            for l in range(64):
                h, g, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
                for hi in (0, 1):
                    row = 16 * G + 4 * h + q + 8 * hi
                    want = off_new(DH, row, 4 * D + 2 * g + (p >> 1)) + 8 * (p & 1)
                    if tr_lane(l, hi) + imm + (NC * 512 if hi else 0) != want:
                        return False
    return True


# ---- gemm6.hip (one wave per SIMD GEMM): the same image with W = stage depth (row-read operands) or W = 256 (transposed-read
# operands, read in the STANDARD MFMA k order: lane (i, h) gets tile rows 16 G + 8 h + 0..7 of column 32 D + i) -----------------
def gemm6_row_read_cycles(W, rows=256):
    """Frag4<false>: lane (r = l & 31, h = l >> 5) reads chunk 2 G + h of tile row 32 D + r through the even / odd-G lane parts."""
    worst = 0
    for D in range(rows // 32):
        for G in range(W // 16):
            addrs = []
            for l in range(64):
                r, h = l & 31, l >> 5
                s = (r >> 2) & 3
                lane_part = (W // 32) * 512 * (r >> 3) + 64 * (r & 7) + 16 * (((2 if G & 1 else 0) + h) ^ s)
                a = lane_part + (W // 32) * 2048 * D + 512 * (G >> 1)
                assert a == off_new(W, 32 * D + r, 2 * G + h)
                addrs.append(a)
            worst = max(worst, cycles(addrs, 16, B128_GROUPS))
    return worst


def gemm6_tr_read_ok(G_count):
    """Frag4<true>: the two ds_read_b64_tr_b16 of fragment (G, D) address rows 16 G + 8 h + q (+ 4) of columns 32 D + 16 g + 4 p ..;
    returns (addresses match img_off, worst LDS cycles per read)."""
    W, worst, ok = 256, 0, True
    for G in range(G_count):
        for D in range(8):
            for hi in (0, 1):
                addrs = []
                for l in range(64):
                    h, g, q, p = l >> 5, (l >> 4) & 1, (l >> 2) & 3, l & 3
                    c = 2 * g + (p >> 1)
                    if hi:
                        lane_part = 4096 * h + 64 * (4 + q) + 16 * (c ^ (2 * h + 1)) + 8 * (p & 1)
                    else:
                        lane_part = 4096 * h + 64 * q + 16 * (c ^ (2 * h)) + 8 * (p & 1)
                    a = lane_part + 8192 * G + 512 * D
                    ok = ok and a == off_new(W, 16 * G + 8 * h + q + 4 * hi, 4 * D + c) + 8 * (p & 1)
                    addrs.append(a)
                worst = max(worst, cycles(addrs, 8, TR_GROUPS))
    return ok, worst


def gemm6_piece_map_ok(KR, BK):
    """piece_voff: piece p, lane L -> (tile row, 16-byte chunk); its LDS position 1024 p + 16 L must be img_off(row, chunk) and
    the pieces must cover the tile exactly once."""
    W = 256 if KR else BK
    rows = BK if KR else 256
    NC, seen = W // 32, set()
    for p in range(BK // 2):
        for L in range(64):
            st = 2 * p + (L >> 5)
            rg, cc = st // NC, st % NC
            w = L & 31
            row = 8 * rg + (w >> 2)
            ch = 4 * cc + ((w & 3) ^ ((row >> 2) & 3))
            if 1024 * p + 16 * L != off_new(W, row, ch) or row >= rows:
                return False
            seen.add((row, ch))
    return len(seen) == rows * W // 8


if __name__ == "__main__":
    for DH in (96, 64):
        for name, off in (("padded rows [.][DH+8] (rounds 1-2)", off_old), ("8x32 subtile image (round 3)", off_new)):
            r, t = read_cycles(DH, off)
            print(f"DH {DH} {name:38s}: ds_read_b128 {r:.1f} LDS cycles (floor 4), ds_read_b64_tr_b16 {t:.1f} (floor 2)")
        print(f"DH {DH}: LDS-DMA fill map inverts img_off: {dma_map_ok(DH)}; transposed-read lane parts: {tr_parts_ok(DH)}")
    for BK in (32, 64):
        print(f"gemm6 BK {BK}: row reads {gemm6_row_read_cycles(BK)} LDS cycles (floor 4); transposed reads {gemm6_tr_read_ok(BK // 16)} (ok, floor 2); "
              f"piece maps KC {gemm6_piece_map_ok(False, BK)} KR {gemm6_piece_map_ok(True, BK)}")
