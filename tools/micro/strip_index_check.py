"""CPU check of gemm256_strip.hip's INDEX arithmetic (not of its synchronisation): restates, formula by formula, which global
16-byte chunk every DMA lane writes to which LDS byte (strip main rows, rows 256-258, weight half-tiles) and which LDS bytes
every lane's fragment reads fetch, then compares what a lane would feed the MFMA with the im2col value it must be --
A[pixel i of the tile][tap (dy, dx)][channel] with zeros outside the image, W[tap][column][channel].  bf16 and e4m3 layouts.

usage: python tools/micro/strip_index_check.py        (numpy only; a few seconds)
"""
import itertools
import sys

import numpy as np


def check(H, W, B, Cin, N, F8, tile_m, tile_n):
    EB = 1 if F8 else 2
    rowb = Cin * EB                       # bytes per pixel row / weight row
    EPC = 16 // EB                        # K elements per 16-byte chunk
    M, hw = B * H * W, H * W
    rng = np.random.default_rng(1)
    # operands as element-id arrays so that every byte position is distinguishable: value = unique integer
    A = rng.integers(1, 1 << 30, size=(M, Cin), dtype=np.int64)
    Wt = rng.integers(1, 1 << 30, size=(9, N, Cin), dtype=np.int64)
    m0, n0 = tile_m * 256, tile_n * 256
    nck = rowb // 128
    STRIP_ROWS = 259

    def a_chunk(pix, ck, chunk):          # the 16-byte chunk `chunk` of the 128-byte slice ck of pixel pix, as EPC element ids
        c0 = (ck * 128 + chunk * 16) // EB
        return A[pix, c0:c0 + EPC]

    for ck, dyi in itertools.product(range(nck), range(3)):
        dy = dyi - 1
        # ---------------- stage the strip exactly as issue_S does
        lds = np.zeros((STRIP_ROWS, 8, EPC), dtype=np.int64)       # [row][16-byte position][element]
        written = np.zeros((STRIP_ROWS, 8), dtype=bool)
        for tid in range(512):
            rho0 = tid >> 3
            chunk0 = (tid & 7) ^ ((rho0 >> 1) & 7)
            for j in range(4):
                s = j * 512 + tid
                rho, pos = s >> 3, s & 7                            # linear LDS destination: byte s * 16
                assert rho == j * 64 + rho0 and pos == (tid & 7)
                pix = m0 + rho0 + 64 * j
                y = (pix % hw) // W
                ok = dy == 0 or (dy < 0 and y > 0) or (dy > 0 and y < H - 1)
                if ok:
                    src_pix = pix + dy * W                          # Ab + soff + 64 j rowb + offS
                    lds[rho, pos] = a_chunk(src_pix, ck, chunk0)
                written[rho, pos] = True
        for lane in range(24):
            ar, pos = lane >> 3, lane & 7
            rho = 256 + ar
            chunk = pos ^ ((rho >> 1) & 7)
            pix = m0 - 1 if ar == 0 else m0 + 256
            if ar < 2 and 0 <= pix < M:
                y = (pix % hw) // W
                ok = dy == 0 or (dy < 0 and y > 0) or (dy > 0 and y < H - 1)
                if ok:
                    lds[rho, pos] = a_chunk(pix + dy * W, ck, chunk)
            written[rho, pos] = True
        assert written.all()
        for dxi in range(3):
            dx = dxi - 1
            tap = dyi * 3 + dxi
            # ---------------- weight half-tiles exactly as issue_B does
            ldsB = np.zeros((2, 128, 8, EPC), dtype=np.int64)
            for h, tid, j in itertools.product(range(2), range(512), range(2)):
                rho0 = tid >> 3
                chunk0 = (tid & 7) ^ ((rho0 >> 1) & 7)
                s = j * 512 + tid
                i, pos = s >> 3, s & 7
                col = n0 + (rho0 >> 5) * 64 + (rho0 & 31) + j * 128 + h * 32        # offB + (j*128 + h*32) rowb
                assert col == n0 + (i >> 5) * 64 + h * 32 + (i & 31)
                c0 = (ck * 128 + chunk0 * 16) // EB
                ldsB[h, i, pos] = Wt[tap, col, c0:c0 + EPC]
            # ---------------- fragment reads of every wave / lane
            for wave in range(8):
                wr, wc = wave >> 2, wave & 3
                for lane in range(64):
                    lr, lh = lane & 31, lane >> 5
                    for sub in range(4):                            # sub = 2 ah + mt
                        i = wr * 128 + sub * 32 + lr
                        pix = m0 + i
                        x = (pix % hw) % W
                        rho = i + dxi - 1
                        if rho < 0:
                            rho = 256
                        if rho > 255:
                            rho = 257
                        if (dxi == 0 and x == 0) or (dxi == 2 and x == W - 1):
                            rho = 258
                        key = (rho >> 1) & 7
                        fbase = rho * 128 + ((((2 * lh) ^ key) if F8 else (lh ^ key)) << 4)
                        # expected im2col row of output pixel i for tap (dy, dx)
                        yy, xx = (pix % hw) // W + dy, x + dx
                        inside = 0 <= yy < H and 0 <= xx < W
                        for ks in range(4):
                            cx = ((4 * (ks >> 1) + (ks & 1)) << 4) if F8 else ((2 * ks) << 4)
                            addr = fbase ^ cx
                            r, pos = addr >> 7, (addr >> 4) & 7
                            got = lds[r, pos]
                            # which K elements of the 128-byte slice this read must be: bf16: chunk 2 ks + lh; e4m3: chunk 4 (ks>>1) + 2 lh + (ks&1)
                            kchunk = (4 * (ks >> 1) + 2 * lh + (ks & 1)) if F8 else (2 * ks + lh)
                            if inside:
                                want = a_chunk(pix + dy * W + dx, ck, kchunk)
                            else:
                                want = np.zeros(EPC, dtype=np.int64)
                            if not np.array_equal(got, want):
                                print(f"A MISMATCH H={H} W={W} F8={F8} ck={ck} dy={dy} dx={dx} wave={wave} lane={lane} sub={sub} ks={ks} "
                                      f"rho={r} pos={pos} inside={inside}")
                                return False
                    # weights: half h, row wc*32 + lr
                    keyb = (lr >> 1) & 7
                    fbB = (wc * 32 + lr) * 128 + ((((2 * lh) ^ keyb) if F8 else ((lh ^ (lr >> 1)) & 7)) << 4)
                    for h in range(2):
                        col = n0 + wc * 64 + h * 32 + lr
                        for ks in range(4):
                            cx = ((4 * (ks >> 1) + (ks & 1)) << 4) if F8 else ((2 * ks) << 4)
                            addr = fbB ^ cx
                            r, pos = addr >> 7, (addr >> 4) & 7
                            kchunk = (4 * (ks >> 1) + 2 * lh + (ks & 1)) if F8 else (2 * ks + lh)
                            c0 = (ck * 128 + kchunk * 16) // EB
                            if not np.array_equal(ldsB[h, r, pos], Wt[tap, col, c0:c0 + EPC]):
                                print(f"B MISMATCH F8={F8} tap={tap} wave={wave} lane={lane} h={h} ks={ks}")
                                return False
    return True


def bank_check():
    """ds_read_b128 service groups (16 lanes each) must hit 16 distinct 16-byte slots of the 256-byte bank row for every run start."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for R0, c, g in itertools.product(range(64), range(8), groups):
        slots = set()
        for lr in g:
            rho = R0 + lr
            addr = rho * 128 + ((c ^ ((rho >> 1) & 7)) << 4)
            slots.add((addr >> 4) & 15)
        if len(slots) != 16:
            print("bank conflict for run start", R0, "chunk", c)
            return False
    return True


if __name__ == "__main__":
    ok = bank_check()
    print("shifted 32-row runs conflict-free for every start row:", ok)
    # (H, W, B, Cin, N, F8, tile_m, tile_n): tiles at an image start, in the middle of an image, at the tensor's end; W = 16 puts 16 image rows in a tile
    cases = [(16, 16, 3, 64, 256, False, 0, 0), (16, 16, 3, 64, 256, False, 1, 0), (16, 16, 3, 64, 256, False, 2, 0),
             (32, 32, 1, 128, 512, False, 1, 1), (32, 32, 1, 128, 512, False, 3, 0), (64, 64, 1, 64, 256, False, 5, 0),
             (32, 32, 1, 128, 256, True, 2, 0), (16, 16, 2, 256, 256, True, 1, 0)]
    for cse in cases:
        r = check(*cse)
        print(cse, "ok" if r else "FAILED")
        ok = ok and r
    sys.exit(0 if ok else 1)
