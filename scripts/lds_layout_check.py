#!/usr/bin/env python3
"""CPU model of the index arithmetic of sgpt_amd/csrc/gemm256w.hip (the 32x32x16-MFMA GEMM): run before spending GPU time.

1. LDS bank conflicts.  ds_read_b128 / ds_write_b128 / ds_write_b64 are serviced in fixed lane groups
   (/opt/skills/guides/MI355X_MICROARCH.md, LDS table); inside a group every lane must touch distinct banks.
2. A functional walk of one 256 x 256 tile: LDS-DMA fill with the source-side swizzle -> fragment reads -> the
   32x32x16 MFMA lane maps (A: row = lane & 31, k = 8 (lane >> 5) + e; C: col = lane & 31, row = 8q + 4h + e) -> the three
   store epilogues through the wave scratch -> global.  The result must equal A . W^T (and its transpose for V^T).
The formulas are typed from the kernel, not imported: this guards the kernel's internal consistency (fill swizzle vs
read swizzle, scratch write vs read-back, coverage of every output element exactly once), not the MFMA hardware maps."""
import itertools

import numpy as np

READ_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ_B128_GROUPS += [[l + 32 for l in g] for g in READ_B128_GROUPS]
WRITE_B128_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]
WRITE_B64_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


def conflicts(addr_of_lane, nbytes, groups, nbanks):
    """max number of lanes of one group on one bank (1 = conflict-free)."""
    worst = 1
    for g in groups:
        use = {}
        for l in g:
            a = addr_of_lane(l)
            for b in range(a // 4, (a + nbytes) // 4):
                use.setdefault(b % nbanks, set()).add(a)     # identical addresses broadcast
        worst = max(worst, max(len(v) for v in use.values()))
    return worst


def check_banks():
    CH = 8
    res = {}
    # k-loop fragment reads: lane (r32, h), sub-step s, block row offset multiple of 32
    for wm, blk, s in itertools.product(range(2), range(4), range(4)):
        def addr(l, wm=wm, blk=blk, s=s):
            r32, h = l & 31, l >> 5
            row = wm * 128 + blk * 32 + r32
            return (row * CH + ((2 * s + h) ^ ((r32 >> 1) & 7))) * 16
        res["fragment ds_read_b128"] = max(res.get("fragment ds_read_b128", 1), conflicts(addr, 16, READ_B128_GROUPS, 64))
    # 16-bit epilogue: write 8 B at row r32 * 144 + (32 j + 8 q + 4 h) * 2 ; read 16 B at (8 k + lane >> 3) * 144 + (lane & 7) * 16
    for j, q in itertools.product(range(2), range(4)):
        res["epi16 ds_write_b64"] = max(res.get("epi16 ds_write_b64", 1), conflicts(
            lambda l: (l & 31) * 144 + (j * 32 + 8 * q + 4 * (l >> 5)) * 2, 8, WRITE_B64_GROUPS, 32))
    for k8 in range(4):
        res["epi16 ds_read_b128"] = max(res.get("epi16 ds_read_b128", 1), conflicts(
            lambda l: (k8 * 8 + (l >> 3)) * 144 + (l & 7) * 16, 16, READ_B128_GROUPS, 64))
    # fp32 epilogue: write 16 B at r32 * 256 + ((c ^ (r32 & 15)) << 4), c = 8 j + 2 q + h ; read row 4 k + (lane >> 4)
    for j, q in itertools.product(range(2), range(4)):
        res["epi32 ds_write_b128"] = max(res.get("epi32 ds_write_b128", 1), conflicts(
            lambda l: (l & 31) * 256 + (((j * 8 + 2 * q + (l >> 5)) ^ ((l & 31) & 15)) << 4), 16, WRITE_B128_GROUPS, 32))
    for k4 in range(8):
        res["epi32 ds_read_b128"] = max(res.get("epi32 ds_read_b128", 1), conflicts(
            lambda l: (k4 * 4 + (l >> 4)) * 256 + (((l & 15) ^ ((k4 * 4 + (l >> 4)) & 15)) << 4), 16, READ_B128_GROUPS, 64))
    # V^T epilogue: write 8 B at r32 * 256 + ((c ^ (r32 & 15)) << 4) + 8 h, c = 4 i + q
    for i, q in itertools.product(range(4), range(4)):
        res["epiVT ds_write_b64"] = max(res.get("epiVT ds_write_b64", 1), conflicts(
            lambda l: (l & 31) * 256 + (((i * 4 + q) ^ ((l & 31) & 15)) << 4) + 8 * (l >> 5), 8, WRITE_B64_GROUPS, 32))
    return res


def check_q8():
    """gemm256q.hip (fp8, 32-byte fragments = chunks 2g and 2g + 1 of row fr): the rotl3(row & 7) swizzle is conflict
    free for both ds_read_b128 of a fragment; the chunk ^ (row & 7) swizzle of the 16-bit kernels is not."""
    def rotl3(x):
        return ((x << 1) & 7) | (x >> 2)
    out = {}
    for name, f in (("rotl3(row&7)", rotl3), ("row&7", lambda x: x)):
        worst = 1
        for blk, half in itertools.product(range(8), range(2)):
            def addr(l, blk=blk, half=half):
                fr, g = l & 15, l >> 4
                row = blk * 16 + fr
                return (row * 8 + ((2 * g + half) ^ f(row & 7))) * 16
            worst = max(worst, conflicts(addr, 16, READ_B128_GROUPS, 64))
        out[name] = worst
    # functional: DMA fill (lane l -> row 8q + (l >> 3), slot l & 7 <- global chunk (l & 7) ^ rotl3(l >> 3)) vs fragment read
    for q, l in itertools.product(range(4), range(64)):
        row, slot = 8 * q + (l >> 3), l & 7
        gchunk = slot ^ rotl3(l >> 3)
        assert gchunk ^ rotl3(row & 7) == slot
    return out


def walk_tile(K=128, seed=0):
    rng = np.random.default_rng(seed)
    M = N = 256
    A = rng.integers(-3, 4, size=(M, K)).astype(np.float64)
    W = rng.integers(-3, 4, size=(N, K)).astype(np.float64)
    CH = 8
    want = A @ W.T
    out16 = np.full((M, N), np.nan)      # row-major 16-bit epilogue
    out32 = np.full((M, N), np.nan)      # fp32 epilogue
    outvt = np.full((N, M), np.nan)      # transposed epilogue
    acc_swap = np.zeros((8, 64, 4, 2, 16))     # [wave][lane][i][j][e], W fragment as the A operand
    acc_vt = np.zeros((8, 64, 4, 2, 16))
    for kt in range(K // 64):
        # LDS-DMA fill of one stage: [row][slot] <- global chunk
        lds_a = np.zeros((256, CH, 8))
        lds_w = np.zeros((256, CH, 8))
        for wave, q, l in itertools.product(range(8), range(4), range(64)):
            row = wave * 32 + q * 8 + (l >> 3)
            lc = (l & 7) ^ (l >> 4) ^ (4 * (q & 1))
            lds_a[row, l & 7] = A[row, kt * 64 + lc * 8: kt * 64 + lc * 8 + 8]
            lds_w[row, l & 7] = W[row, kt * 64 + lc * 8: kt * 64 + lc * 8 + 8]
        for wave, l in itertools.product(range(8), range(64)):
            wm, wn, r32, h = wave >> 2, wave & 3, l & 31, l >> 5
            swz = (r32 >> 1) & 7
            for s in range(4):
                cs = (2 * s + h) ^ swz
                af = [lds_a[wm * 128 + i * 32 + r32, cs] for i in range(4)]      # 8 k-values: k = 16 s + 8 h + e
                wf = [lds_w[wn * 64 + j * 32 + r32, cs] for j in range(2)]
                # the MFMA contracts over (h, e) across the two half-waves: emulate by scattering per-lane partial rows
                for i, j in itertools.product(range(4), range(2)):
                    pass
            # (the contraction needs all lanes: done below with the gathered fragments)
        # gather fragments per wave and contract like the instruction: D[row][col] = sum_k Aop[row][k] Bop[k][col]
        for wave in range(8):
            wm, wn = wave >> 2, wave & 3
            for s in range(4):
                Aop = np.zeros((4, 32, 16))
                Wop = np.zeros((2, 32, 16))
                for l in range(64):
                    r32, h = l & 31, l >> 5
                    cs = (2 * s + h) ^ ((r32 >> 1) & 7)
                    for i in range(4):
                        Aop[i, r32, 8 * h: 8 * h + 8] = lds_a[wm * 128 + i * 32 + r32, cs]
                    for j in range(2):
                        Wop[j, r32, 8 * h: 8 * h + 8] = lds_w[wn * 64 + j * 32 + r32, cs]
                for i, j in itertools.product(range(4), range(2)):
                    D_swap = Wop[j] @ Aop[i].T          # rows n, cols m
                    D_vt = Aop[i] @ Wop[j].T            # rows m, cols n
                    for l in range(64):
                        r32, h = l & 31, l >> 5
                        for e in range(16):
                            row = (e & 3) + 8 * (e >> 2) + 4 * h
                            acc_swap[wave, l, i, j, e] += D_swap[row, r32]
                            acc_vt[wave, l, i, j, e] += D_vt[row, r32]
    # ---- epilogues ----
    for wave in range(8):
        wm, wn = wave >> 2, wave & 3
        for i in range(4):
            # 16-bit row-major: scratch rows of 144 B, 2-byte elements (model: element index = byte / 2)
            scr = np.full(32 * 72, np.nan)
            for l, j, q in itertools.product(range(64), range(2), range(4)):
                r32, h = l & 31, l >> 5
                base = (r32 * 144 + (j * 32 + 8 * q + 4 * h) * 2) // 2
                scr[base: base + 4] = acc_swap[wave, l, i, j, 4 * q: 4 * q + 4]
            for k8, l in itertools.product(range(4), range(64)):
                row, rchunk = k8 * 8 + (l >> 3), l & 7
                v = scr[(row * 144 + rchunk * 16) // 2: (row * 144 + rchunk * 16) // 2 + 8]
                m = wm * 128 + i * 32 + row
                assert np.isnan(out16[m, wn * 64 + rchunk * 8: wn * 64 + rchunk * 8 + 8]).all()
                out16[m, wn * 64 + rchunk * 8: wn * 64 + rchunk * 8 + 8] = v
            # fp32 row-major: scratch rows of 256 B, 4-byte elements, chunk ^ (row & 15)
            scr = np.full(32 * 64, np.nan)
            for l, j, q in itertools.product(range(64), range(2), range(4)):
                r32, h = l & 31, l >> 5
                c = j * 8 + 2 * q + h
                base = (r32 * 256 + ((c ^ (r32 & 15)) << 4)) // 4
                scr[base: base + 4] = acc_swap[wave, l, i, j, 4 * q: 4 * q + 4]
            for k4, l in itertools.product(range(8), range(64)):
                row, rchunk = k4 * 4 + (l >> 4), l & 15
                base = (row * 256 + ((rchunk ^ (row & 15)) << 4)) // 4
                m = wm * 128 + (l >> 4) + i * 32 + k4 * 4
                assert np.isnan(out32[m, wn * 64 + rchunk * 4: wn * 64 + rchunk * 4 + 4]).all()
                out32[m, wn * 64 + rchunk * 4: wn * 64 + rchunk * 4 + 4] = scr[base: base + 4]
        for j in range(2):
            scr = np.full(32 * 128, np.nan)
            for l, i, q in itertools.product(range(64), range(4), range(4)):
                r32, h = l & 31, l >> 5
                c = i * 4 + q
                base = (r32 * 256 + ((c ^ (r32 & 15)) << 4) + 8 * h) // 2
                scr[base: base + 4] = acc_vt[wave, l, i, j, 4 * q: 4 * q + 4]
            for k4, l in itertools.product(range(8), range(64)):
                row, rchunk = k4 * 4 + (l >> 4), l & 15
                base = (row * 256 + ((rchunk ^ (row & 15)) << 4)) // 2
                n = wn * 64 + j * 32 + row
                assert np.isnan(outvt[n, wm * 128 + rchunk * 8: wm * 128 + rchunk * 8 + 8]).all()
                outvt[n, wm * 128 + rchunk * 8: wm * 128 + rchunk * 8 + 8] = scr[base: base + 8]
    assert np.array_equal(out16, want), "16-bit row-major epilogue"
    assert np.array_equal(out32, want), "fp32 row-major epilogue"
    assert np.array_equal(outvt, want.T), "V^T epilogue"
    # threshold-filter epilogue index: document column of (j, e, h)
    for wave, l, i, j, e in itertools.product(range(8), (0, 17, 40, 63), range(4), range(2), range(16)):
        wm, wn, r32, h = wave >> 2, wave & 3, l & 31, l >> 5
        m, n = wm * 128 + i * 32 + r32, wn * 64 + j * 32 + 8 * (e >> 2) + 4 * h + (e & 3)
        assert acc_swap[wave, l, i, j, e] == want[m, n]
    return True


if __name__ == "__main__":
    for k, v in check_banks().items():
        print(f"{k:26s} worst lanes-per-bank in a service group: {v}")
    for k, v in check_q8().items():
        print(f"fp8 fragment reads, swizzle {k:13s} worst lanes-per-bank: {v}")
    print("tile walk (fill swizzle / fragment reads / C maps / three epilogues):", "ok" if walk_tile() else "FAILED")
