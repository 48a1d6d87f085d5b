#!/usr/bin/env python3
"""CPU emulation of gemm.hip's index math (staging map, LDS swizzle, MFMA fragment maps,
epilogue map) for one work-group, using the gfx950 MFMA lane layouts from
/opt/skills/guides/cdna_hip_programming.md §3.  Also counts ds_read_b128 bank conflicts.
Run: python tools/emulate_gemm_tile.py   (no GPU needed)"""
import numpy as np

BM = BN = 128
ROWB = 128


def swz(row, chunk):
    return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4)


def emulate(esize, M=128, N=128, K=256, seed=0):
    epc = 16 // esize              # elements per 16-byte chunk
    kpt = ROWB // esize            # K elements per tile
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, K))
    W = rng.standard_normal((N, K))
    C = np.zeros((M, N))
    acc = np.zeros((4, 2, 2, 64, 16))  # wave, i, j, lane, reg
    for kt in range(K // kpt):
        ldsA = {}
        ldsB = {}
        for tid in range(256):
            sc, sr = tid & 7, tid >> 3
            for i in range(4):
                r = sr + 32 * i
                off = swz(r, sc)
                ldsA[off] = A[r, kt * kpt + sc * epc: kt * kpt + (sc + 1) * epc]
                ldsB[off] = W[r, kt * kpt + sc * epc: kt * kpt + (sc + 1) * epc]
        for wave in range(4):
            wm, wn = wave >> 1, wave & 1
            for ks in range(4):
                for i in range(2):
                    for j in range(2):
                        a = np.zeros((64, epc))
                        b = np.zeros((64, epc))
                        for lane in range(64):
                            half = lane >> 5
                            ch = 2 * ks + half
                            a[lane] = ldsA[swz(wm * 64 + (lane & 31) + 32 * i, ch)]
                            b[lane] = ldsB[swz(wn * 64 + (lane & 31) + 32 * j, ch)]
                        # MFMA 32x32: D[row,col] += sum_k A[row,k] B[k,col]
                        # A operand: lane l holds row l&31, k-block l>>5 ; B operand: col l&31, k-block l>>5
                        if esize == 2:   # 32x32x16: 8 k per lane-half
                            Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                            for lane in range(64):
                                Am[lane & 31, 8 * (lane >> 5): 8 * (lane >> 5) + 8] = a[lane]
                                Bm[8 * (lane >> 5): 8 * (lane >> 5) + 8, lane & 31] = b[lane]
                            D = Am @ Bm
                        else:            # four 32x32x2: element e of each lane's float4
                            D = np.zeros((32, 32))
                            for e in range(4):
                                Am = np.zeros((32, 2)); Bm = np.zeros((2, 32))
                                for lane in range(64):
                                    Am[lane & 31, lane >> 5] = a[lane, e]
                                    Bm[lane >> 5, lane & 31] = b[lane, e]
                                D += Am @ Bm
                        for lane in range(64):
                            for r in range(16):
                                row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                                acc[wave, i, j, lane, r] += D[row, lane & 31]
    for wave in range(4):
        wm, wn = wave >> 1, wave & 1
        for i in range(2):
            for j in range(2):
                for lane in range(64):
                    col = wn * 64 + j * 32 + (lane & 31)
                    for r in range(16):
                        row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        C[row, col] = acc[wave, i, j, lane, r]
    ref = A @ W.T
    return np.abs(C - ref).max()


def bank_conflicts():
    """ds_read_b128: 4 groups of 16 lanes; bank = (addr/4) % 64; count max distinct 16B slots/ conflicts."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              [32 + x for x in (list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)))],
              [32 + x for x in (list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)))]]
    worst = 0
    for ks in range(4):
        for base in (0, 32, 64, 96):
            for g in groups:
                slots = {}
                for lane in g:
                    off = swz(base + (lane & 31), 2 * ks + (lane >> 5))
                    slot = (off // 16) % 16
                    slots.setdefault(slot, set()).add(off)
                worst = max(worst, max(len(v) for v in slots.values()))
    return worst


if __name__ == "__main__":
    print("bf16 tile max err", emulate(2))
    print("f32  tile max err", emulate(4))
    print("ds_read_b128 worst-case ways per 16B slot:", bank_conflicts())
