"""Line-by-line Python emulation of conzic_amd/csrc/bridge.hip working on the SAME tables
(conzic_amd.bridge.BridgeArrays).  Lets the CPU test-suite check the table builder and the
device algorithm's logic against the HF-generated golden before any GPU is involved."""
import numpy as np

MAXB, MAXSYM, LEN = 512, 64, 77


def emulate_row(t, row_ids):
    merges = {}
    for r, (a, b, o) in enumerate(zip(t.merge_left.tolist(), t.merge_right.tolist(), t.merge_out.tolist())):
        merges.setdefault((a, b), (r, o))
    txt, cls = [], []
    first = True
    for i in row_ids:
        fl = int(t.piece_flags[i])
        if fl & 1:
            continue
        o0, o1 = int(t.piece_off[i]), int(t.piece_off[i + 1])
        if len(txt) + (o1 - o0) + 2 > MAXB:
            raise OverflowError
        if first:
            if fl & 2:
                txt += [ord("#")] * 2
                cls += [2 | 4] * 2
        elif not (fl & 6):
            txt.append(32)
            cls.append(3 | 4)
        txt += t.piece_bytes[o0:o1].tolist()
        cls += t.piece_class[o0:o1].tolist()
        first = False
    n = len(txt)
    out = [t.bos_id]
    i = 0
    while i < n and len(out) - 1 < LEN - 2:
        c = cls[i] & 3
        if c == 3:
            i += 1
            continue
        j = 0
        if txt[i] == 39 and i + 1 < n:
            c1 = chr(txt[i + 1])
            if c1 in "stmd":
                j = i + 2
            elif i + 2 < n:
                c2 = chr(txt[i + 2])
                if (c1, c2) in (("r", "e"), ("v", "e"), ("l", "l")):
                    j = i + 3
        if j == 0:
            j = i + 1
            if c == 0:
                while j < n and (cls[j] & 3) == 0:
                    j += 1
            elif c == 1:
                while j < n and not (cls[j] & 4):
                    j += 1
            else:
                while j < n and (cls[j] & 3) == 2:
                    j += 1
        m = j - i
        assert m <= MAXSYM
        sym = [int(t.byte_sym[b]) for b in txt[i:j]]
        sym[-1] = int(t.byte_sym_eow[txt[j - 1]])
        while len(sym) > 1:
            best, bi, bo = None, -1, 0
            for q in range(len(sym) - 1):
                hit = merges.get((sym[q], sym[q + 1]))
                if hit is not None and (best is None or hit[0] < best):
                    best, bi, bo = hit[0], q, hit[1]
            if bi < 0:
                break
            sym[bi:bi + 2] = [bo]
        for s in sym:
            if len(out) - 1 < LEN - 2:
                out.append(s)
        i = j
    out.append(t.eos_id)
    return out
