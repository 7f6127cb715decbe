"""Round 6: a CPU model of the block finder's stage 2 on a real DEFLATE stream — how many steps the code-length walk of every
stage-1 survivor takes, how many steps a wavefront of 64 of them takes (its longest member's), and what sorting the survivors
by a class computed from the first three code-length-code widths (symbols 16, 17, 18: the repeat codes) makes of that.
  arrival order: 204 steps per batch for a mean of 43 per candidate; classes (16 quantiles of E2): 108; exact sort: 104.
Usage: python tools/exp/find2_model.py [Mbit]   (pure python over the bits: a minute per 10 Mbit)"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
TH = [432, 568, 652, 832, 944, 1076, 1420, 1604, 2008, 2710, 3016, 4978, 5184, 5664, 9820]      # = make_find_cls(), lfx_decode_kernels.hip


def main():
    import synth
    synth.build()
    nbits = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 20_000_000
    comp = zlib.compress(synth.text(24 << 20).tobytes(), 6)[2:]
    bits = np.unpackbits(np.frombuffer(comp, dtype=np.uint8), bitorder="little")[:nbits]
    n = len(bits) - 6000

    def val(pos, k):
        v = 0
        for i in range(k):
            v |= int(bits[pos + i]) << i
        return v

    cand = np.nonzero((bits[:n] == 0) & (bits[1:n + 1] == 0) & (bits[2:n + 2] == 1))[0]
    surv = []
    for p in cand:
        hlit, hdist, hclen = val(p + 3, 5), val(p + 8, 5), val(p + 13, 4)
        if hlit > 29 or hdist > 29:
            continue
        cl, k, nz = [0] * 19, 0, 0
        for i in range(hclen + 4):
            w = val(p + 17 + 3 * i, 3)
            cl[ORDER[i]] = w
            if w:
                k += 128 >> w
                nz += 1
        if k == 128 and nz >= 2:
            surv.append((p, hlit + 257, hdist + 1, hclen + 4, cl))

    def walk(p, nl, nd, nc, cl):
        cnt = [0] * 8
        for w in cl:
            if w:
                cnt[w] += 1
        code, nxt = 0, [0] * 8
        for w in range(1, 8):
            code = (code + (cnt[w - 1] if w > 1 else 0)) << 1
            nxt[w] = code
        tab = {}
        for s, w in enumerate(cl):
            if w:
                tab[(w, nxt[w])] = s
                nxt[w] += 1
        pos, have, kl, last, steps = p + 17 + 3 * nc, 0, 0, 0, 0
        while have < nl:
            steps += 1
            code, sym = 0, None
            for w in range(1, 8):
                code = (code << 1) | int(bits[pos + w - 1])
                if (w, code) in tab:
                    sym = tab[(w, code)]
                    pos += w
                    break
            if sym is None:
                return steps
            if sym < 16:
                rep, v = 1, sym
            elif sym == 16:
                if have == 0:
                    return steps
                rep, v = 3 + val(pos, 2), last
                pos += 2
            elif sym == 17:
                rep, v = 3 + val(pos, 3), 0
                pos += 3
            else:
                rep, v = 11 + val(pos, 7), 0
                pos += 7
            if have + rep > nl + nd:
                return steps
            if v:
                kl += min(rep, nl - have) * (32768 >> v)
            if kl > 32768:
                return steps
            have += rep
            last = v
        return steps

    st = np.array([walk(*s) for s in surv])
    cl = np.array([s[4] for s in surv])

    def p(l):
        return np.where(l > 0, 128 >> l, 0)

    p16, p17, p18 = p(cl[:, 16]), p(cl[:, 17]), p(cl[:, 18])
    e2 = p18 * 149 + p17 * 13 + p16 * 9 + 2 * (128 - p16 - p17 - p18)
    cls = np.searchsorted(np.array(TH), e2, side="right")

    def per_batch(idx):
        s = st[idx]
        m = len(s) // 64 * 64
        return s[:m].reshape(-1, 64).max(axis=1).sum() / (m // 64)

    print(f"{len(surv)} survivors of {n} offsets ({len(surv) / n:.2e}); mean walk {st.mean():.1f} steps")
    print(f"steps per batch of 64: arrival order {per_batch(np.arange(len(st))):.1f}, by class {per_batch(np.argsort(cls, kind='stable')):.1f}, "
          f"by E2 exactly {per_batch(np.argsort(e2, kind='stable')):.1f}, by the walk itself {per_batch(np.argsort(st, kind='stable')):.1f}")
    print("class sizes:", np.bincount(cls, minlength=16))


if __name__ == "__main__":
    main()
