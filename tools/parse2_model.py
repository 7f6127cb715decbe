"""CPU model of the lazy-length parse (libflate_amd/csrc/lfx_parse2.hip), statement for statement.

Not a product path and not the oracle: a plain-Python restatement of WHAT the four parse kernels compute (lane loops
instead of wavefronts), used by tests/test_host_pipeline.py to check the algorithm — speculative group walks, the
in-wavefront chain of entries with its serial repair, the segment chain (fixseg / fix) and the rebuild of code words
from visit bits — against the oracle's DefaultLz77Encoder::flush (libflate_lz77/src/default.rs:69-109) on the CPU.
"""
import numpy as np

U = 52                 # PARSE_GROUP
SEG = 64 * U           # PARSE_SEG
MAX_WINDOW = 32768


def candidates(buf, window=MAX_WINDOW):
    """cd[p] = distance to the most recent earlier occurrence of buf[p:p+3] (0 = none inside the window) — what
    lfx_match3.hip writes (default.rs:76-87: every position < end is inserted, in order)."""
    n = len(buf)
    end = max(n, 3) - 3
    cd = np.zeros(n, dtype=np.uint32)
    last = {}
    for p in range(end):
        k = bytes(buf[p:p + 3])
        q = last.get(k)
        if q is not None and p - q <= window:
            cd[p] = p - q
        last[k] = p
    return cd


class Chunk:
    def __init__(self, buf, window=MAX_WINDOW, max_len=258):
        self.buf = bytes(buf)
        self.n = len(buf)
        self.end = max(self.n, 3) - 3
        self.max_len = max_len
        self.cd = candidates(self.buf, window)
        self.n_seg = (self.n + SEG - 1) // SEG

    def step(self, pos):                      # walk_step / gwalk_step
        d = int(self.cd[pos])
        if d == 0:
            return 1
        lim = min(self.n - (pos + 3), self.max_len - 3)
        l = 0
        b = self.buf
        while l < lim and b[pos + 3 + l] == b[pos + 3 - d + l]:
            l += 1
        return 3 + l


def resolve(ch, in_, a, stop, mask, exit_spec):
    walked = 0
    pos = in_
    while True:
        if pos >= stop:
            return walked, pos
        r = pos - a
        if (mask >> r) & 1:
            return walked | (mask & (~0 << r) & ((1 << 64) - 1)), exit_spec
        walked |= 1 << r
        pos += ch.step(pos)


def walk_segment(ch, sidx, stats=None):
    """parse_walk_kernel for one wavefront → (vis[64], seg_exit, seg_count, staged codes)."""
    s0 = sidx * SEG
    if s0 >= ch.end:
        return [0] * 64, s0, 0, []
    s1 = min(s0 + SEG, ch.end)
    nact = (s1 - s0 + U - 1) // U
    a = [s0 + L * U for L in range(64)]
    stop = [min(a[L] + U, s1) for L in range(64)]
    have = [a[L] < s1 for L in range(64)]
    mask = [0] * 64
    exit_spec = list(a)
    for L in range(64):
        pos = a[L]
        while have[L] and pos < stop[L]:
            mask[L] |= 1 << (pos - a[L])
            pos += ch.step(pos)
        exit_spec[L] = pos
    used_in = [a[0]] + exit_spec[:63]
    m_fin = list(mask)
    x_fin = list(exit_spec)
    for L in range(1, 64):
        if have[L]:
            m_fin[L], x_fin[L] = resolve(ch, used_in[L], a[L], stop[L], mask[L], exit_spec[L])
    repairs = 0
    while True:
        bad = [L for L in range(1, nact) if used_in[L] != x_fin[L - 1]]
        if not bad:
            break
        j = bad[0]
        tin = x_fin[j - 1]
        used_in[j] = tin
        m_fin[j], x_fin[j] = resolve(ch, tin, a[j], stop[j], mask[j], exit_spec[j])
        for k in range(j + 1, nact):              # groups the walk jumps over entirely
            if stop[k] <= x_fin[j]:
                used_in[k], m_fin[k], x_fin[k] = x_fin[j], 0, x_fin[j]
        repairs += 1
    if stats is not None:
        stats["repairs"] = stats.get("repairs", 0) + repairs
    for L in range(64):
        if not have[L]:
            m_fin[L] = 0
    total = sum(bin(m).count("1") for m in m_fin)
    seg_x = x_fin[nact - 1]
    staged = []
    for L in range(64):
        m = m_fin[L]
        while m:
            b = (m & -m).bit_length() - 1
            m &= m - 1
            p = a[L] + b
            nxt = a[L] + ((m & -m).bit_length() - 1) if m else x_fin[L]
            d = int(ch.cd[p])
            staged.append(((nxt - p) << 16) | d if d else ch.buf[p] << 16)
    assert len(staged) == total
    return m_fin, seg_x, total, staged


def rewalk(ch, vw, s0, s1, e, cnt, ex):
    """parse_rewalk → (cnt, ex, mpos, kspec); rewrites vw."""
    pos, walked, spec_below, merge_pos, merged = e, 0, 0, s1, False
    for g in range(64):
        if merged:
            break
        base = s0 + g * U
        if base >= s1:
            break
        stop = min(base + U, s1)
        V = vw[g]
        if pos >= stop:
            spec_below += bin(V).count("1")
            vw[g] = 0
            continue
        T, mr = 0, 64
        while pos < stop:
            r = pos - base
            if (V >> r) & 1:
                merged, mr, merge_pos = True, r, pos
                break
            T |= 1 << r
            walked += 1
            pos += ch.step(pos)
        keep = (V & ~((1 << mr) - 1)) if mr < 64 else 0
        spec_below += bin(V & ~keep).count("1")
        vw[g] = T | keep
    if merged:
        return cnt - spec_below + walked, ex, merge_pos, spec_below
    return walked, pos, s1, cnt


def emit_segment(ch, vw, s0, total, seg_exit2, mpos_in, kspec, staged):
    """parse_emit_kernel for one segment → its codes."""
    s1 = min(s0 + SEG, ch.end)
    out = []
    mpos = min(mpos_in, s1) if s0 < s1 else s0
    if mpos > s0:
        first = [s0 + g * U + ((vw[g] & -vw[g]).bit_length() - 1) if vw[g] else 0xFFFFFFFF for g in range(64)]
        nxt_g = [min(first[g + 1:] + [0xFFFFFFFF]) for g in range(64)]
        nxt_g = [min(x, seg_exit2) for x in nxt_g]
        for g in range(64):
            base = s0 + g * U
            if base >= mpos:
                break
            V = vw[g]
            m = V
            if mpos - base < 64:
                m &= (1 << (mpos - base)) - 1
            for lane in range(64):
                if (m >> lane) & 1:
                    i = base + lane
                    above = V >> (lane + 1)
                    nxt = i + 1 + ((above & -above).bit_length() - 1) if above else nxt_g[g]
                    d = int(ch.cd[i])
                    out.append(((nxt - i) << 16) | d if d else ch.buf[i] << 16)
    n2 = total - len(out)
    out += staged[kspec:kspec + n2]
    assert len(out) == total
    return out


def parse_chunk(buf, window=MAX_WINDOW, max_len=258, stats=None):
    """The whole pipeline for one chunk → code words (val << 16 | dist), as lfo_lz77_chunk returns them."""
    ch = Chunk(buf, window, max_len)
    ns = ch.n_seg
    vis, seg_exit, seg_count, staged = [], [], [], []
    for s in range(ns):
        v, x, c, st = walk_segment(ch, s, stats)
        vis.append(v); seg_exit.append(x); seg_count.append(c); staged.append(st)
    # fixseg
    seg_exit2, seg_mpos, seg_kspec = [0] * ns, [0] * ns, [0] * ns
    for s in range(ns):
        s0 = s * SEG
        s1 = min(s0 + SEG, ch.end)
        e = seg_exit[s - 1] if s else 0
        f = (seg_count[s], seg_exit[s], s0, 0)
        if s0 >= ch.end:
            f = (0, e, s0, 0)
        elif e != s0:
            f = rewalk(ch, vis[s], s0, s1, e, seg_count[s], seg_exit[s])
        seg_count[s], seg_exit2[s], seg_mpos[s], seg_kspec[s] = f
    # fix (serial statement of parse_fix_kernel)
    e_last = 0
    redo = 0
    for s in range(ns):
        assumed = seg_exit[s - 1] if s else 0
        if assumed != e_last:
            s0 = s * SEG
            s1 = min(s0 + SEG, ch.end)
            if s0 >= ch.end:
                f = (0, e_last, 0, 0)
            else:
                f = rewalk(ch, vis[s], s0, s1, e_last, seg_count[s], seg_exit2[s])
            seg_count[s], seg_exit2[s] = f[0], f[1]
            seg_mpos[s] = 0xFFFFFFFF
            redo += 1
        e_last = seg_exit2[s]
    if stats is not None:
        stats["redo"] = stats.get("redo", 0) + redo
    codes = []
    for s in range(ns):
        codes += emit_segment(ch, vis[s], s * SEG, seg_count[s], seg_exit2[s], seg_mpos[s], seg_kspec[s], staged[s])
    pos = e_last if ns else 0
    codes += [ch.buf[i] << 16 for i in range(pos, ch.n)]          # default.rs:105-107
    return np.array(codes, dtype=np.uint32)
