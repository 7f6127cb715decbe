"""CPU model of the block header's code-length decode as parse_header runs it since round 5 (lfx_inflate_fast.hip): a wavefront
takes a WINDOW of 64 bit offsets at a time; lane i decodes the symbol that would start at offset i; a scalar walk marks the
offsets the true chain visits; output positions, symbol 16's "previous width", the checks and the stores are data-parallel over
the marked lanes.  `serial()` is the walk the reference does (decode.rs:166-223 via symbol.rs:245-331: widths 0-15, 16 = repeat
the previous width 3-6 times, 17 / 18 = 3-10 / 11-138 zeros), one symbol after the other.  tests/test_host_pipeline.py runs the
two against each other on random headers, clean and damaged.  Not product code."""

REP_BITS = {16: 2, 17: 3, 18: 7}
REP_BASE = {16: 3, 17: 3, 18: 11}


def build_table(cl_widths):
    """cl_widths[s] = width of code-length symbol s (0 = unused) -> 128-entry table: 7 stream bits -> (symbol, width) or None.
    Canonical codes (symbol.rs:354-369), stored bit-reversed as the stream has them."""
    codes = {}
    code = 0
    for w in range(1, 8):
        for s in range(19):
            if cl_widths[s] == w:
                codes[s] = (code, w)
                code += 1
        code <<= 1
    tab = [None] * 128
    for s, (c, w) in codes.items():
        r = int(format(c, "0%db" % w)[::-1], 2)
        for hi in range(1 << (7 - w)):
            tab[r | hi << w] = (s, w)
    return tab


def bits_at(bits, pos, n):
    v = 0
    for k in range(n):
        v |= (bits[pos + k] if pos + k < len(bits) else 0) << k
    return v


def serial(bits, tab, total):
    """-> (lengths, end position) or None when the sequence is damaged"""
    lens, pos, last = [], 0, 0
    while len(lens) < total:
        e = tab[bits_at(bits, pos, 7)]
        if e is None:
            return None
        sym, used = e
        pos += used
        if sym < 16:
            rep, val = 1, sym
        else:
            if sym == 16 and not lens:
                return None
            rep = REP_BASE[sym] + bits_at(bits, pos, REP_BITS[sym])
            pos += REP_BITS[sym]
            val = last if sym == 16 else 0
        if len(lens) + rep > total:
            return None
        lens += [val] * rep
        last = val
    return lens, pos


def windowed(bits, tab, total):
    """the same by windows of 64 bit offsets, step for step what the kernel's lanes do"""
    lens = [0] * total
    rel, have, last = 0, 0, 0
    while True:
        # every lane: the symbol that would start at its offset
        sym, used, rep, nxt, bad_code = [0] * 64, [0] * 64, [0] * 64, [0] * 64, [False] * 64
        for i in range(64):
            e = tab[bits_at(bits, rel + i, 7)]
            if e is None:
                bad_code[i], sym[i], used[i] = True, 31, 7
            else:
                sym[i], used[i] = e
            nbx = REP_BITS.get(sym[i], 0)
            rep[i] = (REP_BASE[sym[i]] if sym[i] in REP_BASE else 1) + bits_at(bits, rel + i + used[i], nbx)
            nxt[i] = i + used[i] + nbx
        # scalar walk: the offsets the chain visits
        marked, i = [], 0
        while i < 64:
            marked.append(i)
            i = nxt[i]
        # data-parallel over the marked lanes
        at, acc = {}, have
        for m in marked:
            at[m] = acc
            acc += rep[m]
        real = [m for m in marked if at[m] < total]
        for m in real:
            if bad_code[m] or (sym[m] == 16 and at[m] == 0) or at[m] + rep[m] > total:
                return None
        definers = [m for m in real if sym[m] != 16]
        val = {}
        for m in real:
            own = sym[m] if sym[m] < 16 else 0
            if sym[m] == 16:
                before = [d for d in definers if d < m]
                val[m] = (sym[before[-1]] if sym[before[-1]] < 16 else 0) if before else last
            else:
                val[m] = own
            if val[m]:
                for k in range(rep[m]):
                    lens[at[m] + k] = val[m]
        lr = real[-1]
        have, last = at[lr] + rep[lr], val[lr]
        rest = [m for m in marked if m not in real]
        if have >= total:
            return lens, rel + (rest[0] if rest else i)
        rel += i


def encode_lengths(lengths, cl_widths, rng):
    """a bit string that decodes to `lengths` under the code `cl_widths` (random choice between runs and single symbols)"""
    codes = {}
    code = 0
    for w in range(1, 8):
        for s in range(19):
            if cl_widths[s] == w:
                codes[s] = (code, w)
                code += 1
        code <<= 1
    out = []

    def put_sym(s):
        c, w = codes[s]
        out.extend(int(b) for b in format(c, "0%db" % w))          # Huffman codes go MSB first (bit.rs)

    def put_bits(v, n):
        out.extend((v >> k) & 1 for k in range(n))                   # extra bits LSB first
    i, n = 0, len(lengths)
    while i < n:
        v = lengths[i]
        run = 1
        while i + run < n and lengths[i + run] == v:
            run += 1
        if v == 0 and run >= 11 and 18 in codes and rng.random() < 0.8:
            r = min(run, 138); put_sym(18); put_bits(r - 11, 7); i += r
        elif v == 0 and run >= 3 and 17 in codes and rng.random() < 0.8:
            r = min(run, 10); put_sym(17); put_bits(r - 3, 3); i += r
        elif i > 0 and lengths[i - 1] == v and run >= 3 and 16 in codes and rng.random() < 0.8:
            r = min(run, 6); put_sym(16); put_bits(r - 3, 2); i += r
        else:
            put_sym(v); i += 1
    return out
