"""Sharded encode across ranks: one gzip/zlib/deflate member, contiguous block ranges per rank
(SURVEY.md §8e).  The only exchange is an all-gather of (total_bits, n_bytes, crc32, adler32) per
rank — 32 bytes each, latency-bound — from which every rank derives its start bit and the last rank
the combined trailer.  No bulk data crosses xGMI on this path."""
from . import _ffi


def layout(infos, header_len, fmt):
    """infos: list of (total_bits, n_bytes, crc32, adler32) in rank order (non-last shards end on a
    block boundary; the last holds BFINAL).  → (start_bits, combined_check, total_n)."""
    L = _ffi.lib()
    start_bits, bit = [], 8 * header_len
    for tb, _n, _c, _a in infos:
        start_bits.append(bit)
        bit += tb
    crc, adler, total = infos[0][2], infos[0][3], infos[0][1]
    for _tb, n, c, a in infos[1:]:
        crc = L.lfx_crc32_combine(crc, c, n)
        adler = L.lfx_adler32_combine(adler, a, n)
        total += n
    check = crc if fmt == _ffi.GZIP else adler
    return start_bits, check, total


def assemble(parts, start_bits):
    """Concatenate per-rank outputs (bytes) into the member: part r starts at byte start_bits[r]//8
    and shares that byte with its predecessor when start_bits[r] % 8 != 0 (OR the halves)."""
    out = bytearray()
    for r, p in enumerate(parts):
        at = start_bits[r] // 8 if r else 0
        if r and at < len(out):
            out[at] |= p[0]
            out += p[1:]
        else:
            out += p
    return bytes(out)
