"""Sharded encode, host side: layout of the ranks' bit ranges inside ONE member, the combined
trailer checksum, and the concatenation of the shards on the writer rank.

The data path has no collective: every rank encodes its own blocks; the ranks exchange 32 bytes each
(`lfx_shard_info`, one all-gather over RCCL / xGMI), emit at their global bit offset, and the shard bytes travel
once to the writer rank (point-to-point over xGMI), where `lfx_shard_place_device` puts them in place — the byte at a
shard boundary is shared by two shards and is OR-ed."""
import ctypes as C

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
    """Host reference of the concatenation (tests): part r starts at byte start_bits[r]//8 and shares that byte
    with its predecessor when start_bits[r] % 8 != 0 (OR the halves)."""
    out = bytearray()
    for r, p in enumerate(parts):
        at = start_bits[r] // 8 if r else 0
        if r and at < len(out):
            out[at] |= p[0]
            out += p[1:]
        else:
            out += p
    return bytes(out)


def member_bytes(start_bits, part_lens):
    """size of the assembled member given every rank's emitted byte count"""
    last = len(start_bits) - 1
    return (start_bits[last] // 8 if last else 0) + part_lens[last]


def gather_begin(ctx, rank, world, d_part, part_len, start_bits, part_lens, d_member, cap, dist=None, staging=None):
    """Start concatenating the shards on rank 0 (the writer) and return a handle for gather_finish().  d_part / d_member /
    staging are torch uint8 tensors on the rank's device; `dist` is torch.distributed (None for world == 1).
    Point-to-point transfers, ALL posted at once (one batch_isend_irecv: on RCCL the shards arrive over different xGMI
    links concurrently — the links are point-to-point, a rank-by-rank receive would use one of seven at a time);
    `staging` must hold the shards of ranks 1..world-1 side by side (256-byte aligned).  Between begin and finish the
    caller is free to do other work on its own shard (the transfers run on RCCL's stream)."""
    L = _ffi.lib()

    def place(src_tensor, r):
        rc = L.lfx_shard_place_device(ctx.handle, d_member.data_ptr(), cap, src_tensor.data_ptr(), part_lens[r],
                                      start_bits[r], int(r == 0))
        if rc:
            raise _ffi.LfxError(rc, ctx.last_error())

    h = {"rank": rank, "world": world, "works": [], "pending": [], "place": place, "host_hop": False,
         "len": member_bytes(start_bits, part_lens) if rank == 0 else 0}
    if world == 1 or dist is None:
        place(d_part, 0)
        return h
    import torch
    host_hop = h["host_hop"] = dist.get_backend() == "gloo"          # (CPU test rigs: gloo moves host tensors)
    if rank == 0:
        place(d_part, 0)
        ops, off = [], 0
        for r in range(1, world):
            need = (part_lens[r] + 255) & ~255
            if off + need > staging.numel():
                raise ValueError("staging holds %d bytes, the shards of ranks 1..%d need more" % (staging.numel(), world - 1))
            buf = staging[off:off + part_lens[r]]
            off += need
            host = None
            if host_hop and buf.device.type != "cpu":
                host = torch.empty(part_lens[r], dtype=torch.uint8)
            ops.append(dist.P2POp(dist.irecv, host if host is not None else buf, r))
            h["pending"].append((r, buf, host))
        h["works"] = dist.batch_isend_irecv(ops)
        return h
    send = d_part[:part_len]
    if host_hop and send.device.type != "cpu":
        send = send.cpu()
    h["keep"] = send
    h["works"] = dist.batch_isend_irecv([dist.P2POp(dist.isend, send, 0)])
    return h


def gather_finish(h):
    """Wait for the transfers of gather_begin(); rank 0 places the received shards.  → member length on rank 0, else 0."""
    for w in h["works"]:
        w.wait()
    if h["pending"]:
        import torch
        if not h["host_hop"] and torch.cuda.is_available():
            # (wait() orders the RCCL stream before torch's current stream, not before the context's own stream on
            #  which the placement kernels run: block the host until the shards have landed)
            torch.cuda.synchronize()
        for r, buf, host in h["pending"]:
            if host is not None:
                buf.copy_(host)
            h["place"](buf, r)
    return h["len"]


def gather_member(ctx, rank, world, d_part, part_len, start_bits, part_lens, d_member, cap, dist=None, staging=None):
    """gather_begin() + gather_finish()."""
    return gather_finish(gather_begin(ctx, rank, world, d_part, part_len, start_bits, part_lens, d_member, cap, dist, staging))
