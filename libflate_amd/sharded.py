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


# ------------------------------------------------------------------------------------------------
# N-GPU decode of ONE member without the encoder's layout (include/lfx.h: lfx_decode_range_scan / lfx_decode_chain /
# lfx_decode_range_emit).  The member is cut by compressed BYTES; the only collective is one all-gather of the ranks'
# candidate tuples (56 bytes each) and one of their slice checksums.
RANGE_TAIL = 4 << 20      # bytes of the right neighbour's range a rank also holds: a block that starts in a rank's range
                          # is scanned to its end (reference-made blocks: <= ~1.1 MB of stream)


def byte_ranges(first_byte, member_len, world):
    """Equal byte ranges of the DEFLATE part [first_byte, member_len) (the trailer's few bytes ride along in the last
    one) → list of (lo, hi); a rank holds [lo, min(hi + RANGE_TAIL, member_len))."""
    span = member_len - first_byte
    cuts = [first_byte + span * r // world for r in range(world)] + [member_len]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def final_from(member_len):
    """the finder's tail rule for a member of member_len bytes: BFINAL headers are looked for in its last eighth (at least
    8 MiB) — the same rule the one-GPU decode applies (lfx_decode.cpp, inflate_member) → member bit"""
    tail = max(member_len // 8, 8 << 20)
    return max(member_len - tail, 0) * 8


def range_scan(ctx, rank, d_part_ptr, n_part, lo, hi, first_bit=None, cap=1 << 16, final_from_bit=0):
    """step 1 on one rank → list of BlkTuple (ctypes array slice)"""
    L = _ffi.lib()
    tuples = (_ffi.BlkTuple * cap)()
    cnt = C.c_uint32(0)
    rc = L.lfx_decode_range_scan(ctx.handle, d_part_ptr, n_part, lo, hi, (1 << 64) - 1 if first_bit is None else first_bit,
                                 final_from_bit, rank, tuples, cap, C.byref(cnt))
    if rc:
        raise _ffi.LfxError(rc, ctx.last_error())
    return tuples, cnt.value


def chain_of(all_tuples, n_all, first_bit):
    """step 3 (host, deterministic) → (chain indices as a ctypes array, length, total output bytes)"""
    L = _ffi.lib()
    chain = (C.c_uint32 * max(n_all, 1))()
    nch, total = C.c_uint32(0), C.c_uint64(0)
    rc = L.lfx_decode_chain(all_tuples, n_all, first_bit, chain, max(n_all, 1), C.byref(nch), C.byref(total))
    if rc:
        raise _ffi.LfxError(rc, "the block chain of the member breaks (a stored / fixed block, or damage): decode it on one GPU")
    return chain, nch.value, total.value


def range_emit(ctx, rank, d_part_ptr, n_part, lo, all_tuples, chain, n_chain, d_out_ptr, cap):
    """step 4 on one rank → (bytes of the slice, its offset in the member's output, state); state 1: the slice is held as
    symbols and needs the window in front of it (steps 5-6: range_map, all-gather, range_finish)"""
    L = _ffi.lib()
    ol, base, state = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
    rc = L.lfx_decode_range_emit(ctx.handle, d_part_ptr, n_part, lo, all_tuples, chain, n_chain, rank, d_out_ptr, cap,
                                 C.byref(ol), C.byref(base), C.byref(state))
    if rc:
        raise _ffi.LfxError(rc, ctx.last_error())
    return ol.value, base.value, state.value


def range_map(ctx, d_map_ptr):
    """step 5 on one rank: the slice's index map (32768 uint16 on the device)"""
    rc = _ffi.lib().lfx_decode_range_map(ctx.handle, d_map_ptr)
    if rc:
        raise _ffi.LfxError(rc, ctx.last_error())


def range_finish(ctx, rank, d_maps_ptr=None):
    """step 6 on one rank → (crc32, adler32) of the slice; d_maps_ptr: every rank's map in rank order (device), or None"""
    crc, ad = C.c_uint32(0), C.c_uint32(0)
    rc = _ffi.lib().lfx_decode_range_finish(ctx.handle, d_maps_ptr, rank, C.byref(crc), C.byref(ad))
    if rc:
        raise _ffi.LfxError(rc, ctx.last_error())
    return crc.value, ad.value


def fold_checks(parts):
    """parts: (length, crc32, adler32) per rank, in rank order → (crc32, adler32) of the concatenation"""
    L = _ffi.lib()
    crc, ad = 0, 1
    for n, c, a in parts:
        crc = L.lfx_crc32_combine(crc, c, n)
        ad = L.lfx_adler32_combine(ad, a, n)
    return crc, ad


def gather_tuples(tuples, cnt, world, dist, device="cpu", group=None, status=0):
    """step 2: all-gather of the ranks' candidate tuples (variable counts: the counts first, then rows padded to the
    longest) → (ctypes array of all tuples in rank order, their number).  `status`: this rank's error code of step 1 —
    it rides with the counts, and a failure on ANY rank is raised on EVERY rank after the collective (a rank
    that raises before a collective leaves the others waiting in it)."""
    if dist is None or world == 1:
        if status:
            raise _ffi.LfxError(status, "range scan failed")
        return tuples, cnt
    import torch
    tsz = C.sizeof(_ffi.BlkTuple)
    dev = "cpu" if dist.get_backend() == "gloo" else device
    counts = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([cnt, status], dtype=torch.int64, device=dev), group=group)
    counts = [[int(v) for v in x.cpu().tolist()] for x in counts]
    bad = [(r, st) for r, (_, st) in enumerate(counts) if st]
    if bad:
        raise _ffi.LfxError(bad[0][1], "range scan failed on rank %d: decode the member on one GPU" % bad[0][0])
    counts = [c0 for c0, _ in counts]
    width = max(max(counts), 1) * tsz
    mine = torch.zeros(width, dtype=torch.uint8)
    if cnt:
        mine[:cnt * tsz] = torch.frombuffer(bytearray(C.string_at(tuples, cnt * tsz)), dtype=torch.uint8)
    mine = mine.to(dev)
    gathered = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)         # RCCL over xGMI: a few KB per rank
    n_all = sum(counts)
    all_t = (_ffi.BlkTuple * max(n_all, 1))()
    at = 0
    for r in range(world):
        raw = gathered[r][:counts[r] * tsz].cpu().numpy().tobytes()
        C.memmove(C.byref(all_t, at * tsz), raw, len(raw))
        at += counts[r]
    return all_t, n_all


def decode_member_ranks(ctx, rank, world, d_part, n_part, lo, hi, first_bit, d_out, cap, dist=None, group=None, member_len=None):
    """One rank's side of the N-GPU decode over torch.distributed (RCCL on GPUs, gloo in CPU rigs): scan → all-gather of
    the tuples → chain → emit → [window hand-over: all-gather of the ranks' 64 KiB index maps, only for members whose blocks
    read earlier blocks] → all-gather of (length, crc, adler).  d_part / d_out: torch uint8 tensors on the rank's
    device; `group`: a process group of its own for these small collectives, so that they do not queue behind bulk
    transfers posted on the default group (bench.py: the member's concatenation is in flight).  Every failure is carried
    through the next collective and raised on ALL ranks.  → (bytes of this rank's slice, its offset in the member's
    output, total output bytes, crc32, adler32 of the whole member's output)."""
    import torch
    # `member_len` (bytes of the whole member, when the caller knows it): the finder looks for the BFINAL header only near
    # the member's end; if the chain then breaks (a last block that starts earlier) every rank scans again without the rule
    ffb = final_from(member_len) if member_len else 0
    while True:
        status = 0
        tuples, cnt = None, 0
        try:
            tuples, cnt = range_scan(ctx, rank, d_part.data_ptr(), n_part, lo, hi, first_bit if rank == 0 else None, final_from_bit=ffb)
        except _ffi.LfxError as e:
            status = e.status or _ffi.E_UNSUPPORTED
        all_t, n_all = gather_tuples(tuples, cnt, world, dist, d_part.device, group, status)
        try:
            chain, nch, total = chain_of(all_t, n_all, first_bit)        # (deterministic: breaks on every rank alike)
            break
        except _ffi.LfxError:
            if not ffb:
                raise
            ffb = 0
    ol = base = state = 0
    crc, ad = 0, 1
    try:
        ol, base, state = range_emit(ctx, rank, d_part.data_ptr(), n_part, lo, all_t, chain, nch, d_out.data_ptr(), cap)
        if state == 0:
            crc, ad = range_finish(ctx, rank)
    except _ffi.LfxError as e:
        status = e.status or _ffi.E_UNSUPPORTED
    if dist is None or world == 1:
        if status:
            raise _ffi.LfxError(status, ctx.last_error())
        if state:
            crc, ad = range_finish(ctx, rank)
        return ol, base, total, crc, ad
    host = dist.get_backend() == "gloo"
    dev = "cpu" if host else d_part.device

    def gather5(vals):
        mine = torch.tensor(vals, dtype=torch.int64, device=dev)
        parts = [torch.empty(len(vals), dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        return [[int(v) for v in p.cpu().tolist()] for p in parts]

    rows = gather5([status, state, ol, crc, ad])
    bad = [(r, row[0]) for r, row in enumerate(rows) if row[0]]
    if bad:
        raise _ffi.LfxError(bad[0][1], "slice decode failed on rank %d: decode the member on one GPU" % bad[0][0])
    if any(row[1] for row in rows):
        # ---- window hand-over (another encoder's member): every rank's slice as one index map, all-gathered
        d_map = torch.empty(32768, dtype=torch.int16, device=d_part.device)
        try:
            range_map(ctx, d_map.data_ptr())
        except _ffi.LfxError as e:
            status = e.status or _ffi.E_UNSUPPORTED
        mine = d_map.cpu() if host else d_map
        maps = [torch.empty(32768, dtype=torch.int16, device=dev) for _ in range(world)]
        dist.all_gather(maps, mine, group=group)                 # RCCL over xGMI: 64 KiB per rank
        d_maps = torch.stack(maps).to(d_part.device).contiguous()
        try:
            if not status:
                crc, ad = range_finish(ctx, rank, d_maps.data_ptr())
        except _ffi.LfxError as e:
            status = e.status or _ffi.E_UNSUPPORTED
        rows = gather5([status, 0, ol, crc, ad])
        bad = [(r, row[0]) for r, row in enumerate(rows) if row[0]]
        if bad:
            raise _ffi.LfxError(bad[0][1], "window hand-over failed on rank %d" % bad[0][0])
    crc_all, ad_all = fold_checks([(row[2], row[3], row[4]) for row in rows])
    return ol, base, total, crc_all, ad_all
