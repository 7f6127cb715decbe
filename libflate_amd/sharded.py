"""N-GPU encode / decode of ONE member, host side — a thin caller of the library's own drivers.

Until round 4 this module WAS the driver (layout, concatenation, the member decode with its retries and its window hand-over);
since round 5 that sequencing lives in the C ABI (include/lfx.h: lfx_sharded_encode_begin / _finish, lfx_sharded_decode and
their exchange steps lfx_sharded_layout / _gather_tuples / _fold; libflate_amd/csrc/lfx_sharded.cpp), so that a Rust or C
caller has the same path.  What is left here: an `lfx_comm` whose five callbacks run over torch.distributed (RCCL on GPUs,
gloo in CPU rigs), and the functions the tests and bench.py call, now forwarding to the library.

The data path has no collective: every rank encodes its own blocks; the ranks exchange 32 bytes each (`lfx_shard_info`, one
all-gather over RCCL / xGMI), emit at their global bit offset, and the shard bytes travel once to the writer rank
(point-to-point over xGMI), where `lfx_shard_place_device` puts them in place — the byte at a shard boundary is shared by two
shards and is OR-ed."""
import ctypes as C

from . import _ffi

RANGE_TAIL = 4 << 20      # bytes of the right neighbour's range a rank also holds (lfx_sharded_byte_range applies it)


class TorchComm:
    """lfx_comm over torch.distributed.  allgather moves HOST bytes (through the device on RCCL: its collectives take
    device tensors); isend / irecv collect transfers of DEVICE buffers, start() launches all of them as ONE batch
    (`batch_isend_irecv`: on RCCL the shards then arrive over different xGMI links concurrently) and returns at once —
    lfx_sharded_encode_begin calls it last, so the shards are in flight while the caller decodes; wait() completes them.
    With gloo (CPU rigs, the one-GPU self-test of bench.py) the buffers hop through host tensors.  A comm of ONE rank
    without a process group has no callbacks at all (NULL: the drivers then exchange nothing)."""

    def __init__(self, dist, rank, world, device=None, group=None):
        import torch
        self.torch, self.dist, self.rank, self.world, self.group = torch, dist, rank, world, group
        self.host = dist is None or dist.get_backend() == "gloo"
        self.device = device if device is not None else "cpu"
        self.ops, self.keep, self.copies, self.works = [], [], [], []
        self.error = None
        if dist is None:
            self._cbs = ()
            self.c = _ffi.Comm(None, rank, world)
            return
        self._cbs = (_ffi.COMM_ALLGATHER(self._allgather), _ffi.COMM_P2P(self._isend), _ffi.COMM_P2P(self._irecv),
                     _ffi.COMM_WAIT(self._wait), _ffi.COMM_WAIT(self._start))
        self.c = _ffi.Comm(None, rank, world, *self._cbs)

    def _wrap(self, ptr, n):
        return (C.c_uint8 * n).from_address(ptr)

    def _allgather(self, _user, send, recv, nbytes):
        try:
            torch = self.torch
            mine = torch.frombuffer(bytearray(self._wrap(send, nbytes)), dtype=torch.uint8)
            dev = "cpu" if self.host else self.device
            parts = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(self.world)]
            self.dist.all_gather(parts, mine.to(dev), group=self.group)
            out = torch.cat([p.cpu() for p in parts]).numpy().tobytes()
            C.memmove(recv, out, len(out))
            return 0
        except Exception as e:  # noqa: BLE001  (an exception must not cross the C frame)
            self.error = e
            return 1

    def _dev_tensor(self, ptr, n):
        """a uint8 tensor over n bytes of device memory at `ptr` (no copy; the caller keeps the allocation alive).  With
        device "cpu" (the CPU rigs of tests/test_sharded_gloo.py) `ptr` is host memory."""
        torch = self.torch
        if str(self.device) == "cpu":
            return torch.frombuffer(self._wrap(ptr, n), dtype=torch.uint8)

        class _Ext:       # __cuda_array_interface__: the way to hand torch a raw device pointer
            pass
        e = _Ext()
        e.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        return torch.as_tensor(e, device=self.device)

    def _isend(self, _user, d_buf, nbytes, to):
        try:
            t = self._dev_tensor(d_buf, nbytes)
            if self.host:
                t = t.cpu()
            self.keep.append(t)
            self.ops.append(self.dist.P2POp(self.dist.isend, t, to, group=self.group))
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1

    def _irecv(self, _user, d_buf, nbytes, frm):
        try:
            dst = self._dev_tensor(d_buf, nbytes)
            if self.host:
                h = self.torch.empty(nbytes, dtype=self.torch.uint8)
                self.copies.append((dst, h))
                self.ops.append(self.dist.P2POp(self.dist.irecv, h, frm, group=self.group))
            else:
                self.keep.append(dst)
                self.ops.append(self.dist.P2POp(self.dist.irecv, dst, frm, group=self.group))
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1

    def _start(self, _user):
        """everything collected so far starts to move; returns without waiting"""
        try:
            if self.ops:
                self.works += self.dist.batch_isend_irecv(self.ops)
                self.ops = []
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1

    def _wait(self, _user):
        try:
            if self._start(_user):
                return 1
            for w in self.works:
                w.wait()
            for dst, h in self.copies:
                dst.copy_(h)
            if not self.host and self.torch.cuda.is_available():
                # (wait() orders the RCCL stream before torch's current stream, not before the context's own stream on
                #  which the placement kernels run: block the host until the shards have landed)
                self.torch.cuda.synchronize()
            self.works, self.keep, self.copies = [], [], []
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1


def _comm(dist, rank, world, device=None, group=None):
    if dist is None and world > 1:
        raise ValueError("a sharded call with world = %d needs a torch.distributed module (dist=None)" % world)
    if world == 1:
        return TorchComm(None, rank, world, device)
    return TorchComm(dist, rank, world, device, group)


def _raise(rc, ctx, cm, what):
    if cm is not None and cm.error is not None:
        raise cm.error
    raise _ffi.LfxError(rc, "%s: %s" % (what, ctx.last_error() if ctx is not None else ""))


# ------------------------------------------------------------------------------------------------ exchange steps
def layout(infos, header_len, fmt):
    """infos: list of (total_bits, n_bytes, crc32, adler32) in rank order (non-last shards end on a block boundary; the last
    holds BFINAL) → (start_bits, combined_check, total_n).  The same arithmetic as lfx_sharded_layout, on a list the caller
    already gathered (tests; bench.py's first rounds)."""
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


def layout_exchange(info, header_len, fmt, rank, world, dist=None, group=None):
    """lfx_sharded_layout: all-gather of the ranks' shard infos (32 bytes each, the ONE collective on the encode's data path)
    → (start_bits of every rank + the end bit, combined check, total uncompressed bytes).  info: (bits, bytes, crc, adler)."""
    cm = _comm(dist, rank, world, group=group)
    mine = _ffi.ShardInfo(*info)
    sb = (C.c_uint64 * (world + 1))()
    check, total = C.c_uint32(0), C.c_uint64(0)
    rc = _ffi.lib().lfx_sharded_layout(C.byref(cm.c), C.byref(mine), header_len, fmt, sb, C.byref(check), C.byref(total))
    if rc:
        _raise(rc, None, cm, "lfx_sharded_layout")
    return list(sb), check.value, total.value


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
    last = len(part_lens) - 1
    return (start_bits[last] // 8 if last else 0) + part_lens[last]


# ------------------------------------------------------------------------------------------------ sharded encode
def encode_begin(ctx, rank, world, fmt, opts, sched, d_in, n, d_part, part_cap, d_member=None, member_cap=0, staging=None,
                 dist=None, group=None):
    """lfx_sharded_encode_begin on torch uint8 device tensors → (handle, ShardedPart).  The shards are travelling to rank 0
    when this returns (posted AND started: lfx_comm.start, TorchComm._start); encode_finish() waits for them and completes
    the member.  Between the two the caller may work on its own shard (d_part)."""
    cm = _comm(dist, rank, world, d_part.device, group)
    st = C.c_void_p(None)
    part = _ffi.ShardedPart()
    rc = _ffi.lib().lfx_sharded_encode_begin(ctx.handle, C.byref(cm.c), fmt, C.byref(opts), C.byref(sched), d_in.data_ptr(), n,
                                             d_part.data_ptr(), part_cap, d_member.data_ptr() if d_member is not None else None,
                                             member_cap, staging.data_ptr() if staging is not None else None,
                                             staging.numel() if staging is not None else 0, C.byref(st), C.byref(part))
    if rc:
        _raise(rc, ctx, cm, "lfx_sharded_encode_begin")
    return {"cm": cm, "state": st, "ctx": ctx}, part


def encode_finish(h):
    """→ member length on rank 0, else 0"""
    ml = C.c_uint64(0)
    rc = _ffi.lib().lfx_sharded_encode_finish(h["ctx"].handle, C.byref(h["cm"].c), h["state"], C.byref(ml))
    if rc:
        _raise(rc, h["ctx"], h["cm"], "lfx_sharded_encode_finish")
    return ml.value


# ------------------------------------------------------------------------------------------------ member decode by byte ranges
def byte_ranges(first_byte, member_len, world):
    """Equal byte ranges of the DEFLATE part [first_byte, member_len) (the trailer's few bytes ride along in the last one) →
    list of (lo, hi); a rank holds [lo, min(hi + RANGE_TAIL, member_len)) (lfx_sharded_byte_range)."""
    out = []
    for r in range(world):
        lo, hi, hold = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _ffi.lib().lfx_sharded_byte_range(first_byte, member_len, r, world, C.byref(lo), C.byref(hi), C.byref(hold))
        out.append((lo.value, hi.value))
    return out


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


def fold_exchange(rank, world, dist, status, state, length, crc, adler, group=None):
    """lfx_sharded_fold: all-gather of (status, state, length, crc32, adler32) → (any rank in state 1?, crc32, adler32, total
    length) of the concatenation; a non-zero status of ANY rank raises on EVERY rank."""
    cm = _comm(dist, rank, world, group=group)
    anyst, c_all, a_all, tot, bad = C.c_uint32(0), C.c_uint32(0), C.c_uint32(1), C.c_uint64(0), C.c_uint32(0)
    rc = _ffi.lib().lfx_sharded_fold(C.byref(cm.c), status, state, length, crc, adler, C.byref(anyst), C.byref(c_all),
                                     C.byref(a_all), C.byref(tot), C.byref(bad))
    if rc:
        if cm.error is not None:
            raise cm.error
        raise _ffi.LfxError(rc, "slice decode failed on rank %d: decode the member on one GPU" % bad.value)
    return anyst.value, c_all.value, a_all.value, tot.value


def gather_tuples(tuples, cnt, world, dist, device="cpu", group=None, status=0, rank=None):
    """step 2 (lfx_sharded_gather_tuples): all-gather of the ranks' candidate tuples (variable counts: the counts first, then
    rows padded to the longest) → (ctypes array of all tuples in rank order, their number).  `status`: this rank's error code
    of step 1 — it rides with the counts, and a failure on ANY rank is raised on EVERY rank after the collective (a
    rank that raises before a collective leaves the others waiting in it)."""
    if rank is None:
        rank = dist.get_rank() if dist is not None and world > 1 else 0
    cm = _comm(dist, rank, world, device, group)
    allp = C.POINTER(_ffi.BlkTuple)()
    n_all, bad = C.c_uint32(0), C.c_uint32(0)
    rc = _ffi.lib().lfx_sharded_gather_tuples(C.byref(cm.c), tuples, cnt, status, C.byref(allp), C.byref(n_all), C.byref(bad))
    if rc:
        if cm.error is not None:
            raise cm.error
        raise _ffi.LfxError(rc, "range scan failed on rank %d: decode the member on one GPU" % bad.value)
    out = (_ffi.BlkTuple * max(n_all.value, 1))()
    if n_all.value:
        C.memmove(out, allp, n_all.value * C.sizeof(_ffi.BlkTuple))
    _ffi.lib().lfx_sharded_free(allp)
    return out, n_all.value


def decode_member_ranks(ctx, rank, world, d_part, n_part, lo, hi, first_bit, d_out, cap, dist=None, group=None, member_len=None):
    """One rank's side of the N-GPU decode (lfx_sharded_decode) over torch.distributed: scan → all-gather of the tuples → chain
    → emit → [window hand-over: all-gather of the ranks' 64 KiB index maps, only for members whose blocks read earlier blocks]
    → all-gather of (status, length, crc, adler).  d_part / d_out: torch uint8 tensors on the rank's device; `group`: a process
    group of its own for these small collectives, so that they do not queue behind bulk transfers posted on the default group
    (bench.py: the member's concatenation is in flight).  Every failure is carried through the next collective and comes back
    on ALL ranks.  → (bytes of this rank's slice, its offset in the member's output, total output bytes, crc32, adler32 of the
    whole member's output)."""
    cm = _comm(dist, rank, world, d_part.device, group)
    sl = _ffi.ShardedSlice()
    rc = _ffi.lib().lfx_sharded_decode(ctx.handle, C.byref(cm.c), d_part.data_ptr(), n_part, lo, hi, first_bit, member_len or 0,
                                       d_out.data_ptr(), cap, C.byref(sl))
    if rc:
        _raise(rc, ctx, cm, "lfx_sharded_decode")
    return sl.out_len, sl.out_base, sl.total_out, sl.crc32, sl.adler32
