"""CPU, world_size 2 over gloo: the sharded-encode exchange.  Each rank contributes the
(total_bits, n_bytes, crc32, adler32) of its shard — produced here by the oracle standing in for the HIP
shard encoder, which cannot run without a GPU — all-gathers them, derives its start bit and the combined
trailer with the PRODUCT logic (libflate_amd.sharded + lfx_crc32_combine), and the assembled member must
equal what ONE encoder emits for the concatenated input."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import lfo_oracle as oracle
        import synth
        from libflate_amd import _ffi, sharded
        n = 2 << 20                                    # per rank; a multiple of the 1 MiB block size
        data = synth.text(n, seed=synth.SEED_BASE + 2 + rank).tobytes()
        # stand-in for lfx_encode_shard_prepare: raw DEFLATE of the shard; a non-last shard is the
        # stream without the final (empty, BFINAL) block and without byte alignment
        full = oracle.encode(oracle.DEFLATE, data, write_size=8192)
        blocks = oracle.scan_blocks(full)
        last = rank == world - 1
        bits = blocks[-1][1] if last else blocks[-2][1]            # end bit of the last kept block
        opts = _ffi.make_opts(mtime=0)
        import ctypes as C
        hdr_len = _ffi.lib().lfx_container_header_len(_ffi.GZIP, C.byref(opts))
        # the only collective on this path, through the LIBRARY's driver (lfx_sharded_layout) with gloo behind its lfx_comm
        start_bits, check, total_n = sharded.layout_exchange((bits, n, oracle.crc32(data), oracle.adler32(data)), hdr_len, _ffi.GZIP,
                                                              rank, world, dist)
        assert len(start_bits) == world + 1 and start_bits[0] == 8 * hdr_len
        # ... and the same arithmetic on a list gathered by hand
        mine = torch.tensor([bits, n, oracle.crc32(data), oracle.adler32(data)], dtype=torch.int64)
        allv = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allv, mine)
        sb2, check2, total2 = sharded.layout([tuple(int(x) for x in t) for t in allv], hdr_len, _ffi.GZIP)
        assert (sb2, check2, total2) == (start_bits[:world], check, total_n)
        # stand-in for lfx_encode_shard_emit: place my bits at start_bits[rank] (bit-granular shift)
        val = int.from_bytes(full, "little") & ((1 << bits) - 1)
        sb = start_bits[rank]
        first_byte = sb // 8 if rank else 0
        shifted = val << (sb - 8 * first_byte)
        nbytes = (sb - 8 * first_byte + bits + 7) // 8
        part = bytearray(shifted.to_bytes(nbytes, "little"))
        if rank == 0:
            part[:hdr_len] = bytes([31, 139, 8, 0, 0, 0, 0, 0, 0, 3])
        if last:
            part += check.to_bytes(4, "little") + (total_n & 0xFFFFFFFF).to_bytes(4, "little")
        parts = [None] * world
        dist.all_gather_object(parts, bytes(part))                  # test-only: gather for the comparison
        if rank == 0:
            member = sharded.assemble(parts, start_bits)
            whole = b"".join(synth.text(n, seed=synth.SEED_BASE + 2 + r).tobytes() for r in range(world))
            want = oracle.encode(oracle.GZIP, whole, write_size=8192)
            q.put(("ok", member == want, len(member), len(want)))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def test_sharded_layout_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == "ok", res
    assert res[1], ("assembled member differs from the single-encoder output", res)


def test_layout_and_combine_single_process(oracle):
    import numpy as np
    from libflate_amd import _ffi, sharded
    rng = np.random.default_rng(2)
    shards = [rng.integers(0, 256, int(k), dtype=np.uint8).tobytes() for k in (70000, 1, 123457, 99)]
    infos = [(1000 + 7 * i, len(s), oracle.crc32(s), oracle.adler32(s)) for i, s in enumerate(shards)]
    sb, crc, tot = sharded.layout(infos, 10, _ffi.GZIP)
    assert sb == [80, 1080, 2087, 3101] and tot == sum(len(s) for s in shards)
    assert crc == oracle.crc32(b"".join(shards))
    _, ad, _ = sharded.layout(infos, 2, _ffi.ZLIB)
    assert ad == oracle.adler32(b"".join(shards))
    # shared boundary bytes are OR-ed
    assert sharded.assemble([b"\x01\x02\x03", b"\x30\x04"], [0, 20]) == b"\x01\x02\x33\x04"


def _decode_worker(rank, world, port, q):
    """N-GPU decode of one member, the exchange: every rank contributes the tuples of the blocks that START in its byte
    range — produced here from the oracle's block scan standing in for lfx_decode_range_scan, which needs a GPU — plus
    a false candidate; the PRODUCT code all-gathers them (sharded.gather_tuples), walks the chain (lfx_decode_chain,
    host only) and folds the slice checksums (sharded.fold_checks)."""
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import ctypes as C
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import lfo_oracle as oracle
        import synth
        from libflate_amd import _ffi, sharded
        data = synth.text(5 << 20).tobytes()
        member = oracle.encode(oracle.GZIP, data, write_size=8192, mtime=0)
        hdr = 10
        blocks = oracle.scan_blocks(member[hdr:-8])              # (start_bit, end_bit, btype, bfinal, out_len) relative to the DEFLATE data
        lo, hi = sharded.byte_ranges(hdr, len(member), world)[rank]
        mine = [b for b in blocks if lo * 8 <= hdr * 8 + b[0] < hi * 8]
        tuples = (_ffi.BlkTuple * (len(mine) + 1))()
        for i, b in enumerate(mine):
            t = tuples[i]
            t.start_bit, t.end_bit, t.n_out, t.btype, t.bfinal, t.status, t.rank, t.slot = hdr * 8 + b[0], hdr * 8 + b[1], b[4], b[2], b[3], 0, rank, i
        f = tuples[len(mine)]                                    # a false candidate in the middle of my range: never on the chain
        f.start_bit, f.end_bit, f.n_out, f.status, f.rank, f.slot = (lo + hi) * 4 + 3, (lo + hi) * 4 + 4000, 999, 0, rank, len(mine)
        all_t, n_all = sharded.gather_tuples(tuples, len(mine) + 1, world, dist)
        chain, nch, total = sharded.chain_of(all_t, n_all, hdr * 8)
        got = [(all_t[chain[k]].start_bit, all_t[chain[k]].rank) for k in range(nch)]
        want = [(hdr * 8 + b[0], next(r for r, (l, h) in enumerate(sharded.byte_ranges(hdr, len(member), world)) if l * 8 <= hdr * 8 + b[0] < h * 8))
                for b in blocks]
        # slice checksums: rank r "decoded" the output of its blocks
        off = sum(b[4] for b in blocks if hdr * 8 + b[0] < lo * 8)
        ln = sum(b[4] for b in mine)
        sl = data[off:off + ln]
        # the slices' checksums folded by the library's driver step (lfx_sharded_fold) over gloo
        any_state, crc, ad, tot = sharded.fold_exchange(rank, world, dist, 0, 0, ln, oracle.crc32(sl), oracle.adler32(sl))
        ok = got == want and total == len(data) and crc == oracle.crc32(data) and ad == oracle.adler32(data)
        ok = ok and any_state == 0 and tot == len(data)
        ok = ok and crc == int.from_bytes(member[-8:-4], "little")
        if rank == 0:
            q.put(("ok", ok, nch, len(blocks), n_all))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def test_member_decode_exchange_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == "ok", res
    assert res[1], ("chain / checksum fold of the gathered tuples is wrong", res)
    assert res[2] == res[3] and res[4] == res[3] + 2          # every block on the chain; the two false candidates are not


def _error_worker(rank, world, port, q):
    """a rank whose scan fails must not leave the others waiting in the collective — the status rides with the
    counts and every rank raises after the all-gather."""
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from libflate_amd import _ffi, sharded
        tuples = (_ffi.BlkTuple * 1)()
        raised = None
        try:
            sharded.gather_tuples(tuples, 0, world, dist, status=_ffi.E_NOSPACE if rank == 1 else 0)
        except _ffi.LfxError as e:
            raised = e.status
        flag = __import__("torch").tensor([1 if raised == _ffi.E_NOSPACE else 0])
        dist.all_reduce(flag)                                   # (both ranks are still in step: no one hangs)
        if rank == 0:
            q.put(("ok", int(flag.item())))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def test_failure_on_one_rank_is_raised_on_all_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_error_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert res == ("ok", 2), res


def _p2p_worker(rank, world, port, q):
    """the comm's point-to-point half.  isend / irecv only COLLECT transfers; `start` launches them and
    returns (the shards then travel while the caller decodes); an all-gather between start and wait must not disturb them;
    wait completes.  Host memory stands in for device buffers (TorchComm with device "cpu"), driven through the C struct's
    function pointers exactly as lfx_sharded_encode_begin / _finish drive them."""
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import ctypes as C
        import numpy as np
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from libflate_amd import sharded
        cm = sharded.TorchComm(dist, rank, world, "cpu")
        n = 1 << 20
        ok = True
        if rank == 0:
            bufs = [np.zeros(n + r, dtype=np.uint8) for r in range(1, world)]
            for r, b in zip(range(1, world), bufs):
                ok &= cm.c.irecv(None, b.ctypes.data, b.size, r) == 0
            ok &= len(cm.ops) == world - 1 and not cm.works          # collected, nothing moving yet
        else:
            mine = (np.arange(n + rank, dtype=np.uint32) * (rank + 7) & 0xFF).astype(np.uint8)
            ok &= cm.c.isend(None, mine.ctypes.data, mine.size, 0) == 0
            ok &= len(cm.ops) == 1 and not cm.works
        ok &= cm.c.start(None) == 0
        ok &= not cm.ops and len(cm.works) == (world - 1 if rank == 0 else 1)      # launched, not awaited
        # a small all-gather while the transfers are in flight (what lfx_sharded_decode does between begin and finish)
        send = (C.c_uint8 * 8)(*([rank + 1] * 8))
        recv = (C.c_uint8 * (8 * world))()
        ok &= cm.c.allgather(None, C.addressof(send), C.addressof(recv), 8) == 0
        ok &= bytes(recv) == b"".join(bytes([r + 1] * 8) for r in range(world))
        ok &= cm.c.wait(None) == 0 and not cm.works
        if rank == 0:
            for r, b in zip(range(1, world), bufs):
                want = (np.arange(n + r, dtype=np.uint32) * (r + 7) & 0xFF).astype(np.uint8)
                ok &= bool((b == want).all())
            q.put(("ok", bool(ok), cm.error))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def test_comm_start_launches_posted_transfers_world3():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert res[0] == "ok" and res[1], res
