#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: GB/s gzip encode+decode on 256 MiB synthetic text per GPU.

One step = gzip-encode the rank's 256 MiB TEXT buffer (write schedule S8K: 8192-byte writes, as
examples/flate.rs does → 1024 LZ77 chunks, 256 one-MiB dynamic blocks + the empty final block) and
decode it back, with input and output resident in HBM.  value = whole-job uncompressed bytes per
second through the round trip: (ranks x 256 MiB) / (encode time + decode time).

N > 1 (one rank per GPU, RCCL): the ranks' buffers form ONE gzip member; the only collective is the
32-byte-per-rank all-gather of (bits, bytes, crc, adler) — scaling is weak.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_BYTES = 256 << 20
WRITE = 8192
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)


# phase (HIP-event bracket in the library) -> the kernel that fills it
PHASE_KERNEL = {"enc:lz77_match": "lz77_match_kernel", "dec:lz77_copy": "blk_materialize_kernel",
                "dec:blk_scan": "blk_scan_kernel", "dec:blk_emit": "blk_emit_kernel"}


def hbm_traffic(phase, n):
    """HBM bytes per launch of the phase's kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, gfx950 x2 correction on FETCH_SIZE: profiles/r01_hbm_traffic.json).
    Only valid for the workload it was measured on; None otherwise."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("workload_bytes") != n:
            return None
        return t["kernels"][PHASE_KERNEL[phase]]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=N_BYTES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", choices=["S8K", "S1"], default="S8K",
                    help="write schedule of the encoder: S8K = 8192-byte writes (the metric's configuration), "
                         "S1 = one write_all (one LZ77 chunk, one block: SURVEY cfg2's second schedule)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as g
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # self-test of the N>1 path on a one-GPU box: every rank on GPU 0, the 32-byte exchange over gloo
    one_gpu_test = os.environ.get("LFX_BENCH_ONE_GPU") == "1"
    if one_gpu_test:
        local = 0
    if rank == 0:
        g.build()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if one_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        dist.barrier()
    import libflate_amd
    from libflate_amd import _ffi, sharded
    import synth
    import ctypes as C

    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    ctx = libflate_amd.Context(local)
    ctx.enable_timing(True)
    n = args.bytes
    data = synth.text(n, seed=synth.SEED_BASE + 2 + rank)
    d_in = torch.from_numpy(data).to(dev)
    write = WRITE if args.schedule == "S8K" else 0       # 0 = one write_all
    opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(write)
    bound = _ffi.lib().lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device=dev)
    d_dec = torch.empty(n, dtype=torch.uint8, device=dev)
    hdr_len = _ffi.lib().lfx_container_header_len(_ffi.GZIP, C.byref(opts))
    L = _ffi.lib()

    def check(rc, what):
        if rc:
            raise RuntimeError("%s failed: %d %s" % (what, rc, ctx.last_error()))

    phase_acc = {}

    def acc_timing(prefix):
        t = ctx.last_timing()
        if t:
            for name, ms in t["phases"]:
                k = prefix + name
                phase_acc.setdefault(k, []).append(ms)

    def step(record=False):
        """→ (t_enc, t_dec, compressed bytes of this rank)"""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if world == 1:
            m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
            t1 = time.perf_counter()
            if record:
                acc_timing("enc:")
            rc, ol, used, msg = ctx.decode_device(_ffi.GZIP, d_out.data_ptr(), m, d_dec.data_ptr(), n)
            if rc or ol != n or used != m:
                raise RuntimeError("decode failed rc=%d out=%d used=%d %s" % (rc, ol, used, msg))
            t2 = time.perf_counter()
            if record:
                acc_timing("dec:")
            return t1 - t0, t2 - t1, m
        info = _ffi.ShardInfo()
        check(L.lfx_encode_shard_prepare(ctx.handle, _ffi.GZIP, C.byref(opts), C.byref(sched), d_in.data_ptr(),
                                         n, int(rank == 0), int(rank == world - 1), C.byref(info)), "shard_prepare")
        mine = torch.tensor([info.total_bits, info.n_bytes, info.crc32, info.adler32], dtype=torch.int64, device=dev)
        allv = torch.empty(world * 4, dtype=torch.int64, device=dev)
        if one_gpu_test:
            parts = [torch.empty(4, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(parts, mine.cpu())
            allv = torch.cat(parts).to(dev)
        else:
            dist.all_gather_into_tensor(allv, mine)    # RCCL over xGMI: 32 B per rank
        infos = [tuple(int(x) for x in row) for row in allv.view(world, 4).cpu().tolist()]
        start_bits, combined, total_n = sharded.layout(infos, hdr_len, _ffi.GZIP)
        m = C.c_uint64(0)
        check(L.lfx_encode_shard_emit(ctx.handle, start_bits[rank], combined, total_n, d_out.data_ptr(), bound,
                                      C.byref(m)), "shard_emit")
        t1 = time.perf_counter()
        if record:
            acc_timing("enc:")
        ol = C.c_uint64(0)
        sb = start_bits[rank]
        first_byte_bits = sb if rank == 0 else sb & 7
        check(L.lfx_decode_shard_device(ctx.handle, d_out.data_ptr(), m.value, first_byte_bits, infos[rank][0],
                                        int(rank == world - 1), d_dec.data_ptr(), n, C.byref(ol)), "shard_decode")
        if ol.value != n:
            raise RuntimeError("shard decode produced %d bytes" % ol.value)
        t2 = time.perf_counter()
        if record:
            acc_timing("dec:")
        return t1 - t0, t2 - t1, m.value

    # ---- correctness gate (untimed): the round trip must reproduce the input bit for bit
    _, _, m = step()
    if not torch.equal(d_dec, d_in):
        raise RuntimeError("round trip mismatch")
    if world == 1:
        import zlib
        comp = d_out[:m].cpu().numpy().tobytes()
        assert int.from_bytes(comp[-8:-4], "little") == zlib.crc32(data.tobytes()), "CRC-32 trailer mismatch"
        if n <= (64 << 20):
            assert zlib.decompress(comp, 31) == data.tobytes()

    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    enc_t = dec_t = 0.0
    for _ in range(args.steps):
        a, b, m = step(record=True)
        enc_t += a
        dec_t += b
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if dist:
        tt = torch.tensor([elapsed, enc_t, dec_t], dtype=torch.float64, device="cpu" if one_gpu_test else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, enc_t, dec_t = (float(x) for x in tt.cpu())
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_bytes = n * world
    value = total_bytes / (elapsed / args.steps) / 1e9
    # ---- roofline of the dominant kernel phase (HIP events on the context's stream)
    avg = {k: sum(v) / len(v) for k, v in phase_acc.items()}
    kernel_phases = {k: v for k, v in avg.items() if k.split(":")[1] not in ("upload", "start", "done")}
    dom = max(kernel_phases, key=kernel_phases.get) if kernel_phases else None
    algo_bytes = n + m     # encode N read + C written; decode C read + N written
    roof = None
    if dom:
        ach = algo_bytes / (avg[dom] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": hbm_traffic(dom, n),
                "avg_launch_ms": round(avg[dom], 4), "algorithmic_bytes": algo_bytes}
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import lfo_oracle as oracle
        sample = min(n, 64 << 20)
        buf = data[:sample].tobytes()
        t0 = time.perf_counter()
        enc = oracle.encode(oracle.GZIP, buf, write_size=write)
        t1 = time.perf_counter()
        rc, out, _, _ = oracle.decode(oracle.GZIP, enc)
        t2 = time.perf_counter()
        assert rc == 0 and out == buf
        # the GPU output must equal the oracle's on the same prefix only if the prefix is the whole input
        if sample == n:
            assert enc == d_out[:m].cpu().numpy().tobytes(), "GPU output differs from the oracle"
        cpu = {"value": round(sample / (t2 - t0) / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port",
               "sample": "first %d MiB of the same buffer, %s, oracle C restatement (encode %.3f GB/s, decode %.3f GB/s)"
                         % (sample >> 20, args.schedule, sample / (t1 - t0) / 1e9, sample / (t2 - t1) / 1e9)}
    line = {
        "metric": "gzip encode+decode throughput on 256 MiB synthetic text per GPU (uncompressed bytes through the round trip)",
        "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "cfg2: gzip::Encoder (DefaultLz77Encoder, default options, mtime=0) + gzip::Decoder on "
                               "TEXT(%d B) per GPU, write schedule %s" % (n, "S8K (8192-byte writes)" if args.schedule == "S8K" else "S1 (one write_all)"),
                   "bytes_per_gpu": n, "schedule": args.schedule, "compressed_bytes": m,
                   "parallelism": "1 rank" if world == 1 else "%d ranks, one member, all-gather of shard infos" % world},
        "encode_GBps": round(total_bytes * args.steps / enc_t / 1e9, 4),
        "decode_GBps": round(total_bytes * args.steps / dec_t / 1e9, 4),
        "phases_ms": {k: round(v, 4) for k, v in sorted(avg.items())},
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
