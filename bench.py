#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: GB/s gzip encode+decode on 256 MiB synthetic text per GPU.

One step = gzip-encode the rank's 256 MiB TEXT buffer (write schedule S8K: 8192-byte writes, as
examples/flate.rs does → 1024 LZ77 chunks, 256 one-MiB dynamic blocks + the empty final block) and
decode it back, with input and output resident in HBM.  value = whole-job uncompressed bytes per
second through the round trip: (ranks x 256 MiB) / (encode time + decode time).

N > 1 (one rank per GPU, RCCL): the ranks' buffers form ONE gzip member (lfx_sharded_encode_begin / _finish).  Collectives on
the path: the 32-byte-per-rank all-gather of (bits, bytes, crc, adler) and the point-to-point transfer of every shard to the
writer rank, which concatenates them (boundary bytes OR-ed) — posted AND started inside begin (lfx_comm.start), so that they are in
flight while the ranks decode; `overlap_ms` = what finish still waited, `gather_alone_ms` = the same wait with nothing in between.
Decode at N > 1 is the N-GPU decode of that ONE member by byte ranges (lfx_sharded_decode: block finder + scan per range, one
all-gather of candidate tuples, the same chain walk on every rank; no bit offset from the encoder).  Scaling is weak by default.

The line also carries (N = 1): `roofline` of the dominant kernel with HBM traffic measured IN THIS RUN (two rocprofv3
--pmc child passes of this script: the L2's memory-side read requests by size, WRITE_SIZE), `whole_path` (2(N+C)/t_step against
the HBM peak), the second write schedule of SURVEY cfg2 (`schedule_S1`), `other_configs` (cfg3, cfg5), `pcie_inclusive`,
`stream_api` (the reference's io::copy protocol through the stream ABI, from C), `raw_deflate`, `match_fallbacks`, and `cpu_baseline` — the oracle on one host core over the SAME 256 MiB (its output must
equal the GPU's byte for byte) plus the "N streams on N cores" figure with the host's core count.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_BYTES = 256 << 20
WRITE = 8192
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec.  The peak a COPY reaches on the box at hand is measured in
                           # every run (measure_copy_peak) and reported beside it: roofline.peak_measured / frac_measured

# phase (HIP-event bracket in the library) -> the kernel that fills it (the match phase: the candidate kernel; its resolver
# and compaction kernels are the rest of that bracket)
PHASE_KERNEL = {"enc:lz77_match": "lz77_match7_kernel", "dec:lz77_copy": "blk_materialize2_kernel",
                "dec:blk_scan": "blk_scan_kernel", "dec:blk_emit": "blk_emit_kernel", "enc:lz77_parse": "parse_walk_kernel",
                # (timing level 6, the main record: the encode's match and parse phases split by kernel)
                "enc:lz77_cand": "lz77_match7_kernel", "enc:lz77_resolve": "lz77_resolve7_kernel", "enc:lz77_walk": "parse_walk_kernel",
                "enc:lz77_chain": "parse_emit_hist_kernel",
                "dec:find1": "find_blocks_stage1", "dec:find2": "find_blocks_stage2", "enc:pack": "pack_kernel",
                "dec:batch_copy": "blk_materialize2_kernel", "dec:fast": "blk_scan_kernel"}
# resident wavefronts per SIMD of those kernels (workgroup size x workgroups per CU / 4), for measure_bound
PHASE_WAVES_PER_SIMD = {"enc:lz77_match": 4, "dec:lz77_copy": 4, "dec:blk_scan": 4, "dec:blk_emit": 4, "enc:lz77_parse": 3,
                        "enc:lz77_cand": 4, "enc:lz77_walk": 3}
# lfx_ctx_enable_timing levels that carry only the two events around ONE phase (the timed steps: that of the dominant kernel)
PHASE_BRACKET_LEVEL = {"enc:lz77_walk": 2, "dec:blk_scan": 3, "enc:lz77_cand": 4, "dec:lz77_copy": 5}
TIMING_FINE = 6
CALIBRATION_KERNEL = "checksum_span_kernel"   # reads its input exactly once with wide coalesced loads


def measure_copy_peak(torch, dev, nbytes=N_BYTES, reps=5):
    """The HBM rate a plain device-to-device copy reaches on THIS box, now ("use the measured peak as the
    denominator too"): bytes read + bytes written per second of the best of `reps` copies of `nbytes`."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.zero_()
    best = None
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.copy_(a)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    del a, b
    return 2.0 * nbytes / best / 1e9


def dominant_phase(timing, skip=("upload", "start", "done")):
    """(phase name, ms) of the longest kernel bracket of one library call's phase list"""
    if not timing:
        return None, None
    tot = {}
    for k, v in timing["phases"]:          # (a batch call brackets the kernels of each of its rounds: same names, summed)
        if k not in skip:
            tot[k] = tot.get(k, 0.0) + v
    return max(tot.items(), key=lambda kv: kv[1]) if tot else (None, None)


def _pmc_pass(counters, n, schedule):
    """one rocprofv3 --pmc child pass of this script's bare timed loop → ({counter: {kernel: [value per dispatch]}}, error)"""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="lfx_pmc_", dir="/tmp")
    try:
        cmd = [rocprof, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", tmp, "--", sys.executable,
                                                     os.path.abspath(__file__), "--child", "--steps", "1", "--warmup", "0",
                                                     "--bytes", str(n), "--schedule", schedule]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=180)
        if r.returncode != 0:
            return None, "rocprofv3 %s pass failed: %s" % ("+".join(counters), (r.stderr or r.stdout)[-300:])
        acc = {}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    c = row.get("Counter_Name") or ""
                    if c not in counters:
                        continue
                    name = (row.get("Kernel_Name") or "").split("(")[0]
                    key = (c, name, row.get("Dispatch_Id"))
                    acc[key] = acc.get(key, 0.0) + float(row.get("Counter_Value") or 0)
        out = {c: {} for c in counters}
        for (c, name, _), v in acc.items():
            out[c].setdefault(name, []).append(v)
        return out, None
    except Exception as e:  # noqa: BLE001
        return None, "pmc pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(kernel, n, schedule):
    """HBM-side bytes per launch of `kernel` and per encode+decode step of every lfx kernel, measured now: two rocprofv3 --pmc
    child passes (reads and writes cannot share a pass, MI355X_MICROARCH.md §PMC slots) of this script's timed loop.

    Reads, round 6: the L2's memory-side request counters BY SIZE — TCC_EA0_RDREQ_{32B,64B,128B}_sum: bytes = 32 a + 64 b + 128 c,
    exact for every access pattern.  (FETCH_SIZE, which the guide's HBM section starts from, tallies this part's 128-byte requests at
    64 bytes — its expression takes them from a counter that reads 0 here — hence the guide's "double it" for wide coalesced
    readers; rounds 1-5 applied that factor, calibrated on checksum_span_kernel, to every kernel, which over-counted the
    scattered readers: parse_emit appeared to move 6.2 TB/s.)  The calibration kernel — it reads the n input bytes exactly once
    per direction — is kept as a check of the method: `calibration_check` = its measured bytes per step / 2 n.  Falls back to FETCH_SIZE x factor when
    the sized counters are not all there.  Writes: WRITE_SIZE (32 / 64-byte requests, tallied as such).  Infinity-Cache hits are
    counted, not excluded (guide): this is fabric traffic.  → dict or {"error": ...}."""
    sized = ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")
    rd, err = _pmc_pass(sized, n, schedule)
    method = "sized"
    if rd is None or not rd.get(sized[2]) or not any(sum(v) for v in rd[sized[2]].values()):
        rd, err2 = _pmc_pass(("FETCH_SIZE",), n, schedule)
        method = "fetch_size"
        if rd is None:
            return {"error": err or err2}
    wr, err = _pmc_pass(("WRITE_SIZE",), n, schedule)
    if wr is None:
        return {"error": err}

    def find(table, frag):
        for k, v in table.items():
            if frag in k:
                return v
        return None
    # bytes read per dispatch, per kernel
    read = {}
    if method == "sized":
        names = set().union(*[set(rd[c]) for c in sized])
        for k in names:
            per = [rd[c].get(k, []) for c in sized]
            nd = max(len(p) for p in per)
            read[k] = [32.0 * (per[0][i] if i < len(per[0]) else 0) + 64.0 * (per[1][i] if i < len(per[1]) else 0) +
                       128.0 * (per[2][i] if i < len(per[2]) else 0) for i in range(nd)]
        factor = None
    else:
        cal = find(rd["FETCH_SIZE"], CALIBRATION_KERNEL)
        if not cal:
            return {"error": "calibration kernel not in the counter output"}
        nsteps_f = max([len(v) for k, v in rd["FETCH_SIZE"].items() if "lz77_match" in k] or [1])
        factor = 2.0 * n * nsteps_f / (sum(cal) * 1024.0)   # ≈ 2 on gfx950 (the kernel reads n bytes per encode, in two launches, and n per decode)
        read = {k: [x * 1024.0 * factor for x in v] for k, v in rd["FETCH_SIZE"].items()}
    write = {k: [x * 1024.0 for x in v] for k, v in wr["WRITE_SIZE"].items()}
    r_k, w_k, cal_r = find(read, kernel), find(write, kernel), find(read, CALIBRATION_KERNEL)
    if r_k is None or w_k is None or not cal_r:
        return {"error": "kernel %s not in the counter output" % kernel}
    # one step = the dispatches of a kernel that runs exactly once per step
    nsteps = max([len(v) for k, v in read.items() if CALIBRATION_KERNEL + "<3>" in k or "lz77_match" in k] or [1])
    per_kernel = {}
    for table in (read, write):
        for name, v in table.items():
            if "lfx::" in name:
                key = name.replace("lfx::", "")
                per_kernel[key] = per_kernel.get(key, 0.0) + sum(v) / nsteps
    read_step = {name.replace("lfx::", ""): sum(v) / nsteps for name, v in read.items() if "lfx::" in name}
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1])[:14]
    fetch, wbytes = sum(r_k) / len(r_k), sum(w_k) / len(w_k)
    out = {"hbm_bytes": int(fetch + wbytes), "fetch_bytes": int(fetch), "write_bytes": int(wbytes),
           "method": ("reads: 32 x TCC_EA0_RDREQ_32B + 64 x _64B + 128 x _128B (exact request sizes); writes: WRITE_SIZE"
                      if method == "sized" else "reads: FETCH_SIZE x %.3f (calibrated on %s); writes: WRITE_SIZE" % (factor, CALIBRATION_KERNEL)),
           "calibration_check": round(sum(cal_r) / (2.0 * nsteps * n), 4),
           "calibration_how": ("%s reads the %d input bytes once per encode (in two launches) and the %d output bytes once per decode: "
                               "measured read bytes of a step / 2 n" % (CALIBRATION_KERNEL, n, n)),
           "step_hbm_bytes_all_kernels": int(sum(per_kernel.values())),
           "step_hbm_bytes_by_kernel": {k: int(v) for k, v in top},
           "step_read_bytes_by_kernel": {k: int(read_step.get(k, 0)) for k, _ in top},
           "source": "two rocprofv3 --pmc child passes of this run"}
    if factor is not None:
        out["fetch_size_correction"] = round(factor, 3)
    return out


_WORKER_BUF = {}


def measure_bound(kernel, n, schedule, waves_per_simd):
    """What `kernel` is bound by, measured now: two more rocprofv3 --pmc child passes (SQ counters only, no trace domain).
    SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts (MI355X_MICROARCH.md §PMC slots:
    WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ≈ WAVE_CYCLES); a SIMD that hosts `waves_per_simd` wavefronts for the whole
    kernel has SQ_WAVE_CYCLES / waves_per_simd quad-cycles.  Reported: the share of a wavefront's life it is parked at a
    wait (s_waitcnt / barrier), stalled at issue, or executing; the VALU / scalar / LDS pipes' busy share of the SIMD
    time; LDS array cycles and the bank-conflict share of them; instruction counts.  → dict ("error" on failure)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    passes = [["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
               "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
               "SQ_BUSY_CU_CYCLES"]]
    per = {}
    try:
        for names in passes:
            tmp = tempfile.mkdtemp(prefix="lfx_pmc_", dir="/tmp")
            try:
                cmd = [rocprof, "--pmc"] + names + ["--output-format", "csv", "-d", tmp, "--", sys.executable,
                                                    os.path.abspath(__file__), "--child", "--steps", "1", "--warmup", "0",
                                                    "--bytes", str(n), "--schedule", schedule]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=180)
                if r.returncode != 0:
                    return {"error": "rocprofv3 SQ pass failed: %s" % (r.stderr or r.stdout)[-300:]}
                acc, disp = {}, set()
                for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                    with open(f, newline="") as fh:
                        for row in csv.DictReader(fh):
                            if kernel not in (row.get("Kernel_Name") or ""):
                                continue
                            disp.add(row.get("Dispatch_Id"))
                            c = row.get("Counter_Name") or ""
                            acc[c] = acc.get(c, 0.0) + float(row.get("Counter_Value") or 0)
                if not disp:
                    return {"error": "kernel %s not in the counter output" % kernel}
                for c in names:
                    per[c] = acc.get(c, 0.0) / float(len(disp))
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        wc = per["SQ_WAVE_CYCLES"]
        if not wc:
            return {"error": "SQ_WAVE_CYCLES is zero"}
        simd_quads = wc / waves_per_simd
        out = {"bound": "latency / dependency (waves parked at s_waitcnt and barriers), then VALU issue",
               "wave_parked_frac": round(per["SQ_WAIT_ANY"] / wc, 4),
               "wave_issue_stall_frac": round(per["SQ_WAIT_INST_ANY"] / wc, 4),
               "wave_executing_frac": round(per["SQ_ACTIVE_INST_ANY"] / wc, 4),
               "valu_busy_frac": round(per["SQ_ACTIVE_INST_VALU"] / simd_quads, 4),
               "scalar_busy_frac": round(per["SQ_ACTIVE_INST_SCA"] / simd_quads, 4),
               "lds_inst_busy_frac": round(per["SQ_ACTIVE_INST_LDS"] / simd_quads, 4),
               "lds_issue_stall_frac": round(per["SQ_WAIT_INST_LDS"] / wc, 4),
               "lds_array_cycles": int(per["SQ_LDS_IDX_ACTIVE"]),
               "lds_bank_conflict_frac": round(per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"], 4) if per["SQ_LDS_IDX_ACTIVE"] else None,
               "simd_quad_cycles": int(simd_quads), "waves_per_simd": waves_per_simd,
               "wavefront_instructions": {c[9:].lower(): int(per[c]) for c in passes[1][:4]},
               "source": "rocprofv3 --pmc, two child passes of this run: " + " ".join(passes[0] + passes[1])}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": "bound measurement failed: %r" % (e,)}


def under_profiler():
    """True when this process runs under rocprofv3 (its tool library rides along into every child process)."""
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def oracle_warm(args):
    """(per worker process) import the oracle and synthesise the stream, outside the timed map"""
    seed, sample, _ = args
    import lfo_oracle as oracle  # noqa: F401
    import synth
    _WORKER_BUF[seed] = synth.text(sample, seed=seed).tobytes()
    return os.getpid()


def oracle_worker(args):
    """one host core: gzip encode + decode of `sample` bytes of TEXT with the oracle → (encode s, decode s)"""
    seed, sample, write = args
    import lfo_oracle as oracle
    import synth
    buf = _WORKER_BUF.get(seed) or synth.text(sample, seed=seed).tobytes()
    t0 = time.perf_counter()
    enc = oracle.encode(oracle.GZIP, buf, write_size=write)
    t1 = time.perf_counter()
    rc, out, _, _ = oracle.decode(oracle.GZIP, enc)
    t2 = time.perf_counter()
    assert rc == 0 and out == buf
    return t1 - t0, t2 - t1


def sub_roofline(prefix, dom, algo_bytes):
    """roofline record of a sub-configuration's dominant kernel: the longest HIP-event bracket of the call (one kernel, or
    one kernel per round of a batch), its algorithmic bytes = the call's (input read + output written)"""
    name, ms = dom
    if not name or not ms:
        return None
    ach = algo_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": prefix + name, "kernel_name": PHASE_KERNEL.get(prefix + name), "achieved": round(ach, 2),
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": None,
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes": algo_bytes}


def sub_pcie(ctx, _ffi, data, write, reps=2):
    """PCIe-inclusive figures (never `value`): lfx_encode_host / lfx_decode_host on HOST buffers — H2D, the device path, D2H —
    called on raw pointers (no Python-side copy inside the timed region).  Two kinds of caller memory: pageable (numpy
    arrays: staged through page-locked slabs by the library's copy threads, lfx_hostio.h) and page-locked (lfx_host_alloc:
    plain DMA).  Best of `reps` calls each after a warm-up call; wall clock around the blocking calls."""
    import numpy as np
    import ctypes as C
    L = _ffi.lib()
    opts, sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(write)
    n = int(data.size)
    bound = L.lfx_encode_bound(n, C.byref(opts), C.byref(sched))

    def run(in_ptr, enc_ptr, dec_ptr):
        best = None
        m = 0
        ctx.encode_host_ptr(_ffi.GZIP, in_ptr, min(n, 8 << 20), enc_ptr, bound, opts, sched)      # (warm-up: slabs, scratch)
        for _ in range(reps):
            t0 = time.perf_counter()
            m = ctx.encode_host_ptr(_ffi.GZIP, in_ptr, n, enc_ptr, bound, opts, sched)
            t1 = time.perf_counter()
            rc, ol, used, _msg = ctx.decode_host_ptr(_ffi.GZIP, enc_ptr, m, dec_ptr, n)
            t2 = time.perf_counter()
            if rc or ol != n or used != m:
                raise RuntimeError("host round trip failed: rc=%d out=%d used=%d" % (rc, ol, used))
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        return best, m

    # pageable
    enc = np.empty(bound, dtype=np.uint8)
    dec = np.empty(n, dtype=np.uint8)
    (t, te, td), m = run(data.ctypes.data, enc.ctypes.data, dec.ctypes.data)
    ok = bool((dec == data).all())
    rec = {"workload": "lfx_encode_host + lfx_decode_host, %d MiB in, %d B compressed, caller's buffers PAGEABLE (numpy): H2D + "
                       "device path + D2H, staged through page-locked slabs by %d copy threads" % (n >> 20, m, 4),
           "value": round(n / t / 1e9, 3), "unit": "GB/s", "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
           "round_trip_ok": ok}
    # page-locked
    pin = [L.lfx_host_alloc(n), L.lfx_host_alloc(bound), L.lfx_host_alloc(n)]
    try:
        if all(pin):
            C.memmove(pin[0], data.ctypes.data, n)
            (t, te, td), m2 = run(pin[0], pin[1], pin[2])
            back = np.ctypeslib.as_array((C.c_uint8 * n).from_address(pin[2]))
            rec["pinned"] = {"workload": "the same calls on page-locked buffers (lfx_host_alloc): plain DMA, no staging copy",
                             "value": round(n / t / 1e9, 3), "unit": "GB/s", "encode_ms": round(te * 1e3, 2),
                             "decode_ms": round(td * 1e3, 2), "round_trip_ok": bool(m2 == m and (back == data).all())}
    finally:
        for p in pin:
            if p:
                L.lfx_host_free(p)
    return rec


def sub_stream_api(ctx, _ffi, data, enc_want=None, chunk=8192):
    """The drop-in surface itself: the reference's io::copy protocol — lfx_encoder_write in 8192-byte
    calls, then lfx_decoder_read with 8192-byte reads (examples/flate.rs:52,96-97; flate_bench/src/main.rs:86-100) — on the
    whole buffer, driven from C (tools/stream_copy.c: 2 x 32768 calls through Python callbacks would time the interpreter).
    Host memory in, host memory out; the bytes must equal the one-shot call's (= the oracle's)."""
    import numpy as np
    import stream_copy
    n = int(data.size)
    opts = _ffi.make_opts(mtime=0)
    enc = np.empty(n + n // 4 + 4096, dtype=np.uint8)
    dec = np.empty(n, dtype=np.uint8)
    enc[:] = 0          # (np.empty's pages are not there yet: first touches would be timed as page faults)
    dec[:] = 0
    # warm-up of both directions with the whole buffer: the page-locked buffers of the context's pool grow to what the later
    # 32 MiB windows need (page-locking ~170 MiB costs more than the run), device scratch
    rc, m0, _ = stream_copy.encode(ctx, _ffi.GZIP, opts, data.ctypes.data, n, chunk, enc.ctypes.data, enc.size)
    stream_copy.decode(ctx, _ffi.GZIP, enc.ctypes.data, m0, chunk, dec.ctypes.data, n)
    rc, m, te = stream_copy.encode(ctx, _ffi.GZIP, opts, data.ctypes.data, n, chunk, enc.ctypes.data, enc.size)
    if rc:
        raise RuntimeError("stream encode failed: %d" % rc)
    enc_gpu_s, enc_gpu_calls = stream_copy.last_split()
    rc, ol, td = stream_copy.decode(ctx, _ffi.GZIP, enc.ctypes.data, m, chunk, dec.ctypes.data, n)
    if rc:
        raise RuntimeError("stream decode failed: %d" % rc)
    dec_gpu_s, dec_gpu_calls = stream_copy.last_split()
    scratch = np.empty(n, dtype=np.uint8)
    t_copy = min(stream_copy.memcpy_seconds(scratch.ctypes.data, data.ctypes.data, n, chunk) for _ in range(2))
    same = None if enc_want is None else bool(m == len(enc_want) and enc[:m].tobytes() == enc_want)
    return {"workload": "io::copy protocol on the stream ABI: lfx_encoder_write in %d-byte calls + lfx_decoder_read with %d-byte "
                        "reads over %d MiB of TEXT (gzip, default options), host memory, C driver tools/stream_copy.c" % (chunk, chunk, n >> 20),
            "value": round(n / (te + td) / 1e9, 3), "unit": "GB/s", "encode_GBps": round(n / te / 1e9, 3),
            "decode_GBps": round(n / td / 1e9, 3), "encode_ms": round(te * 1e3, 2), "decode_ms": round(td * 1e3, 2),
            "compressed_bytes": int(m), "round_trip_ok": bool(ol == n and (dec == data).all()),
            "equals_one_shot_output": same,
            # where the time goes: the calls that wait for GPU work (a write() that closes a batch and collects the one before,
            # finish(); a read() that decodes the next window) against the protocol's own copying
            "encode_ms_in_batch_calls": round(enc_gpu_s * 1e3, 2), "encode_batch_calls": enc_gpu_calls,
            "decode_ms_in_window_calls": round(dec_gpu_s * 1e3, 2), "decode_window_calls": dec_gpu_calls,
            "host_memcpy_ms": round(t_copy * 1e3, 2),
            "host_memcpy_how": "one plain pass over the %d MiB in %d-byte memcpy calls on this host, one thread: what each of "
                               "the protocol's four copies (into the encoder, sink, cursor, out of the decoder) costs at best" % (n >> 20, chunk)}


def sub_raw_deflate(ctx, torch, _ffi, C, d_in, n, write, reps=3):
    """The reference's only published protocol (flate_bench/src/main.rs:33-62: raw DEFLATE, no container, no checksum;
    README.md:60-67) next to its context figures: deflate::Encoder + deflate::Decoder on the resident buffer."""
    L = _ffi.lib()
    opts, sched = _ffi.make_opts(), _ffi.make_schedule(write)
    bound = L.lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device=d_in.device)
    d_dec = torch.empty(n, dtype=torch.uint8, device=d_in.device)
    te = td = None
    m = 0
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = ctx.encode_device(_ffi.DEFLATE, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
        t1 = time.perf_counter()
        rc, ol, used, _msg = ctx.decode_device(_ffi.DEFLATE, d_out.data_ptr(), m, d_dec.data_ptr(), n)
        t2 = time.perf_counter()
        if rc or ol != n:
            raise RuntimeError("raw deflate round trip failed rc=%d" % rc)
        te = t1 - t0 if te is None else min(te, t1 - t0)
        td = t2 - t1 if td is None else min(td, t2 - t1)
    return {"workload": "raw DEFLATE (deflate::Encoder / deflate::Decoder, no container, no checksum) on TEXT(%d MiB), %s, resident "
                        "in HBM — the protocol of flate_bench/src/main.rs:33-62" % (n >> 20, "8192-byte writes" if write else "one write_all"),
            "encode_MBps": round(n / te / 1e6, 1), "decode_MBps_of_output": round(n / td / 1e6, 1), "compressed_bytes": int(m),
            "ratio": round(m / n, 4), "round_trip_ok": bool(torch.equal(d_dec, d_in)),
            "reference_context": "README.md:60-67 (unknown hardware, 1 thread, enwiki titles): libflate encode 34.1 MB/s, decode "
                                 "204.7 MB/s of output — quoted, not re-measured"}


def sub_cfg3(ctx, torch, synth, _ffi, C, dev, reps=3):
    """BASELINE cfg3 as a sub-record: 4096 independent 64 KiB zlib streams decoded by ONE lfx_decode_batch_device call; wall
    clock around the blocking call, streams and outputs resident in HBM; GB/s of output bytes.  As SURVEY §8d asks, half of
    the streams are reference-format (one write_all each — made by ONE lfx_encode_batch_device call, whose rate is reported
    too; a sample is compared with the oracle's bytes), half are python-zlib's (level 6: blocks that read earlier blocks)."""
    import zlib
    import numpy as np
    L = _ffi.lib()
    count, size = 4096, 65536
    half = count // 2
    big = synth.text(count * size, seed=synth.SEED_BASE + 3)
    d_plain = torch.from_numpy(big).to(dev)
    opts, sched = _ffi.make_opts(), _ffi.make_schedule(0)
    bound = L.lfx_encode_bound(size, C.byref(opts), C.byref(sched)) & ~3
    d_streams = torch.zeros(count * bound, dtype=torch.uint8, device=dev)
    in_len = np.zeros(count, dtype=np.uint64)
    # ---- the reference-format half: one batch encode call (timed: cfg3's encode side)
    e_in_off = (np.arange(half, dtype=np.uint64) * np.uint64(size))
    e_in_len = np.full(half, size, dtype=np.uint64)
    e_out_off = (np.arange(half, dtype=np.uint64) * np.uint64(bound))
    e_out_cap = np.full(half, bound, dtype=np.uint64)
    e_out_len = np.zeros(half, dtype=np.uint64)
    e_status = np.zeros(half, dtype=np.int32)
    enc_best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.lfx_encode_batch_device(ctx.handle, _ffi.ZLIB, C.byref(opts), C.byref(sched), half, d_plain.data_ptr(), e_in_off.ctypes.data,
                                       e_in_len.ctypes.data, d_streams.data_ptr(), e_out_off.ctypes.data, e_out_cap.ctypes.data,
                                       e_out_len.ctypes.data, e_status.ctypes.data)
        dt = time.perf_counter() - t0
        enc_best = dt if enc_best is None else min(enc_best, dt)
        if rc or e_status.any():
            raise RuntimeError("batch encode failed: %d %s" % (rc, ctx.last_error()))
    in_len[:half] = e_out_len
    sample_ok = True
    try:
        import lfo_oracle as oracle
        host_streams = d_streams[:half * bound].cpu().numpy()
        for i in (0, 1, half // 2, half - 1):
            want = oracle.encode(oracle.ZLIB, big[i * size:(i + 1) * size].tobytes(), write_size=0)
            sample_ok &= host_streams[i * bound:i * bound + int(e_out_len[i])].tobytes() == want
    except Exception:   # (no oracle: the streams are still checked by the round trip below)
        sample_ok = None
    # ---- the python-zlib half
    host = np.zeros((count - half) * bound, dtype=np.uint8)
    for k in range(count - half):
        i = half + k
        z = zlib.compress(big[i * size:(i + 1) * size].tobytes(), 6)
        host[k * bound:k * bound + len(z)] = np.frombuffer(z, dtype=np.uint8)
        in_len[i] = len(z)
    d_streams[half * bound:] = torch.from_numpy(host).to(dev)
    in_off = (np.arange(count, dtype=np.uint64) * np.uint64(bound))
    out_off = (np.arange(count, dtype=np.uint64) * np.uint64(size))
    out_cap = np.full(count, size, dtype=np.uint64)
    out_len = np.zeros(count, dtype=np.uint64)
    status = np.zeros(count, dtype=np.int32)
    d_out = torch.zeros(count * size, dtype=torch.uint8, device=dev)
    best = None
    dom = (None, None)
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.lfx_decode_batch_device(ctx.handle, _ffi.ZLIB, count, d_streams.data_ptr(), in_off.ctypes.data, in_len.ctypes.data,
                                       d_out.data_ptr(), out_off.ctypes.data, out_cap.ctypes.data, out_len.ctypes.data,
                                       status.ctypes.data)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, dom = dt, dominant_phase(ctx.last_timing())
    ok = rc == 0 and not status.any() and bool((out_len == size).all()) and torch.equal(d_out, d_plain)
    comp = int(in_len.sum())
    return {"roofline": sub_roofline("dec:", dom, comp + count * size),
            "workload": "cfg3: %d independent %d KiB zlib streams (half reference-format, half python-zlib level 6), one "
                        "lfx_decode_batch_device call" % (count, size >> 10),
            "value": round(count * size / best / 1e9, 3), "unit": "GB/s of output", "ms": round(best * 1e3, 3),
            "compressed_bytes": comp, "round_trip_ok": ok,
            "hbm_frac_algorithmic": round((comp + count * size) / best / 1e9 / HBM_PEAK_GBPS, 5),
            "batch_encode": {"workload": "%d x %d KiB through ONE lfx_encode_batch_device call" % (half, size >> 10),
                             "value": round(half * size / enc_best / 1e9, 3), "unit": "GB/s of input", "ms": round(enc_best * 1e3, 3),
                             "sample_equals_oracle": sample_ok}}


def sub_cfg5(ctx, torch, synth, _ffi, C, dev, reps=2):
    """BASELINE cfg5 as a sub-record: zlib encode of 1 GiB of low-entropy binary (runs and fixed records: every other
    match is 258 long), 8192-byte writes; then one decode of the result.  GB/s of input bytes."""
    L = _ffi.lib()
    n = 1 << 30
    data = synth.lowent(n, seed=synth.SEED_BASE + 5)
    d_in = torch.from_numpy(data).to(dev)
    opts, sched = _ffi.make_opts(), _ffi.make_schedule(WRITE)
    bound = L.lfx_encode_bound(n, C.byref(opts), C.byref(sched)) & ~3
    d_out = torch.empty(bound, dtype=torch.uint8, device=dev)
    best = None
    m = 0
    dom = (None, None)
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = ctx.encode_device(_ffi.ZLIB, d_in.data_ptr(), n, d_out.data_ptr(), bound, opts, sched)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, dom = dt, dominant_phase(ctx.last_timing())
    d_dec = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc, ol, used, msg = ctx.decode_device(_ffi.ZLIB, d_out.data_ptr(), m, d_dec.data_ptr(), n)
    tdec = time.perf_counter() - t0
    dom_dec = dominant_phase(ctx.last_timing())
    ok = rc == 0 and ol == n and used == m and torch.equal(d_dec, d_in)
    return {"roofline": sub_roofline("enc:", dom, n + m), "decode_roofline": sub_roofline("dec:", dom_dec, n + m),
            "workload": "cfg5: zlib::Encoder on LOWENT(1 GiB), 8192-byte writes", "value": round(n / best / 1e9, 3),
            "unit": "GB/s of input", "ms": round(best * 1e3, 3), "compressed_bytes": int(m), "decode_ms": round(tdec * 1e3, 3),
            "round_trip_ok": ok, "hbm_frac_algorithmic": round((n + m) / best / 1e9 / HBM_PEAK_GBPS, 5)}


def strong_share(total_bytes, world):
    """--scaling strong: a rank's share of the job — whole blocks, since a shard boundary has to be a block boundary
    (deflate/encode.rs:12, DEFAULT_BLOCK_SIZE = 1 MiB; 8192-byte writes divide it)."""
    return max(total_bytes // world // (1 << 20) * (1 << 20), 1 << 20)


def launch_plan(gpus, env, device_count, argv):
    """Decide how `bench.py --gpus N` runs.  → ("inline", None) when this process IS the job (N = 1, or a launcher
    already set WORLD_SIZE: the driver's `python -m torch.distributed.run ... bench.py --gpus N`), ("spawn", cmd) when
    --gpus N > 1 was given to a bare `python bench.py`: the script re-executes itself under torch.distributed.run with one
    rank per GPU (RCCL), so that `python bench.py --gpus 8` measures eight GPUs and not one.  Raises when the box has
    fewer than N devices, unless LFX_BENCH_ONE_GPU=1 (every rank on GPU 0 over gloo: the self-test on a one-GPU box)."""
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if gpus != world:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
        return "inline", None
    if gpus <= 1:
        return "inline", None
    if device_count < gpus and env.get("LFX_BENCH_ONE_GPU") != "1":
        raise SystemExit("bench.py: --gpus %d but only %d device(s) visible (LFX_BENCH_ONE_GPU=1 runs all ranks on "
                         "GPU 0 over gloo as a self-test)" % (gpus, device_count))
    port = env.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node.  Without a launcher (no WORLD_SIZE in the environment) N > 1 re-executes "
                         "this script under torch.distributed.run with one rank per GPU (RCCL)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=N_BYTES,
                    help="uncompressed bytes PER RANK with --scaling weak (the default), of the WHOLE JOB with --scaling strong.  "
                         "Default 256 MiB = the metric's configuration; the cfg4 shape (8-way sharded gzip encode of 8 GiB) is "
                         "`--gpus 8 --bytes 1073741824` (and rides along as a sub-record of every --gpus 8 run)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank holds --bytes (256 MiB per GPU: the driver's scaling curve).  strong: --bytes is the "
                         "whole job, split evenly over the ranks (BASELINE.json's metric read literally: 256 MiB at 1/2/4/8 GPUs; "
                         "at 32 MiB per GPU the per-call fixed costs show)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes")
    ap.add_argument("--no-s1", action="store_true", help="skip the S1 sub-record")
    ap.add_argument("--no-subs", action="store_true", help="skip the cfg3 / cfg5 sub-records")
    ap.add_argument("--child", action="store_true", help="(internal) bare timed loop for a profiler pass")
    ap.add_argument("--schedule", choices=["S8K", "S1"], default="S8K",
                    help="write schedule of the encoder: S8K = 8192-byte writes (the metric's configuration), "
                         "S1 = one write_all (one LZ77 chunk, one block: SURVEY cfg2's second schedule)")
    args = ap.parse_args()

    import numpy as np  # noqa: F401
    import torch
    mode, cmd = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), sys.argv[1:])
    if mode == "spawn":
        # one rank per GPU; the ranks' stdout is ours, rank 0 prints the JSON line last
        sys.stdout.flush()
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)
    import __graft_entry__ as g
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # self-test of the N>1 path on a one-GPU box: every rank on GPU 0, the exchange over gloo
    one_gpu_test = os.environ.get("LFX_BENCH_ONE_GPU") == "1"
    # take the sharded path (all-gather + concatenation) even with one rank: exercises RCCL on a one-GPU box
    force_sharded = os.environ.get("LFX_BENCH_FORCE_SHARDED") == "1"
    if one_gpu_test:
        local = 0
    if rank == 0:
        g.build()
    dist = None
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        if one_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        dist.barrier()
    sharded_path = dist is not None
    # the decode's two small all-gathers get a communicator of their own: on the default one they would queue behind the
    # member's concatenation (7 x 130 MB into rank 0), which is still in flight when the decode starts
    small_group = dist.new_group() if sharded_path and world > 1 else None
    import libflate_amd
    from libflate_amd import _ffi, sharded
    import synth
    import ctypes as C

    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    ctx = libflate_amd.Context(local)
    ctx.enable_timing(TIMING_FINE)
    n = args.bytes if args.scaling == "weak" else strong_share(args.bytes, world)
    data = synth.text(n, seed=synth.SEED_BASE + 2 + rank)
    d_in = torch.from_numpy(data).to(dev)
    L = _ffi.lib()

    class Run:
        """buffers + one step of the round trip for a write schedule"""

        def __init__(self, schedule, n=n, d_in=d_in):
            self.n, self.d_in = n, d_in
            self.schedule = schedule
            self.write = WRITE if schedule == "S8K" else 0       # 0 = one write_all
            self.opts, self.sched = _ffi.make_opts(mtime=0), _ffi.make_schedule(self.write)
            self.bound = L.lfx_encode_bound(self.n, C.byref(self.opts), C.byref(self.sched)) & ~3
            self.d_out = torch.empty(self.bound, dtype=torch.uint8, device=dev)
            self.d_dec = torch.empty(self.n, dtype=torch.uint8, device=dev)
            self.hdr_len = L.lfx_container_header_len(_ffi.GZIP, C.byref(self.opts))
            self.phase_acc = {}
            self.member_len = 0
            self.finish_s = []          # sharded path: seconds spent in lfx_sharded_encode_finish per timed step
            if sharded_path and rank == 0:
                self.d_member = torch.empty(self.bound * world, dtype=torch.uint8, device=dev)
                # (the shards of the other ranks side by side: they are received concurrently)
                self.staging = torch.empty((self.bound + 256) * max(world - 1, 1), dtype=torch.uint8, device=dev)
            else:
                self.d_member = self.staging = None

        def acc_timing(self, prefix):
            t = ctx.last_timing()
            if t:
                for name, ms in t["phases"]:
                    self.phase_acc.setdefault(prefix + name, []).append(ms)

        def check(self, rc, what):
            if rc:
                raise RuntimeError("%s failed: %d %s" % (what, rc, ctx.last_error()))

        def step(self, record=False):
            """→ (t_enc, t_dec, compressed bytes of this rank)"""
            n, d_in = self.n, self.d_in
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if not sharded_path:
                m = ctx.encode_device(_ffi.GZIP, d_in.data_ptr(), n, self.d_out.data_ptr(), self.bound, self.opts, self.sched)
                t1 = time.perf_counter()
                if record:
                    self.acc_timing("enc:")
                rc, ol, used, msg = ctx.decode_device(_ffi.GZIP, self.d_out.data_ptr(), m, self.d_dec.data_ptr(), n)
                if rc or ol != n or used != m:
                    raise RuntimeError("decode failed rc=%d out=%d used=%d %s" % (rc, ol, used, msg))
                t2 = time.perf_counter()
                if record:
                    self.acc_timing("dec:")
                return t1 - t0, t2 - t1, m
            # ---- the library's own N-GPU driver (lfx_sharded_encode_begin: prepare → 32-byte all-gather → emit at the rank's
            #      bit offset → the shards are posted AND started — lfx_comm.start, one batch_isend_irecv — before it returns);
            #      the collectives are this script's torch.distributed calls behind an lfx_comm (libflate_amd/sharded.py: RCCL
            #      over xGMI, or gloo in the one-GPU self-test).  The time lfx_sharded_encode_finish then still needs is reported
            #      as `overlap_ms`: ≈ 0 when the gather hid behind the decode
            gh, part = sharded.encode_begin(ctx, rank, world, _ffi.GZIP, self.opts, self.sched, d_in, n, self.d_out, self.bound,
                                            self.d_member, self.bound * world if rank == 0 else 0, self.staging,
                                            dist if world > 1 else None)
            m = C.c_uint64(part.part_len)
            combined, total_n = part.check, part.total_n
            if record:
                self.acc_timing("enc:")
            t1 = time.perf_counter()
            # ---- N-GPU decode of the ONE member, no bit offset from the encoder (lfx_sharded_decode): every rank holds a byte
            #      range of the member (here: the bytes it emitted — the cut a consumer reading the member in parallel would also
            #      know), finds and scans the blocks that start in it, the ranks all-gather their candidate tuples (56 bytes each
            #      over RCCL), walk the same chain and materialise their own blocks; the transfers of the concatenation run
            #      meanwhile.  Arbitrary byte cuts: tests/test_gpu_round3.py::test_member_decode_on_virtual_ranks
            lo = part.start_bit // 8 if rank else 0
            hi = part.end_bit // 8 if rank + 1 < world else lo + m.value
            ol, base, total_out, crc_all, _ad = sharded.decode_member_ranks(ctx, rank, world, self.d_out, m.value, lo, hi,
                                                                            8 * self.hdr_len, self.d_dec, n, dist if world > 1 else None,
                                                                            small_group, member_len=part.member_len)
            if ol != n or base != rank * n or total_out != total_n or crc_all != combined:
                raise RuntimeError("member decode: slice %d bytes at %d of %d, crc %08x vs %08x" % (ol, base, total_out, crc_all, combined))
            t2 = time.perf_counter()
            if record:
                self.acc_timing("dec:")
            self.member_len = sharded.encode_finish(gh)      # (the member is complete on rank 0 when the step ends)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if record:
                self.finish_s.append(t3 - t2)
            return (t1 - t0) + (t3 - t2), t2 - t1, m.value

        def gather_alone(self):
            """sharded path, outside the timed region: begin → finish with NOTHING in between — what the concatenation costs
            when it cannot hide behind anything; `overlap_ms` (finish's wait behind the decode, in the timed steps) is read
            against it"""
            torch.cuda.synchronize()
            gh, _part = sharded.encode_begin(ctx, rank, world, _ffi.GZIP, self.opts, self.sched, self.d_in, self.n, self.d_out, self.bound,
                                             self.d_member, self.bound * world if rank == 0 else 0, self.staging,
                                             dist if world > 1 else None)
            t0 = time.perf_counter()
            sharded.encode_finish(gh)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        def timed(self, steps, warmup, record=True):
            _, _, m = self.step()
            if not torch.equal(self.d_dec, self.d_in):
                raise RuntimeError("round trip mismatch")
            self.step_s = []
            for _ in range(warmup):
                self.step()
            # which phase is the longest kernel's?  (one fully instrumented step; its events go nowhere else)
            self.bracket_level, self.bracket_phase = 2, "enc:lz77_walk"
            if record:
                keep, self.phase_acc = self.phase_acc, {}
                ctx.enable_timing(TIMING_FINE)
                self.step(record=True)
                one = {k: v[-1] for k, v in self.phase_acc.items() if k in PHASE_BRACKET_LEVEL}
                self.phase_acc = keep
                if one:
                    self.bracket_phase = max(one, key=one.get)
                    self.bracket_level = PHASE_BRACKET_LEVEL[self.bracket_phase]
            self.gather_alone_s = None
            if sharded_path and record:
                ga = min(self.gather_alone() for _ in range(2))
                tt = torch.tensor([ga], dtype=torch.float64, device="cpu" if one_gpu_test else dev)
                if dist and world > 1:
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                self.gather_alone_s = float(tt.cpu()[0])
            if dist:
                dist.barrier()
            torch.cuda.synchronize()
            # The timed steps carry TWO events per step (timing levels 2..5: around the phase of the longest kernel, picked above)
            # — an event record between two kernels is ~6 us of idle GPU (kernel trace), and the two dozen of a fully
            # instrumented step were 3 % of it.  Every phase is measured in extra, untimed steps behind the loop.
            ctx.enable_timing(self.bracket_level if record else False)
            t_start = time.perf_counter()
            enc_t = dec_t = 0.0
            for _ in range(steps):
                a, b, m = self.step(record=record)
                enc_t += a
                dec_t += b
                self.step_s.append(a + b)
            torch.cuda.synchronize()
            if dist:
                dist.barrier()
            elapsed = time.perf_counter() - t_start
            self.phase_timed = {k: v for k, v in self.phase_acc.items() if k == self.bracket_phase}   # from the timed steps
            self.phase_acc = {}
            ctx.enable_timing(TIMING_FINE)
            if record:
                keep = self.finish_s
                for _ in range(max(2, min(5, steps))):
                    self.step(record=True)
                self.finish_s = keep[:steps]
            if dist:
                fin = sum(self.finish_s) / len(self.finish_s) if self.finish_s else 0.0
                tt = torch.tensor([elapsed, enc_t, dec_t, fin], dtype=torch.float64, device="cpu" if one_gpu_test else dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elapsed, enc_t, dec_t, self.finish_max_s = (float(x) for x in tt.cpu())
            return elapsed, enc_t, dec_t, m

    run = Run(args.schedule)
    if args.child:                       # profiler pass: the timed loop and nothing else
        run.timed(args.steps, args.warmup, record=False)
        return

    elapsed, enc_t, dec_t, m = run.timed(args.steps, args.warmup)
    comp = None
    if not sharded_path:
        import zlib
        comp = run.d_out[:m].cpu().numpy().tobytes()
        assert int.from_bytes(comp[-8:-4], "little") == zlib.crc32(data.tobytes()), "CRC-32 trailer mismatch"
    elif rank == 0 and world * n <= (1 << 30):
        import zlib
        member = run.d_member[:run.member_len].cpu().numpy().tobytes()
        whole = b"".join(synth.text(n, seed=synth.SEED_BASE + 2 + r).tobytes() for r in range(world))
        assert zlib.decompress(member, 31) == whole, "the concatenated member does not inflate to the ranks' input"
    # ---- BASELINE cfg4 (8-way sharded gzip encode of 8 GiB, CRC-32 combine over RCCL) as a sub-record of every 8-rank run:
    #      1 GiB per rank through the same sharded step (all ranks take part; LFX_BENCH_CFG4_BYTES sizes the self-test)
    cfg4 = None
    n4 = int(os.environ.get("LFX_BENCH_CFG4_BYTES", str(1 << 30)))
    if sharded_path and (world == 8 or "LFX_BENCH_CFG4_BYTES" in os.environ) and not args.no_subs and n4 != n:
        try:
            d_in4 = torch.from_numpy(synth.text(n4, seed=synth.SEED_BASE + 40 + rank)).to(dev)
            r4 = Run(args.schedule, n4, d_in4)
            k4 = 2
            e4, en4, de4, m4 = r4.timed(k4, 1, record=False)
            cfg4 = {"workload": "cfg4: %d-way sharded gzip encode (+ N-GPU decode) of %d MiB of TEXT, %d MiB per rank, one member, "
                                "CRC-32 combine and concatenation over %s" % (world, (n4 * world) >> 20, n4 >> 20,
                                                                              "gloo (one-GPU self-test)" if one_gpu_test else "RCCL / xGMI"),
                    "value": round(n4 * world / (e4 / k4) / 1e9, 4), "unit": "GB/s", "ms_per_step": round(e4 / k4 * 1e3, 3),
                    "encode_GBps": round(n4 * world * k4 / en4 / 1e9, 4), "decode_GBps": round(n4 * world * k4 / de4 / 1e9, 4),
                    "steps": k4, "member_bytes": int(r4.member_len) if rank == 0 else None}
            del r4, d_in4
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001  (a sub-record must not take the metric down — but every rank must agree)
            cfg4 = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_bytes = n * world
    steps_sorted = sorted(run.step_s)
    median_s = steps_sorted[len(steps_sorted) // 2] if steps_sorted else None     # (this rank's encode + decode calls of one step)
    peak_measured = None
    try:
        peak_measured = measure_copy_peak(torch, dev, min(n, N_BYTES))
    except Exception:   # noqa: BLE001
        pass
    value = total_bytes / (elapsed / args.steps) / 1e9
    # ---- roofline of the dominant kernel phase (HIP events on the context's stream, inside the timed region)
    avg = {k: sum(v) / len(v) for k, v in run.phase_acc.items()}        # (the extra, fully instrumented steps)
    kernel_phases = {k: v for k, v in avg.items() if k.split(":")[1] not in ("upload", "start", "done")}
    dom = max(kernel_phases, key=kernel_phases.get) if kernel_phases else None
    timed_avg = {k: sum(v) / len(v) for k, v in getattr(run, "phase_timed", {}).items() if v}
    dom_from_timed = dom in timed_avg
    if dom_from_timed:
        avg[dom] = timed_avg[dom]                 # HIP events over the timed region itself
    algo_bytes = n + m     # encode N read + C written; decode C read + N written
    roof = None
    step_traffic = None
    if dom:
        ach = algo_bytes / (avg[dom] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "kernel_name": PHASE_KERNEL.get(dom), "achieved": round(ach, 2),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": None,
                "peak_measured": round(peak_measured, 1) if peak_measured else None,
                "frac_measured": round(ach / peak_measured, 5) if peak_measured else None,
                "peak_measured_how": "device-to-device copy of %d MiB in this run: bytes read + written per second" % (min(n, N_BYTES) >> 20),
                "avg_launch_ms": round(avg[dom], 4), "algorithmic_bytes": algo_bytes,
                "avg_launch_how": ("HIP events around this phase in the %d timed steps" % args.steps if dom_from_timed else
                                   "HIP events in extra, fully instrumented steps behind the timed loop (the timed steps only carry "
                                   "the events of " + str(getattr(run, "bracket_phase", None)) + ")")}
        if under_profiler():
            roof["traffic_error"] = "not measured: this run is itself under a profiler"
        elif world == 1 and not args.no_traffic and dom in PHASE_KERNEL:
            t = measure_traffic(PHASE_KERNEL[dom], n, args.schedule)
            if t and "hbm_bytes" in t:
                roof["traffic"] = t["hbm_bytes"]
                roof["traffic_detail"] = t
                step_traffic = t.get("step_hbm_bytes_all_kernels")
            else:
                roof["traffic_error"] = (t or {}).get("error", "unknown")
            # what the dominant kernel is actually bound by: the HBM figures above are the contract's
            if dom in PHASE_WAVES_PER_SIMD:
                roof["bound_detail"] = measure_bound(PHASE_KERNEL[dom], n, args.schedule, waves_per_simd=PHASE_WAVES_PER_SIMD[dom])
    whole_ach = 2.0 * algo_bytes * world / (elapsed / args.steps) / 1e9
    whole = {"achieved": round(whole_ach, 2), "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
             "frac": round(whole_ach / (HBM_PEAK_GBPS * world), 5), "algorithmic_bytes_per_step": 2 * algo_bytes * world,
             "frac_measured": round(whole_ach / (peak_measured * world), 5) if peak_measured else None,
             "traffic": step_traffic,
             "traffic_over_algorithmic": round(step_traffic / (2.0 * algo_bytes), 2) if step_traffic else None}
    # ---- the other write schedule of SURVEY cfg2 in the same run (fewer steps: it is a sub-record, not the metric)
    s1 = None
    if world == 1 and not sharded_path and not args.no_s1:
        other = "S1" if args.schedule == "S8K" else "S8K"
        r2 = Run(other)
        e2, en2, de2, m2 = r2.timed(max(1, min(3, args.steps)), 1, record=False)
        k2 = max(1, min(3, args.steps))
        s1 = {"schedule": other, "value": round(n / (e2 / k2) / 1e9, 4), "unit": "GB/s", "ms_per_step": round(e2 / k2 * 1e3, 3),
              "encode_GBps": round(n * k2 / en2 / 1e9, 4), "decode_GBps": round(n * k2 / de2 / 1e9, 4), "compressed_bytes": m2,
              "steps": k2}
        del r2
    # ---- the other single-GPU configurations of BASELINE.json, as sub-records of the same line
    subs = None
    if world == 1 and not sharded_path and not args.no_subs and n == N_BYTES:
        subs = {}
        for name, fn in (("cfg3_batch_decode", sub_cfg3), ("cfg5_lowent_encode", sub_cfg5)):
            try:
                subs[name] = fn(ctx, torch, synth, _ffi, C, dev)
            except Exception as e:      # noqa: BLE001  (a sub-record must not take the metric down)
                subs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
    pcie = stream_api = raw_deflate = None
    if world == 1 and not sharded_path and not args.no_subs:
        for name, fn in (("pcie", lambda: sub_pcie(ctx, _ffi, data, run.write)),
                         ("stream_api", lambda: sub_stream_api(ctx, _ffi, data, comp if args.schedule == "S8K" else None)),
                         ("raw_deflate", lambda: sub_raw_deflate(ctx, torch, _ffi, C, d_in, n, run.write))):
            try:
                rec = fn()
            except Exception as e:      # noqa: BLE001  (a sub-record must not take the metric down)
                rec = {"error": "%s: %s" % (type(e).__name__, e)}
            if name == "pcie":
                pcie = rec
            elif name == "stream_api":
                stream_api = rec
            else:
                raw_deflate = rec
    cpu = None
    if not args.no_cpu_baseline and world == 1 and not sharded_path:
        import multiprocessing as mp
        import lfo_oracle as oracle
        write = run.write
        buf = data.tobytes()
        t0 = time.perf_counter()
        enc = oracle.encode(oracle.GZIP, buf, write_size=write)
        t1 = time.perf_counter()
        rc, out, _, _ = oracle.decode(oracle.GZIP, enc)
        t2 = time.perf_counter()
        assert rc == 0 and out == buf
        assert enc == comp, "GPU output differs from the oracle on the benchmarked buffer"     # bit-exactness gate
        host_cores = os.cpu_count() or 1
        cores = usable_cores()
        sample = 16 << 20
        jobs = [(synth.SEED_BASE + 100, sample, write)] * cores      # (same seed: whichever worker takes a job has it warm)
        ncore = None
        # (a profiler that preloads itself into every child process makes a pool of spawned workers crawl or hang:
        #  the N-core figure is then left out instead of stalling the run; every wait is bounded)
        if not under_profiler():
            pool = mp.get_context("spawn").Pool(cores)
            try:
                pool.map_async(oracle_warm, jobs, chunksize=1).get(timeout=120)   # spawn, import, synthesise: not timed
                tw0 = time.perf_counter()
                pool.map_async(oracle_worker, jobs, chunksize=1).get(timeout=120)
                tw = time.perf_counter() - tw0
                ncore = {"value": round(cores * sample / tw / 1e9, 5), "unit": "GB/s", "cores": cores,
                         "sample": "%d independent %d MiB TEXT streams, one worker process per usable host core "
                                   "(affinity mask / cgroup quota; the box reports %d logical CPUs), encode+decode"
                                   % (cores, sample >> 20, host_cores)}
            except Exception as e:      # noqa: BLE001  (timeout or a dead worker: report, do not stall)
                ncore = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            finally:
                pool.terminate()
                pool.join()
        cpu = {"value": round(n / (t2 - t0) / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port", "host_cores": host_cores,
               "usable_cores": cores,
               "sample": "the whole benchmarked buffer (%d MiB, %s), oracle C restatement on one core: encode %.3f GB/s, "
                         "decode %.3f GB/s; its output equals the GPU's byte for byte" % (n >> 20, args.schedule,
                                                                                         n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9),
               "n_streams_on_n_cores": ncore}
    line = {
        "metric": "gzip encode+decode throughput on 256 MiB synthetic text per GPU (uncompressed bytes through the round trip)",
        "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "ms_per_step_median": round(median_s * 1e3, 3) if median_s else None,
        "value_median": round(total_bytes / median_s / 1e9, 4) if median_s else None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "cfg2: gzip::Encoder (DefaultLz77Encoder, default options, mtime=0) + gzip::Decoder on "
                               "TEXT(%d B) per GPU, write schedule %s" % (n, "S8K (8192-byte writes)" if args.schedule == "S8K" else "S1 (one write_all)"),
                   "bytes_per_gpu": n, "schedule": args.schedule, "compressed_bytes": m,
                   "parallelism": "1 rank" if not sharded_path else
                   "%d ranks, one gzip member: all-gather of shard infos + shards concatenated on rank 0 (RCCL point-to-point, "
                   "all posted at once, in flight while the ranks decode); decode = N-GPU decode of the one member by byte ranges: "
                   "block finder + scan per range, one all-gather of candidate tuples, chain walk on every rank, no bit "
                   "offset from the encoder" % world},
        "encode_GBps": round(total_bytes * args.steps / enc_t / 1e9, 4),
        "decode_GBps": round(total_bytes * args.steps / dec_t / 1e9, 4),
        "phases_ms": {k: round(v, 4) for k, v in sorted(avg.items())},
        "phases_ms_how": "HIP events behind every phase in extra steps behind the timed loop (an event record between two kernels costs "
                         "~6 us of idle GPU: the timed steps carry only the two around the longest kernel's phase — " +
                         str(getattr(run, "bracket_phase", None)) + ", picked from one instrumented step in front of them — whose figure "
                         "here is theirs); the encode's match and parse phases are split by kernel (timing level 6): lz77_cand = "
                         "lz77_match7_kernel, lz77_resolve = compaction + resolver, lz77_walk = parse_walk_kernel, lz77_chain = parse_fixseg + "
                         "parse_fix + parse_emit_hist",
        "roofline": roof, "whole_path": whole, "schedule_S1" if args.schedule == "S8K" else "schedule_S8K": s1,
        "other_configs": subs,
        "pcie_inclusive": pcie,
        "stream_api": stream_api,
        "raw_deflate": raw_deflate,
        "match_fallbacks": ctx.match_fallbacks(),
        "cfg4": cfg4,
        "cpu_baseline": cpu,
    }
    if sharded_path:
        # what lfx_sharded_encode_finish still had to wait for after the decode (max over ranks, mean over steps): the part of
        # the concatenation that did NOT hide behind the decode, plus placing the shards on rank 0
        fin = getattr(run, "finish_max_s", None)
        line["overlap_ms"] = round(fin * 1e3, 3) if fin is not None else None
        line["overlap_frac_of_step"] = round(fin / (elapsed / args.steps), 4) if fin is not None else None
        # the same wait with nothing between begin and finish (outside the timed steps): the transfers' full duration.  The
        # difference is what the decode hid.  (Over gloo on one box — LFX_BENCH_ONE_GPU — the shards hop through host memory
        # and loopback TCP and take several decodes' time: most of the wait remains; over xGMI 7 x 130 MB are ≈ 1.3 ms.)
        ga = getattr(run, "gather_alone_s", None)
        line["gather_alone_ms"] = round(ga * 1e3, 3) if ga is not None else None
    # The JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio, whose buffer (when stdout is
    # a pipe or a file) would otherwise be flushed at process exit, behind this line.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist:
        dist.destroy_process_group()       # (the other ranks left theirs right after the timed region)
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
