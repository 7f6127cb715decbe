"""CPU (hipcc cross-compiles): the kernels whose loops run over the data hold no "one load, one full wait" loop — the pattern
behind the window resolution's 0.59 ms (round 4) and behind `tile_bits` / `pack` / `parse_emit` / the checksum sweep (round 3),
see tools/isa_scan.py and DESIGN.md §4 / §5."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_no_serialized_loads_in_the_data_loops():
    import isa_scan
    watched = {
        "lfx_inflate_fast.hip": ("window_chain_kernel", "window_compose_kernel", "window_groups_kernel", "window_apply_kernel",
                                 "window_rank_map_kernel", "window_ranks_kernel", "sym_substitute_kernel", "blk_scan_kernel",
                                 "blk_emit_kernel", "find_blocks_stage2"),
        "lfx_decode_kernels.hip": ("find_blocks_stage1",),
        "lfx_match7.hip": ("lz77_match7_kernel", "lz77_compact7_kernel", "lz77_resolve7_kernel"),
        "lfx_encode_kernels.hip": ("checksum_span_kernel", "checksum_ranges_kernel", "histogram_kernel", "huffman_kernel"),
    }
    for fname, kernels in watched.items():
        asm = isa_scan.compile_to_asm(os.path.join(ROOT, "libflate_amd", "csrc", fname))
        assert all(any(k in line for line in asm.split("\n") if line.startswith("_Z")) for k in kernels), fname   # (the names still exist)
        bad = [f for f in isa_scan.serialized_load_loops(asm) if any(k in f[0] for k in kernels)]
        # known and left: the copy of a stream decoder's 32 KiB of history into LDS at the start of the two chain kernels
        # (thirty-two trips per LAUNCH, not per step; no load at all at the start of a member)
        bad = [f for f in bad if not (("window_chain_kernel" in f[0] or "window_groups_kernel" in f[0]) and f[2] <= 16 and f[4] == 0)]
        assert not bad, bad
