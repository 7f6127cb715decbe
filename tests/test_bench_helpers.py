"""CPU: the parts of bench.py that decide what runs where (no GPU needed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_usable_cores_and_profiler_detection(monkeypatch):
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    for k in list(os.environ):
        if k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.under_profiler()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")      # what rocprofv3 exports into the profiled process tree
    assert bench.under_profiler()                             # → no nested counter passes, no worker pool


def test_phase_kernel_table_names_existing_kernels():
    import bench
    src = ""
    for f in ("lfx_match3.hip", "lfx_parse2.hip", "lfx_encode_kernels.hip", "lfx_inflate_fast.hip", "lfx_decode_kernels.hip"):
        src += open(os.path.join(ROOT, "libflate_amd", "csrc", f)).read()
    for phase, kernel in bench.PHASE_KERNEL.items():
        assert "void %s(" % kernel in src, (phase, kernel)    # a renamed kernel must not silently lose its traffic figure
    assert "void %s(" % bench.CALIBRATION_KERNEL in src
