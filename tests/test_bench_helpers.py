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
    for f in ("lfx_match7.hip", "lfx_match5.hip", "lfx_parse2.hip", "lfx_encode_kernels.hip", "lfx_inflate_fast.hip", "lfx_decode_kernels.hip"):
        src += open(os.path.join(ROOT, "libflate_amd", "csrc", f)).read()
    for phase, kernel in bench.PHASE_KERNEL.items():
        assert "void %s(" % kernel in src, (phase, kernel)    # a renamed kernel must not silently lose its traffic figure
    assert set(bench.PHASE_WAVES_PER_SIMD) <= set(bench.PHASE_KERNEL)
    assert "void %s(" % bench.CALIBRATION_KERNEL in src


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus N` without a launcher must start N ranks, not run one."""
    import bench
    import pytest
    assert bench.launch_plan(1, {}, 0, []) == ("inline", None)
    # the driver's own launcher already made the ranks: run inline, but the flag must agree with the world size
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8, []) == ("inline", None)
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {"WORLD_SIZE": "2"}, 8, [])
    mode, cmd = bench.launch_plan(8, {}, 8, ["--gpus", "8", "--steps", "3"])
    assert mode == "spawn"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")
    with pytest.raises(SystemExit):             # fewer devices than ranks: fail loudly ...
        bench.launch_plan(8, {}, 1, [])
    mode, cmd = bench.launch_plan(3, {"LFX_BENCH_ONE_GPU": "1"}, 1, ["--gpus", "3"])    # ... unless it is the one-GPU self-test
    assert mode == "spawn" and cmd[cmd.index("--nproc-per-node") + 1] == "3"


def test_strong_scaling_share_is_whole_blocks():
    import bench
    for total in (256 << 20, 1 << 30, 100 << 20):
        for world in (1, 2, 3, 4, 7, 8):
            n = bench.strong_share(total, world)
            assert n % (1 << 20) == 0 and n >= 1 << 20 and n * world <= max(total, world << 20)
