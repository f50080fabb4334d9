"""utils/: seeding, rocprofv3 command rules, kernel-trace summary, debug ledger."""
import os
import sqlite3

import pytest
import torch

from distributed_deep_learning_amd import utils
from distributed_deep_learning_amd.utils import profiling


def test_seed_everything_is_deterministic_and_rank_offset():
    utils.seed_everything(7)
    a = torch.rand(4)
    utils.seed_everything(7)
    assert torch.equal(a, torch.rand(4))
    assert utils.seed_everything(7, rank=3, per_rank=True) == 10
    assert utils.seed_everything(7, rank=3, per_rank=False) == 7


def test_rocprof_command_rules():
    cmd = utils.rocprof_command(["python3", "bench.py", "--steps", "5"], "out", "t")
    assert cmd[:2] == ["rocprofv3", "--kernel-trace"] and "--stats" in cmd and cmd[-4:] == ["python3", "bench.py", "--steps", "5"]
    assert not any(c in cmd for c in ("--sys-trace", "--hip-trace", "--hsa-trace", "-s", "-r"))
    for grp in utils.pmc_groups().values():                      # every group fits ONE hardware pass
        c = utils.rocprof_command(["python3", "x.py"], "out", pmc=grp)
        assert "--pmc" in c and "--stats" not in c and "--sys-trace" not in c
    with pytest.raises(ValueError):
        utils.rocprof_command(["python3", "x.py"], "out", pmc=[f"SQ_C{i}" for i in range(9)])
    with pytest.raises(ValueError):
        utils.rocprof_command(["env", "A=1", "python3", "x.py"], "out")


def test_kernel_trace_summary(tmp_path):
    db = sqlite3.connect(str(tmp_path / "t.db"))
    db.execute("create table kernels (name text, start integer, end integer)")
    t = 0
    for _step in range(4):
        for name, dur in (("void conv_ring_kernel<128>(ConvParams)", 3000), ("bn_apply_kernel<true>(x)", 1000),
                          ("flat_sgd_kernel(float*)", 500)):
            db.execute("insert into kernels values (?,?,?)", (name, t, t + dur))
            t += dur + 100
    db.commit()
    db.close()
    s = profiling.summarize_kernel_trace(str(tmp_path / "t.db"), steps=3)
    assert abs(s["step_span_ms"] - 4800e-6) < 1e-9
    assert s["kernels"][0]["name"] == "conv_ring_kernel<128>" and abs(s["kernels"][0]["ms_per_step"] - 3000e-6) < 1e-9
    assert abs(s["gpu_busy_ms"] - 4500e-6) < 1e-9
    assert "conv_ring_kernel" in profiling.format_summary(s)


def test_region_timer_cpu():
    t = utils.RegionTimer(torch.device("cpu"))
    for _ in range(3):
        with t.region("a"):
            sum(range(1000))
    s = t.summary()
    assert s["a"]["calls"] == 3 and s["a"]["total_ms"] >= 0.0


def test_check_finite():
    from distributed_deep_learning_amd.parallel import ParamArena
    m = torch.nn.Linear(4, 3)
    a = ParamArena(m.named_parameters())
    assert utils.check_finite(a) and utils.check_finite(a, "master")
    a.grad[1] = float("nan")
    assert not utils.check_finite(a)


def test_native_comm_library_resolves_rccl_at_run_time():
    """csrc/comm.cpp has no link-time RCCL dependency: it maps torch's copy with dlopen (works on a CPU box too)."""
    from distributed_deep_learning_amd.ops import _lib
    from distributed_deep_learning_amd.parallel.native_comm import _rccl_path
    if not _lib.available() or _rccl_path() is None:
        import pytest
        pytest.skip("kernel library or torch's librccl not present")
    lib = _lib.require()
    assert lib.ddl_comm_load(_rccl_path()) == 0
    assert lib.ddl_comm_version() > 20000
    assert lib.ddl_comm_all_reduce(None, None, None, 0, 0, 0, None) != 0      # bad arguments are refused, not crashed on


def test_ddl_lib_selects_another_build_of_the_kernel_library(tmp_path):
    """`DDL_LIB=path` (docs/DEBUG_SWITCHES.md) makes ops/_lib.py bind that file instead of the in-tree library: the way
    variant builds with parts of a kernel compiled out are timed (bench/stem_abl.sh).  Checked in a child process,
    because the path is read at import."""
    import subprocess
    import sys
    fake = tmp_path / "libddl_variant.so"
    code = ("from distributed_deep_learning_amd.ops import _lib; import sys; "
            "print(_lib.LIB_PATH); sys.exit(0)")
    env = dict(os.environ, DDL_LIB=str(fake))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == str(fake)
    env.pop("DDL_LIB")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip().splitlines()[-1].endswith(os.path.join("lib", "libddl_hip.so"))
