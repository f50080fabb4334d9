"""Profiling helpers (the reference has only epoch timestamps).

* :class:`RegionTimer` — HIP-event timing of named regions with no host sync until ``summary()``;
* :func:`rocprof_command` — builds a ``rocprofv3`` command line that respects this pool's rules: counters (``--pmc``)
  are collected in a run of their own, never together with API / system traces, at most one hardware pass per run,
  and the profiled program is the interpreter itself (no ``env`` / shell hop between ``--`` and the program);
* :func:`pmc_groups` — counter sets that each fit one pass (<= 8 SQ, <= 4 TCC, <= 2 GRBM counters);
* :func:`summarize_kernel_trace` — per-kernel time per training step out of the rocpd SQLite file that
  ``rocprofv3 --kernel-trace`` writes, plus how much of the wall span had any kernel running (overlap / gaps).

On gfx950 ``FETCH_SIZE`` reports half the bytes of wide streaming reads (MI355X guide): double it before comparing
with a byte count.
"""
from __future__ import annotations

import re
import sqlite3
import sys
import time
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import torch


class RegionTimer:
    def __init__(self, device: Optional[torch.device] = None):
        self.gpu = (device.type == "cuda") if device is not None else torch.cuda.is_available()
        self._open: Dict[str, object] = {}
        self._spans: Dict[str, List] = defaultdict(list)

    def start(self, name: str) -> None:
        if self.gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._open[name] = ev
        else:
            self._open[name] = time.perf_counter()

    def stop(self, name: str) -> None:
        a = self._open.pop(name)
        if self.gpu:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self._spans[name].append((a, b))
        else:
            self._spans[name].append((time.perf_counter() - a) * 1e3)

    class _Ctx:
        def __init__(self, t, name):
            self.t, self.name = t, name

        def __enter__(self):
            self.t.start(self.name)

        def __exit__(self, *exc):
            self.t.stop(self.name)

    def region(self, name: str):
        return RegionTimer._Ctx(self, name)

    def summary(self) -> Dict[str, Dict[str, float]]:
        if self.gpu:
            torch.cuda.synchronize()
        out = {}
        for name, spans in self._spans.items():
            ms = [a.elapsed_time(b) for a, b in spans] if self.gpu else list(spans)
            if ms:
                out[name] = {"calls": len(ms), "total_ms": sum(ms), "mean_ms": sum(ms) / len(ms), "max_ms": max(ms)}
        return out


def pmc_groups() -> Dict[str, List[str]]:
    return {
        "issue": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                  "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA"],
        "stalls": ["SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_LDS",
                   "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"],
        "memory": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "GRBM_GUI_ACTIVE",
                   "GRBM_COUNT"],
    }


def rocprof_command(program: Sequence[str], out_dir: str, name: str = "trace", pmc: Optional[Sequence[str]] = None,
                    stats: bool = True) -> List[str]:
    """``program`` = e.g. ``["python3", "bench.py", "--steps", "5"]``.  With ``pmc`` the run collects ONLY those
    counters (plus the kernel trace that names the dispatches)."""
    if program and program[0] in ("env", "bash", "sh", "taskset", "numactl"):
        raise ValueError("profile the program itself: no env / shell / launcher hop after '--'")
    cmd = ["rocprofv3", "--kernel-trace", "-d", out_dir, "-o", name]
    if pmc:
        sq = sum(c.startswith("SQ_") for c in pmc)
        tcc = sum(c.startswith("TCC_") for c in pmc)
        grbm = sum(c.startswith("GRBM_") for c in pmc)
        if sq > 8 or tcc > 4 or grbm > 2:
            raise ValueError("counter set needs more than one hardware pass: split it (<= 8 SQ, 4 TCC, 2 GRBM)")
        cmd += ["--pmc", *pmc]
    elif stats:
        cmd.append("--stats")
    return cmd + ["--", *program]


def _short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:72]


def summarize_kernel_trace(db_path: str, steps: int = 5, step_marker: Sequence[str] = ("flat_sgd", "flat_adam"),
                           top: int = 32) -> Dict[str, object]:
    """Averages over the last ``steps`` training steps; a step ends at the fused optimizer kernel."""
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if any(m in r[0] for m in step_marker)]
    if len(marks) < steps + 1:
        raise ValueError(f"trace holds {len(marks)} optimizer steps, need {steps + 1}")
    a, b = marks[-steps - 1], marks[-1]
    span = (rows[b][1] - rows[a][1]) / steps
    tot, cnt, events = defaultdict(float), defaultdict(int), []
    for n, s, e in rows[a:b]:
        k = _short(n)
        tot[k] += (e - s) / steps
        cnt[k] += 1
        events += [(s, 1), (e, -1)]
    events.sort()
    busy = depth = 0
    last = None
    for t, d in events:
        if depth > 0:
            busy += t - last
        depth += d
        last = t
    kernels = [{"name": k, "ms_per_step": t / 1e6, "launches_per_step": cnt[k] / steps}
               for k, t in sorted(tot.items(), key=lambda kv: -kv[1])[:top]]
    return {"step_span_ms": span / 1e6, "kernel_sum_ms": sum(tot.values()) / 1e6, "gpu_busy_ms": busy / steps / 1e6,
            "kernels": kernels}


def format_summary(s: Dict[str, object]) -> str:
    lines = [f"step span {s['step_span_ms']:.3f} ms | sum of kernel durations {s['kernel_sum_ms']:.3f} ms | "
             f"GPU non-idle {s['gpu_busy_ms']:.3f} ms"]
    for k in s["kernels"]:
        lines.append(f"{k['ms_per_step']:8.3f} ms {k['launches_per_step']:7.1f}  {k['name']}")
    return "\n".join(lines)


if __name__ == "__main__":
    print(format_summary(summarize_kernel_trace(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)))
