"""Host-side checks of bench.py (no GPU): the algorithmic-byte formulas it prices the scan kernels with are SURVEY.md 8(d)'s,
and the command-line contract of the driver (`--gpus N --steps K --warmup W`, defaults that finish within minutes)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_survey_8d():
    b = _bench()
    # 5 x 268.44 MB + 8.39 MB + 33.55 MB + 0.07 MB = 1.384 GB per forward call
    assert b.scan_bytes(256, 1024, 256, 16) == 5 * 256 * 1024 * 256 * 4 + 2 * 256 * 16 * 256 * 4 + 256 * 1024 * 32 * 4 + (1024 * 16 + 2048) * 4 == 1384194048
    # backward: (5 reads + 3 writes [+ 1 with the out_z recompute]) B D L s + 2 B N L (s + 4) + x  ->  2.20 / 2.47 GB
    assert b.scan_bwd_bytes(256, 1024, 256, 16) == 2466324480
    assert b.scan_bwd_bytes(256, 1024, 256, 16, recompute_out_z=False) == 2466324480 - 256 * 1024 * 256 * 4 == 2197889024
    # half-precision I/O halves the activation terms only; two 2048-chunks double the chunk-state term
    assert b.scan_bytes(2, 64, 4096, 16, s=2) == 5 * 2 * 64 * 4096 * 2 + 2 * 2 * 16 * 4096 * 2 + 2 * 64 * 2 * 32 * 4 + (64 * 16 + 128) * 4
    assert b.HBM_PEAK_GBPS == 8000.0


def test_command_line_contract(monkeypatch):
    b = _bench()
    seen = {}

    class Stop(Exception):
        pass

    def fake_bench(args):
        seen["args"] = args
        raise Stop

    monkeypatch.setattr(b, "Bench", fake_bench)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    try:
        b.main()
    except Stop:
        pass
    a = seen["args"]
    assert (a.gpus, a.steps, a.warmup, a.mode, a.model, a.batch, a.nfe, a.sample_batch) == (1, 10, 2, "all", "DiM-L/2", 256, 250, 128)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    try:
        b.main()
    except Stop:
        pass
    a = seen["args"]
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
