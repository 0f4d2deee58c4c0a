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
    # a launch without the `out` / `x` stores (the model's inference calls) is priced with what it moves: 8(d)'s 1.082 GB
    assert b.scan_bytes(256, 1024, 256, 16, has_out=False, has_x=False) == 4 * 256 * 1024 * 256 * 4 + 2 * 256 * 16 * 256 * 4 + (1024 * 16 + 2048) * 4 == 1082204160
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
    monkeypatch.setenv("WORLD_SIZE", "8")           # as a rank of torch.distributed.run (without it: the launcher, below)
    try:
        b.main()
    except Stop:
        pass
    a = seen["args"]
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)


def test_gpus_n_without_rank_env_launches_child_ranks(monkeypatch):
    """`python bench.py --gpus 2` as the driver may invoke it (no RANK / WORLD_SIZE): the process becomes a launcher BEFORE
    anything touches the GPU -- N fresh children through torch.distributed.run (scripts/eval.sh:73's torchrun), the same
    argv handed down, their exit code handed up. With the rank environment present it runs as a rank instead."""
    import pytest
    b = _bench()
    calls = []

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return Done()

    def no_bench(args):
        raise AssertionError("the launcher process must not construct Bench (that is where the GPU is first touched)")

    monkeypatch.setattr(b.subprocess, "run", fake_run)
    monkeypatch.setattr(b, "Bench", no_bench)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 7                                    # the children's exit code is the launcher's
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

    # inside a rank (WORLD_SIZE set by the launcher): no second launch
    calls.clear()
    seen = {}

    class Stop(Exception):
        pass

    def fake_bench(args):
        seen["gpus"] = args.gpus
        raise Stop

    monkeypatch.setattr(b, "Bench", fake_bench)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(Stop):
        b.main()
    assert seen["gpus"] == 2 and not calls


def test_launcher_runs_real_children_and_relays_rank0_line(tmp_path):
    """the launch path for real, without a GPU: a stand-in script (not bench.py: that needs the device) is started by
    bench.launch_ranks' command line as 2 ranks over gloo; rank 0's line comes back on the inherited stdout and a failing
    rank makes the launcher return non-zero."""
    import subprocess
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, json\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'world': dist.get_world_size(), 'argv': sys.argv[1:]}), flush=True)\n"
        "dist.barrier()\n"
        "dist.destroy_process_group()\n"
        "sys.exit(3 if '--fail' in sys.argv and os.environ['RANK'] == '1' else 0)\n")
    prog = ("import sys, importlib.util\n"
            f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r})\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            f"b.__file__ = {str(child)!r}\n"
            "sys.exit(b.launch_ranks(2, sys.argv[1:]))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for attempt in range(2):          # (the rendezvous port is picked, released and re-bound by the children: one retry covers a lost race)
        ok = subprocess.run([sys.executable, "-c", prog, "--gpus", "2"], capture_output=True, text=True, env=env, timeout=600)
        if ok.returncode == 0:
            break
    assert ok.returncode == 0, ok.stderr[-2000:]
    import json
    line = [l for l in ok.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and json.loads(line[0]) == {"world": 2, "argv": ["--gpus", "2"]}
    bad = subprocess.run([sys.executable, "-c", prog, "--gpus", "2", "--fail"], capture_output=True, text=True, env=env, timeout=600)
    assert bad.returncode != 0


def _diag_worker(rank, world, port, q):
    import importlib.util
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    from dimsum_amd.sample_ddp import sample_batch

    class Field(torch.nn.Module):
        in_channels, num_classes = 2, 10

        def forward(self, x, t, y=None):
            return -x * (1 + y.view(-1, 1, 1, 1).float())

    torch.manual_seed(rank)
    z, y = torch.randn(4, 2, 4, 4), torch.full((4,), rank)
    stats = {}
    t0 = time.perf_counter()
    full = sample_batch(Field(), z, y, num_steps=2, world_size=world, stats=stats)            # --nfe 2
    issued = time.perf_counter() - t0
    if rank == 1:
        time.sleep(0.2)                                                                       # a straggler the summary must name
    own = time.perf_counter() - t0
    summary = b.rank_summary(own, issued, 1, stats)
    q.put((rank, summary, tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_line_explains_itself():
    """what bench.py adds to its line at N > 1 (rank_summary + sample_batch(stats=...)), on 2 gloo ranks: per-rank step times with the
    slowest rank named, the all-gather's own time, the host-issue fraction, the thread count, and the in-line check that every rank's
    block of the gathered latents is its own output"""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_diag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, shape0), (_, s1, shape1) = res
    assert s0 == s1 and shape0 == shape1 == (8, 2, 4, 4)                 # every rank reports the same summary
    pr = s0["per_rank_ms_per_step"]
    assert pr["rank_of_max"] == 1 and pr["max"] >= pr["min"] + 150 and len(pr["all"]) == 2
    assert s0["gathered_block_equals_own_output"] is True
    assert 0 <= s0["all_gather_ms"]["min"] <= s0["all_gather_ms"]["max"]
    assert 0 < s0["host_issue_fraction_max"] <= 1.0 and s0["torch_threads_per_rank"]
