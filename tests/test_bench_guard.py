"""bench.py's multi-rank safety net, on CPU with two gloo ranks: the captured-hipGraph phase is an optimisation that
must never cost the bench line.  graph_phase() returns the graph timing only when EVERY rank captured; when a rank
never comes back (a collective stuck inside a replay) each rank's deadline fires, rank 0 prints the already measured
eager line as the ONE JSON line on stdout, and all ranks exit 0."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import importlib.util, json, os, sys, time
    import torch, torch.distributed as dist
    spec = importlib.util.spec_from_file_location("bench", os.path.join(%(root)r, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    rank, scenario = int(os.environ["RANK"]), os.environ["SCENARIO"]
    bench.keep_stdout_for_json()
    print("library banner on stdout")                       # must not reach the real stdout
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))

    def run():
        if scenario == "raise" and rank == 1:
            raise RuntimeError("capture failed on this rank")
        if scenario == "hang" and rank == 1:
            time.sleep(120)
        return 0.5 + rank

    fallback = {"value": 1.0, "config": {"launch": "eager"}}
    dt = bench.graph_phase(None, rank, fallback, run, torch.device("cpu"))
    if rank == 0:
        bench.emit({"dt": dt, "scenario": scenario})
    dist.destroy_process_group()
''') % {"root": ROOT}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(scenario, deadline):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SCENARIO=scenario, MNK_GRAPH_DEADLINE_S=str(deadline), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=100) for p in procs]
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("scenario", ["ok", "raise"])
def test_graph_phase_ranks_agree(scenario):
    rcs, outs = _run(scenario, 60)
    assert rcs == [0, 0], outs
    lines = outs[0][0].splitlines()
    assert len(lines) == 1, "stdout of rank 0 must be exactly the JSON line: %r" % outs[0][0]
    got = json.loads(lines[0])
    assert got["dt"] == (0.5 if scenario == "ok" else None)   # one failed capture -> every rank keeps the eager number
    assert outs[1][0] == ""
    assert "library banner" in outs[0][1]


def test_graph_phase_deadline_prints_the_eager_line():
    rcs, outs = _run("hang", 3)
    assert rcs == [0, 0], outs
    lines = outs[0][0].splitlines()
    got = json.loads(lines[0])       # the eager measurement, and LOUDLY so: capture_failed in the line itself
    assert len(lines) == 1 and got["value"] == 1.0 and got["capture_failed"] is True, outs[0]
    assert got["config"]["launch"].startswith("eager (the hipGraph phase exceeded")
    assert outs[1][0] == "" and "deadline" in outs[0][1] and "deadline" in outs[1][1]


def test_gpus_flag_self_launches_n_ranks():
    """`python bench.py --gpus 2` without a rank environment must start 2 ranks by itself (round-1 verdict: the flag was
    parsed and ignored).  --launcher-selftest swaps the GPU workload for a gloo all-reduce so this runs on CPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    got = json.loads(lines[0])
    assert got == {"selftest": True, "n_gpus": 2, "ranks_seen": 2}


def test_print_launch_is_the_drivers_command():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--launcher-selftest",
                          "--print-launch"], capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k != "RANK"})
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.strip()
    assert "-m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1" in cmd
    assert cmd.endswith("bench.py --gpus 4 --steps 3 --launcher-selftest --print-launch")


def test_recon_l1_leg_of_the_bench_line(be):
    """bench.py's "recon L1 vs CPU ref" leg (the second half of BASELINE's metric) on the kernels' CPU emulation with the tiny
    configuration: the HIP-path eval forward and the oracle's give the same reconstruction L1 (north_star: within 1e-4)."""
    import importlib.util
    import torch
    from oracle import cases
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = bench.recon_l1_vs_cpu(cases.TINY, 32, be.device, batch=2)
    be.sync()
    assert set(rec) >= {"hip", "cpu_ref", "abs_diff", "max_abs_frame_diff", "sample"}
    assert 0.0 < rec["cpu_ref"] < 1.0 and rec["abs_diff"] < 1e-4 and rec["max_abs_frame_diff"] < 1e-3
