"""bench.py's launch path without a GPU: ``python bench.py --gpus N`` outside a launcher starts its
own N ranks (it used to run ONE rank silently and print n_gpus 1), joins the ranks a
launcher made otherwise, and refuses to measure when the GPUs are not there.  ``--rendezvous-only``
runs everything up to the measurement (process group, device census, one frame-sized gather to
rank 0) over gloo."""
import json
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, **env):
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_self_launch_two_ranks():
    res = _run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"])
    assert res.returncode == 0, res.stderr[-3000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1  # rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 2 and out["self_launched"] is True and out["gather_checked"] is True
    assert len(out["devices"]) == 2 and len(set(out["devices"])) == 2
    assert "launching 2 ranks" in res.stderr


def test_self_launch_eight_ranks():
    """BASELINE configs[3]'s rank count: eight ranks rendezvous, census their devices and gather one
    frame-sized payload each to rank 0 (gloo here; RCCL on an 8-GPU node, which this container lacks)."""
    res = _run([sys.executable, BENCH, "--gpus", "8", "--rendezvous-only"], OMP_NUM_THREADS="1")
    assert res.returncode == 0, res.stderr[-3000:]
    out = _json_lines(res.stdout)
    assert len(out) == 1 and out[0]["n_gpus"] == 8 and out[0]["gather_checked"] is True
    assert len(set(out[0]["devices"])) == 8 and out[0]["gather_ms_first"] > 0


def test_under_torchrun_joins_the_launchers_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                "--rendezvous-only"])
    assert res.returncode == 0, res.stderr[-3000:]
    out = _json_lines(res.stdout)
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["self_launched"] is False


def test_world_size_mismatch_is_an_error():
    res = _run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1",
                "--nproc-per-node", "2", BENCH, "--gpus", "4", "--rendezvous-only"])
    assert res.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in res.stderr


def test_measurement_refuses_without_enough_gpus():
    """A real run (no --rendezvous-only) with fewer GPUs than ranks must fail loudly, never fall
    back to fewer ranks or to the CPU."""
    res = _run([sys.executable, BENCH, "--gpus", "64", "--steps", "2", "--warmup", "1"])
    assert res.returncode != 0
    assert not _json_lines(res.stdout)
    if torch.cuda.is_available():
        assert "only %d GPU(s) are visible" % torch.cuda.device_count() in res.stderr
    else:
        assert "needs MI355X" in res.stderr


def test_in_flight_layout():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.in_flight_layout(8, 8) == (1, 1)  # configs[3]: one frame per rank
    assert bench.in_flight_layout(8, 4) == (2, 1)
    assert bench.in_flight_layout(8, 2) == (2, 2)
    assert bench.in_flight_layout(8, 1) == (2, 4)
    assert bench.pick_batch(20, 10) == 10 and bench.pick_batch(7, 10) == 7 and bench.pick_batch(22, 10) == 2
    assert bench.pick_batch(20, None) == 20 and bench.pick_batch(48, None) == 16 and bench.pick_batch(25, None) == 25
    assert bench.pick_batch(96, None) == 32 and bench.pick_batch(48, 8) == 8 and bench.pick_batch(64, None) == 32  # one rule (docstring)
    try:
        bench.in_flight_layout(8, 3)
    except SystemExit:
        pass
    else:
        raise AssertionError("8 frames over 3 ranks must be refused")


def test_rank_cpu_affinity_helpers():
    """parallel.pin_to_gpu_numa (bench.py: config.cpu_affinity on N > 1): the sysfs cpulist parser, and the fallback
    when the GPU's PCI device has no NUMA answer (here: no GPU at all) -- the allowed cores split evenly by local
    rank, disjoint between ranks, the mask applied to the process and restorable."""
    from monoport_amd import parallel
    assert parallel._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert parallel._parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    try:
        if len(before) >= 2:
            a = parallel.pin_to_gpu_numa(0, 0, 2)
            got_a = os.sched_getaffinity(0)
            os.sched_setaffinity(0, before)
            b = parallel.pin_to_gpu_numa(1, 1, 2)
            got_b = os.sched_getaffinity(0)
            assert got_a and got_b and not (got_a & got_b) and (got_a | got_b) <= before
            assert a["cpus"] == len(got_a) and b["cpus"] == len(got_b) and "even split" in a["source"]
        os.sched_setaffinity(0, before)
        one = parallel.pin_to_gpu_numa(0, 0, 1)  # a single process is left alone
        assert os.sched_getaffinity(0) == before and one["cpus"] == len(before)
    finally:
        os.sched_setaffinity(0, before)
