"""Frame-parallel sharding + gather, world_size 2 over gloo on CPU (the GPU path uses the same
code with backend nccl = RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_recon(frame):
    """Stand-in for one reconstruction: a deterministic [5,5,3] 'render' per frame id."""
    g = torch.Generator().manual_seed(1000 + frame)
    return torch.rand((5, 5, 3), generator=g)


def _worker(rank, world, port, n_frames, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from monoport_amd import parallel
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = parallel.frames_of_rank(rank, world, n_frames)
    gather = parallel.FrameGather((5, 5, 3))
    for k in range(parallel.rounds(world, n_frames)):
        fid = k * world + rank
        gather.push(k, _fake_recon(fid) if fid in mine else None)
    import torch.distributed as dist
    dist.barrier()
    if rank == 0:
        torch.save(gather.ordered(n_frames), out_path)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [4, 5])
def test_two_rank_frame_parallel_gather(tmp_path, n_frames):
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_frames, out), nprocs=2, join=True)
    got = torch.load(out)
    want = torch.stack([_fake_recon(i) for i in range(n_frames)])
    assert torch.equal(got, want)  # identical to the single-process result, in frame order


def test_single_process_path():
    from monoport_amd import parallel
    assert parallel.frames_of_rank(1, 4, 10) == [1, 5, 9]
    assert parallel.rounds(4, 10) == 3
    g = parallel.FrameGather((2, 2))
    g.push(0, torch.ones(2, 2))
    g.push(1, 2 * torch.ones(2, 2))
    assert torch.equal(g.ordered(2), torch.stack([torch.ones(2, 2), 2 * torch.ones(2, 2)]))


def _one_rank_worker(rank, port, out_path):
    import torch.distributed as dist
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoport_amd import parallel
    r, w = parallel.init_from_env(backend="gloo", force=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1
    g = parallel.FrameGather((2, 2))
    assert g._collective  # the gather goes through the backend, not the single-process shortcut
    g.push(0, torch.ones(2, 2))
    g.push(1, 2 * torch.ones(2, 2))
    assert torch.equal(g.received(0), 2 * torch.ones(2, 2))
    torch.save(g.ordered(2), out_path)
    dist.destroy_process_group()


def test_forced_one_rank_group_gathers_through_the_backend(tmp_path):
    """bench.py's MONOPORT_BENCH_FORCE_GROUP hook (a one-GPU box running the N > 1 collectives on RCCL): a group
    of ONE rank is a real group -- init_from_env(force=True) makes it, FrameGather gathers through it."""
    out = str(tmp_path / "one.pt")
    mp.spawn(_one_rank_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert torch.equal(torch.load(out), torch.stack([torch.ones(2, 2), 2 * torch.ones(2, 2)]))
