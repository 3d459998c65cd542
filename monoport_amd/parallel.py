"""Frame-parallel sharding across the GPUs of one node.

Frames are independent (no temporal state in netG / the octree engine), so rank r of N
reconstructs frames r, r+N, r+2N, ... with a full model replica and the only communication is a
gather of each frame's fixed-size result to rank 0 -- RCCL over xGMI on the GPUs (backend
"nccl"), gloo in the CPU tests.  The reference has no equivalent: its two-GPU mode is a
hand-written functional split (RTL/main.py:87-99) and output order comes from its FIFO queues
(RTL/dataloader.py:883-888); here order is restored from the frame index.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun sets them).
    Returns (rank, world).  A single process needs no group; ``force`` makes one anyway (a
    ONE-rank RCCL communicator: how a one-GPU box exercises the collective calls of the N > 1
    path on the real backend, tests/test_dropin_gpu.py)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only a forced single-process group gets here without one
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def frames_of_rank(rank, world, n_frames):
    """Frame ids handled by ``rank``: round-robin, rank r gets r, r+world, ..."""
    return list(range(rank, n_frames, world))


def rounds(world, n_frames):
    """Number of gather rounds needed so every rank takes part in every collective."""
    return (n_frames + world - 1) // world


class FrameGather:
    """Collects per-frame results of fixed shape on rank 0, restoring global frame order."""

    def __init__(self, shape, dtype=torch.float32, device="cpu", dst=0, store=True):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.dst = dst
        self.shape = tuple(shape)
        self.dtype = dtype
        # gloo has no gather for device tensors: stage through the host (CPU tests and the
        # one-GPU test hook of bench.py only; the GPU path is RCCL on device buffers)
        self._stage = (dist.is_initialized() and dist.get_backend() == "gloo"
                       and torch.device(device).type != "cpu")
        self.device = device
        if self._stage:
            device = "cpu"
        self._bufs = ([torch.empty(self.shape, dtype=dtype, device=device)
                       for _ in range(self.world)] if self.rank == dst else None)
        self._pad = torch.zeros(self.shape, dtype=dtype, device=device)
        self.store = store  # False: only the newest round stays in the receive buffers
        # a process group of ONE rank still gathers through its backend (bench.py's forced RCCL group)
        self._collective = dist.is_initialized()
        self.results = {}  # frame id -> tensor (rank dst only)

    def push(self, round_idx, result):
        """Every rank calls this once per round with its frame's result (or None past the end).
        Frame id of rank r in round k is k*world + r."""
        payload = self._pad if result is None else result.to(self.dtype).contiguous()
        if self.world == 1 and not self._collective:
            if result is not None and self.store:
                self.results[round_idx] = payload.clone()
            return
        if self._stage:
            payload = payload.cpu()
        dist.gather(payload, self._bufs, dst=self.dst)
        if self.rank == self.dst and self.store:
            for r in range(self.world):
                self.results[round_idx * self.world + r] = self._bufs[r].to(self.device, copy=True)

    def received(self, rank):
        """Rank ``rank``'s payload of the newest round (rank dst only; on the collective's stream)."""
        return self._bufs[rank]

    def ordered(self, n_frames):
        """[n_frames, *shape] in frame order (rank dst)."""
        return torch.stack([self.results[i] for i in range(n_frames)])
