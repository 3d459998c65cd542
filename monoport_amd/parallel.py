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


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(device_index, local_rank=0, local_world=1, sysfs="/sys/bus/pci/devices"):
    """Pin the calling process (every thread it starts later inherits the mask) to the host cores next to its GPU.
    One process per GPU: the stage threads of a rank (eight Python threads feeding one device) should not
    migrate across sockets, nor share cores with the seven other ranks.  The cores come from the GPU's PCI device
    (``local_cpulist`` in sysfs, i.e. the CPUs of its NUMA node), divided among the ranks that share that node; when
    sysfs has no answer (containers, one-socket boxes) the cores the process may use are split evenly by local
    rank.  Returns a dict for the bench line (``config.cpu_affinity``); never raises -- affinity is an
    optimisation, not a requirement."""
    info = {"source": "unchanged", "cpus": None, "numa_node": None}
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return info
    local = None
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        base = os.path.join(sysfs, bdf)
        with open(os.path.join(base, "local_cpulist")) as f:
            local = _parse_cpulist(f.read()) & allowed
        with open(os.path.join(base, "numa_node")) as f:
            info["numa_node"] = int(f.read().strip())
    except Exception:  # noqa: BLE001 -- no GPU properties / no sysfs entry
        local = None
    if local and len(local) < len(allowed):
        # ranks on the same NUMA node share its cores: give each an equal slice (ranks of one node are those whose
        # GPUs report the same list; without a census we slice by the rank's position among `local_world` ranks
        # assuming GPUs are spread evenly over the nodes, which is how 8-GPU MI355X hosts are built)
        nodes = max(1, round(len(allowed) / len(local)))
        per_node = max(1, -(-local_world // nodes))
        cpus = sorted(local)
        k = local_rank % per_node
        share = cpus[k * len(cpus) // per_node:(k + 1) * len(cpus) // per_node] or cpus
        info["source"] = "sysfs local_cpulist of the GPU's PCI device, sliced over %d rank(s) per NUMA node" % per_node
    elif local_world > 1:
        cpus = sorted(allowed)
        share = cpus[local_rank * len(cpus) // local_world:(local_rank + 1) * len(cpus) // local_world] or cpus
        info["source"] = "even split of the allowed cores over %d local ranks (no NUMA information)" % local_world
    else:
        info["cpus"] = len(allowed)
        return info
    try:
        os.sched_setaffinity(0, share)
        info["cpus"] = len(share)
        info["first_cpu"], info["last_cpu"] = share[0], share[-1]
    except OSError as e:
        info["source"] = "unchanged (sched_setaffinity failed: %s)" % e
    return info


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
