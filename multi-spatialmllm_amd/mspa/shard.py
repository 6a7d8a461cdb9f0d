"""One process per GPU: shard frame pairs / scenes across ranks, collate records over RCCL.

The geometry path shards embarrassingly: frame pairs (K3), (scene, image)
projections (K1) and scenes (K2, which needs all bitsets of one scene) share nothing, so there is
NO collective on the data path.  The one exchange step is the collation of the per-pair / per-scene
numeric records at the end: an all_gather of the record counts followed by an all_gather of the
padded fixed-width record tensors (RCCL over xGMI when the backend is "nccl"; the same code runs on
gloo/CPU tensors for the world_size-2 tests).  Text templating happens on the host after it.
"""
from __future__ import annotations

import dataclasses
import datetime
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclasses.dataclass
class DistContext:
    rank: int
    world: int
    device: torch.device
    group: object = None
    owns_process_group: bool = False
    backend: str = "nccl"

    @property
    def collective_device(self) -> torch.device:
        """Where collectives run: the GPU for RCCL, host memory for gloo (CPU tests, 1-GPU boxes)."""
        return self.device if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.device.type == "cuda" and self.backend == "nccl":
            dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            dist.barrier(group=self.group)

    def max_over_ranks(self, x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=self.collective_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self):
        if self.owns_process_group and dist.is_initialized():
            dist.destroy_process_group()


def init_distributed(device: torch.device, backend: str = None, timeout_s: int = 600) -> DistContext:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run)."""
    backend = backend or ("nccl" if device.type == "cuda" else "gloo")   # "nccl" IS RCCL on ROCm
    owns = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:      # a single process asking for a communicator
            import socket                                                  # (MSPA_BENCH_FORCE_DIST, tests): world of one
            os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        kwargs = {}
        if device.type == "cuda" and backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
        owns = True
    return DistContext(dist.get_rank(), dist.get_world_size(), device, None, owns, dist.get_backend())


def partition(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) range of a flat work list (config-2 style pair lists)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def lpt_assign(costs: Sequence[float], world: int, rank0_share: float = 1.0) -> List[List[int]]:
    """Longest-processing-time-first assignment of scenes to ranks (cost ~ F^2*N/64 + F*N).  ``rank0_share`` < 1 starts
    rank 0 with a handicap so that it ends with about that fraction of what each other rank gets (rank 0 also writes the
    split's files: mspa/sweep.py); every rank computes the same assignment."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    if world > 1 and rank0_share < 1.0:
        # balanced loads with a handicap h on rank 0: the others end at L, rank 0 at L - h = share * L, and
        # (world - 1 + share) * L = total
        load[0] = max(0.0, (1.0 - float(rank0_share)) * float(sum(costs)) / (world - 1 + float(rank0_share)))
    bins: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += costs[i]
    for b in bins:
        b.sort()
    return bins


def scene_cost(n_frames: int, n_points: int) -> float:
    return float(n_frames) ** 2 * n_points / 64.0 + float(n_frames) * n_points


def collate_records_async(local: torch.Tensor, ctx: DistContext, out: torch.Tensor = None):
    """Fixed-size collation (every rank contributes the same number of records, e.g. the bench's
    per-step batches): ONE all_gather, enqueued asynchronously -- no count exchange, no host
    synchronisation.  Returns (gathered [world*n, k], work); ``work.wait()`` orders the caller's stream
    after the collective (it does not block the host on RCCL)."""
    if local.dim() != 2 or not local.is_contiguous():
        raise ValueError("collate_records_async: a contiguous [n, k] record tensor is required")
    if out is None:
        out = torch.empty((ctx.world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    if local.device != ctx.collective_device:          # gloo: collectives on host memory (CPU tests, ranks sharing one GPU)
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(ctx.world)]
        dist.all_gather(parts, local.cpu(), group=ctx.group)
        out.copy_(torch.cat(parts, 0))

        class _Done:
            def wait(self):
                return True
        return out, _Done()
    work = dist.all_gather_into_tensor(out, local, group=ctx.group, async_op=True)
    return out, work


def collate_records(local: torch.Tensor, ctx: DistContext, dst: int = None) -> torch.Tensor:
    """Collate a [n_local, k] record tensor whose n_local may differ per rank.

    Two collectives: counts (one int64 per rank), then the records padded to the largest count.  ``dst=None``: all_gather,
    the concatenation in rank order comes back identical on every rank.  ``dst=r``: gather to rank r only (the pipeline's
    tapes, which only rank 0 reads: fabric traffic world x less); the other ranks get an empty [0, k] tensor."""
    if local.dim() != 2:
        raise ValueError("collate_records: a [n, k] record tensor is required")
    if local.device != ctx.collective_device:
        local = local.to(ctx.collective_device)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(ctx.world)]
    dist.all_gather(counts, n_local, group=ctx.group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:                                   # nobody has a record: no second collective
        return local[:0]
    if n_max == local.shape[0]:
        padded = local.contiguous()
    else:
        padded = torch.zeros((n_max, local.shape[1]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    if dst is not None:
        parts = [torch.empty_like(padded) for _ in range(ctx.world)] if ctx.rank == dst else None
        dist.gather(padded, parts, dst=dst, group=ctx.group)
        if ctx.rank != dst:
            return local[:0]
        return torch.cat([parts[r][:c] for r, c in enumerate(counts)], dim=0)
    gathered = torch.empty((ctx.world * n_max, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=ctx.group)
    if all(c == n_max for c in counts):
        return gathered
    return torch.cat([gathered[r * n_max:r * n_max + c] for r, c in enumerate(counts)], dim=0)


def gather_bytes(payload, ctx: DistContext, dst: int = 0):
    """Collate one byte string per rank on rank ``dst``: finished UTF-8 text (JSONL records, warning lines) or an arrow IPC
    stream built by its owner -- text stays where it was produced and crosses the fabric once, as bytes.  Two collectives,
    like ``collate_records``: the lengths (all_gather of one int64), then a gather of uint8 tensors padded to the longest.
    Returns a list of ``world`` uint8 NumPy arrays on rank ``dst`` (zero-copy views of the received buffers), None elsewhere."""
    import numpy as np
    arr = np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload.reshape(-1).view(np.uint8)
    dev = ctx.collective_device
    n_local = torch.tensor([arr.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(ctx.world)]
    dist.all_gather(counts, n_local, group=ctx.group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:
        return [np.zeros(0, np.uint8) for _ in range(ctx.world)] if ctx.rank == dst else None
    padded = torch.zeros(n_max, dtype=torch.uint8, device=dev)
    if arr.size:
        padded[:arr.size] = torch.from_numpy(np.array(arr, copy=True) if not arr.flags.writeable else arr).to(dev)
    parts = [torch.empty_like(padded) for _ in range(ctx.world)] if ctx.rank == dst else None
    dist.gather(padded, parts, dst=dst, group=ctx.group)
    if ctx.rank != dst:
        return None
    return [parts[r][:c].cpu().numpy() for r, c in enumerate(counts)]


def raise_together(ctx: "DistContext | None", failure: "BaseException | None", what: str = "a sharded step"):
    """One int over the ranks: did ANY rank fail in the step that just ended?  The failing rank raises its own exception,
    every other rank a RuntimeError -- instead of walking into the next collective and standing there until the
    communicator times out.  Call it on every rank where a rank-local step ends and a collective begins."""
    if ctx is None:
        if failure is not None:
            raise failure
        return
    flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device=ctx.collective_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctx.group)
    if int(flag.item()):
        if failure is not None:
            raise failure
        raise RuntimeError(f"{what}: another rank failed (its own traceback says why)")


def broadcast_object(obj, ctx: "DistContext | None", src: int = 0):
    """``obj`` of rank ``src`` on every rank (pickled; a few hundred bytes to a few KB: generator states, counts).  With no
    communicator the object itself."""
    if ctx is None:
        return obj
    box = [obj if ctx.rank == src else None]
    dist.broadcast_object_list(box, src=src, group=ctx.group, device=ctx.collective_device)
    return box[0]


def context_from_env(device=None) -> "DistContext | None":
    """The communicator of a job launched with one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK in the environment, as
    torch.distributed.run sets them), or None for a plain single-process run: what the drop-in entry points
    (``run_split``, ``generate_qa_training_data``, ``mspa.pipeline``) call when no context is handed to them.
    ``MSPA_DIST_BACKEND`` overrides the backend (``gloo``: ranks that share one GPU, CPU tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not dist.is_initialized():
        return None
    if device is None:
        if torch.cuda.is_available():
            local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            device = torch.device("cuda", local)
        else:
            device = torch.device("cpu")
    ctx = init_distributed(torch.device(device), backend=os.environ.get("MSPA_DIST_BACKEND"))
    if ctx.owns_process_group:                          # created here: torn down when the script ends, whichever entry point came first
        import atexit
        atexit.register(lambda: dist.is_initialized() and dist.destroy_process_group())
        ctx.owns_process_group = False
    return ctx if ctx.world > 1 else None
