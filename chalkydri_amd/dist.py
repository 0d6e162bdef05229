"""Multi-GPU sharding of the hot path: one process per GPU, camera streams (or contiguous frame blocks) sharded with no
data-path collective; the only exchange is the final gather of 64-byte pose records (layout of whacknet's
VisionMeasurement, crates/whacknet/src/lib.rs:43-66) — one all_gather per batch (RCCL over xGMI on GPUs, gloo on CPU).
"""
import os

import numpy as np

RECORD_BYTES = 64


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_frames(n_frames, rank, world):
    """Contiguous block [lo, hi) of a batch owned by `rank` (configs C5-style sharding)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def stream_of_rank(rank, n_streams, world):
    """Streams owned by `rank` when n_streams virtual cameras are spread over `world` GPUs (C4: 1 stream per GPU)."""
    return [s for s in range(n_streams) if s % world == rank]


def init(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def gather_records(records, world, force=False):
    """records: uint8 torch tensor [n, 64] (device tensor with nccl, CPU tensor with gloo).  Returns [world*n, 64] on every rank.
    A single rank skips the collective unless `force` (used by the one-rank RCCL rehearsal in tests/test_gpu_pose.py)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return records
    out = torch.empty((world * records.shape[0], RECORD_BYTES), dtype=torch.uint8, device=records.device)
    dist.all_gather_into_tensor(out, records.contiguous())
    return out


def records_to_numpy(t):
    """[n,64] uint8 -> structured array with the VisionMeasurement fields."""
    dt = np.dtype([("pose_x", "<f8"), ("pose_y", "<f8"), ("pose_rot", "<f8"), ("std_x", "<f8"), ("std_y", "<f8"), ("std_rot", "<f8"),
                   ("ts", "<u8"), ("camera_id", "u1"), ("tag_count", "u1"), ("reserved", "u1", 6)])
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return np.ascontiguousarray(a).view(dt).reshape(-1)
