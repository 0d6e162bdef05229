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


def shard_rows(n_frames, world):
    """Rows every rank contributes to the gather when n_frames are sharded with shard_frames(): the largest shard.  Shorter
    shards are padded with zero records (tag_count = 0 — what a frame without a pose publishes, crates/apriltags/src/lib.rs:365-376)."""
    return -(-n_frames // world)


def gather_records(records, world, force=False, rows=None):
    """records: uint8 torch tensor [n, 64] (device tensor with nccl, CPU tensor with gloo).  Returns [world*rows, 64] on every
    rank, rank r's records at [r*rows, r*rows+n_r).  `rows` (default n) is the common row count of the collective: ranks whose
    shard is shorter (shard_frames() with n_frames % world != 0) are padded with zero records; trim with trim_gathered().
    A single rank skips the collective unless `force` (used by the one-rank RCCL rehearsal in tests/test_gpu_pose.py)."""
    import torch
    import torch.distributed as dist
    n = records.shape[0]
    rows = n if rows is None else rows
    if n > rows:
        raise ValueError(f"{n} records do not fit the common row count {rows}")
    if world == 1 and not force:
        return records
    send = records.contiguous()
    if n < rows:
        send = torch.zeros((rows, RECORD_BYTES), dtype=torch.uint8, device=records.device)
        send[:n] = records
    out = torch.empty((world * rows, RECORD_BYTES), dtype=torch.uint8, device=records.device)
    dist.all_gather_into_tensor(out, send)
    return out


def trim_gathered(gathered, n_frames, world):
    """Drops the padding of gather_records(rows=shard_rows(...)): returns the n_frames records in frame order."""
    import torch
    rows = shard_rows(n_frames, world)
    parts = []
    for r in range(world):
        lo, hi = shard_frames(n_frames, r, world)
        parts.append(gathered[r * rows:r * rows + (hi - lo)])
    return torch.cat(parts) if parts else gathered[:0]


class PoseComm:
    """The C ABI's collective (ck_comm_create / ck_gather_poses: one ncclAllGather of n x 64 bytes on the handle's stream).
    The 128-byte RCCL id of rank 0 travels through the torch.distributed group that init() made, whatever its backend."""

    def __init__(self, detector, rank, world, device=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _abi as A
        from ._lib import check
        L = detector._L
        L.ck_comm_unique_id.argtypes = [C.c_void_p]
        L.ck_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.ck_comm_destroy.argtypes = [C.c_void_p]
        L.ck_comm_destroy.restype = None
        L.ck_gather_poses.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        L.ck_comm_sync.argtypes = [C.c_void_p]
        L.ck_backend.argtypes = [C.c_void_p]
        self._L, self._det, self.world, self.rank = L, detector, world, rank
        ident = np.zeros(128, np.uint8)
        if rank == 0:
            check(L.ck_comm_unique_id(ident.ctypes.data), "ck_comm_unique_id")
        if world > 1:
            t = torch.from_numpy(ident)
            if dist.get_backend() == "nccl":
                t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
            dist.broadcast(t, 0)
            ident = t.cpu().numpy().copy()
        c = C.c_void_p()
        check(L.ck_comm_create(detector._h, ident.ctypes.data, world, rank, C.byref(c)), "ck_comm_create")
        self._c = c
        self._A = A

    def gather(self, n, out_ptr=None, sync=True, rows=None):
        """All-gather of the n records of the handle's last process call.  rows (default n): the common row count of the
        collective, the same on every rank; a rank whose shard is shorter (n < rows) sends empty records behind its own — the
        library pads, the send buffer being the handle's.  out_ptr: device or host address of world*rows records (default: a
        fresh host array, returned as [world*rows, 64] uint8)."""
        from ._lib import check
        rows = n if rows is None else rows
        if out_ptr is not None:
            check(self._L.ck_gather_poses(self._det._h, self._c, n, rows, out_ptr, 1 if sync else 0), "ck_gather_poses")
            return None
        buf = np.zeros((self.world * rows, RECORD_BYTES), np.uint8)
        check(self._L.ck_gather_poses(self._det._h, self._c, n, rows, buf.ctypes.data, 1), "ck_gather_poses")
        return buf

    def sync(self):
        from ._lib import check
        check(self._L.ck_comm_sync(self._c), "ck_comm_sync")

    def library(self):
        """Path of the librccl this communicator's calls resolved to (ck_comm_library: dladdr of ncclAllGather)."""
        import ctypes as C
        self._L.ck_comm_library.restype = C.c_char_p
        self._L.ck_comm_library.argtypes = [C.c_void_p]
        return (self._L.ck_comm_library(self._c) or b"").decode()

    def close(self):
        if getattr(self, "_c", None):
            self._L.ck_comm_destroy(self._c)
            self._c = None


def records_to_numpy(t):
    """[n,64] uint8 -> structured array with the VisionMeasurement fields."""
    dt = np.dtype([("pose_x", "<f8"), ("pose_y", "<f8"), ("pose_rot", "<f8"), ("std_x", "<f8"), ("std_y", "<f8"), ("std_rot", "<f8"),
                   ("ts", "<u8"), ("camera_id", "u1"), ("tag_count", "u1"), ("reserved", "u1", 6)])
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return np.ascontiguousarray(a).view(dt).reshape(-1)
