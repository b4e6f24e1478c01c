"""RCCL communicator driven directly through its C API (ctypes on the librccl.so that torch already has in the process),
so that the training step's single collective — the all-reduce(sum) of the flat fp32 gradient buffer —
is ENQUEUED ON THE LIBRARY'S OWN LAUNCH STREAM, between the gradient-graph replay and the fused RMSProp kernel.

Why not ``torch.distributed.all_reduce``: ProcessGroupNCCL runs collectives on its own internal stream and joins it to the
caller's stream with events; a second active hardware queue makes every node of the replayed graphs ~1 us slower on this
stack, and the event round trips sit on the critical path of a ~12 ms step.  One stream = one
queue = kernel order is the only synchronisation.

The reference is single-device (scripts/experiment.py:68,72); this file has no counterpart there.  Rendezvous: rank 0
draws the ``ncclUniqueId`` and the existing ``torch.distributed`` process group (any backend, gloo included) carries its
128 bytes to the other ranks — bootstrap only, nothing on the data path.
"""
from __future__ import annotations

import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128
NCCL_FLOAT32 = 7   # ncclDataType_t: ncclFloat32 / ncclFloat
NCCL_SUM = 0       # ncclRedOp_t


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def lib():
    """librccl.so as bundled with torch (already mapped by libtorch_hip; a second copy from /opt/rocm would bring a
    second HIP runtime into the process)."""
    global _lib
    if _lib is None:
        import torch
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"]
        err = None
        for p in cands:
            try:
                l = C.CDLL(p)
                break
            except OSError as e:  # noqa: PERF203
                err = e
        else:
            raise ImportError("librccl.so not found ({})".format(err))
        l.ncclGetErrorString.restype = C.c_char_p
        l.ncclGetErrorString.argtypes = [C.c_int]
        l.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        l.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        l.ncclCommDestroy.argtypes = [C.c_void_p]
        l.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.ncclCommUserRank.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.ncclGetVersion.argtypes = [C.POINTER(C.c_int)]
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("{} failed: {}".format(what, lib().ncclGetErrorString(rc).decode()))


def get_unique_id():
    uid = _UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES)


class RcclComm(object):
    """One rank of an RCCL communicator bound to ``device``; collectives take the HIP stream to run on."""

    def __init__(self, rank, world, unique_id, device):
        import torch
        self.rank, self.world = int(rank), int(world)
        self.device = torch.device(device)
        uid = _UniqueId()
        C.memmove(C.byref(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        n = C.c_int()
        _check(lib().ncclCommCount(self._comm, C.byref(n)), "ncclCommCount")
        self.n_ranks = int(n.value)
        assert self.n_ranks == self.world

    @classmethod
    def from_process_group(cls, device):
        """Builds the communicator over the ranks of the default ``torch.distributed`` group (bootstrap only)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
            box = [get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        else:
            rank, world, uid = 0, 1, get_unique_id()
        return cls(rank, world, uid, device)

    def all_reduce_sum_(self, tensor, stream):
        """In-place sum over the ranks of a contiguous fp32 device tensor, enqueued on ``stream`` (a raw hipStream_t /
        ``torch.cuda.Stream``); returns immediately."""
        import torch
        assert tensor.is_cuda and tensor.dtype == torch.float32 and tensor.is_contiguous()
        s = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        with torch.cuda.device(self.device):
            _check(lib().ncclAllReduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), NCCL_FLOAT32, NCCL_SUM,
                                       self._comm, C.c_void_p(s)), "ncclAllReduce")
        return tensor

    def version(self):
        v = C.c_int()
        _check(lib().ncclGetVersion(C.byref(v)), "ncclGetVersion")
        return int(v.value)

    def destroy(self):
        if self._comm:
            lib().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
