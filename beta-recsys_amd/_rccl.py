"""What the C step drivers need from RCCL, bound with ctypes to the librccl that PyTorch already loaded: a
communicator of their own over a torch.distributed process group (the unique id travels through one broadcast of
that group) and the ADDRESSES of ncclAllReduce -- which ``hiprec_mf_bpr_dp_epoch_fused_range`` calls from C between
the step launches -- and of ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd, with which
``hiprec_shard_planned_steps`` posts the row-sharded step's two exchanges (libhiprec itself does not link RCCL).

Plumbing, like torch.distributed: nothing here computes.  Every failure (library not found, a symbol missing, an
init that does not return ncclSuccess) makes :func:`create_communicator` return None on that rank; the caller then
agrees with its peers -- over the torch process group -- whether everybody got one."""
import ctypes
import glob
import os

import torch

from . import _dist as dist

NCCL_UNIQUE_ID_BYTES = 128


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _load():
    """librccl.so as torch ships it (falls back to the ROCm one); None if neither loads."""
    global _lib
    if _lib is not None:
        return _lib or None
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
    cands += ["/opt/rocm/lib/librccl.so", "librccl.so"]
    for path in cands:
        try:
            lib = ctypes.CDLL(path)
            lib.ncclGetUniqueId.restype = ctypes.c_int
            lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
            lib.ncclCommInitRank.restype = ctypes.c_int
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
            lib.ncclCommDestroy.restype = ctypes.c_int
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            lib.ncclAllReduce.restype = ctypes.c_int
            lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            # point-to-point entry points of the row-sharded planned step (optional: older builds may lack them)
            for name in ("ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd"):
                if hasattr(lib, name):
                    getattr(lib, name).restype = ctypes.c_int
            # the watchdog's two calls (optional as well)
            if hasattr(lib, "ncclCommGetAsyncError"):
                lib.ncclCommGetAsyncError.restype = ctypes.c_int
                lib.ncclCommGetAsyncError.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
            if hasattr(lib, "ncclCommAbort"):
                lib.ncclCommAbort.restype = ctypes.c_int
                lib.ncclCommAbort.argtypes = [ctypes.c_void_p]
            _lib = lib
            return lib
        except (OSError, AttributeError):
            continue
    _lib = False
    return None


class Communicator:
    """An RCCL communicator + the addresses of the collectives the C step drivers call."""

    def __init__(self, lib, comm, world, rank):
        self._lib, self.comm, self.world, self.rank = lib, comm, world, rank
        self.all_reduce_fn = ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p).value
        addr = lambda name: ctypes.cast(getattr(lib, name), ctypes.c_void_p).value if hasattr(lib, name) else None  # noqa: E731
        self.send_fn, self.recv_fn = addr("ncclSend"), addr("ncclRecv")
        self.group_start_fn, self.group_end_fn = addr("ncclGroupStart"), addr("ncclGroupEnd")

    def has_send_recv(self):
        """True when the grouped send / recv the row-sharded step driver calls from C are all there."""
        return all((self.send_fn, self.recv_fn, self.group_start_fn, self.group_end_fn))

    def all_reduce_sum_(self, tensor):
        """In-place fp32 sum of a contiguous device tensor over the communicator, enqueued on the current stream by a
        direct ncclAllReduce call (no torch.distributed host path: ~2 us instead of 10-14 per call)."""
        rc = self._lib.ncclAllReduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), 7, 0, self.comm,
                                     torch.cuda.current_stream(tensor.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed with code {rc}")

    def async_error(self):
        """ncclCommGetAsyncError: 0 (ncclSuccess) while the communicator is healthy, None when this librccl does not
        export it.  ncclInProgress (7) is not an error."""
        if not self.comm or not hasattr(self._lib, "ncclCommGetAsyncError"):
            return None
        code = ctypes.c_int(0)
        rc = self._lib.ncclCommGetAsyncError(self.comm, ctypes.byref(code))
        return rc if rc != 0 else code.value

    def wait(self, stream, timeout_s=300.0, what="the enqueued exchanges"):
        """Host-side wait for `stream` that cannot hang for ever: the steps' exchanges are posted from C (grouped
        ncclSend / ncclRecv) long before anybody reads a result, and a peer that posts a different size -- or none --
        leaves the matching kernel spinning.  Polls the stream, asks the communicator for asynchronous errors, and after
        `timeout_s` aborts the communicator (ncclCommAbort releases the spinning kernels) and raises instead of
        blocking in a synchronize (the loopback tests time out, real RCCL would have hung)."""
        import time

        t0 = time.monotonic()
        pause = 1e-5
        while not stream.query():
            err = self.async_error()
            if err not in (None, 0, 7):
                self.abort()
                raise RuntimeError(f"RCCL reported asynchronous error {err} while waiting for {what}")
            if time.monotonic() - t0 > timeout_s:
                self.abort()
                raise RuntimeError(f"{what} did not complete within {timeout_s:.0f} s on rank {self.rank} of {self.world} "
                                   "(a peer posted a different exchange, or none): the communicator was aborted")
            time.sleep(pause)
            pause = min(pause * 2, 1e-3)

    def abort(self):
        if self.comm and hasattr(self._lib, "ncclCommAbort"):
            self._lib.ncclCommAbort(self.comm)
            self.comm = None

    def destroy(self):
        if self.comm:
            self._lib.ncclCommDestroy(self.comm)
            self.comm = None


def create_communicator(group, device):
    """Collective over ``group`` (every rank calls it, with its own cuda ``device`` current): a Communicator, or
    None where it could not be made.  Use :func:`all_ranks_agree` before relying on it.  A group that carries its
    own collectives (``_dist.own``: the single-GPU loopback world of the tests) makes its own communicator -- the
    same surface, the addresses of ITS send / recv / group / all-reduce entry points."""
    c = dist.own(group)
    if c is not None:
        return c.create_communicator(device)
    lib = _load()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    uid = _UniqueId()
    ok = lib is not None
    if ok and rank == 0:
        ok = lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    # the id (and rank 0's verdict) travel through one broadcast of the group; every rank takes part whatever its state
    payload = torch.zeros(NCCL_UNIQUE_ID_BYTES + 1, dtype=torch.uint8)
    if rank == 0:
        payload[:NCCL_UNIQUE_ID_BYTES] = torch.frombuffer(bytearray(bytes(uid.internal)), dtype=torch.uint8)
        payload[NCCL_UNIQUE_ID_BYTES] = 1 if ok else 0
    backend = dist.get_backend(group)
    wire = payload.to(device) if backend == "nccl" else payload
    dist.broadcast(wire, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    payload = wire.cpu()
    if not ok or int(payload[NCCL_UNIQUE_ID_BYTES]) != 1:
        return None
    ctypes.memmove(uid.internal, bytes(payload[:NCCL_UNIQUE_ID_BYTES].tolist()), NCCL_UNIQUE_ID_BYTES)
    comm = ctypes.c_void_p()
    try:
        with torch.cuda.device(device):
            rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
            if rc != 0 or not comm.value:
                return None
            # one small sum through the new communicator, the way the C driver will call it (float32 = 7, sum = 0,
            # in place, on the current stream): every rank contributes rank + 1
            probe = torch.full((8,), float(rank + 1), dtype=torch.float32, device=device)
            rc = lib.ncclAllReduce(probe.data_ptr(), probe.data_ptr(), probe.numel(), 7, 0, comm,
                                   torch.cuda.current_stream(device).cuda_stream)
            torch.cuda.current_stream(device).synchronize()
            if rc != 0 or not bool((probe == world * (world + 1) / 2).all()):
                lib.ncclCommDestroy(comm)
                return None
    except Exception:  # noqa: BLE001  (a binding that does not fit this librccl: keep the torch.distributed loop)
        return None
    return Communicator(lib, comm, world, rank)


def all_ranks_agree(flag, group, device):
    """True iff ``flag`` is true on every rank of the group (one small all-reduce of the torch group)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.cpu()[0]) == 1)
