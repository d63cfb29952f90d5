"""TEST INFRASTRUCTURE: R ranks as R host threads on ONE GPU.

The multi-GPU engines talk to their peers through two seams: ``beta_recsys_amd._dist`` (the torch.distributed calls
of the host code) and the RCCL function pointers their C step drivers are handed (``hiprec_nccl_fns``, the
``ncclAllReduce`` pointer of the data-parallel epoch).  ``VirtualWorld`` fills both with loopback implementations
-- python collectives between threads for the first, ``tests/native/loopback_rccl.hip`` for the second -- so that
the world-size > 1 branches, including the exchanges the C drivers post themselves, run on the single-GPU box:

    world = VirtualWorld(4)
    results = world.run(lambda group: train(ShardedMFEngine(cfg, process_group=group)))   # one call per rank

Every rank thread has its own stream (current inside ``run``), its own engine and shard; a rank that raises aborts
the world (its peers return from their rendezvous with an error instead of waiting), and ``run`` re-raises the first
exception.  A recv whose size differs from the matching send, or a peer that never posts, fails after
``timeout`` seconds: a wrong plan makes a test FAIL, not hang."""
import ctypes
import os
import sys
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "native", "libloopback_rccl.so")
_lib = None


def native():
    """tests/native/libloopback_rccl.so (built by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(LIB_PATH)
        lib.loopback_world_create.restype = ctypes.c_void_p
        lib.loopback_world_create.argtypes = [ctypes.c_int]
        lib.loopback_world_destroy.argtypes = [ctypes.c_void_p]
        lib.loopback_comm_create.restype = ctypes.c_void_p
        lib.loopback_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.loopback_comm_destroy.argtypes = [ctypes.c_void_p]
        lib.loopback_set_timeout_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.loopback_abort.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        lib.loopback_failed.restype = ctypes.c_int
        lib.loopback_failed.argtypes = [ctypes.c_void_p]
        lib.loopback_last_error.restype = ctypes.c_char_p
        lib.loopback_last_error.argtypes = [ctypes.c_void_p]
        lib.loopback_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        lib.loopback_all_reduce.restype = ctypes.c_int
        lib.loopback_all_reduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


EXPORTS = ("loopback_world_create", "loopback_world_destroy", "loopback_comm_create", "loopback_comm_destroy",
           "loopback_set_timeout_ms", "loopback_abort", "loopback_failed", "loopback_last_error", "loopback_counters",
           "loopback_group_start", "loopback_group_end", "loopback_send", "loopback_recv", "loopback_all_reduce")


class LoopbackCommunicator:
    """What ``_rccl.Communicator`` is to librccl, for the loopback library: a communicator handle + the ADDRESSES of
    the entry points the C step drivers call."""

    def __init__(self, world, rank):
        lib = native()
        self._lib, self._world = lib, world
        self.world, self.rank = world.size, rank
        self.comm = ctypes.c_void_p(lib.loopback_comm_create(world.handle, rank))
        addr = lambda name: ctypes.cast(getattr(lib, name), ctypes.c_void_p).value  # noqa: E731
        self.all_reduce_fn = addr("loopback_all_reduce")
        self.send_fn, self.recv_fn = addr("loopback_send"), addr("loopback_recv")
        self.group_start_fn, self.group_end_fn = addr("loopback_group_start"), addr("loopback_group_end")

    def has_send_recv(self):
        return True

    def all_reduce_sum_(self, tensor):
        rc = self._lib.loopback_all_reduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), 7, 0, self.comm,
                                           torch.cuda.current_stream(tensor.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"loopback all-reduce failed with code {rc}: {self._world.last_error()}")

    def destroy(self):
        if self.comm:
            self._lib.loopback_comm_destroy(self.comm)
            self.comm = None


class _Shared:
    """What the ranks of one group share: a slot per rank and a barrier."""

    def __init__(self, size):
        self.slots = [None] * size
        self.barrier = threading.Barrier(size)
        self.children = []          # new_group(): the k-th call of every rank gets the same child
        self.lock = threading.Lock()


class LoopbackGroup:
    """One rank's end of a group of the virtual world.  ``hiprec_collectives`` makes ``beta_recsys_amd._dist`` route
    the engines' collectives here instead of torch.distributed."""

    def __init__(self, world, rank, shared):
        self._world, self._rank, self._sh = world, rank, shared
        self._n_children = 0
        self.hiprec_collectives = self

    # ---- identity -------------------------------------------------------------------------------------------
    def size(self):
        return self._world.size

    def rank(self):
        return self._rank

    def ranks(self):
        return list(range(self._world.size))

    def backend(self):
        # device tensors, direct communicators: what RCCL groups are to the engines (a CPU world: what gloo is)
        return "nccl" if self._world.device.type == "cuda" else "gloo"

    def new_group(self):
        sh, k = self._sh, self._n_children
        self._n_children += 1
        with sh.lock:
            while len(sh.children) <= k:
                sh.children.append(_Shared(self._world.size))
        return LoopbackGroup(self._world, self._rank, sh.children[k])

    def create_communicator(self, device):
        return LoopbackCommunicator(self._world, self._rank)

    # ---- collectives: deposit, meet, read the peers' tensors, meet again ---------------------------------------
    def _sync(self, *tensors):
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
                return

    def _wait(self):
        try:
            self._sh.barrier.wait(self._world.timeout)
        except threading.BrokenBarrierError:
            raise RuntimeError("loopback world: a peer left the collective (aborted or timed out)") from None

    def _meet(self, item, *tensors):
        self._sync(*tensors)
        self._sh.slots[self._rank] = item
        self._wait()
        return list(self._sh.slots)

    def _leave(self, *tensors):
        self._sync(*tensors)
        self._wait()

    def all_reduce(self, tensor, op=None):
        name = str(op).split(".")[-1].upper() if op is not None else "SUM"
        peers = self._meet(tensor, tensor)
        stack = torch.stack([p.to(tensor.device) for p in peers])
        if name == "SUM":
            out = stack.sum(0)
        elif name == "MAX":
            out = stack.max(0).values
        elif name == "MIN":
            out = stack.min(0).values
        else:
            raise NotImplementedError(f"loopback all_reduce: {op}")
        self._leave(out)          # everybody has read every input
        tensor.copy_(out.to(tensor.dtype))
        self._leave(tensor)

    def all_to_all_single(self, output, input, output_split_sizes=None, input_split_sizes=None):  # noqa: A002
        R, me = self._world.size, self._rank
        if input_split_sizes is None:
            assert input.shape[0] % R == 0, "all_to_all_single without split sizes needs dim 0 divisible by the world"
            input_split_sizes = [input.shape[0] // R] * R
        if output_split_sizes is None:
            assert output.shape[0] % R == 0
            output_split_sizes = [output.shape[0] // R] * R
        assert sum(input_split_sizes) == input.shape[0] and sum(output_split_sizes) == output.shape[0], \
            "all_to_all_single: split sizes do not add up to dim 0"
        peers = self._meet((input, list(input_split_sizes)), input)
        off = 0
        for q in range(R):
            src, splits = peers[q]
            a = sum(splits[:me])
            n = splits[me]
            if n != output_split_sizes[q]:
                self._world.abort(f"rank {me} expects {output_split_sizes[q]} rows from rank {q}, which sends {n}")
                raise RuntimeError(f"loopback all_to_all_single: rank {me} expects {output_split_sizes[q]} rows from "
                                   f"rank {q}, which sends {n}")
            output[off:off + n].copy_(src[a:a + n])
            off += n
        self._leave(output)

    def all_gather(self, tensor_list, tensor):
        peers = self._meet(tensor, tensor)
        for dst, src in zip(tensor_list, peers):
            dst.copy_(src)
        self._leave(tensor)

    def broadcast(self, tensor, src=0):
        peers = self._meet(tensor, tensor)
        if self._rank != src:
            tensor.copy_(peers[src])
        self._leave(tensor)


class VirtualWorld:
    """R virtual ranks on one device.  ``run(fn)`` calls ``fn(group)`` on R threads."""

    def __init__(self, size, device="cuda:0", timeout=30.0):
        self.size, self.device, self.timeout = size, torch.device(device), timeout
        self.handle = None          # device "cpu": the python collectives only (host-logic tests), no native world
        if self.device.type == "cuda":
            lib = native()
            self.handle = ctypes.c_void_p(lib.loopback_world_create(size))
            assert self.handle.value, "loopback_world_create failed"
            lib.loopback_set_timeout_ms(self.handle, int(timeout * 1000))
        self._root = _Shared(size)

    def last_error(self):
        return native().loopback_last_error(self.handle).decode() if self.handle else ""

    def failed(self):
        return bool(self.handle and native().loopback_failed(self.handle))

    def counters(self):
        """{sends, recvs, bytes, all_reduces, groups} the native entry points served so far."""
        out = (ctypes.c_int64 * 5)()
        native().loopback_counters(self.handle, out)
        return dict(zip(("sends", "recvs", "bytes", "all_reduces", "groups"), out))

    def abort(self, why="a rank raised"):
        if self.handle:
            native().loopback_abort(self.handle, why.encode())

        def walk(sh):
            sh.barrier.abort()
            for c in list(sh.children):
                walk(c)
        walk(self._root)

    def run(self, fn):
        results, errors = [None] * self.size, [None] * self.size
        saved_stdout = sys.stdout

        def body(rank):
            try:
                if self.device.type != "cuda":
                    results[rank] = fn(LoopbackGroup(self, rank, self._root))
                    return
                torch.cuda.set_device(self.device)
                stream = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(stream):
                    results[rank] = fn(LoopbackGroup(self, rank, self._root))
                    stream.synchronize()
            except BaseException as e:  # noqa: BLE001
                errors[rank] = e
                self.abort(f"rank {rank} raised {type(e).__name__}: {e}")

        threads = [threading.Thread(target=body, args=(r,), name=f"vrank{r}") for r in range(self.size)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        sys.stdout = saved_stdout    # engines redirect stdout while they build their models; threads interleave that
        first = [e for e in errors if e is not None and "a peer left the collective" not in str(e)]
        first = first or [e for e in errors if e is not None]
        if first:
            raise first[0]
        return results

    def close(self):
        if self.handle:
            torch.cuda.synchronize(self.device)
            native().loopback_world_destroy(self.handle)
            self.handle = None
