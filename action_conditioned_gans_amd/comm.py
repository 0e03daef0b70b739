"""Gradient all-reduce transport: this process's own RCCL communicator (GPU) or a gloo process group (CPU tests).

The data-parallel step needs exactly one collective - a sum all-reduce of contiguous float32 ranges of the flat gradient
buffers - plus a few tiny ones in the optional exact-global-batch mode.  On the GPU they are issued as
plain ``ncclAllReduce`` calls on a ``hipStream_t`` of the caller's choice through ctypes on the librccl.so that PyTorch
ships (the one HIP runtime of the process): stream-ordered, capturable into the step's HIP graph, no helper threads,
no events, and an ordinary ``ncclCommDestroy`` at the end.  ``torch.distributed`` is used for the *bootstrap* only
(the 128-byte unique id travels through a gloo group or a TCPStore) - never for device traffic, so no
ProcessGroupNCCL (and no watchdog thread polling events beside a stream capture) exists in the process.
"""
import contextlib
import ctypes
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128
NCCL_SUM, NCCL_MAX = 0, 2
NCCL_INT32, NCCL_FLOAT32, NCCL_FLOAT64, NCCL_BFLOAT16 = 2, 7, 8, 9
_DTYPES = {torch.float32: NCCL_FLOAT32, torch.float64: NCCL_FLOAT64, torch.int32: NCCL_INT32, torch.bfloat16: NCCL_BFLOAT16}


class CommError(RuntimeError):
    pass


class _UniqueId(ctypes.Structure):
    # c_ubyte, not c_char: ctypes reads a c_char array field as a C string and stops at the first NUL, and a real
    # ncclUniqueId has NULs from byte 9 on (magic, then a sockaddr: AF_INET = 02 00).
    _fields_ = [('internal', ctypes.c_ubyte * NCCL_UNIQUE_ID_BYTES)]


def pack_unique_id(uid):
    """All 128 bytes of the struct, NULs included."""
    raw = ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)
    assert len(raw) == NCCL_UNIQUE_ID_BYTES
    return raw


def unpack_unique_id(raw, uid=None):
    """The struct from what pack_unique_id produced on rank 0; anything but exactly 128 bytes is a bootstrap error."""
    if not isinstance(raw, (bytes, bytearray)) or len(raw) != NCCL_UNIQUE_ID_BYTES:
        raise CommError('unique-id bootstrap delivered %s, expected %d bytes'
                        % ('%d bytes' % len(raw) if isinstance(raw, (bytes, bytearray)) else type(raw).__name__,
                           NCCL_UNIQUE_ID_BYTES))
    uid = uid if uid is not None else _UniqueId()
    ctypes.memmove(ctypes.byref(uid), bytes(raw), NCCL_UNIQUE_ID_BYTES)
    return uid


_RCCL = None


def load_rccl(path):
    """A library exporting the seven nccl* entry points this module calls, with their signatures declared.  The product
    loads the librccl.so PyTorch ships (_rccl below); tests load a stub that fails on demand (tests/stub_rccl.c)."""
    if not os.path.exists(path):
        raise CommError('librccl.so not found (%s): no multi-GPU transport' % path)
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    P = ctypes.c_void_p
    lib.ncclGetUniqueId.argtypes, lib.ncclGetUniqueId.restype = [ctypes.POINTER(_UniqueId)], ctypes.c_int
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(P), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclCommInitRank.restype = ctypes.c_int
    lib.ncclAllReduce.argtypes = [P, P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, P, P]
    lib.ncclAllReduce.restype = ctypes.c_int
    lib.ncclCommDestroy.argtypes, lib.ncclCommDestroy.restype = [P], ctypes.c_int
    lib.ncclGetErrorString.argtypes, lib.ncclGetErrorString.restype = [ctypes.c_int], ctypes.c_char_p
    lib.ncclGetVersion.argtypes, lib.ncclGetVersion.restype = [ctypes.POINTER(ctypes.c_int)], ctypes.c_int
    for q in ('ncclCommCount', 'ncclCommUserRank'):       # optional (reporting only: RcclCommunicator.reported)
        if hasattr(lib, q):
            getattr(lib, q).argtypes, getattr(lib, q).restype = [P, ctypes.POINTER(ctypes.c_int)], ctypes.c_int
    return lib


def _rccl():
    """librccl.so next to torch's libamdhip64.so: the process keeps ONE HIP runtime (see _lib.py on import order)."""
    global _RCCL
    if _RCCL is None:
        _RCCL = load_rccl(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))
    return _RCCL


@contextlib.contextmanager
def _stdout_to_stderr():
    """File descriptor 1 points at stderr for the duration: librccl prints a version banner ("RCCL version : ...", five lines)
    on STDOUT from inside ncclCommInitRank - in front of whatever the program itself reports there (bench.py's one JSON line)."""
    import sys
    try:
        sys.stdout.flush()
        saved = os.dup(1)
    except (OSError, ValueError, AttributeError):
        yield
        return
    try:
        os.dup2(2, 1)
        yield
    finally:
        try:
            libc = ctypes.CDLL(None)
            libc.fflush(None)              # whatever the C side still holds goes where fd 1 points NOW
        except OSError:
            pass
        os.dup2(saved, 1)
        os.close(saved)


def _check(rc, what, lib=None):
    if rc != 0:
        raise CommError('%s failed: %s (ncclResult %d)' % (what, (lib or _rccl()).ncclGetErrorString(rc).decode(), rc))


def _share_unique_id(uid_bytes, world_size, rank, process_group):
    """Rank 0's unique id to every rank, over CPU only: an initialised non-NCCL process group, else a TCPStore at
    MASTER_ADDR:MASTER_PORT (under torch.distributed.run the agent already serves one there)."""
    import torch.distributed as dist
    if process_group is not None or (dist.is_available() and dist.is_initialized()):
        if dist.get_backend(process_group) == 'nccl':
            raise CommError('bootstrap needs a CPU process group (gloo): ProcessGroupNCCL is not used by this package')
        box = [uid_bytes if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=process_group)
        return box[0]
    addr, port = os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')
    if not addr or not port:
        raise CommError('multi-rank communicator needs a gloo process group or MASTER_ADDR / MASTER_PORT')
    agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '').lower() == 'true'
    store = dist.TCPStore(addr, int(port), world_size, is_master=(rank == 0 and not agent_store))
    key = 'acg/rccl_unique_id/%s' % os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')
    if rank == 0:
        store.set(key, uid_bytes)
        import time
        t0 = time.time()
        while store.add(key + '/ack', 0) < world_size - 1:      # the server lives in THIS process: stay until every rank has read
            if time.time() - t0 > 600:
                raise CommError('unique-id bootstrap: %d of %d ranks did not fetch the id within 600 s'
                                % (world_size - 1 - store.add(key + '/ack', 0), world_size - 1))
            time.sleep(0.01)
        return uid_bytes
    got = bytes(store.get(key))
    store.add(key + '/ack', 1)
    return got


class Communicator:
    """``all_reduce(tensor, op, stream)``: in place, asynchronous, ordered on ``stream`` (None = the current one)."""
    world_size, rank = 1, 0
    capturable = False

    def all_reduce(self, tensor, op='sum', stream=None):
        raise NotImplementedError

    def destroy(self):
        pass


class RcclCommunicator(Communicator):
    capturable = True      # ncclAllReduce on a capturing stream becomes graph nodes (validated: tests/dp_one_rank.py)

    def __init__(self, device, world_size=1, rank=0, process_group=None, lib=None):
        """``lib``: tests only - a stand-in for librccl.so (load_rccl); with it the device may be the CPU, so that the error
        paths of the bootstrap and of the collectives run without a GPU."""
        device = torch.device(device)
        if lib is None and (device.type != 'cuda' or not torch.cuda.is_available()):
            raise CommError('RcclCommunicator needs a GPU device, got %s' % device)
        self.device, self.world_size, self.rank = device, int(world_size), int(rank)
        self._comm = None
        lib = lib if lib is not None else _rccl()
        self._lib = lib
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId', lib)
        if self.world_size > 1:
            raw = _share_unique_id(pack_unique_id(uid) if self.rank == 0 else None, self.world_size, self.rank, process_group)
            unpack_unique_id(raw, uid)
        comm = ctypes.c_void_p()
        import contextlib
        with (torch.cuda.device(device) if device.type == 'cuda' else contextlib.nullcontext()), _stdout_to_stderr():
            _check(lib.ncclCommInitRank(ctypes.byref(comm), self.world_size, uid, self.rank), 'ncclCommInitRank', lib)
        self._comm = comm
        self.calls = 0

    def reported(self):
        """(ranks in the communicator, this rank) as RCCL itself answers - ncclCommCount / ncclCommUserRank - or None where the
        library does not export them: what bench.py prints so that a multi-GPU run can be checked against RCCL having seen N ranks."""
        if self._comm is None or not hasattr(self._lib, 'ncclCommCount') or not hasattr(self._lib, 'ncclCommUserRank'):
            return None
        n, r = ctypes.c_int(-1), ctypes.c_int(-1)
        _check(self._lib.ncclCommCount(self._comm, ctypes.byref(n)), 'ncclCommCount', self._lib)
        _check(self._lib.ncclCommUserRank(self._comm, ctypes.byref(r)), 'ncclCommUserRank', self._lib)
        return n.value, r.value

    def all_reduce_ptr(self, ptr, count, nccl_dtype, nccl_op, stream_ptr):
        """The raw form the launch lists use: everything precomputed, one C call."""
        rc = self._lib.ncclAllReduce(ptr, ptr, count, nccl_dtype, nccl_op, self._comm, stream_ptr)
        if rc != 0:
            _check(rc, 'ncclAllReduce', self._lib)
        self.calls += 1

    def all_reduce(self, tensor, op='sum', stream=None):
        on_gpu = self.device.type == 'cuda'
        if not ((tensor.is_cuda or not on_gpu) and tensor.is_contiguous()):
            raise CommError('all_reduce: contiguous GPU tensor expected')
        sp = stream if (stream is not None or not on_gpu) else ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self.all_reduce_ptr(ctypes.c_void_p(tensor.data_ptr()), tensor.numel(), _DTYPES[tensor.dtype],
                            NCCL_MAX if op == 'max' else NCCL_SUM, sp)

    def destroy(self):
        if self._comm is not None and self._comm.value:
            if self.device.type == 'cuda':
                torch.cuda.synchronize(self.device)    # nothing of ours may still be queued on the communicator
            comm, self._comm = self._comm, None        # a failing destroy is not retried by a second close()
            _check(self._lib.ncclCommDestroy(comm), 'ncclCommDestroy', self._lib)


class ProcessGroupCommunicator(Communicator):
    """CPU stand-in (world_size-2 gloo tests): the same interface over torch.distributed, synchronous."""

    def __init__(self, process_group=None, world_size=None, rank=None):
        import torch.distributed as dist
        self._dist, self.group = dist, process_group
        self.world_size = world_size if world_size is not None else dist.get_world_size(process_group)
        self.rank = rank if rank is not None else dist.get_rank(process_group)

    def all_reduce(self, tensor, op='sum', stream=None):
        d = self._dist
        d.all_reduce(tensor, op=d.ReduceOp.MAX if op == 'max' else d.ReduceOp.SUM, group=self.group)


def create(device, world_size=1, rank=0, process_group=None):
    """The transport for ``device``: RCCL on a GPU, the given (gloo) process group on the CPU."""
    device = torch.device(device)
    if device.type == 'cuda':
        return RcclCommunicator(device, world_size, rank, process_group)
    return ProcessGroupCommunicator(process_group, world_size, rank)
