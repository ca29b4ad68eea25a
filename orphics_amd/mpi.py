"""Task distribution + communicator adapters (mirrors orphics/mpi.py).

The reference distributes Monte-Carlo tasks over MPI ranks (mpi.py:78-102) and
reduces with mpi4py.  Here one process drives one GPU and the communicator is
``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU box, "gloo"
in CPU tests).  :class:`TorchComm` exposes the small mpi4py-like surface that
:mod:`orphics_amd.stats` needs, so the statistics containers keep the
reference's semantics unchanged.
"""
import contextlib
import os

import numpy as np


class fakeMpiComm(object):
    """Single-process stand-in with the communicator surface the containers use (role of mpi.py:41-57)."""
    rank, size = 0, 1

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def Barrier(self):
        return None

    def Abort(self, errorcode=1):
        return None

    def allgatherv(self, x):
        return x


def mpi_distribute(num_tasks, avail_cores, allow_empty=False):
    """Split tasks 0..num_tasks-1 into ``avail_cores`` contiguous blocks whose sizes differ by at most one, the
    LARGER blocks on the LAST ranks so that rank 0 (which usually also collects) never gets an extra job -- the
    split rule of mpi.py:78-91.  Returns (block sizes, list of task lists)."""
    if not allow_empty and avail_cores > num_tasks:
        raise AssertionError("more ranks (%d) than tasks (%d); pass allow_empty=True to leave ranks idle" % (avail_cores, num_tasks))
    base, extra = divmod(int(num_tasks), int(avail_cores))
    sizes = np.full(avail_cores, base, dtype=int)
    if extra:
        sizes[avail_cores - extra:] += 1
    ends = np.cumsum(sizes)
    blocks = [list(range(int(e - n), int(e))) for n, e in zip(sizes, ends)]
    return sizes, blocks


class TorchComm(object):
    """mpi4py-flavoured facade over an initialised torch.distributed group.

    Buffers handed to Allreduce/Send/Recv are NumPy arrays (host) or torch
    tensors; with the nccl (RCCL) backend host arrays are staged through the
    current GPU, with gloo they go as CPU tensors."""

    SUM = "sum"
    IN_PLACE = "in_place"

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def Barrier(self):
        self.dist.barrier(group=self.group)

    def Abort(self, code=1):
        """Tear the job down: a rank that failed must not leave the others waiting in the next collective.
        The process group is abandoned (not destroyed: destroy would itself wait for the peers) and the process
        exits non-zero; torchrun / bench.py's launcher then terminates the remaining ranks."""
        import sys
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(int(code) if code else 1)

    def _stage(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr)) if isinstance(arr, np.ndarray) else arr
        if self.backend == "nccl" and not t.is_cuda:
            t = t.cuda()
        return t

    def Allreduce(self, sendbuf, recvbuf, op=None):
        """In-place SUM all-reduce (Statistics.allreduce, stats.py:1215-1228)."""
        import torch
        buf = recvbuf if (sendbuf is self.IN_PLACE or sendbuf is None) else sendbuf
        if isinstance(buf, torch.Tensor):
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM, group=self.group)
            if buf is not recvbuf and recvbuf is not None:
                recvbuf.copy_(buf)
            return
        t = self._stage(np.array(buf, copy=True))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        out = t.cpu().numpy()
        if isinstance(recvbuf, np.ndarray) and recvbuf.shape == out.shape and recvbuf.ndim > 0:
            recvbuf[...] = out
        return out

    def allreduce_array(self, arr):
        """Functional SUM all-reduce of a NumPy array (returns the reduced copy)."""
        t = self._stage(np.array(arr, copy=True))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy().reshape(np.shape(arr))

    def allgather(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def gather_arrays(self, arr, root=0):
        """Variable-length gather of float64 arrays to ``root`` (legacy
        Stats.get_stats Send/Recv pattern, stats.py:693-735)."""
        out = [None] * self.size if self.rank == root else None
        self.dist.gather_object(np.asarray(arr, dtype=np.float64), out, dst=root, group=self.group)
        return out


_WORLD = None


def get_world():
    """COMM_WORLD equivalent: TorchComm if torch.distributed is initialised
    (or can be from the torchrun env), else the fake single-rank comm."""
    global _WORLD
    if _WORLD is not None:
        return _WORLD
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            _WORLD = TorchComm()
            return _WORLD
    except Exception:
        pass
    return fakeMpiComm()


@contextlib.contextmanager
def mpi_abort_on_exception(comm):
    """``with mpi_abort_on_exception(comm): <Monte-Carlo loop>`` -- role of mpi.py:31-39: an exception on ANY rank
    prints its traceback (rank 0 prints on behalf of a single-rank job too) and aborts the whole job with a
    non-zero exit code instead of dead-locking the other ranks in their next collective."""
    try:
        yield
    except Exception as e:      # noqa: BLE001 -- the point is to catch everything a user loop can raise
        import sys
        import traceback
        sys.stderr.write("rank %d: %s: %s\n" % (comm.Get_rank(), type(e).__name__, e))
        traceback.print_exc()
        comm.Abort(1)
        raise                   # fake communicators return from Abort: propagate instead of swallowing


def distribute(njobs, verbose=True, comm=None, **kwargs):
    """mpi.py:95-102."""
    comm = get_world() if comm is None else comm
    rank = comm.Get_rank()
    numcores = comm.Get_size()
    num_each, each_tasks = mpi_distribute(njobs, numcores, **kwargs)
    if rank == 0 and verbose:
        print("At most ", max(num_each), " tasks...")
    return comm, rank, each_tasks[rank]
