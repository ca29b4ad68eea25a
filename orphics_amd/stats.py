"""orphics.stats hot-path surface: bin2D (HIP), Stats / Statistics (host logic,
RCCL/gloo reduce through :class:`orphics_amd.mpi.TorchComm`).

Signatures mirror /root/reference/orphics/stats.py (cited per method).
"""
from collections import defaultdict
from pathlib import Path

import numpy as np

from . import mpi as _mpi


def _torch():
    import torch
    return torch


class HalfPlane(object):
    """A Fourier-space plane stored on the non-redundant half grid (ny, kpitch)
    of a real field's transform (device tensor ``t``).  ``full()`` expands by
    Hermitian (complex) / even (real) symmetry."""

    def __init__(self, t, eng):
        self.t = t
        self.eng = eng

    @property
    def is_complex(self):
        return self.t.is_complex()

    def full(self):
        lead = self.t.shape[:-2]
        if len(lead) == 0:
            return self.eng.hc_to_full(self.t) if self.is_complex else self.eng.hcreal_to_full(self.t)
        torch = _torch()
        flat = self.t.reshape((-1,) + tuple(self.t.shape[-2:]))
        outs = [self.eng.hc_to_full(f) if self.is_complex else self.eng.hcreal_to_full(f) for f in flat]
        return torch.stack(outs).reshape(tuple(lead) + (self.eng.ny, self.eng.nx))

    def numpy(self):
        return self.full().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, idx):
        return HalfPlane(self.t[idx], self.eng)

    @property
    def shape(self):
        return tuple(self.t.shape[:-2]) + (self.eng.ny, self.eng.nx)


class bin2D(object):
    """stats.py:782-811.  ``np.digitize(..., right=True)`` runs once on the GPU
    (float64 comparisons, bit-exact ids); every ``bin`` call is one streaming
    histogram kernel.  Accepted data: NumPy (Ny,Nx) -> NumPy results (drop-in);
    CUDA tensor (Ny,Nx) or :class:`HalfPlane` -> results as NumPy vectors too
    (they are nbins long)."""

    def __init__(self, modrmap, bin_edges):
        from . import engine as E
        torch = _torch()
        bin_edges = np.asarray(bin_edges, dtype=np.float64)
        if bin_edges.ndim != 1 or bin_edges.size < 2 or not np.all(np.diff(bin_edges) > 0):
            raise ValueError("bin_edges must be 1-D and strictly increasing")
        self.centers = (bin_edges[1:] + bin_edges[:-1]) / 2.
        self.cents = self.centers  # backwards compatibility
        self.bin_edges = bin_edges
        self.modrmap = modrmap
        self._dev = E.cuda_device()
        self._edges_d = torch.as_tensor(bin_edges, device=self._dev)
        if isinstance(modrmap, torch.Tensor):
            m = modrmap.to(device=self._dev, dtype=torch.float64)
        else:
            m = torch.as_tensor(np.ascontiguousarray(modrmap, dtype=np.float64), device=self._dev)
        self._shape = tuple(m.shape)
        self._ids = E.dev_digitize(m.reshape(-1), self._edges_d)
        self._nids = bin_edges.size + 1
        self._digitized = None
        self._ids_half = None
        # H3 quirk (stats.py:796-797): np.bincount has no minlength, so if nothing
        # overflows the last edge the ``[1:-1]`` slice drops a real bin.
        self._maxid = int(self._ids.max().item()) if self._ids.numel() else 0

    @property
    def digitized(self):
        if self._digitized is None:
            self._digitized = self._ids.cpu().numpy().astype(np.int64)
        return self._digitized

    def _half_ids(self, eng):
        """ids restricted to the hc grid (pad columns -> -1); requires a
        symmetric modrmap (true for any |ell| map)."""
        torch = _torch()
        if self._ids_half is None:
            ny, nx = self._shape
            if (ny, nx) != (eng.ny, eng.nx):
                raise ValueError("HalfPlane geometry does not match the binner's modrmap")
            full = self._ids.reshape(ny, nx)
            # symmetry check: id(-l) == id(l)
            flipped = torch.roll(torch.flip(full, dims=(0, 1)), shifts=(1, 1), dims=(0, 1))
            if not torch.equal(full, flipped):
                raise ValueError("modrmap is not symmetric under l -> -l; half-plane binning is invalid")
            h = torch.full((ny, eng.kp), -1, dtype=torch.int32, device=self._dev)
            h[:, :nx // 2 + 1] = full[:, :nx // 2 + 1]
            self._ids_half = h.contiguous()
        return self._ids_half

    def _slice(self, arr):
        # reference: np.bincount(...)[1:-1] on an array of length maxid+1
        return arr[1:self._maxid]

    def _raw(self, data2d, weights=None, aux=None, mode=0, skip_nan=False):
        from . import engine as E
        torch = _torch()
        if isinstance(data2d, HalfPlane):
            if weights is not None:
                raise ValueError("weights are not supported with HalfPlane data")
            ids = self._half_ids(data2d.eng)
            d = data2d.t.contiguous()
            if d.is_complex():
                raise TypeError("cannot bin complex data")
            return E.dev_bin(d, ids, self._nids, aux=aux, mode=mode, skip_nan=skip_nan,
                             herm_pitch=data2d.eng.kp, herm_nxh=data2d.eng.nxh)
        if isinstance(data2d, torch.Tensor):
            d = data2d.to(self._dev)
            if d.dtype not in (torch.float32, torch.float64):
                d = d.to(torch.float64)
        else:
            a = np.asarray(data2d)
            if np.iscomplexobj(a):
                raise TypeError("cannot bin complex data")
            d = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64 if a.dtype != np.float32 else np.float32), device=self._dev)
        d = d.contiguous()
        if d.numel() != self._ids.numel():
            raise ValueError("data2d size does not match modrmap")
        w = None
        if weights is not None:
            w = weights.to(self._dev) if isinstance(weights, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(weights), device=self._dev)
            w = w.to(d.dtype).contiguous()
        return E.dev_bin(d.reshape(-1), self._ids, self._nids, weights=None if w is None else w.reshape(-1), aux=aux,
                         mode=mode, skip_nan=skip_nan)

    def bin(self, data2d, weights=None, err=False, get_count=False, mask_nan=False):
        """stats.py:790-811.  ``err=True`` returns the standard error of the
        bin mean sqrt(sum (x-mu_b)^2/(c-1)/c); the reference shifts mu_b by one bin
        (loop index bug, stats.py:799-801) -- the intended statistic is computed."""
        torch = _torch()
        if weights is None:
            sums, counts = self._raw(data2d, skip_nan=mask_nan)
            count = self._slice(counts.cpu().numpy())
            with np.errstate(divide="ignore", invalid="ignore"):
                res = self._slice(sums.cpu().numpy()) / count
            if err:
                cfull = counts.to(torch.float64)
                mean = torch.where(cfull > 0, sums / cfull, torch.zeros_like(sums))
                ssq, _ = self._raw(data2d, aux=mean, mode=1, skip_nan=mask_nan)
                with np.errstate(divide="ignore", invalid="ignore"):
                    std = np.sqrt(self._slice(ssq.cpu().numpy()) / (count - 1) / count)
        else:
            # reference ignores mask_nan on the weighted path (stats.py:802-804)
            sums, wsums = self._raw(data2d, weights=weights)
            count = self._slice(wsums.cpu().numpy())
            with np.errstate(divide="ignore", invalid="ignore"):
                res = self._slice(sums.cpu().numpy()) / count
            if err:
                raise NotImplementedError("weights with err=True is undefined in the reference (stats.py:802-810)")
        if get_count:
            assert not err  # need to make more general (stats.py:806)
            return self.centers, res, count
        if err:
            assert not get_count
            return self.centers, res, std
        return self.centers, res


def bin_in_annuli(data2d, modrmap, bin_edges):
    """stats.py:853-855."""
    return bin2D(modrmap, bin_edges).bin(data2d)


def cov2corr(cov):
    d = np.sqrt(np.diagonal(cov))
    return cov / np.outer(d, d)


def get_stats(binned_vectors):
    """stats.py:859-898: mean, cov, covmean, err, errmean, corr."""
    arr = np.asarray(binned_vectors)
    N = arr.shape[0]
    ret = {}
    ret['mean'] = np.nanmean(arr, axis=0)
    ret['cov'] = np.cov(arr.transpose())
    ret['covmean'] = ret['cov'] / N
    if arr.shape[1] == 1:
        ret['err'] = np.sqrt(ret['cov'])
    else:
        ret['err'] = np.sqrt(np.diagonal(ret['cov']))
    ret['errmean'] = ret['err'] / np.sqrt(N)
    if arr.shape[1] == 1:
        ret['corr'] = 1.
    else:
        ret['corr'] = cov2corr(ret['cov'])
    return ret


class Stats(object):
    """Legacy container (stats.py:577-735): vectors gathered to ``root`` and
    reduced with :func:`get_stats`; stacks summed to ``root``."""

    def __init__(self, comm=None, root=0, loopover=None, tag_start=333):
        self.comm = comm if comm is not None else _mpi.fakeMpiComm()
        self.rank = self.comm.Get_rank()
        self.numcores = self.comm.Get_size()
        self.columns = {}
        self.vectors = {}
        self.little_stack = {}
        self.little_stack_count = {}
        self.tag_start = tag_start
        self.root = root
        self.loopover = list(range(root + 1, self.numcores)) if loopover is None else loopover

    def add_to_stats(self, label, vector, exclude=False):
        """stats.py:614-631."""
        assert label != 'stats', "Sorry, 'stats' is a forbidden label."
        vector = np.asarray(vector)
        if np.iscomplexobj(vector):
            print("ERROR: stats on complex arrays not supported. Do the real and imaginary parts separately.")
            raise TypeError
        if label not in self.vectors:
            self.vectors[label] = []
            self.columns[label] = vector.shape
        if not exclude:
            self.vectors[label].append(vector)

    def add_to_stack(self, label, arr, exclude=False):
        """stats.py:634-650."""
        assert label != 'stats', "Sorry, 'stats' is a forbidden label."
        if np.iscomplexobj(arr):
            print("ERROR: stacking of complex arrays not supported. Stack the real and imaginary parts separately.")
            raise TypeError
        if label not in self.little_stack:
            self.little_stack[label] = arr * 0.
            self.little_stack_count[label] = 0
        if not exclude:
            self.little_stack[label] += arr
            self.little_stack_count[label] += 1

    def get_stacks(self, verbose=True):
        """stats.py:653-691 (sum to root then divide by the total count)."""
        self.stacks = {}
        self.stack_count = {}
        for label in self.little_stack.keys():
            local = np.array(self.little_stack[label]).astype(np.float64)
            cnt = np.array([self.little_stack_count[label]], dtype=np.int64)
            if self.numcores > 1:
                local = self.comm.allreduce_array(local)
                cnt = self.comm.allreduce_array(cnt)
            if self.rank == self.root:
                self.stack_count[label] = int(cnt[0])
                self.stacks[label] = local / self.stack_count[label]

    def get_stats(self, verbose=True, skip_stats=False):
        """stats.py:693-735."""
        self.stats = {}
        for label in list(self.vectors.keys()):
            mine = np.array(self.vectors[label], dtype=np.float64).reshape((-1,) + tuple(self.columns[label]))
            if self.numcores > 1:
                parts = self.comm.gather_arrays(mine, root=self.root)
                if self.rank != self.root:
                    continue
                parts = [p for p in parts if p.shape[0] > 0]
                self.vectors[label] = np.concatenate(parts, axis=0) if parts else mine
            else:
                self.vectors[label] = mine
            if not skip_stats:
                self.stats[label] = get_stats(self.vectors[label])

    def dump(self, path):
        """stats.py:737-743."""
        for d, name in zip([self.vectors, self.stacks], ['vectors', 'stack']):
            for key in d.keys():
                np.save(f"{path}/mstats_dump_{name}_{key}.npy", d[key])
        for key in self.stats.keys():
            for skey in self.stats[key].keys():
                np.savetxt(f"{path}/mstats_dump_stats_{key}_{skey}.txt", np.atleast_1d(self.stats[key][skey]))


class Statistics(object):
    """stats.py:918-1530: one-pass (n, sum, cross) moments and stack sums with
    a SUM all-reduce.  ``comm`` is None (single process), an mpi4py
    communicator, or :class:`orphics_amd.mpi.TorchComm` (RCCL / gloo)."""

    def __init__(self, comm=None, dtype=np.float64):
        self.comm = comm
        self.dtype = np.dtype(dtype)
        self._n = defaultdict(int)
        self._sum = {}
        self._cross = {}
        self._dim_stats = {}
        self._k = defaultdict(int)
        self._stack_sum = {}
        self._shape_stack = {}
        self._N, self._SUM, self._CROSS = {}, {}, {}
        self._K, self._STACK_SUM = {}, {}
        self._reduced = False

    @property
    def mpi_enabled(self):
        return self.comm is not None

    def _ensure_stats_label(self, label, d):
        if label in self._shape_stack:
            raise ValueError(f"Label {label!r} already used in stack mode.")
        if label not in self._dim_stats:
            self._dim_stats[label] = int(d)
            self._sum[label] = np.zeros(d, dtype=self.dtype)
            self._cross[label] = np.zeros((d, d), dtype=self.dtype)
        elif self._dim_stats[label] != d:
            raise ValueError(f"Stats dim mismatch for {label!r}: {self._dim_stats[label]} vs {d}")

    def _ensure_stack_label(self, label, shape):
        if label in self._dim_stats:
            raise ValueError(f"Label {label!r} already used in stats mode.")
        if label not in self._shape_stack:
            self._shape_stack[label] = tuple(int(s) for s in shape)
            self._stack_sum[label] = np.zeros(shape, dtype=self.dtype)
        elif self._shape_stack[label] != tuple(shape):
            raise ValueError(f"Stack shape mismatch for {label!r}: {self._shape_stack[label]} vs {tuple(shape)}")

    def add(self, label, x):
        """stats.py:1068-1090."""
        x = np.asarray(x, dtype=self.dtype).ravel()
        d = x.shape[0]
        self._ensure_stats_label(label, d)
        self._n[label] += 1
        self._sum[label] += x
        self._cross[label] += np.outer(x, x)

    def extend(self, label, X):
        """stats.py:1092-1120."""
        X = np.asarray(list(X) if not hasattr(X, "shape") else X, dtype=self.dtype)
        if X.ndim == 1:
            self.add(label, X)
            return
        if X.ndim != 2:
            raise ValueError("X must be (m, d) or (d,).")
        m, d = X.shape
        self._ensure_stats_label(label, d)
        self._n[label] += m
        self._sum[label] += X.sum(axis=0)
        self._cross[label] += X.T @ X

    def add_moments(self, label, n, S, C):
        """Merge externally accumulated moments (device-side MC accumulators)."""
        S = np.asarray(S, dtype=self.dtype)
        self._ensure_stats_label(label, S.shape[0])
        self._n[label] += int(n)
        self._sum[label] += S
        self._cross[label] += np.asarray(C, dtype=self.dtype)

    def add_stack(self, label, arr):
        """stats.py:1124-1150."""
        A = np.asarray(arr, dtype=self.dtype)
        shape = () if A.ndim == 0 else A.shape
        self._ensure_stack_label(label, shape)
        self._k[label] += 1
        self._stack_sum[label] += A

    def add_stack_sum(self, label, total, count):
        """Merge an externally accumulated stack (sum of ``count`` arrays)."""
        A = np.asarray(total, dtype=self.dtype)
        self._ensure_stack_label(label, A.shape)
        self._k[label] += int(count)
        self._stack_sum[label] += A

    def _union_dims(self):
        """stats.py:1153-1182."""
        local = {"stats": [(lab, d) for lab, d in self._dim_stats.items()],
                 "stack": [(lab, shp) for lab, shp in self._shape_stack.items()]}
        if not self.mpi_enabled:
            return dict(self._dim_stats), dict(self._shape_stack)
        all_lists = self.comm.allgather(local)
        stats_union, stack_union = {}, {}
        for entry in all_lists:
            for lab, d in entry["stats"]:
                if lab in stats_union and stats_union[lab] != d:
                    raise ValueError(f"Stats dim mismatch for {lab!r} across ranks.")
                if lab in stack_union:
                    raise ValueError(f"Label {lab!r} used in stats and stack across ranks.")
                stats_union[lab] = d
            for lab, shp in entry["stack"]:
                shp = tuple(shp)
                if lab in stack_union and stack_union[lab] != shp:
                    raise ValueError(f"Stack shape mismatch for {lab!r} across ranks.")
                if lab in stats_union:
                    raise ValueError(f"Label {lab!r} used in stats and stack across ranks.")
                stack_union[lab] = shp
        return stats_union, stack_union

    def _allreduce(self, arr):
        if not self.mpi_enabled:
            return arr
        if hasattr(self.comm, "allreduce_array"):
            return self.comm.allreduce_array(arr)
        from mpi4py import MPI  # pragma: no cover - mpi4py path
        buf = np.array(arr, copy=True)
        self.comm.Allreduce(MPI.IN_PLACE, buf, op=MPI.SUM)
        return buf

    def allreduce(self):
        """stats.py:1184-1232."""
        stats_union, stack_union = self._union_dims()
        for lab, d in stats_union.items():
            if lab not in self._dim_stats:
                self._ensure_stats_label(lab, d)
        for lab, shp in stack_union.items():
            if lab not in self._shape_stack:
                self._ensure_stack_label(lab, shp)
        for lab in stats_union:
            n_loc = np.array([self._n.get(lab, 0)], dtype=np.int64)
            self._N[lab] = int(np.asarray(self._allreduce(n_loc)).ravel()[0])
            self._SUM[lab] = self._allreduce(self._sum[lab])
            self._CROSS[lab] = self._allreduce(self._cross[lab])
        for lab in stack_union:
            k_loc = np.array([self._k.get(lab, 0)], dtype=np.int64)
            self._K[lab] = int(np.asarray(self._allreduce(k_loc)).ravel()[0])
            self._STACK_SUM[lab] = self._allreduce(self._stack_sum[lab])
        self._reduced = True

    def labels_stats(self):
        return list(self._SUM.keys()) if self._reduced else list(self._dim_stats.keys())

    def labels_stack(self):
        return list(self._STACK_SUM.keys()) if self._reduced else list(self._shape_stack.keys())

    def _check_reduced(self):
        if not self._reduced:
            raise RuntimeError("Call .allreduce() before requesting global stats/stack.")

    def count(self, label):
        self._check_reduced()
        if label not in self._N:
            raise KeyError(f"{label!r} is not a stats-mode label.")
        return self._N[label]

    def stack_count(self, label):
        self._check_reduced()
        if label not in self._K:
            raise KeyError(f"{label!r} is not a stack-mode label.")
        return self._K[label]

    def mean(self, label):
        self._check_reduced()
        if label not in self._SUM:
            raise KeyError(f"{label!r} is not a stats-mode label.")
        n = self._N[label]
        return self._SUM[label] / n if n > 0 else np.full(self._SUM[label].shape, np.nan, dtype=self.dtype)

    def cov(self, label, ddof=1):
        self._check_reduced()
        if label not in self._CROSS:
            raise KeyError(f"{label!r} is not a stats-mode label.")
        n = self._N[label]
        if n <= ddof:
            d = self._SUM[label].shape[0]
            return np.full((d, d), np.nan, dtype=self.dtype)
        S, C = self._SUM[label], self._CROSS[label]
        return (C - np.outer(S, S) / n) / (n - ddof)

    def var(self, label, ddof=1):
        self._check_reduced()
        if label not in self._CROSS:
            raise KeyError(f"{label!r} is not a stats-mode label.")
        n = self._N[label]
        if n <= ddof:
            return np.full(self._SUM[label].shape[0], np.nan, dtype=self.dtype)
        S, C = self._SUM[label], self._CROSS[label]
        return (np.diag(C) - (S * S) / n) / (n - ddof)

    def stack_sum(self, label):
        self._check_reduced()
        if label not in self._STACK_SUM:
            raise KeyError(f"{label!r} is not a stack-mode label.")
        return self._STACK_SUM[label]

    def save_reduced(self, path, compressed=False, root_rank=0):
        """stats.py:1455-1480 .npz schema; additionally stores ``stack/<label>/K``
        (the reference forgets the stack count, stats.py:1507-1527)."""
        self._check_reduced()
        if self.mpi_enabled and not (self.comm.Get_rank() == root_rank):
            return
        arrays = {}
        for lab in self._SUM.keys():
            arrays[f"stats/{lab}/N"] = np.array(self._N[lab], dtype=np.int64)
            arrays[f"stats/{lab}/SUM"] = self._SUM[lab]
            arrays[f"stats/{lab}/CROSS"] = self._CROSS[lab]
        for lab in self._STACK_SUM.keys():
            arrays[f"stack/{lab}/SUM"] = self._STACK_SUM[lab]
            arrays[f"stack/{lab}/K"] = np.array(self._K[lab], dtype=np.int64)
        saver = np.savez_compressed if compressed else np.savez
        saver(Path(path), **arrays)

    @classmethod
    def load_reduced(cls, path, comm=None, dtype=np.float64):
        """stats.py:1483-1530."""
        data = np.load(Path(path), allow_pickle=False)
        acc = cls(comm=comm, dtype=dtype)
        for key in data.files:
            parts = key.split("/")
            lab = parts[1]
            if parts[0] == "stats":
                if parts[2] == "N":
                    acc._N[lab] = int(data[key])
                elif parts[2] == "SUM":
                    acc._SUM[lab] = np.array(data[key])
                    acc._dim_stats[lab] = acc._SUM[lab].shape[0]
                elif parts[2] == "CROSS":
                    acc._CROSS[lab] = np.array(data[key])
            elif parts[0] == "stack":
                if parts[2] == "SUM":
                    acc._STACK_SUM[lab] = np.array(data[key])
                    acc._shape_stack[lab] = acc._STACK_SUM[lab].shape
                elif parts[2] == "K":
                    acc._K[lab] = int(data[key])
        acc._reduced = True
        return acc
