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

    def bin(self, data2d, weights=None, err=False, get_count=False, mask_nan=False, err_reference_indexing=False):
        """stats.py:790-811.  ``err=True`` returns the standard error of the bin mean
        sqrt(sum_b (x - mu_b)^2 / (c_b - 1) / c_b).  DEVIATION, on purpose: the reference's loop runs its index over
        0..nbins-1 against the 1-based bin ids (stats.py:799-801), so it subtracts the mean of the NEXT bin from
        the members of bin i (constant-per-bin data does not give 0 there).  The intended statistic is the default;
        ``err_reference_indexing=True`` reproduces the reference's shifted one (pinned by the fixture
        ``a_err_ref_shifted`` made from the real orphics.stats.bin2D)."""
        torch = _torch()
        if weights is None:
            sums, counts = self._raw(data2d, skip_nan=mask_nan)
            count = self._slice(counts.cpu().numpy())
            with np.errstate(divide="ignore", invalid="ignore"):
                res = self._slice(sums.cpu().numpy()) / count
            if err:
                cfull = counts.to(torch.float64)
                mean = torch.where(cfull > 0, sums / cfull, torch.zeros_like(sums))
                if err_reference_indexing:
                    # reference: meanmap[digitized == i] = res[i], res = bincount[1:-1]  ->  members of id i get the
                    # mean of id i + 1 (ids 0..nbins-1 only); everything else keeps meanmap = 0
                    nb = self.bin_edges.size - 1
                    shifted = torch.zeros_like(mean)
                    hi = min(nb, self._maxid - 1)
                    shifted[:hi] = mean[1:hi + 1]
                    mean = shifted
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
    """Legacy per-realisation container (contract of stats.py:577-735): every rank keeps the vectors it was
    given, ``get_stats`` gathers them on ``root`` and summarises them with :func:`get_stats`; ``get_stacks``
    returns ensemble means of running array sums.  Here the gather / sum go through
    :class:`orphics_amd.mpi.TorchComm` (gloo or RCCL) instead of tagged Send/Recv pairs."""

    RESERVED = "stats"

    def __init__(self, comm=None, root=0, loopover=None, tag_start=333):
        self.comm = _mpi.fakeMpiComm() if comm is None else comm
        self.rank, self.numcores = self.comm.Get_rank(), self.comm.Get_size()
        self.root = root
        self.tag_start = tag_start                       # kept for signature compatibility (no tagged messages here)
        self.loopover = [r for r in range(self.numcores) if r != root] if loopover is None else loopover
        self.vectors, self.columns = {}, {}
        self.little_stack, self.little_stack_count = {}, {}
        self.stats, self.stacks, self.stack_count = {}, {}, {}

    @classmethod
    def _admit(cls, label, data, what):
        if label == cls.RESERVED:
            raise AssertionError("the label %r is reserved" % cls.RESERVED)
        data = np.asarray(data)
        if np.iscomplexobj(data):
            raise TypeError("%s of complex arrays is not supported: pass real and imaginary parts under two labels" % what)
        return data

    def add_to_stats(self, label, vector, exclude=False):
        """Record one realisation of a 1-D quantity (stats.py:614-631).  ``exclude`` only declares the label
        (so that every rank knows its width) without contributing a sample."""
        vector = self._admit(label, vector, "statistics")
        if label not in self.vectors:
            self.vectors[label], self.columns[label] = [], vector.shape
        if not exclude:
            self.vectors[label].append(vector)

    def add_to_stack(self, label, arr, exclude=False):
        """Add an array to a running sum (stats.py:634-650)."""
        arr = self._admit(label, arr, "stacking")
        if label not in self.little_stack:
            self.little_stack[label] = np.zeros(arr.shape, dtype=np.float64)
            self.little_stack_count[label] = 0
        if not exclude:
            self.little_stack[label] = self.little_stack[label] + arr
            self.little_stack_count[label] += 1

    def get_stacks(self, verbose=True):
        """Ensemble mean of every stack on ``root`` (stats.py:653-691): sum of sums / sum of counts."""
        for label, part in self.little_stack.items():
            total = np.array(part, dtype=np.float64)
            count = np.array([self.little_stack_count[label]], dtype=np.int64)
            if self.numcores > 1:
                total, count = self.comm.allreduce_array(total), self.comm.allreduce_array(count)
            if self.rank == self.root:
                self.stack_count[label] = int(count[0])
                self.stacks[label] = total / self.stack_count[label]

    def get_stats(self, verbose=True, skip_stats=False):
        """Collect all realisations on ``root`` and summarise (stats.py:693-735)."""
        for label in list(self.vectors):
            width = tuple(self.columns[label])
            local = np.asarray(self.vectors[label], dtype=np.float64).reshape((-1,) + width)
            if self.numcores > 1:
                pieces = self.comm.gather_arrays(local, root=self.root)
                if self.rank != self.root:
                    continue
                pieces = [p for p in pieces if len(p)]
                local = np.concatenate(pieces, axis=0) if pieces else local
            self.vectors[label] = local
            if not skip_stats:
                self.stats[label] = get_stats(local)

    def dump(self, path):
        """One file per item (format of stats.py:737-743, read back by :func:`load_stats`)."""
        for kind, table in (("vectors", self.vectors), ("stack", self.stacks)):
            for label, arr in table.items():
                np.save(f"{path}/mstats_dump_{kind}_{label}.npy", arr)
        for label, summary in self.stats.items():
            for name, val in summary.items():
                np.savetxt(f"{path}/mstats_dump_stats_{label}_{name}.txt", np.atleast_1d(val))


def load_stats(path):
    """Read a directory written by :meth:`Stats.dump` (stats.py:745-772): an object with ``vectors``,
    ``stacks`` (label -> array) and ``stats`` (label -> {mean, cov, ...}; one-element summaries come back as scalars)."""
    import glob
    import os
    import types
    out = types.SimpleNamespace(vectors={}, stacks={}, stats={})
    for kind, table in (("vectors", out.vectors), ("stack", out.stacks)):
        prefix = "mstats_dump_%s_" % kind
        for f in glob.glob(os.path.join(path, prefix + "*.npy")):
            table[os.path.basename(f)[len(prefix):-len(".npy")]] = np.load(f)
    prefix = "mstats_dump_stats_"
    summary_names = ("mean", "cov", "covmean", "errmean", "err", "corr")       # longest match first for *_errmean vs *_err
    for f in glob.glob(os.path.join(path, prefix + "*.txt")):
        stem = os.path.basename(f)[len(prefix):-len(".txt")]
        name = next((n for n in summary_names if stem.endswith("_" + n)), None)
        if name is None:
            label, _, name = stem.rpartition("_")
        else:
            label = stem[:-(len(name) + 1)]
        arr = np.loadtxt(f)
        out.stats.setdefault(label, {})[name] = arr.ravel()[0] if arr.size == 1 else arr
    return out


# ---- one-pass ensemble moments -----------------------------------------------------------------------------------
class _Moments(object):
    """count, sum x, sum x x^T of d-vectors.  ``xp`` is numpy (host) or torch (device tensors, float64)."""
    __slots__ = ("dim", "count", "first", "second", "cell")

    def __init__(self, dim, first, second):
        self.dim, self.count, self.first, self.second = int(dim), 0, first, second
        self.cell = None        # device mode: the int64 counter the accumulation kernels increment


class _Stack(object):
    __slots__ = ("shape", "count", "total", "support")

    def __init__(self, shape, total):
        self.shape, self.count, self.total = tuple(int(v) for v in shape), 0, total
        self.support = None     # (rows, cols): the stack is zero outside rows y < rows or y > ny - rows, columns < cols


def _pack_support(t, support):
    """the possibly non-zero part of a (ny, kp, ...) device stack as one contiguous tensor"""
    import torch
    rb, w = support
    ny = t.shape[0]
    if rb <= 0 or 2 * rb - 1 >= ny:
        return t[:, :w].contiguous()
    return torch.cat((t[:rb, :w], t[ny - rb + 1:, :w])).contiguous()


def _unpack_support(sub, shape, support):
    import torch
    rb, w = support
    ny = shape[0]
    out = torch.zeros(shape, dtype=sub.dtype, device=sub.device)
    if rb <= 0 or 2 * rb - 1 >= ny:
        out[:, :w] = sub
    else:
        out[:rb, :w] = sub[:rb]
        out[ny - rb + 1:, :w] = sub[rb:]
    return out


class _RegionSum(object):
    """Reduced stack known to vanish outside its support: only the support travelled through the all-reduce; the full
    plane is materialised when somebody asks for it."""

    def __init__(self, sub, shape, support):
        self.sub, self.shape, self.support = sub, tuple(shape), support

    def full(self):
        return _unpack_support(self.sub, self.shape, self.support)


class Statistics(object):
    """One-pass ensemble statistics with a single SUM reduction at the end (the contract of stats.py:918-1530).

    Two kinds of labels: *stats* labels collect d-vectors x_i as (n, sum x, sum x x^T) -- enough for mean, variance
    and covariance without keeping the samples; *stack* labels keep (k, sum A) of equally shaped arrays.
    ``allreduce()`` sums everything over the ranks of ``comm`` (``None`` = single process, a
    :class:`orphics_amd.mpi.TorchComm` = gloo / RCCL, or an mpi4py communicator); only then are ``mean / cov / var /
    count / stack_sum / stack_count`` available.

    MI355X specifics: with ``device=`` the accumulators are float64 tensors on that GPU, fed by the HIP kernels
    ``oa_moments_add`` / ``oa_moments_add_binned`` / ``oa_stack_add`` from device-resident samples (a Monte-Carlo
    loop never copies a bandpower vector to the host), and the reduction is ONE packed all-reduce of all labels
    (plus one for the integer counts) instead of three per label.  Without ``device`` everything is host NumPy.
    """

    def __init__(self, comm=None, dtype=np.float64, device=None):
        self.comm = comm
        self.dtype = np.dtype(dtype)
        self.device = device
        self._vec, self._pile = {}, {}          # label -> _Moments / _Stack (this rank's share)
        self._world = None                      # after allreduce(): (vec, pile) dicts of reduced HOST copies

    # -- storage back ends ------------------------------------------------------------------------------------
    @property
    def mpi_enabled(self):
        return self.comm is not None

    def _zeros(self, shape):
        if self.device is None:
            return np.zeros(shape, dtype=self.dtype)
        import torch
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    @staticmethod
    def _is_device_tensor(x):
        return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)

    def _slot_vec(self, label, dim):
        if label in self._pile:
            raise ValueError("label %r already collects stacked arrays; it cannot also collect vectors" % (label,))
        m = self._vec.get(label)
        if m is None:
            m = self._vec[label] = _Moments(dim, self._zeros((dim,)), self._zeros((dim, dim)))
        elif m.dim != dim:
            raise ValueError("label %r collects vectors of length %d, got length %d" % (label, m.dim, dim))
        return m

    def _slot_pile(self, label, shape):
        if label in self._vec:
            raise ValueError("label %r already collects vectors; it cannot also collect stacked arrays" % (label,))
        shape = tuple(int(v) for v in shape)
        p = self._pile.get(label)
        if p is None:
            p = self._pile[label] = _Stack(shape, self._zeros(shape))
        elif p.shape != shape:
            raise ValueError("label %r stacks arrays of shape %s, got %s" % (label, p.shape, shape))
        return p

    def _to_store(self, a):
        """sample / array in any container -> the accumulator's container (float64 on the accumulator's device)"""
        if self.device is None:
            if self._is_device_tensor(a) or type(a).__module__.startswith("torch"):
                a = a.detach().cpu().numpy()
            return np.asarray(a, dtype=self.dtype)
        import torch
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64))
        return t.to(device=self.device, dtype=torch.float64)

    # -- accumulation -----------------------------------------------------------------------------------------
    def add(self, label, x):
        """One sample of a stats label (stats.py:1068-1090): n += 1, S += x, C += x x^T."""
        if self.device is not None and self._is_device_tensor(x) and x.dtype.is_floating_point and x.element_size() == 8:
            x = x.reshape(-1).contiguous()
            m = self._slot_vec(label, x.numel())
            self._device_add(m, x)
            return
        v = self._to_store(x).reshape(-1)
        m = self._slot_vec(label, v.shape[0])
        m.count += 1
        m.first += v
        m.second += (v[:, None] * v[None, :])

    def _counter_cell(self, m):
        if m.cell is None:
            import torch
            m.cell = torch.zeros(1, dtype=torch.int64, device=self.device)
        return m.cell

    def _device_add(self, m, x):
        from ._lib import check, load
        from .engine import _ptr, _stream
        check(load().oa_moments_add(_ptr(x), m.dim, _ptr(self._counter_cell(m)), _ptr(m.first), _ptr(m.second), _stream()))
        m.count += 1

    def add_binned(self, label, sums, counts):
        """Device shortcut for the Monte-Carlo loop: the sample is the vector of bin means sums/counts (the output
        of ``Engine.bin_power`` / ``oa_bin``; stats.bin2D.bin semantics), formed inside the accumulation kernel."""
        if self.device is None or not self._is_device_tensor(sums):
            c = np.asarray(counts.cpu() if hasattr(counts, "cpu") else counts, dtype=np.float64)
            t = np.asarray(sums.cpu() if hasattr(sums, "cpu") else sums, dtype=np.float64)
            return self.add(label, t / c)
        from ._lib import check, load
        from .engine import _ptr, _stream
        m = self._slot_vec(label, sums.numel())
        check(load().oa_moments_add_binned(_ptr(sums), _ptr(counts), m.dim, _ptr(self._counter_cell(m)), _ptr(m.first), _ptr(m.second), _stream()))
        m.count += 1

    def device_moments(self, label, dim):
        """(n int64[1], S f64[dim], C f64[dim, dim]) device accumulators of a stats label, for kernels that add
        samples themselves (``oa_qe_tt_moments``, ``oa_mc_run``); report how many with :meth:`note_samples`."""
        if self.device is None:
            raise RuntimeError("device_moments() needs Statistics(device=...)")
        m = self._slot_vec(label, dim)
        return self._counter_cell(m), m.first, m.second

    def note_samples(self, label, k):
        self._vec[label].count += int(k)

    def device_stack(self, label, shape, support=None):
        """float64 device accumulator of a stack label (``oa_stack_add`` / ``oa_mc_run`` mean field); report the
        number of arrays added with :meth:`note_stacked`.  ``support = (rows, cols)``: the caller only ever adds to the
        rows y < rows or y > ny - rows, columns < cols of the (ny, kp, ...) plane (kappa_hat's active region): the
        all-reduce then moves that region only (14 MB instead of a 135 MB plane and its copy at 4096^2)."""
        if self.device is None:
            raise RuntimeError("device_stack() needs Statistics(device=...)")
        p = self._slot_pile(label, shape)
        if support is not None:
            support = (int(support[0]), int(support[1]))
            if p.support not in (None, support):
                raise ValueError("label %r already stacks with support %s" % (label, p.support))
            p.support = support
        return p.total

    def note_stacked(self, label, k):
        self._pile[label].count += int(k)

    def extend(self, label, X):
        """Many samples at once (stats.py:1092-1120): rows of an (m, d) array, or one (d,) sample."""
        X2 = self._to_store(X if hasattr(X, "shape") else list(X))
        if X2.ndim == 1:
            return self.add(label, X2)
        if X2.ndim != 2:
            raise ValueError("extend() takes an (m, d) block of samples or a single (d,) sample, got %d dimensions" % X2.ndim)
        m = self._slot_vec(label, X2.shape[1])
        m.count += int(X2.shape[0])
        m.first += X2.sum(0)
        m.second += X2.T @ X2

    def add_moments(self, label, n, S, C):
        """Merge moments accumulated elsewhere (another accumulator, a device-side Monte-Carlo loop)."""
        S = self._to_store(S).reshape(-1)
        m = self._slot_vec(label, S.shape[0])
        m.count += int(n)
        m.first += S
        m.second += self._to_store(C)

    def add_stack(self, label, arr):
        """One array of a stack label (stats.py:1124-1150): k += 1, total += arr."""
        if self.device is not None and self._is_device_tensor(arr) and arr.dtype.is_floating_point:
            from ._lib import OA_F32, OA_F64, check, load
            from .engine import _ptr, _stream
            a = arr.contiguous()
            p = self._slot_pile(label, tuple(a.shape))
            code = OA_F32 if a.element_size() == 4 else OA_F64
            check(load().oa_stack_add(code, _ptr(a), _ptr(p.total), a.numel(), _stream()))
            p.count += 1
            return
        a = self._to_store(arr)
        p = self._slot_pile(label, tuple(a.shape))
        p.count += 1
        p.total += a

    def add_stack_sum(self, label, total, count):
        """Merge a stack accumulated elsewhere (sum of ``count`` arrays)."""
        a = self._to_store(total)
        p = self._slot_pile(label, tuple(a.shape))
        p.count += int(count)
        p.total += a

    # -- reduction ----------------------------------------------------------------------------------------------
    def _schema(self):
        return {"vec": {lab: m.dim for lab, m in self._vec.items()}, "pile": {lab: p.shape for lab, p in self._pile.items()},
                "support": {lab: p.support for lab, p in self._pile.items() if p.support is not None}}

    def _agree_on_schema(self):
        """Union of the labels of all ranks (a label may be missing on some: it contributes zeros there,
        stats.py:1153-1182); contradictory uses are an error on EVERY rank."""
        mine = self._schema()
        everyone = self.comm.allgather(mine) if (self.mpi_enabled and self.comm.Get_size() > 1) else [mine]
        vec, pile = {}, {}
        for sch in everyone:
            for lab, dim in sch["vec"].items():
                if vec.setdefault(lab, dim) != dim:
                    raise ValueError("label %r has vector length %d on one rank and %d on another" % (lab, vec[lab], dim))
            for lab, shape in sch["pile"].items():
                if pile.setdefault(lab, tuple(shape)) != tuple(shape):
                    raise ValueError("label %r stacks shape %s on one rank and %s on another" % (lab, pile[lab], tuple(shape)))
        both = set(vec) & set(pile)
        if both:
            raise ValueError("label(s) %s collect vectors on one rank and stacked arrays on another" % sorted(both))
        self._supports = {}
        for sch in everyone:
            for lab, sup in sch.get("support", {}).items():
                if self._supports.setdefault(lab, tuple(sup)) != tuple(sup):
                    raise ValueError("label %r stacks with support %s on one rank and %s on another" % (lab, self._supports[lab], tuple(sup)))
        return vec, pile

    def _sum_over_ranks(self, buf):
        """SUM all-reduce of one flat buffer (NumPy array or torch tensor); identity without a communicator."""
        if not self.mpi_enabled or self.comm.Get_size() == 1:
            return buf
        if hasattr(self.comm, "dist"):                          # TorchComm: tensors stay where they are (RCCL on GPU)
            import torch
            if isinstance(buf, torch.Tensor):
                if buf.is_cuda or self.comm.backend != "nccl":
                    self.comm.dist.all_reduce(buf, op=self.comm.dist.ReduceOp.SUM, group=self.comm.group)
                    return buf
                return self.comm.allreduce_array(buf.numpy())
            return self.comm.allreduce_array(buf)
        from mpi4py import MPI                                   # pragma: no cover - mpi4py communicators
        out = np.array(buf, copy=True)
        self.comm.Allreduce(MPI.IN_PLACE, out, op=MPI.SUM)
        return out

    PACK_LIMIT = 1 << 16      # device tensors with more elements than this are reduced in place, not packed

    def allreduce(self):
        """Sum counts, first and second moments and stacks over all ranks (stats.py:1184-1232) -- here as ONE
        all-reduce of the integer counts and ONE of a packed float64 buffer holding every small item of every label;
        large device-resident stacks (mean-field planes) are reduced in place, one collective each, and stay on
        the GPU until asked for."""
        vec, pile = self._agree_on_schema()
        for lab, dim in vec.items():
            self._slot_vec(lab, dim)
        for lab, shape in pile.items():
            self._slot_pile(lab, shape)
        order_v, order_p = sorted(vec), sorted(pile)
        counts = np.array([self._vec[l].count for l in order_v] + [self._pile[l].count for l in order_p] + [0], dtype=np.int64)
        counts = np.asarray(self._sum_over_ranks(counts))
        items = []                                   # (kind, label, field, array)
        for l in order_v:
            items += [("vec", l, "first", self._vec[l].first), ("vec", l, "second", self._vec[l].second)]
        for l in order_p:
            items.append(("pile", l, "total", self._pile[l].total))
        on_device = self.device is not None
        small = [it for it in items if not (on_device and it[3].numel() > self.PACK_LIMIT)]
        reduced = {}
        if small:
            if on_device:
                import torch
                flat = self._sum_over_ranks(torch.cat([it[3].reshape(-1) for it in small])).detach().cpu().numpy()
            else:
                flat = np.asarray(self._sum_over_ranks(np.concatenate([np.asarray(it[3]).reshape(-1) for it in small])))
            at = 0
            for kind, l, field, arr in small:
                n = int(np.prod(tuple(arr.shape), dtype=np.int64)) if len(arr.shape) else 1
                reduced[(kind, l, field)] = flat[at:at + n].reshape(tuple(arr.shape)).copy()
                at += n
        for kind, l, field, arr in items:
            if (kind, l, field) not in reduced:
                sup = getattr(self, "_supports", {}).get(l) if kind == "pile" else None
                if sup is not None:      # only the region that can be non-zero travels; no copy of the full plane
                    reduced[(kind, l, field)] = _RegionSum(self._sum_over_ranks(_pack_support(arr, sup)), tuple(arr.shape), sup)
                else:
                    reduced[(kind, l, field)] = self._sum_over_ranks(arr.clone())
        gv, gp = {}, {}
        for i, l in enumerate(order_v):
            g = _Moments(vec[l], reduced[("vec", l, "first")], reduced[("vec", l, "second")])
            g.count = int(counts[i])
            gv[l] = g
        for i, l in enumerate(order_p):
            g = _Stack(pile[l], reduced[("pile", l, "total")])
            g.count = int(counts[len(order_v) + i])
            gp[l] = g
        self._world = (gv, gp)

    # -- reduced results ----------------------------------------------------------------------------------------
    def _reduced(self, kind, label):
        if self._world is None:
            raise RuntimeError("global results exist only after allreduce()")
        table = self._world[0 if kind == "vec" else 1]
        if label not in table:
            raise KeyError("%r is not a %s label" % (label, "stats" if kind == "vec" else "stack"))
        return table[label]

    def labels_stats(self):
        return list(self._world[0]) if self._world is not None else list(self._vec)

    def labels_stack(self):
        return list(self._world[1]) if self._world is not None else list(self._pile)

    def count(self, label):
        return self._reduced("vec", label).count

    def stack_count(self, label):
        return self._reduced("pile", label).count

    def stack_sum(self, label, on_device=False):
        """Reduced sum of a stack label (NumPy; ``on_device=True`` returns the GPU tensor of a device-resident stack)."""
        t = self._reduced("pile", label).total
        if isinstance(t, _RegionSum):
            t = t.full()
        if hasattr(t, "detach"):
            return t if on_device else t.detach().cpu().numpy()
        return t

    def mean(self, label):
        """sum / n; NaN when no rank contributed a sample."""
        g = self._reduced("vec", label)
        return g.first / g.count if g.count else np.full(g.dim, np.nan, dtype=self.dtype)

    def cov(self, label, ddof=1):
        """(sum x x^T - sum x sum x^T / n) / (n - ddof); NaN when n <= ddof."""
        g = self._reduced("vec", label)
        if g.count <= ddof:
            return np.full((g.dim, g.dim), np.nan, dtype=self.dtype)
        return (g.second - np.outer(g.first, g.first) / g.count) / (g.count - ddof)

    def var(self, label, ddof=1):
        g = self._reduced("vec", label)
        if g.count <= ddof:
            return np.full(g.dim, np.nan, dtype=self.dtype)
        return (np.diagonal(g.second) - g.first ** 2 / g.count) / (g.count - ddof)

    # -- on-disk format (interoperable with the reference's post-processing, stats.py:1455-1530) ------------------
    def save_reduced(self, path, compressed=False, root_rank=0):
        """``.npz`` with keys ``stats/<label>/{N,SUM,CROSS}`` and ``stack/<label>/{SUM,K}`` (K: the stack count,
        which the reference's writer omits and its reader therefore cannot restore)."""
        if self._world is None:
            raise RuntimeError("global results exist only after allreduce()")
        if self.mpi_enabled and self.comm.Get_rank() != root_rank:
            return
        blob = {}
        for lab, g in self._world[0].items():
            blob["stats/%s/N" % lab] = np.array(g.count, dtype=np.int64)
            blob["stats/%s/SUM" % lab] = g.first
            blob["stats/%s/CROSS" % lab] = g.second
        for lab, g in self._world[1].items():
            blob["stack/%s/SUM" % lab] = self.stack_sum(lab)
            blob["stack/%s/K" % lab] = np.array(g.count, dtype=np.int64)
        (np.savez_compressed if compressed else np.savez)(Path(path), **blob)

    @classmethod
    def load_reduced(cls, path, comm=None, dtype=np.float64):
        data = np.load(Path(path), allow_pickle=False)
        fields = defaultdict(dict)
        for key in data.files:
            kind, lab, item = key.split("/", 2)
            fields[(kind, lab)][item] = np.array(data[key])
        self = cls(comm=comm, dtype=dtype)
        gv, gp = {}, {}
        for (kind, lab), f in fields.items():
            if kind == "stats":
                g = _Moments(f["SUM"].shape[0], f["SUM"], f["CROSS"])
                g.count = int(f["N"])
                gv[lab] = g
            elif kind == "stack":
                g = _Stack(f["SUM"].shape, f["SUM"])
                g.count = int(f["K"]) if "K" in f else 0
                gp[lab] = g
        self._world = (gv, gp)
        return self
