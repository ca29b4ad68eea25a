"""NumPy restatement of the orphics.stats / orphics.mpi pieces on the hot path
(TEST INFRASTRUCTURE ONLY).

Pinned: ``tests/golden/make_golden.py`` imports the real
/root/reference/orphics/stats.py + mpi.py in the build container and stores
their outputs in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
this restatement against them (bin ids / counts bit-exact, sums <= 1e-15).
"""
import numpy as np


class bin2D(object):
    """stats.py:782-811.  Bins are (e[i-1], e[i]]; id 0 = underflow,
    len(edges) = overflow; ``[1:-1]`` strips both -- and silently drops the last
    real bin when nothing overflows (no ``minlength``), which
    is reproduced here."""

    def __init__(self, modrmap, bin_edges):
        bin_edges = np.asarray(bin_edges)
        self.centers = (bin_edges[1:] + bin_edges[:-1]) / 2.
        self.cents = self.centers
        self.digitized = np.digitize(np.asarray(modrmap).reshape(-1), bin_edges, right=True)
        self.bin_edges = bin_edges
        self.modrmap = modrmap

    def bin(self, data2d, weights=None, err=False, get_count=False, mask_nan=False):
        data2d = np.asarray(data2d)
        if weights is None:
            if mask_nan:
                keep = ~np.isnan(data2d.reshape(-1))
            else:
                keep = np.ones((data2d.size,), dtype=bool)
            count = np.bincount(self.digitized[keep])[1:-1]
            with np.errstate(divide="ignore", invalid="ignore"):
                res = np.bincount(self.digitized[keep], data2d.reshape(-1)[keep])[1:-1] / count
            if err:
                # Reference loops i over 0..nbins-1 against 1-based ids
                # (stats.py:799-801) -> mean map shifted by one bin.  That is
                # a reference bug; ``err_reference_shifted`` reproduces it,
                # this branch computes the intended std of the mean.
                meanmap = np.zeros(self.digitized.shape)
                for i in range(res.size):
                    meanmap[self.digitized == (i + 1)] = res[i]
                with np.errstate(divide="ignore", invalid="ignore"):
                    std = np.sqrt(np.bincount(self.digitized[keep], ((data2d.reshape(-1) - meanmap) ** 2.)[keep])[1:-1]
                                  / (count - 1) / count)
        else:
            count = np.bincount(self.digitized, np.asarray(weights).reshape(-1))[1:-1]
            with np.errstate(divide="ignore", invalid="ignore"):
                res = np.bincount(self.digitized, (data2d * weights).reshape(-1))[1:-1] / count
        if get_count:
            assert not err
            return self.centers, res, count
        if err:
            return self.centers, res, std
        return self.centers, res

    def err_reference_shifted(self, data2d):
        """Bug-compatible stats.py:798-801 (mean map indexed by ``i`` not
        ``i+1``); kept only so the fixture of the reference's own output can be
        matched."""
        data2d = np.asarray(data2d)
        count = np.bincount(self.digitized)[1:-1]
        res = np.bincount(self.digitized, data2d.reshape(-1))[1:-1] / count
        meanmap = np.asarray(self.modrmap).copy().reshape(-1) * 0
        for i in range(self.centers.size):
            if i < res.size:
                meanmap[self.digitized == i] = res[i]
        std = np.sqrt(np.bincount(self.digitized, ((data2d - meanmap.reshape(data2d.shape)) ** 2.).reshape(-1))[1:-1]
                      / (count - 1) / count)
        return self.centers, res, std


def bin_in_annuli(data2d, modrmap, bin_edges):
    """stats.py:853-855."""
    return bin2D(modrmap, bin_edges).bin(data2d)


def cov2corr(cov):
    d = np.sqrt(np.diagonal(cov))
    return cov / np.outer(d, d)


def get_stats(binned_vectors):
    """stats.py:859-898."""
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
    ret['corr'] = 1. if arr.shape[1] == 1 else cov2corr(ret['cov'])
    return ret


def moments_merge(parts):
    """Statistics.allreduce semantics (stats.py:1184-1232): SUM of (n, S, C)."""
    n = sum(p[0] for p in parts)
    S = sum(p[1] for p in parts)
    C = sum(p[2] for p in parts)
    return n, S, C


def moments_mean_cov(n, S, C, ddof=1):
    """stats.py:1311-1370: mean = S/n ; cov = (C - S S^T/n)/(n-ddof)."""
    mean = S / n if n > 0 else np.full(S.shape, np.nan)
    if n <= ddof:
        cov = np.full(C.shape, np.nan)
    else:
        cov = (C - np.outer(S, S) / n) / (n - ddof)
    return mean, cov


def mpi_distribute(num_tasks, avail_cores, allow_empty=False):
    """mpi.py:78-91: contiguous blocks, remainder to the LAST ranks."""
    if not allow_empty:
        assert avail_cores <= num_tasks
    min_each, rem = divmod(num_tasks, avail_cores)
    num_each = np.array([min_each] * avail_cores)
    if rem > 0:
        num_each[-rem:] += 1
    task_range = list(range(num_tasks))
    cumul = np.cumsum(num_each).tolist()
    task_dist = [task_range[x:y] for x, y in zip([0] + cumul[:-1], cumul)]
    return num_each, task_dist
