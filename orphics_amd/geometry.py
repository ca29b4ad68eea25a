"""Explicit flat-sky geometry value object.

The reference gets pixel steps / area from an astropy WCS through pixell
(``enmap.area`` maps.py:1605, ``enmap.lmap`` maps.py:1607, ``wcs.wcs.cdelt``
maps.py:2182).  pixell is not a dependency here, so every class takes either

* a :class:`FlatGeometry` (signed steps in radians + patch area), or
* any object exposing ``wcs.wcs.cdelt`` in degrees (astropy / pixell WCS):
  ``step_x = cdelt[0]``, ``step_y = cdelt[1]``, planar area.  A pixell user
  wanting the exact spherical area passes ``FlatGeometry(..., area=enmap.area())``.
"""
import numpy as np


class FlatGeometry(object):
    def __init__(self, shape, step_y, step_x, area=None):
        self.shape = tuple(int(s) for s in shape)
        self.step_y = float(step_y)
        self.step_x = float(step_x)
        Ny, Nx = self.shape[-2:]
        self.area = float(area) if area is not None else Ny * Nx * abs(self.step_y) * abs(self.step_x)

    @classmethod
    def from_res(cls, shape, res_arcmin, area=None):
        """Standard CAR orientation: y increasing, x (RA) decreasing."""
        r = res_arcmin * np.pi / 180. / 60.
        return cls(shape, r, -r, area)

    def with_shape(self, shape):
        return FlatGeometry(shape, self.step_y, self.step_x, self.area)

    @property
    def pixarea(self):
        Ny, Nx = self.shape[-2:]
        return self.area / (Ny * Nx)

    # pixell enmap.laxes / lmap / modlmap (float64, NumPy op order)
    def laxes(self):
        Ny, Nx = self.shape[-2:]
        ly = np.fft.fftfreq(Ny, self.step_y) * 2 * np.pi
        lx = np.fft.fftfreq(Nx, self.step_x) * 2 * np.pi
        return ly, lx

    def lmap(self):
        ly, lx = self.laxes()
        out = np.empty((2,) + self.shape[-2:])
        out[0] = ly[:, None]
        out[1] = lx[None, :]
        return out

    def modlmap(self):
        ly, lx = self.laxes()
        return np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)

    def __repr__(self):
        return "FlatGeometry(shape=%r, step_y=%g, step_x=%g, area=%g)" % (self.shape, self.step_y, self.step_x, self.area)


def as_geometry(shape, wcs):
    """Accept FlatGeometry or a WCS-like object (``wcs.wcs.cdelt`` in degrees)."""
    if isinstance(wcs, FlatGeometry):
        g = wcs
        if tuple(g.shape[-2:]) != tuple(shape[-2:]):
            # same pixelisation, different patch size: rescale planar area
            Ny, Nx = shape[-2:]
            return FlatGeometry(shape, g.step_y, g.step_x, g.pixarea * Ny * Nx)
        return FlatGeometry(shape, g.step_y, g.step_x, g.area)
    inner = getattr(wcs, "wcs", None)
    cdelt = getattr(inner, "cdelt", None)
    if cdelt is None:
        raise TypeError("wcs must be a FlatGeometry or expose wcs.wcs.cdelt (degrees)")
    return FlatGeometry(shape, float(cdelt[1]) * np.pi / 180., float(cdelt[0]) * np.pi / 180.)


def rect_geometry(width_arcmin=None, width_deg=None, px_res_arcmin=0.5, pol=False, height_deg=None,
                  height_arcmin=None, **kwargs):
    """maps.rect_geometry (maps.py:1472-1498) for a planar patch: returns
    (shape, FlatGeometry).  The pixel count is ``width/res`` rounded to nearest."""
    if width_deg is not None:
        width_arcmin = 60. * width_deg
    if height_deg is not None:
        height_arcmin = 60. * height_deg
    if height_arcmin is None:
        height_arcmin = width_arcmin
    Nx = int(round(width_arcmin / px_res_arcmin))
    Ny = int(round(height_arcmin / px_res_arcmin))
    shape = (Ny, Nx)
    geom = FlatGeometry.from_res(shape, px_res_arcmin)
    if pol:
        shape = (3,) + shape
    return shape, geom
