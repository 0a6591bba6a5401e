"""Minimal labelled array used where the reference uses xarray.DataArray.

The reference front end (xinvert/apps.py) takes xarray.DataArray forcings.  xarray is not
installed in the build/test image, so the host layer works on this small container and
converts real xarray objects at the boundary when xarray is importable (`from_any`/`to_like`).
Only what the hot path's callers need is implemented: values, ordered dims, 1-D coordinate
vectors per dim.
"""
import numpy as np


class Field:
    """values + ordered dim names + one 1-D coordinate vector per dim."""

    def __init__(self, values, dims, coords=None, name=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        if self.values.ndim != len(self.dims):
            raise ValueError('values.ndim %d != len(dims) %d' % (self.values.ndim, len(self.dims)))
        coords = {} if coords is None else dict(coords)
        self.coords = {}
        for ax, d in enumerate(self.dims):
            c = coords.get(d)
            c = np.arange(self.values.shape[ax], dtype=np.float64) if c is None else np.asarray(c)
            if c.ndim != 1 or c.shape[0] != self.values.shape[ax]:
                raise ValueError('coordinate %r does not match axis length' % d)
            self.coords[d] = c
        self.name = name

    @property
    def shape(self):
        return self.values.shape

    def axis(self, dim):
        return self.dims.index(dim)

    def like(self, values, name=None):
        return Field(values, self.dims, self.coords, name=name if name is not None else self.name)

    def __getitem__(self, dim):
        """F['lat'] -> coordinate vector, as xarray does."""
        return self.coords[dim]

    def __repr__(self):
        return 'Field(name=%r, dims=%r, shape=%r)' % (self.name, self.dims, self.shape)


def undef_as(dtype, undef):
    """The undefined value as an array of `dtype` stores it, as a Python float (float32 arrays: float(float32(undef)))."""
    dt = np.dtype(dtype)
    return float(np.asarray(undef, dtype=dt)) if dt.kind == 'f' else float(undef)


class LazyForcing(Field):
    """The forcing the kernels read, described instead of materialised:

        values = where(masked(raw), undef_tmp, raw * scale[along dim])

    with masked(raw) = isnan(raw) (the caller's undef is NaN) or raw == undef.  The host-pointer
    entry points evaluate this on the device (xinv_options.prep_flags: one pass at HBM rate instead
    of the reference's numpy passes, apps.py:2112-2159, 1409-1411); `.values` evaluates it with
    numpy on demand, for every consumer that is not the solver."""

    def __init__(self, raw, dims, coords, undef_in, undef_tmp, scale=None, scale_dim=None, name=None):
        self.raw = np.asarray(raw)
        self.dims = tuple(dims)
        self.coords = {d: np.asarray(coords[d]) for d in self.dims}
        self.name = name
        self.undef_in, self.undef_tmp = undef_in, undef_tmp
        self.scale = None if scale is None else np.asarray(scale, dtype=np.float64)
        self.scale_dim = scale_dim
        self._values = None

    @property
    def shape(self):
        return self.raw.shape

    @property
    def values(self):
        if self._values is None:
            r = np.asarray(self.raw, dtype=np.float64)
            # (the caller's undefined value as the raw array's OWN dtype holds it: a float32 forcing filled with 1e20 or
            #  9.96921e36 carries float32(undef), which differs from the Python float -- the reference's
            #  `F.where(F != undef)` compares in float32, apps.py:2124-2128)
            masked = (np.isnan(r) if np.isnan(self.undef_in) else (r == undef_as(self.raw.dtype, self.undef_in))) | (r == self.undef_tmp)
            v = r if self.scale is None else r * along(self.scale, self, self.scale_dim)
            self._values = np.where(masked, self.undef_tmp, v)
        return self._values

    @values.setter
    def values(self, v):
        self._values = np.asarray(v)

    def scaled(self, vec, dim):
        """The same forcing multiplied along `dim` by `vec` (and re-masked)."""
        if self.scale is not None:
            raise ValueError('a lazy forcing takes one scale')
        return LazyForcing(self.raw, self.dims, self.coords, self.undef_in, self.undef_tmp, vec, dim, self.name)

    def like(self, values, name=None):
        return Field(values, self.dims, self.coords, name=name if name is not None else self.name)


def along(vec, field, dim):
    """Broadcast a 1-D per-`dim` vector against `field` (xarray's alignment by dim name)."""
    shape = [1] * len(field.dims)
    shape[field.axis(dim)] = -1
    return np.asarray(vec).reshape(shape)


def aligned(param, field):
    """A model parameter lined up against `field` the way xarray arithmetic would: scalars pass
    through; labelled arrays (Field / DataArray) are transposed and reshaped by dim NAME, absent
    dims becoming length-1 axes; bare ndarrays follow numpy's trailing-axis broadcasting."""
    if np.isscalar(param):
        return param
    if hasattr(param, 'dims') and hasattr(param, 'values'):
        pd = tuple(param.dims)
        for d in pd:
            if d not in field.dims:
                raise Exception('parameter dimension %r is not a dimension of the forcing %r' % (d, field.dims))
        v = np.asarray(param.values, dtype=np.float64)
        order = sorted(range(len(pd)), key=lambda a: field.axis(pd[a]))
        v = np.transpose(v, order)
        shape = [1] * len(field.dims)
        for a in order:
            shape[field.axis(pd[a])] = param.values.shape[a]
        return v.reshape(shape)
    v = np.asarray(param, dtype=np.float64)
    np.broadcast_shapes(v.shape, field.shape)
    return v


def full(value, field):
    """`zero + value` of the reference: the value broadcast to the forcing's full shape.  Axes the
    value does not depend on get stride 0, which the solver front end (core._prep_coef) turns
    into one shared slice on the device."""
    return np.broadcast_to(np.asarray(value, dtype=np.float64), field.shape)


def from_any(obj, dims=None):
    """Field from a Field, an xarray.DataArray (if xarray is importable) or (ndarray, dims)."""
    if isinstance(obj, Field):
        return obj
    if hasattr(obj, 'dims') and hasattr(obj, 'coords') and hasattr(obj, 'values'):
        coords = {d: np.asarray(obj.coords[d].values) for d in obj.dims if d in obj.coords}
        return Field(np.asarray(obj.values), obj.dims, coords, name=getattr(obj, 'name', None))
    if dims is None:
        raise TypeError('need a Field, an xarray.DataArray or an ndarray with dims')
    return Field(np.asarray(obj), dims)


def to_like(field, template):
    """Return `field` in the caller's container type (xarray in -> xarray out)."""
    if isinstance(template, Field) or not hasattr(template, 'coords'):
        return field
    try:
        import xarray as xr
    except ImportError:                      # pragma: no cover - xarray absent in this image
        return field
    return xr.DataArray(field.values, dims=field.dims,
                        coords={d: field.coords[d] for d in field.dims}, name=field.name)
