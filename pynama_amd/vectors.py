"""Device-resident vector with the PETSc.Vec methods the reference calls:
set, setValues, setValue, assemble, getArray, getValues, copy, duplicate, axpy, scale, norm, dot,
reciprocal, a-b, a+b, a*b, +=, *=, unary -, getOwnershipRange/owner_range, size, setName/getName.
Every operation runs on the GPU through the C ABI; getArray()/getValues() copy to the host."""
import numpy as np

NORM_1, NORM_2, NORM_INFINITY = 1, 2, 3


class Vec:
    def __init__(self, ctx, bs, name=None, _id=None):
        self.ctx = ctx
        self.bs = int(bs)
        self.id = ctx.vec_create(self.bs) if _id is None else _id
        self._name = name

    def __del__(self):
        try:
            if self.ctx is not None and self.ctx.h:
                self.ctx.vec_destroy(self.id)
        except Exception:
            pass

    def destroy(self):
        if self.ctx is not None and self.ctx.h:
            self.ctx.vec_destroy(self.id)
        self.ctx = None

    # -- naming / sizes
    def setName(self, name):
        self._name = name

    def getName(self):
        return self._name

    @property
    def local_size(self):
        return self.ctx.n_owned * self.bs

    def getLocalSize(self):
        return self.local_size

    def getSize(self):
        return int(self.ctx.allreduce([self.local_size])[0]) if self.ctx.nranks > 1 else self.local_size

    size = property(getSize)

    def getSizes(self):
        return (self.local_size, self.getSize())

    def getOwnershipRange(self):
        r0 = getattr(self.ctx, "row_start", 0) * self.bs
        return (r0, r0 + self.local_size)

    owner_range = property(getOwnershipRange)

    # -- fill / access
    def set(self, value):
        self.ctx.vec_fill(self.id, value)

    def setArray(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert arr.size == self.local_size
        self.ctx.vec_set(self.id, arr)

    def getArray(self, readonly=True):
        return self.ctx.vec_get(self.id, self.bs)

    array = property(getArray)

    def __array__(self, dtype=None, copy=None):
        a = self.getArray()
        return a.astype(dtype) if dtype is not None else a

    def _to_local(self, indices):
        idx = np.asarray(indices, dtype=np.int64).ravel()
        return idx - self.getOwnershipRange()[0]

    def setValuesLocal(self, local_indices, values, addv=False):
        self.ctx.vec_scatter(self.id, np.asarray(local_indices, dtype=np.int64).astype(np.int32),
                             np.asarray(values, dtype=np.float64).ravel(), add=bool(addv))

    def setValues(self, indices, values, addv=False):
        loc = self._to_local(indices)
        vals = np.asarray(values, dtype=np.float64).ravel()
        keep = (loc >= 0) & (loc < self.local_size)
        self.setValuesLocal(loc[keep], vals[keep], addv)

    def setValue(self, index, value, addv=False):
        self.setValues([index], [value], addv)

    def getValues(self, indices):
        return self.getArray()[self._to_local(indices)]

    def assemble(self):
        return None

    assemblyBegin = assemblyEnd = assemble

    # -- algebra
    def duplicate(self):
        return Vec(self.ctx, self.bs, name=self._name)

    def copy(self, result=None):
        out = result if result is not None else self.duplicate()
        self.ctx.vec_axpby(out.id, 1.0, self.id, 0.0, self.id)
        return out

    def axpy(self, alpha, x):
        self.ctx.vec_axpby(self.id, float(alpha), x.id, 1.0, self.id)

    def aypx(self, alpha, x):
        self.ctx.vec_axpby(self.id, 1.0, x.id, float(alpha), self.id)

    def scale(self, alpha):
        self.ctx.vec_axpby(self.id, float(alpha), self.id, 0.0, self.id)

    def reciprocal(self):
        self.ctx.vec_reciprocal(self.id)

    def pointwiseMult(self, x, y):
        self.ctx.vec_pointwise_mult(self.id, x.id, y.id)

    def dot(self, other):
        return self.ctx.vec_dot(self.id, other.id)

    def norm(self, norm_type=NORM_2):
        return self.ctx.vec_norm(self.id, int(norm_type) if norm_type is not None else NORM_2)

    def __add__(self, o):
        out = self.duplicate()
        self.ctx.vec_axpby(out.id, 1.0, self.id, 1.0, o.id)
        return out

    def __sub__(self, o):
        out = self.duplicate()
        self.ctx.vec_axpby(out.id, 1.0, self.id, -1.0, o.id)
        return out

    def __mul__(self, o):
        out = self.duplicate()
        if isinstance(o, Vec):
            self.ctx.vec_pointwise_mult(out.id, self.id, o.id)
        else:
            self.ctx.vec_axpby(out.id, float(o), self.id, 0.0, self.id)
        return out

    __rmul__ = __mul__

    def __neg__(self):
        return self * -1.0

    def __iadd__(self, o):
        if isinstance(o, Vec):
            self.axpy(1.0, o)
        else:
            raise TypeError("Vec += scalar is not used by the reference path")
        return self

    def __isub__(self, o):
        self.axpy(-1.0, o)
        return self

    def __imul__(self, a):
        self.scale(a)
        return self
