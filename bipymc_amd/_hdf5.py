"""The few calls of the HDF5 C library the checkpoint layout needs, through ctypes.

The reference writes its checkpoints with h5py (`chain.py:59-70`: one gzip dataset `/chains/chain_id_<i>` of shape (T, dim)
float64 per chain; read back by `chain.py:82-93` and `mc_plot/vis_mcmc_chains.py:16-41`).  h5py is a binding of libhdf5; where
h5py is not installed but the library is (this image: HDF5 1.10.6 under /opt/conda/lib), the same files can be produced and read
with the library itself -- what this module does.  Only what `bipymc_amd/checkpoint.py` uses is bound:
create / open a file, create a group, write and read a 1-D or 2-D float64 / int64 dataset (chunked + deflate like
`create_dataset(..., compression="gzip")`), test a link.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_lib = None
_CANDIDATES = ("libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310", "libhdf5_serial.so.103",
               "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so")

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
hid_t, hsize_t, herr_t, htri_t = C.c_int64, C.c_uint64, C.c_int, C.c_int


class Hdf5Error(IOError):
    pass


def load():
    """The library, or None when no libhdf5 (1.10 or later: 64-bit identifiers) can be loaded."""
    global _lib
    if _lib is not None:
        return _lib or None
    names = []
    if os.environ.get("BPM_HDF5_LIB"):
        names.append(os.environ["BPM_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5")
    if found:
        names.append(found)
    names.extend(_CANDIDATES)
    for n in names:
        try:
            lib = C.CDLL(n)
            maj, mi, rel = C.c_uint(), C.c_uint(), C.c_uint()
            if lib.H5open() < 0 or lib.H5get_libversion(C.byref(maj), C.byref(mi), C.byref(rel)) < 0:
                continue
            if (maj.value, mi.value) < (1, 10):
                continue
            _bind(lib)
            lib._version = (maj.value, mi.value, rel.value)
            _lib = lib
            return lib
        except (OSError, AttributeError):
            continue
    _lib = False
    return None


def _bind(lib):
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (herr_t, [hid_t]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]), "H5Pclose": (herr_t, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dclose": (herr_t, [hid_t]),
        "H5Lexists": (htri_t, [hid_t, C.c_char_p, hid_t]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)                       # no error stack on stderr: failures become Hdf5Error
    lib._f64 = hid_t.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
    lib._i64 = hid_t.in_dll(lib, "H5T_NATIVE_INT64_g").value
    lib._dcpl = hid_t.in_dll(lib, "H5P_CLS_DATASET_CREATE_ID_g").value


def _ck(v, what):
    if v < 0:
        raise Hdf5Error("libhdf5: %s failed" % what)
    return v


class File(object):
    """`with File(path, "w") as f: f.create_group("/chains"); f.write("/chains/chain_id_0", arr, gzip=True)`"""

    def __init__(self, path, mode="r"):
        self.lib = load()
        if self.lib is None:
            raise Hdf5Error("no libhdf5 could be loaded")
        p = os.fsencode(path)
        if mode == "w":
            self.id = _ck(self.lib.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate(%s)" % path)
        else:
            self.id = _ck(self.lib.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen(%s)" % path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.id is not None:
            self.lib.H5Fclose(self.id)
            self.id = None

    def create_group(self, name):
        g = _ck(self.lib.H5Gcreate2(self.id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2(%s)" % name)
        self.lib.H5Gclose(g)

    def exists(self, name):
        """every component of an absolute path must exist (H5Lexists fails on a missing intermediate group)"""
        parts = [p for p in name.split("/") if p]
        cur = ""
        for p in parts:
            cur += "/" + p
            if self.lib.H5Lexists(self.id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def write(self, name, data, gzip=False):
        """what h5py's create_dataset(name, data=data[, compression="gzip"]) stores: native little-endian float64 / int64, the
        array's shape; with gzip a chunked layout (one chunk of at most 1 MiB rows) with the deflate filter at level 4 (h5py's default)"""
        a = np.ascontiguousarray(data)
        if a.dtype.kind == "f":
            a, tid = a.astype(np.float64, copy=False), self.lib._f64
        else:
            a, tid = a.astype(np.int64, copy=False), self.lib._i64
        shape = a.shape if a.ndim else (1,)
        a = a.reshape(shape)
        dims = (hsize_t * len(shape))(*shape)
        space = _ck(self.lib.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")
        dcpl = H5P_DEFAULT
        if gzip and a.size > 0:
            dcpl = _ck(self.lib.H5Pcreate(self.lib._dcpl), "H5Pcreate")
            row_bytes = max(1, a.itemsize * int(np.prod(shape[1:], dtype=np.int64)))
            chunk = (max(1, min(shape[0], (1 << 20) // row_bytes)),) + tuple(shape[1:])
            _ck(self.lib.H5Pset_chunk(dcpl, len(chunk), (hsize_t * len(chunk))(*chunk)), "H5Pset_chunk")
            _ck(self.lib.H5Pset_deflate(dcpl, 4), "H5Pset_deflate")
        try:
            d = _ck(self.lib.H5Dcreate2(self.id, name.encode(), tid, space, H5P_DEFAULT, dcpl, H5P_DEFAULT), "H5Dcreate2(%s)" % name)
            try:
                _ck(self.lib.H5Dwrite(d, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), "H5Dwrite(%s)" % name)
            finally:
                self.lib.H5Dclose(d)
        finally:
            if dcpl != H5P_DEFAULT:
                self.lib.H5Pclose(dcpl)
            self.lib.H5Sclose(space)

    def read(self, name, dtype=np.float64):
        """the whole dataset as a NumPy array (converted by the library to float64 / int64 whatever it was stored as)"""
        d = _ck(self.lib.H5Dopen2(self.id, name.encode(), H5P_DEFAULT), "H5Dopen2(%s)" % name)
        try:
            space = _ck(self.lib.H5Dget_space(d), "H5Dget_space")
            try:
                nd = _ck(self.lib.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims")
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    _ck(self.lib.H5Sget_simple_extent_dims(space, dims, None), "H5Sget_simple_extent_dims")
                shape = tuple(int(dims[i]) for i in range(nd))
            finally:
                self.lib.H5Sclose(space)
            out = np.empty(shape, dtype=dtype)
            tid = self.lib._f64 if np.dtype(dtype).kind == "f" else self.lib._i64
            if out.size:
                _ck(self.lib.H5Dread(d, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), "H5Dread(%s)" % name)
            return out
        finally:
            self.lib.H5Dclose(d)
