"""Minimal read-side HDF5 access through ctypes (h5py is not installed): enough to walk the loop file tree."""
import ctypes
import os

import numpy as np

_CANDS = [os.environ.get("MUGIQ_HIP_HDF5_LIB"), "libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so"]


class H5:
    def __init__(self):
        self.lib = None
        for c in _CANDS:
            if not c:
                continue
            try:
                self.lib = ctypes.CDLL(c)
                break
            except OSError:
                pass
        if self.lib is None:
            raise ImportError("libhdf5 not found")
        L = self.lib
        hid = ctypes.c_int64
        L.H5open.restype = ctypes.c_int
        L.H5Fopen.restype, L.H5Fopen.argtypes = hid, [ctypes.c_char_p, ctypes.c_uint, hid]
        L.H5Fclose.argtypes = [hid]
        L.H5Dopen2.restype, L.H5Dopen2.argtypes = hid, [hid, ctypes.c_char_p, hid]
        L.H5Dclose.argtypes = [hid]
        L.H5Dget_space.restype, L.H5Dget_space.argtypes = hid, [hid]
        L.H5Dget_type.restype, L.H5Dget_type.argtypes = hid, [hid]
        L.H5Tget_size.restype, L.H5Tget_size.argtypes = ctypes.c_size_t, [hid]
        L.H5Sget_simple_extent_dims.argtypes = [hid, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
        L.H5Sget_simple_extent_ndims.argtypes = [hid]
        L.H5Dread.argtypes = [hid, hid, hid, hid, hid, ctypes.c_void_p]
        L.H5Lexists.restype, L.H5Lexists.argtypes = ctypes.c_int, [hid, ctypes.c_char_p, hid]
        L.H5Gget_num_objs.argtypes = [hid, ctypes.POINTER(ctypes.c_ulonglong)]
        L.H5Gopen2.restype, L.H5Gopen2.argtypes = hid, [hid, ctypes.c_char_p, hid]
        L.H5Gclose.argtypes = [hid]
        L.H5Gget_objname_by_idx.restype = ctypes.c_ssize_t
        L.H5Gget_objname_by_idx.argtypes = [hid, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.c_size_t]
        L.H5open()
        self.f64 = hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
        self.f32 = hid.in_dll(L, "H5T_NATIVE_FLOAT_g").value

    def open(self, path):
        fid = self.lib.H5Fopen(path.encode(), 0, 0)
        assert fid >= 0, path
        return fid

    def close(self, fid):
        self.lib.H5Fclose(fid)

    def exists(self, fid, path):
        cur = ""
        for part in path.strip("/").split("/"):
            cur += "/" + part
            if self.lib.H5Lexists(fid, cur.encode(), 0) <= 0:
                return False
        return True

    def children(self, fid, path):
        g = self.lib.H5Gopen2(fid, path.encode(), 0)
        assert g >= 0, path
        n = ctypes.c_ulonglong()
        self.lib.H5Gget_num_objs(g, ctypes.byref(n))
        out = []
        for i in range(n.value):
            buf = ctypes.create_string_buffer(256)
            self.lib.H5Gget_objname_by_idx(g, i, buf, 256)
            out.append(buf.value.decode())
        self.lib.H5Gclose(g)
        return out

    def read(self, fid, path):
        d = self.lib.H5Dopen2(fid, path.encode(), 0)
        assert d >= 0, path
        sp = self.lib.H5Dget_space(d)
        nd = self.lib.H5Sget_simple_extent_ndims(sp)
        dims = (ctypes.c_ulonglong * nd)()
        self.lib.H5Sget_simple_extent_dims(sp, dims, None)
        tsize = self.lib.H5Tget_size(self.lib.H5Dget_type(d))
        dt, mem = (np.float64, self.f64) if tsize == 8 else (np.float32, self.f32)
        out = np.zeros(tuple(dims), dtype=dt)
        assert self.lib.H5Dread(d, mem, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)) >= 0
        self.lib.H5Dclose(d)
        return out
