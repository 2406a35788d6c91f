"""Checkpoint files (reference: demc.py:198-233, chain.py:59-93).

On-disk layout of the reference: HDF5, one gzip dataset `/chains/chain_id_<global_id>` of shape (T, dim) float64 per chain
(`McmcChain.write_chain_h5`, chain.py:59-70; readers chain.py:82-93 and mc_plot/vis_mcmc_chains.py:16-41).  Three backends
write / read exactly that layout, tried in this order:

  1. h5py, when it is importable (what the reference itself uses);
  2. the HDF5 C library through ctypes (`bipymc_amd/_hdf5.py`) -- h5py is a binding of the same library, the files are the same;
  3. a NumPy `.npz` archive with the same keys (`chains/chain_id_<i>`) and shapes, when neither is available or the path ends in `.npz`.

The adaptation state the reference forgets (p_cr, delta_m, n_cr_updates, generation counter, seed) goes to the side group
`/bipymc_amd` (files without it -- the reference's own -- read fine).  `read()` decides by what is ON DISK: an HDF5 signature means
HDF5, whatever this interpreter could have written.
"""
import os

import numpy as np

ADAPT_KEYS = ("t_abs", "seed", "p_cr", "delta_m", "n_cr_updates")


def _have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


def _have_libhdf5():
    from . import _hdf5
    return _hdf5.load() is not None


def hdf5_backend():
    """"h5py", "libhdf5" (ctypes) or None"""
    if _have_h5py():
        return "h5py"
    if _have_libhdf5():
        return "libhdf5"
    return None


def _is_hdf5(path):
    """HDF5 signature (the 8 bytes every HDF5 file starts with, at offset 0 for files h5py / libhdf5 write)."""
    try:
        with open(path, "rb") as f:
            return f.read(8) == b"\x89HDF\r\n\x1a\n"
    except (IOError, OSError):
        return False


def write(path, hist, adapt=None):
    """hist: (T, N, dim) float64.  Returns the path actually written."""
    hist = np.asarray(hist, dtype=np.float64)
    T, N, d = hist.shape
    adapt = adapt or {}
    path = str(path)
    backend = None if path.endswith(".npz") else hdf5_backend()
    if backend == "h5py":
        import h5py
        with h5py.File(path, "w") as f:
            for i in range(N):
                f.create_dataset("/chains/chain_id_" + str(i), data=hist[:, i, :], compression="gzip")
            g = f.create_group("/bipymc_amd")
            for k, v in adapt.items():
                g.create_dataset(k, data=np.asarray(v))
        return path
    if backend == "libhdf5":
        from . import _hdf5
        with _hdf5.File(path, "w") as f:
            f.create_group("/chains")
            for i in range(N):
                f.write("/chains/chain_id_" + str(i), hist[:, i, :], gzip=True)
            f.create_group("/bipymc_amd")
            for k, v in adapt.items():
                f.write("/bipymc_amd/" + k, np.atleast_1d(np.asarray(v)))
        return path
    arrays = {"chains/chain_id_" + str(i): hist[:, i, :] for i in range(N)}
    for k, v in adapt.items():
        arrays["bipymc_amd/" + k] = np.asarray(v)
    target = path if path.endswith(".npz") else path + ".npz"
    np.savez_compressed(target, **arrays)
    return target


def read(path, n_chains, dim):
    """-> (hist (T, N, dim), adapt dict).  The format is decided by what is ON DISK, not by what this interpreter could
    write: `path` itself if it is an HDF5 file (the reference's write_chain_h5 layout, chain.py:59-70; needs h5py or libhdf5), else
    the NumPy twin `path` (when it ends in .npz) or `path + ".npz"`."""
    adapt = {}
    chains = []
    path = str(path)
    if os.path.exists(path) and _is_hdf5(path):
        backend = hdf5_backend()
        if backend is None:
            raise IOError("checkpoint %s is an HDF5 file but neither h5py nor libhdf5 can be loaded here; install one or convert it to "
                          "the .npz layout (keys chains/chain_id_<i>)" % path)
        if backend == "h5py":
            import h5py
            with h5py.File(path, "r") as f:
                for i in range(n_chains):
                    chains.append(f["/chains/chain_id_" + str(i)][:])
                if "bipymc_amd" in f:
                    for k in f["bipymc_amd"]:
                        adapt[k] = f["bipymc_amd"][k][()]
        else:
            from . import _hdf5
            with _hdf5.File(path, "r") as f:
                for i in range(n_chains):
                    chains.append(f.read("/chains/chain_id_" + str(i)))
                for k in ADAPT_KEYS:
                    if f.exists("/bipymc_amd/" + k):
                        if k in ("t_abs", "seed"):
                            adapt[k] = int(f.read("/bipymc_amd/" + k, dtype=np.int64).reshape(-1)[0])
                        else:
                            adapt[k] = f.read("/bipymc_amd/" + k)
    else:
        target = path if path.endswith(".npz") else path + ".npz"
        if not os.path.exists(target):
            raise IOError("checkpoint not found: neither an HDF5 file %s nor its NumPy twin %s" % (path, target))
        with np.load(target) as f:
            for i in range(n_chains):
                chains.append(f["chains/chain_id_" + str(i)])
            for k in f.files:
                if k.startswith("bipymc_amd/"):
                    adapt[k.split("/", 1)[1]] = f[k][()]
    T = chains[0].shape[0]
    for c in chains:
        if c.shape != (T, dim):
            raise RuntimeError("checkpoint chains have unequal shapes")      # demc.py:229-232
    return np.stack(chains, axis=1), adapt
