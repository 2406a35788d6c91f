"""Checkpoint files (reference: demc.py:198-233, chain.py:59-93).

On-disk layout of the reference: HDF5, one gzip dataset `/chains/chain_id_<global_id>` of
shape (T, dim) float64 per chain.  Written/read with h5py when it is importable; this image
has no h5py, so the same logical layout is also supported as a NumPy `.npz` archive (keys
`chains/chain_id_<i>`), chosen by file extension or as the fallback.  The adaptation state
the reference forgets (p_cr, delta_m, n_cr_updates, generation counter, seed) goes to the
side group `/bipymc_amd`.
"""
import numpy as np


def _have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


def _use_h5(path):
    return (not str(path).endswith(".npz")) and _have_h5py()


def write(path, hist, adapt=None):
    """hist: (T, N, dim) float64."""
    hist = np.asarray(hist, dtype=np.float64)
    T, N, d = hist.shape
    adapt = adapt or {}
    if _use_h5(path):
        import h5py
        with h5py.File(path, "w") as f:
            for i in range(N):
                f.create_dataset("/chains/chain_id_" + str(i), data=hist[:, i, :], compression="gzip")
            g = f.create_group("/bipymc_amd")
            for k, v in adapt.items():
                g.create_dataset(k, data=np.asarray(v))
        return path
    arrays = {"chains/chain_id_" + str(i): hist[:, i, :] for i in range(N)}
    for k, v in adapt.items():
        arrays["bipymc_amd/" + k] = np.asarray(v)
    target = path if str(path).endswith(".npz") else str(path) + ".npz"
    np.savez_compressed(target, **arrays)
    return target


def _is_hdf5(path):
    """HDF5 signature (the 8 bytes every HDF5 file starts with, at offset 0 for files h5py writes)."""
    try:
        with open(path, "rb") as f:
            return f.read(8) == b"\x89HDF\r\n\x1a\n"
    except (IOError, OSError):
        return False


def read(path, n_chains, dim):
    """-> (hist (T, N, dim), adapt dict).  The format is decided by what is ON DISK, not by what this interpreter could
    write: `path` itself if it is an HDF5 file (the reference's write_chain_h5 layout, chain.py:59-70; needs h5py), else the
    NumPy twin `path` (when it ends in .npz) or `path + ".npz"`."""
    import os
    adapt = {}
    chains = []
    path = str(path)
    if os.path.exists(path) and _is_hdf5(path):
        if not _have_h5py():
            raise IOError("checkpoint %s is an HDF5 file but h5py is not importable here; install h5py or convert it to the "
                          ".npz layout (keys chains/chain_id_<i>)" % path)
        import h5py
        with h5py.File(path, "r") as f:
            for i in range(n_chains):
                chains.append(f["/chains/chain_id_" + str(i)][:])
            if "bipymc_amd" in f:
                for k in f["bipymc_amd"]:
                    adapt[k] = f["bipymc_amd"][k][()]
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
