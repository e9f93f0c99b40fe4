"""Append-able HDF5 run log (reference: helper_functions/data_management.py:10-214, datasets listed in SURVEY.md 5).

h5py is an optional dependency: it is imported on first use so that `save=False` runs need nothing but NumPy.
"""
import numpy as np


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError as exc:                                    # pragma: no cover - depends on the environment
        raise ImportError('save=True needs h5py for the HDF5 run log; install h5py or call Grape(..., save=False)') from exc


def H5File(path, mode='a'):
    """Open (create) a run log; returns an h5py.File subclass instance with add()/append()."""
    h5py = _h5py()

    class _RunLog(h5py.File):
        def add(self, key, data):
            data = np.array(data)
            if data.dtype.kind == 'U':
                data = data.astype('S')
            if key in self:
                del self[key]
            self.create_dataset(key, shape=data.shape, maxshape=tuple([None] * len(data.shape)), dtype=data.dtype)
            self[key][...] = data

        def append(self, key, data, force_append=False):
            data = np.array(data)
            if key not in self:
                self.create_dataset(key, shape=(1,) + data.shape, maxshape=(None,) + data.shape, dtype=data.dtype)
                self[key][0] = data
                return
            ds = self[key]
            ds.resize(ds.shape[0] + 1, axis=0)
            ds[-1] = data

    return _RunLog(path, mode)
