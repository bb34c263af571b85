"""``DSEC`` (``src/dagr/data/dsec_data.py:58-184``) reads ``events_2x.h5`` (blosc-compressed HDF5) through the
third-party ``dsec-det`` package, h5py and hdf5plugin -- none of which exist in this stack.
The class is declared so that ``from dagr.data.dsec_data import DSEC`` resolves and fails with a precise message at
construction; ``dagr.data.synthetic_data.SyntheticWindows`` offers the same dataset interface on synthetic streams."""


class DSEC:
    def __init__(self, *args, **kwargs):
        missing = []
        for name in ("h5py", "hdf5plugin", "dsec_det"):
            try:
                __import__(name)
            except ImportError:
                missing.append(name)
        raise RuntimeError("the DSEC reader needs " + ", ".join(missing or ["dsec_det"]) + " (blosc HDF5 event files); "
                           "use dagr.data.synthetic_data.SyntheticWindows or feed Data objects with the same fields "
                           "(data/utils.py:to_data)")
