#!/usr/bin/env python3
"""Convert the reference's shipped mocap clip into a plain .npz fixture.

Runs ONLY in the build container (needs /root/reference). Nothing from the
reference's Python is imported or executed: the pickle
(`clips/transform_snips_groom.p`, a pickled `mocap_preprocess.ReferenceClip`
whose leaves are jax arrays, see mocap_preprocess.py:326-340) is read with a
restricted Unpickler that maps the two non-numpy globals to inert stand-ins and
refuses everything else.  The result is data only (13 float32 arrays).

Usage:  python tools/make_fixtures.py
Writes: tests/golden/groom_clip.npz
"""
import io
import os
import pickle
import sys

import numpy as np

REF = "/root/reference/clips/transform_snips_groom.p"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "groom_clip.npz")


class _Bag:
    """Inert stand-in for the pickled dataclass; receives fields via __setstate__/__dict__."""

    def __setstate__(self, state):
        self.__dict__.update(state)


def _reconstruct_array(fun, args, arr_state, aval_state):
    # jax._src.array._reconstruct_array(fun, args, arr_state, aval_state):
    # fun/args rebuild the underlying numpy ndarray.
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return np.asarray(arr)


class _Restricted(pickle.Unpickler):
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"),
        ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"),
        ("numpy", "dtype"),
    }

    def find_class(self, module, name):
        if (module, name) == ("mocap_preprocess", "ReferenceClip"):
            return _Bag
        if (module, name) == ("jax._src.array", "_reconstruct_array"):
            return _reconstruct_array
        if (module, name) in self._ALLOWED:
            mod = __import__(module, fromlist=[name])
            return getattr(mod, name)
        raise pickle.UnpicklingError(f"refused global {module}.{name}")


def main():
    with open(REF, "rb") as f:
        bag = _Restricted(io.BytesIO(f.read())).load()
    fields = {k: np.asarray(v, dtype=np.float32) for k, v in vars(bag).items()}
    for k, v in sorted(fields.items()):
        print(f"{k:20s} {v.shape} {v.dtype}")
    np.savez_compressed(OUT, **fields)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
