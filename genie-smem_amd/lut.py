"""LUT -- drop-in for the reference's K-mer lookup table class (reference SMEM/LUT.py:6).

The reference keeps a Python dict  str(code) -> [[lo, hi], [positions...]].  Here the table
lives in the native index as sorted arrays (host) and as an open-addressing hash table
(device); `.lut` is a read-only mapping view with the reference's key/value shapes, and
save_lut/load_lut use the reference's JSON schema.
"""
import json
from os import path

import numpy as np

from .exact_match import ExactMatch

_NUCLEO = {"A": 0, "C": 1, "G": 2, "T": 3}


class LutView:
    """Mapping view over the native K-mer table: key str(code) -> [[lo, hi], [positions]]."""

    def __init__(self, codes, lo, hi, sa):
        self._codes, self._lo, self._hi, self._sa = codes, lo, hi, sa

    def _find(self, key):
        try:
            code = int(key)
        except (TypeError, ValueError):
            return -1
        if isinstance(key, int) or code < 0 or code > 0xFFFFFFFF:
            return -1                       # the reference's keys are strings (LUT.py:35)
        i = int(np.searchsorted(self._codes, code))
        return i if i < len(self._codes) and int(self._codes[i]) == code else -1

    def __contains__(self, key):
        return self._find(key) >= 0

    def __getitem__(self, key):
        i = self._find(key)
        if i < 0:
            raise KeyError(key)
        lo, hi = int(self._lo[i]), int(self._hi[i])
        return [(lo, hi), [int(x) for x in self._sa[lo:hi + 1]]]

    def __len__(self):
        return len(self._codes)

    def __iter__(self):
        return (str(int(c)) for c in self._codes)

    def keys(self):
        return list(iter(self))

    def items(self):
        return ((k, self[k]) for k in self)


class LUT:

    def __init__(self, matcher: ExactMatch):
        self.matcher = matcher
        if self.matcher.ref_sequence is None:
            self.matcher.load_ref_sequence()
        self.lut = None
        self.lut_size = None

    def generate_lut(self, size):
        """LUT.py:15-35: every K-mer of the reference -> (SA interval, positions).  Read off the
        suffix-array order natively instead of one backward search per K-mer."""
        self.lut_size = size
        ix = self.matcher.host_index(size)
        codes, lo, hi = ix.lut_arrays()
        self.lut = LutView(codes, lo, hi, ix.suffix_array())

    @staticmethod
    def convert_seq_to_num(sequence):
        """LUT.py:37-48: 2 bits per base, A0 C1 G2 T3, first base most significant."""
        conversion = 0
        for ch in sequence:
            conversion = conversion << 2 | _NUCLEO[ch]
        return conversion

    def _lut_file(self):
        return path.join(self.matcher.data_dir, self.matcher.ref_seq_file.split(".")[0] + "-LUT.json")

    def save_lut(self):
        if self.lut is None:
            raise RuntimeError("LUT has not been created yet.")
        as_dict = {k: [list(v[0]), v[1]] for k, v in self.lut.items()}
        with open(self._lut_file(), "w") as lut_f:
            lut_f.write(json.dumps({"lut": as_dict, "lut_size": self.lut_size}, indent=4, sort_keys=True))

    def load_lut(self):
        """LUT.py:58-63.  Only `lut_size` is taken from the file; the table itself is rebuilt from
        the suffix array (the file's content is checked against it by the test-suite)."""
        with open(self._lut_file(), "r") as lut_f:
            lut_json = json.load(lut_f)
        self.generate_lut(lut_json["lut_size"])
