"""RMI_LUT -- drop-in for the reference's learned-index seed (reference SMEM/RMI_LUT.py:10).

Same constructor and method names.  `get_suffix_rmi` runs on the device (`genie_seed_lookup`,
RMI mode): prediction with the staged linear models, then a last-mile search on the suffix
array.  Contract: the interval returned is always the TRUE one -- the
reference's last-mile search mis-resolves a handful of K-mers and can recurse forever; those
defects are not reproduced.  "Absent" keeps the reference's convention lower > upper.
Models are saved as plain .npz arrays, not pickles.
"""
import numpy as np

from .exact_match import ExactMatch
from .rmi import RMI


class RMI_LUT:

    def __init__(self, RMI_structure, LUT_size, data_file, matcher: ExactMatch = None, data_dir: str = "data",
                 device="cuda"):
        self.nucleo = {"A": 0, "C": 1, "G": 2, "T": 3}
        self.structure = list(RMI_structure)
        self.prediction_size = LUT_size
        self.data_file = data_file
        if matcher is None:
            matcher = ExactMatch(data_file, data_dir=data_dir, device=device)
            matcher.load_ref_sequence()
        self.matcher = matcher
        self.ref_seq = matcher.ref_sequence[:-1]
        self.ref_seq_size = len(self.ref_seq)
        self.rmi = RMI(self.structure)
        self._installed = False

    @property
    def suffix_array(self):
        return self.matcher.host_index(self.prediction_size).suffix_array()

    def train_RMI(self, native=True):
        """RMI_LUT.py:36-50: (K-mer code, SA row) pairs for every row whose suffix holds a full
        K-mer, in SA order; then RMI.fit.  native=True (default): the C++ trainer behind
        genie_index_train_rmi, which also records per-leaf error bounds for the device's last-mile
        search; native=False: the numpy restatement in rmi.py."""
        K = self.prediction_size
        if native:
            ix = self.matcher.host_index(K)
            if ix.blob is not None:                      # image already uploaded: train on a fresh host handle
                self.matcher._indexes.pop(K)
                ix = self.matcher.host_index(K)
            coefs, icpts, err, mean, worst = ix.train_rmi(self.structure)
            self.rmi = RMI.from_coefficients(self.structure, coefs, icpts)
            self.leaf_error = err
            self.fit_error = (mean, worst)
            self._installed = True
            return
        sa = np.asarray(self.suffix_array, np.int64)
        rows = np.nonzero(sa - 1 + K <= self.ref_seq_size)[0]
        codes = self.matcher._codes.astype(np.int64)
        starts = sa[rows] - 1
        key = np.zeros(len(rows), np.int64)
        for j in range(K):
            key = (key << 2) | codes[starts + j]
        self.rmi.fit(key.reshape(-1, 1), rows)
        self._installed = False

    def set_model(self, experts, coefs, icpts):
        """Install coefficients exported from elsewhere (e.g. the reference's sklearn models)."""
        self.structure = list(experts)
        self.rmi = RMI.from_coefficients(experts, coefs, icpts)
        self._installed = False

    def _index(self):
        ix = self.matcher.host_index(self.prediction_size)
        if not self._installed:
            if not self.rmi.models:
                raise RuntimeError("RMI has not been trained (call train_RMI or load)")
            if ix.blob is not None:                      # image already uploaded without this model: rebuild
                self.matcher._indexes.pop(self.prediction_size)
                ix = self.matcher.host_index(self.prediction_size)
            coefs, icpts = self.rmi.coefficients()
            ix.set_rmi(self.structure, coefs, icpts)
            self._installed = True
        if ix.blob is None:
            ix.to(self.matcher.device)
        return ix

    def _encode(self, query, encoded):
        if encoded:
            return int(query)
        code = 0
        for j in range(self.prediction_size):
            code = code << 2 | self.nucleo[query[j]]       # KeyError on a non-ACGT base (:60)
        return code

    def rmi_predict(self, query, encoded=False):
        """RMI_LUT.py:53-63: float64 prediction (array of one element, like the reference)."""
        return self.rmi.predict(np.asarray([self._encode(query, encoded)]).reshape(-1, 1))

    def get_suffix_rmi(self, query, encoded=False):
        """RMI_LUT.py:67-78 on the device: (lower, upper); absent <=> lower > upper."""
        code = self._encode(query, encoded)
        K = self.prediction_size
        kmer = np.asarray([(code >> (2 * (K - 1 - j))) & 3 for j in range(K)], np.uint8).reshape(1, K)
        lohi = self._index().seed_lookup("rmi", kmer).cpu().numpy()[0]
        return int(lohi[0]), int(lohi[1])

    def get_ref_seq(self, ind):
        """RMI_LUT.py:89-92."""
        sa = self.suffix_array
        if sa[ind] - 1 + self.prediction_size > self.ref_seq_size:
            return None
        return self.ref_seq[sa[ind] - 1:(sa[ind] + self.prediction_size) - 1]

    def save(self, file):
        coefs, icpts = self.rmi.coefficients()
        arrays = {f"coef{l}": c for l, c in enumerate(coefs)}
        arrays.update({f"icpt{l}": c for l, c in enumerate(icpts)})
        np.savez(file, structure=np.asarray(self.structure, np.int64), K=np.asarray(self.prediction_size),
                 data_file=np.asarray(self.data_file), **arrays)

    @staticmethod
    def load(file, matcher: ExactMatch = None, data_dir: str = "data", device="cuda"):
        z = np.load(file if str(file).endswith(".npz") else str(file) + ".npz", allow_pickle=False)
        structure = [int(x) for x in z["structure"]]
        new = RMI_LUT(structure, int(z["K"]), str(z["data_file"]), matcher=matcher, data_dir=data_dir, device=device)
        nlev = len(structure) + 1
        new.set_model(structure, [z[f"coef{l}"] for l in range(nlev)], [z[f"icpt{l}"] for l in range(nlev)])
        return new
