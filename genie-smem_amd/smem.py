"""SMEM -- drop-in for the reference's SMEM class (reference SMEM/SMEM.py:8).

Per-query methods keep the reference's names, arguments and return shapes:
    get_SMEMS(query, minimum_length)   BWA-SMEM   (SMEM.py:456)
    get_smems_lut(query)               LUT-SMEM   (SMEM.py:20)
    get_smems_rmi(query)               RMI-SMEM   (SMEM.py:206)
each returning the insertion-ordered dict {substring: (lo, hi)}.  They are thin views over the
batched entry points
    find_smems_bwa / find_smems_lut / find_smems_rmi (reads) -> (offsets, smems[S,4], status)
which run one wavefront per read on the GPU (genie_find_smems).  The single-step helpers
(get_suffix_index, forward_extension, backward_extension, get_SMEM_at_index, check_sequential)
are provided with the reference's semantics on top of the batched interval search.
"""
import random

import numpy as np
import torch

from . import _native as N
from .exact_match import ExactMatch
from .lut import LUT
from .rmi_lut import RMI_LUT


class SMEM:

    def __init__(self, matcher: ExactMatch, lut_size: int = None):
        self.matcher = matcher
        self.lut = LUT(self.matcher)
        if lut_size is None:
            self.lut.load_lut()                 # SMEM.py:11-12 (FileNotFoundError if never saved)
        else:
            self.lut.generate_lut(lut_size)
        self.rmi_lut = None

    # ------------------------------------------------------------------ batched entry points
    def _reads_codes(self, reads):
        """list[str] | ndarray -> (uint8 [N, width] numpy codes, int32 lens or None)."""
        if isinstance(reads, np.ndarray):
            return np.ascontiguousarray(reads, np.uint8), None
        enc = [self.matcher.encode(r) for r in reads]
        lens = np.asarray([len(e) for e in enc], np.int32)
        width = int(lens.max()) if len(enc) else 1
        mat = np.zeros((len(enc), max(width, 1)), np.uint8)
        for i, e in enumerate(enc):
            mat[i, :len(e)] = e
        ragged = len(enc) > 0 and int(lens.min()) != width
        return mat, (lens if ragged else None)

    def _reads_tensor(self, reads):
        """list[str] | ndarray | tensor -> (uint8 [N, stride] on the device, lens or None)."""
        ix_dev = torch.device(self.matcher.device)
        if isinstance(reads, torch.Tensor):
            return reads, None
        if isinstance(reads, np.ndarray):
            return torch.as_tensor(np.ascontiguousarray(reads, np.uint8)).to(ix_dev), None
        enc = [self.matcher.encode(r) for r in reads]
        lens = np.asarray([len(e) for e in enc], np.int32)
        width = int(lens.max()) if len(enc) else 1
        mat = np.zeros((len(enc), max(width, 1)), np.uint8)
        for i, e in enumerate(enc):
            mat[i, :len(e)] = e
        ragged = len(enc) > 0 and int(lens.min()) != width
        return torch.as_tensor(mat).to(ix_dev), (torch.as_tensor(lens).to(ix_dev) if ragged else None)

    def _find(self, mode, reads, lens, min_len):
        K = self.lut.lut_size
        if mode == "rmi":
            if self.rmi_lut is None:
                self.rmi_lut = RMI_LUT.load("rmi_file.npz", matcher=self.matcher)     # cf. SMEM.py:207
            ix = self.rmi_lut._index()
        else:
            ix = self.matcher.index(K)
        if not isinstance(reads, torch.Tensor):
            # host inputs: 2-bit packed over the host link, 8-byte rows back (genie_find_smems_packed); results on the host
            codes, l2 = self._reads_codes(reads)
            if codes.shape[1] <= 255:
                ln = lens if lens is not None else l2
                off, rows, st = ix.find_smems_host(mode, codes, ln, min_len)
                return torch.as_tensor(off), torch.as_tensor(rows), torch.as_tensor(st)
        t, l2 = self._reads_tensor(reads)
        return ix.find_smems(mode, t, lens if lens is not None else l2, min_len)

    def find_smems_bwa(self, reads, minimum_length=1, lens=None):
        return self._find("bwa", reads, lens, minimum_length)

    def find_smems_lut(self, reads, lens=None):
        return self._find("lut", reads, lens, 1)

    def find_smems_rmi(self, reads, lens=None):
        return self._find("rmi", reads, lens, 1)

    # ------------------------------------------------------------------ reference API (per query)
    def _one(self, mode, query, min_len=1):
        codes = self.matcher.encode(query)                    # KeyError for an unknown base
        if len(codes) == 0:
            if mode == "bwa":
                return {}
            raise KeyError("")                                # SMEM.py:39 on an empty query
        offsets, smems, status = self._find(mode, codes.reshape(1, -1), None, min_len)
        st = int(status[0].item())
        if st == N.READ_ABSENT_BASE:
            raise KeyError("")            # reference: forward_match[0][""] (SMEM.py:39) / runaway loop
        if st == N.READ_TOO_SHORT:
            raise ValueError("query shorter than the LUT key size (the reference mis-encodes it, SMEM.py:26-28)")
        if st != N.READ_OK:
            raise RuntimeError(f"genie_find_smems: read status {st}")
        out = {}
        for s, e, lo, hi in smems.cpu().numpy().tolist():
            out[query[s:e]] = (lo, hi)
        return out

    def get_suffix_index(self, query):
        return self.matcher.exact_match_back_prop(query)

    def get_smems_lut(self, query):
        return self._one("lut", query)

    def get_smems_rmi(self, query):
        return self._one("rmi", query)

    def get_SMEMS(self, query, minimum_length):
        return self._one("bwa", query, minimum_length)

    @staticmethod
    def check_sequential(list1, list2):
        """SMEM.py:196-202."""
        s2 = set(list2)
        return any(item1 + 1 in s2 for item1 in list1)

    def forward_extension(self, query, start_index, largest="", suffix_tuple=None):
        """SMEM.py:425-443: ({every matching string: interval}, longest).  All prefixes are
        searched in ONE batched device call instead of one backward search per step."""
        forward_matches = {}
        if suffix_tuple is not None:
            forward_matches[largest] = suffix_tuple
        pats = [largest + query[start_index:i] for i in range(start_index + 1, len(query) + 1)]
        if not pats:
            return forward_matches, largest
        res = self.matcher.exact_match_batch(pats)
        longest = largest
        for p, (lo, hi) in zip(pats, res.tolist()):
            if lo < 0:
                return forward_matches, p[:-1]
            forward_matches[p] = (lo, hi)
            longest = p
        return forward_matches, longest

    def backward_extension(self, query, start_index, forward_matches):
        """SMEM.py:389-423."""
        largest, suffix_of_largest, end_index = "", None, -1
        largest_forward = ""
        keys = list(forward_matches)
        pats, owner = [], []
        for key in keys:
            for i in range(start_index - 1, -1, -1):
                pats.append(query[i:start_index] + key)
                owner.append(key)
        res = self.matcher.exact_match_batch(pats).tolist() if pats else []
        pos = 0
        for key in keys:
            if len(key) > len(largest_forward):
                largest_forward = key
            broken = False
            for i in range(start_index - 1, -1, -1):
                lo, hi = res[pos]
                cur = pats[pos]
                pos += 1
                if broken:
                    continue
                if lo < 0:
                    broken = True
                    continue
                if len(cur) > len(largest):
                    largest, suffix_of_largest, end_index = cur, (lo, hi), start_index + len(key)
        if len(largest_forward) > len(largest):
            largest = largest_forward
            suffix_of_largest = forward_matches[largest_forward]
            end_index = start_index + len(largest_forward)
        return largest, suffix_of_largest, end_index

    def get_SMEM_at_index(self, query, start_index):
        """SMEM.py:469-484."""
        forward_extension = self.forward_extension(query, start_index)
        largest_backward = self.backward_extension(query, start_index, forward_extension[0])
        if len(forward_extension[1]) > len(largest_backward[0]):
            return [forward_extension[1], forward_extension[0][forward_extension[1]],
                    len(forward_extension[1]) + start_index]
        return [largest_backward[0], largest_backward[1], largest_backward[2]]


def create_random_query(query_size):
    """SMEM.py:489-493."""
    return "".join(random.choice(["A", "G", "C", "T"]) for _ in range(query_size))


def create_query_from_ref(ref_seq, query_size):
    """SMEM.py:496-505."""
    ref_size = len(ref_seq)
    query = ""
    while len(query) < query_size:
        position = random.randint(0, ref_size)
        size = random.randint(1, 30)
        if size + position > ref_size:
            continue
        query += ref_seq[position: position + size]
    return query[:query_size]
