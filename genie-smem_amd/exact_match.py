"""ExactMatch -- drop-in for the reference class of the same name (reference SMEM/ExactMatch.py:7).

Same constructor, attribute and method names, same return conventions (0-based inclusive
suffix-array interval tuple, the int -1 for "absent", KeyError for a symbol that does not
occur in the reference).  The search itself is the HIP kernel behind `genie_sa_interval`;
index construction is the native suffix-array builder instead of the reference's O(n^2)
rotation sort.  Files use the reference's own formats (FASTA, `<stem>-FM.json`).
"""
import json
import random
from os import path

import numpy as np

from .index import GenieIndex


class ExactMatch:

    def __init__(self, reference_sequence_file: str, query_sequence_file: str = None, data_dir: str = "data",
                 device="cuda"):
        self.ref_seq_file = reference_sequence_file
        self.data_dir = data_dir          # the reference hard-codes "data" relative to cwd
        self.device = device

        self.ref_sequence = None
        self.ref_size = None
        self.query_sequence = None
        self.fm_file = self.ref_seq_file.split(".")[0] + "-FM.json"
        self.fm_index = {}
        self.alphabet = None              # sorted symbols -> codes 0..3
        self._codes = None
        self._sa_from_file = None
        self._indexes = {}                # K -> GenieIndex

        if query_sequence_file is not None:
            self.load_query(query_sequence_file)

    # ------------------------------------------------------------------ files
    def load_ref_sequence(self):
        """ExactMatch.py:43-50: skip the header line, concatenate stripped lines, append '$'."""
        with open(path.join(self.data_dir, self.ref_seq_file), "r") as ref_file:
            ref_file.readline()
            self.ref_sequence = "".join(line.strip() for line in ref_file)
        self.set_reference(self.ref_sequence)

    def set_reference(self, sequence: str):
        """Use an in-memory reference string (without '$')."""
        seq = sequence[:-1] if sequence.endswith("$") else sequence
        symbols = sorted(set(seq))
        if set(symbols) <= set("ACGT"):
            symbols = list("ACGT")         # keep the LUT's fixed A0 C1 G2 T3 map (LUT.py:37-48)
        if len(symbols) > 4 or "$" in symbols:
            raise ValueError("the device path packs 2 bits per base: at most 4 distinct symbols, got "
                             + repr("".join(symbols)))
        self.alphabet = "".join(symbols)
        self._present = set(seq)
        lut = np.full(256, 255, np.uint8)
        for i, ch in enumerate(self.alphabet):
            lut[ord(ch)] = i
        self._enc = lut
        self._codes = lut[np.frombuffer(seq.encode("latin-1"), np.uint8)]
        self.ref_sequence = seq + "$"
        self.ref_size = len(self.ref_sequence)
        self._indexes = {}

    def encode(self, seq: str):
        """Symbols -> codes; KeyError for a symbol that never occurs in the reference, as the
        reference's count_dic lookup does (ExactMatch.py:139)."""
        if self._codes is None:
            self.load_ref_sequence()
        raw = np.frombuffer(seq.encode("latin-1", "replace"), np.uint8)
        codes = self._enc[raw]
        if codes.size and codes.max() == 255 or not set(seq) <= self._present:
            for ch in seq:
                if ch not in self._present:
                    raise KeyError(ch)
        return codes

    def decode(self, codes):
        return "".join(self.alphabet[int(c)] for c in codes)

    def index(self, K: int = 0) -> GenieIndex:
        """The device index for key size K (built once per K, SA reused from a loaded FM file)."""
        if self._codes is None:
            self.load_ref_sequence()
        if K not in self._indexes:
            sa = self._sa_from_file
            if sa is None and self._indexes:
                sa = next(iter(self._indexes.values())).suffix_array()
            ix = GenieIndex.build(self._codes, K, sa_one_based=sa)
            self._indexes[K] = ix
        ix = self._indexes[K]
        if ix.blob is None:
            ix.to(self.device)
        return ix

    def host_index(self, K: int = 0) -> GenieIndex:
        """Index without a device image (host arrays only: SA, LUT table)."""
        if self._codes is None:
            self.load_ref_sequence()
        if K not in self._indexes:
            sa = self._sa_from_file
            if sa is None and self._indexes:
                sa = next(iter(self._indexes.values())).suffix_array()
            self._indexes[K] = GenieIndex.build(self._codes, K, sa_one_based=sa)
        return self._indexes[K]

    def create_fm_index(self):
        """ExactMatch.py:22-33: build the index and write `<stem>-FM.json` in the reference's schema
        (bwt_array, suffix_array, occurance_matrix, count_dic, ref_size)."""
        self.load_ref_sequence()
        sa = np.asarray(self.host_index(0).suffix_array())
        text = np.frombuffer(self.ref_sequence.encode("latin-1"), np.uint8)
        n1 = len(text)
        bwt = text[(sa.astype(np.int64) - 2) % n1]                  # rotation[-1]
        first = text[(sa.astype(np.int64) - 1) % n1]                # rotation[0]
        occ = {}
        for sym in sorted(set(bwt.tolist())):
            occ[chr(sym)] = np.cumsum(bwt == sym).astype(np.int64).tolist()
        count_dic = {}
        for sym in sorted(set(first.tolist())):
            count_dic[chr(sym)] = int(np.argmax(first == sym))
        count_dic[""] = int(n1)
        self.fm_index = {"bwt_array": [chr(c) for c in bwt], "suffix_array": sa.tolist(),
                         "occurance_matrix": occ, "count_dic": count_dic, "ref_size": self.ref_size}
        with open(path.join(self.data_dir, self.fm_file), "w") as fm_file:
            fm_file.write(json.dumps(self.fm_index, indent=4, sort_keys=True))

    def load_fm_index(self):
        """ExactMatch.py:35-41.  The suffix array in the file is adopted (after validation)."""
        try:
            with open(path.join(self.data_dir, self.fm_file), "r") as fm_file:
                self.fm_index = json.load(fm_file)
                self.ref_size = self.fm_index["ref_size"]
        except FileNotFoundError:
            raise FileNotFoundError("No FM index file found. Run ExactMatch.createFMIndex to create an FM index.")
        if self._codes is None:
            self.load_ref_sequence()
        self._sa_from_file = np.asarray(self.fm_index["suffix_array"], np.int32)
        self._indexes = {}

    def load_query(self, query_seq_file):
        with open(path.join(self.data_dir, query_seq_file), "r") as q_file:
            self.query_sequence = "".join(line.strip() for line in q_file)

    def create_query(self, query_size, query_output_file=None):
        """ExactMatch.py:112-128."""
        if self.ref_sequence is None:
            self.load_ref_sequence()
        query_start = random.randint(0, self.ref_size - query_size - 1)
        query = self.ref_sequence[query_start: query_start + query_size]
        if query_output_file is not None:
            with open(path.join(self.data_dir, query_output_file), "w+") as q_out_f:
                q_out_f.write(">query from '" + self.ref_seq_file + "' zero-index location: " + str(query_start) + "\n")
                q_out_f.write("\n".join(query[i:i + 50] for i in range(0, len(query), 50)))
        self.query_sequence = query
        return query

    # ------------------------------------------------------------------ search (device)
    def exact_match_back_prop(self, query_seq: str):
        """ExactMatch.py:132-151: (lo, hi) 0-based inclusive SA rows, or -1."""
        codes = self.encode(query_seq)
        lohi = self.index(self._any_k()).sa_interval(codes.reshape(1, -1)).cpu().numpy()[0]
        return -1 if lohi[0] < 0 else (int(lohi[0]), int(lohi[1]))

    def exact_match_batch(self, patterns):
        """Batched form: list of strings -> int32 [N, 2] on the host ((-1,-1) = absent)."""
        enc = [self.encode(p) for p in patterns]
        width = max([len(e) for e in enc] + [1])
        mat = np.zeros((len(enc), width), np.uint8)
        lens = np.zeros(len(enc), np.int32)
        for i, e in enumerate(enc):
            mat[i, :len(e)] = e
            lens[i] = len(e)
        return self.index(self._any_k()).sa_interval(mat, lens).cpu().numpy()

    def exact_match_positions_batch(self, patterns):
        """Batched exact_match (ExactMatch.py:174-192): for every pattern the ascending 1-based
        positions of its occurrences ([] if absent), resolved on the device (genie_sa_interval +
        genie_locate)."""
        enc = [self.encode(p) for p in patterns]
        width = max([len(e) for e in enc] + [1])
        mat = np.zeros((len(enc), width), np.uint8)
        lens = np.zeros(len(enc), np.int32)
        for i, e in enumerate(enc):
            mat[i, :len(e)] = e
            lens[i] = len(e)
        ix = self.index(self._any_k())
        off, pos = ix.locate(ix.sa_interval(mat, lens), sort=True)
        off, pos = off.cpu().numpy(), pos.cpu().numpy()
        return [pos[off[i]:off[i + 1]].tolist() for i in range(len(enc))]

    def exact_match_back_prop_add_one(self, char, prev_suffix_tuple):
        """ExactMatch.py:155-171: one backward-search step.  Host-side helper (not used by the
        batched path): rows of `char`'s bucket are ordered by the rank of the suffix that follows,
        so the new interval is a range of ranks inside that bucket (LF-mapping via the inverse SA)."""
        if char not in self._present_with_dollar():
            raise KeyError(char)
        sa = np.asarray(self.host_index(self._any_k()).suffix_array(), np.int64)
        if getattr(self, "_isa", None) is None or len(self._isa) != len(sa):
            self._isa = np.empty(len(sa), np.int64)
            self._isa[sa - 1] = np.arange(len(sa))
            text = np.frombuffer(self.ref_sequence.encode("latin-1"), np.uint8)
            self._first = text[sa - 1]
        rows = np.nonzero(self._first == ord(char))[0]               # contiguous bucket of `char`
        nxt = self._isa[sa[rows] % len(sa)]                          # rank of the following suffix
        a = int(np.searchsorted(nxt, prev_suffix_tuple[0], "left"))
        b = int(np.searchsorted(nxt, prev_suffix_tuple[1], "right"))
        if a >= b:
            return -1
        return int(rows[a]), int(rows[b - 1])

    def exact_match(self, query_seq: str = None):
        """ExactMatch.py:174-188: sorted 1-based positions of all matches."""
        if query_seq is None:
            if self.query_sequence is None:
                print("No query sequence. Either input sequence, load_file, or generate.")
                return
            query_seq = self.query_sequence
        start, end = self.exact_match_back_prop(query_seq)            # TypeError on -1, as the reference
        return sorted(self.get_positions(start, end))

    def get_position(self, suffix_array_index):
        return int(self.host_index(self._any_k()).suffix_array()[suffix_array_index])

    def get_positions(self, suffix_start, suffix_end):
        sa = self.host_index(self._any_k()).suffix_array()
        return [int(x) for x in sa[suffix_start:suffix_end + 1]]

    # ------------------------------------------------------------------ helpers
    def _any_k(self):
        return next(iter(self._indexes)) if self._indexes else 0

    def _present_with_dollar(self):
        if self._codes is None:
            self.load_ref_sequence()
        return self._present | {"$"}
