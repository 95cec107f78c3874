"""ctypes binding of libcoati_host.so -- the C++ host layer (models, sequence
preparation, synthetic workload).  Plumbing for tests and bench.py."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent
LIB_PATH = _ROOT / "_build" / "libcoati_host.so"

DEFAULT_PI = (0.308, 0.185, 0.199, 0.308)
_lib = None


class CoatiHostError(ValueError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        path = Path(os.environ.get("COATI_HOST_LIB", LIB_PATH))
        if not path.exists():
            raise ImportError(f"{path} not found: build it with `make host` (or __graft_entry__.build())")
        _lib = C.CDLL(str(path))
        _lib.coati_host_last_error.restype = C.c_char_p
    return _lib


def _check(rc):
    if rc != 0:
        raise CoatiHostError(rc, load().coati_host_last_error().decode(errors="replace"))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f4(v):
    return np.ascontiguousarray(v, np.float32)


def gap_consts(gap_open=0.001, gap_extend=None) -> np.ndarray:
    e = np.float32(1.0) - np.float32(1.0) / np.float32(6.0) if gap_extend is None else np.float32(gap_extend)
    out = np.zeros(4, np.float32)
    _check(load().coati_host_gap_consts(C.c_float(np.float32(gap_open)), C.c_float(e), _p(out)))
    return out


def mg94_p(br_len=0.0133, omega=0.2, pi=DEFAULT_PI, sigma=None) -> np.ndarray:
    out = np.zeros((61, 61), np.float32)
    sg = None if sigma is None else _f4(sigma)
    _check(load().coati_host_mg94_p(C.c_float(np.float32(br_len)), C.c_float(np.float32(omega)), _p(_f4(pi)), _p(sg),
                                    _p(out)))
    return out


def ecm_p(br_len=0.0133, omega=0.2) -> np.ndarray:
    out = np.zeros((61, 61), np.float32)
    _check(load().coati_host_ecm_p(C.c_float(np.float32(br_len)), C.c_float(np.float32(omega)), _p(out)))
    return out


def gtr_q(pi, sigma) -> np.ndarray:
    out = np.zeros((4, 4), np.float32)
    _check(load().coati_host_gtr_q(_p(_f4(pi)), _p(_f4(sigma)), _p(out)))
    return out


def marginal_p(P, pi=DEFAULT_PI, amb_best=False, sub_max=False) -> np.ndarray:
    out = np.zeros((183, 15), np.float32)
    _check(load().coati_host_marginal_p(_p(_f4(P)), _p(_f4(pi)), int(amb_best), int(sub_max), _p(out)))
    return out


def set_subst(model="mar-mg", br_len=0.0133, omega=0.2, pi=DEFAULT_PI, sigma=None, amb_best=False,
              sub_max=False) -> np.ndarray:
    out = np.zeros((183, 15), np.float32)
    sg = None if sigma is None else _f4(sigma)
    _check(load().coati_host_set_subst(model.encode(), C.c_float(np.float32(br_len)), C.c_float(np.float32(omega)),
                                       _p(_f4(pi)), _p(sg), int(amb_best), int(sub_max), _p(out)))
    return out


def encode(anc: str, des: str):
    ab, db = anc.encode(), des.encode()  # (buffers sized by BYTES: non-ASCII input must fail in the library, not overflow here)
    a = np.zeros(max(len(ab), 1), np.uint8)
    b = np.zeros(max(len(db), 1), np.uint8)
    _check(load().coati_host_encode(ab, db, _p(a), _p(b)))
    return a[:len(ab)].copy(), b[:len(db)].copy()


def synth_encoded(first: int, n: int, seed_base: int = 0xC0A71, n_codons: int = 334):
    """Encoded synthetic pairs [first, first+n) -> (a_cat, a_off, b_cat, b_off)."""
    a_off = np.zeros(n + 1, np.uint64)
    b_off = np.zeros(n + 1, np.uint64)
    fn = load().coati_host_synth_encoded
    _check(fn(C.c_ulonglong(first), C.c_ulonglong(n), C.c_ulonglong(seed_base), C.c_uint(n_codons), None, _p(a_off),
              None, _p(b_off)))
    a_cat = np.zeros(max(int(a_off[-1]), 1), np.uint8)
    b_cat = np.zeros(max(int(b_off[-1]), 1), np.uint8)
    _check(fn(C.c_ulonglong(first), C.c_ulonglong(n), C.c_ulonglong(seed_base), C.c_uint(n_codons), _p(a_cat),
              _p(a_off), _p(b_cat), _p(b_off)))
    return a_cat, a_off, b_cat, b_off


def synth_lengths(first: int, n: int, seed_base: int = 0xC0A71, n_codons: int = 334):
    """(len_a[n], len_b[n]) of the synthetic pairs [first, first+n) without materialising them."""
    a_off = np.zeros(n + 1, np.uint64)
    b_off = np.zeros(n + 1, np.uint64)
    _check(load().coati_host_synth_encoded(C.c_ulonglong(first), C.c_ulonglong(n), C.c_ulonglong(seed_base),
                                           C.c_uint(n_codons), None, _p(a_off), None, _p(b_off)))
    return np.diff(a_off), np.diff(b_off)


def synth_raw(index: int, seed_base: int = 0xC0A71, n_codons: int = 334):
    cap = n_codons * 3 * 3 + 64
    a = C.create_string_buffer(cap)
    d = C.create_string_buffer(cap)
    _check(load().coati_host_synth_raw(C.c_ulonglong(index), C.c_ulonglong(seed_base), C.c_uint(n_codons), a, d,
                                       C.c_ulonglong(cap)))
    return a.value.decode(), d.value.decode()


def trim_end_stops(s0: str, s1: str):
    cap = max(len(s0.encode()), len(s1.encode())) + 8
    bufs = [C.create_string_buffer(cap) for _ in range(4)]
    _check(load().coati_host_trim_end_stops(s0.encode(), s1.encode(), *bufs, C.c_ulonglong(cap)))
    t0, t1, st0, st1 = (b.value.decode() for b in bufs)
    return [t0, t1], [st0, st1]


def restore_end_stops(aln0: str, aln1: str, stop0: str, stop1: str, score=0.0, gap_open=0.001, gap_extend=None):
    e = np.float32(1.0) - np.float32(1.0) / np.float32(6.0) if gap_extend is None else np.float32(gap_extend)
    cap = max(len(aln0), len(aln1)) + 16
    b0 = C.create_string_buffer(aln0.encode(), cap)
    b1 = C.create_string_buffer(aln1.encode(), cap)
    sc = C.c_float(np.float32(score))
    _check(load().coati_host_restore_end_stops(b0, b1, stop0.encode(), stop1.encode(), C.c_float(np.float32(gap_open)),
                                               C.c_float(e), C.byref(sc), C.c_ulonglong(cap)))
    return [b0.value.decode(), b1.value.decode()], np.float32(sc.value)


def rng_seed(seeds) -> np.ndarray:
    """Lehmer64Fast state (lo, hi) after rand.Seed(string_seed_seq(seeds))."""
    arr = (C.c_char_p * len(seeds))(*[x.encode() for x in seeds])
    out = np.zeros(2, np.uint64)
    _check(load().coati_host_rng_seed(arr, len(seeds), _p(out)))
    return out


def rng_f24(state, n: int):
    """n draws; returns (draws, advanced state)."""
    st = np.array(state, np.uint64)
    out = np.zeros(n, np.float32)
    _check(load().coati_host_rng_f24(_p(st), n, _p(out)))
    return out, st


def extract_file_type(path: str):
    a, b = C.create_string_buffer(len(path) + 8), C.create_string_buffer(len(path) + 8)
    _check(load().coati_host_extract_file_type(path.encode(), a, b, C.c_ulonglong(len(path) + 8)))
    return a.value.decode(), b.value.decode()


def convert(in_path: str, out_path: str, score: float = float("nan")):
    _check(load().coati_host_convert(str(in_path).encode(), str(out_path).encode(), C.c_float(score)))


def write_json_array(in_path: str, out_path: str, count: int):
    _check(load().coati_host_write_json_array(str(in_path).encode(), str(out_path).encode(), C.c_uint(count)))


def json_number(v) -> str:
    buf = C.create_string_buffer(64)
    _check(load().coati_host_json_number(C.c_float(np.float32(v)), buf, C.c_ulonglong(64)))
    return buf.value.decode()


def alignment_score(aln_anc: str, aln_des: str, model="mar-mg", gap_open=0.001, gap_extend=None, gap_len=1):
    e = np.float32(1.0) - np.float32(1.0) / np.float32(6.0) if gap_extend is None else np.float32(gap_extend)
    out = C.c_float()
    _check(load().coati_host_alignment_score(aln_anc.encode(), aln_des.encode(), model.encode(),
                                             C.c_float(np.float32(gap_open)), C.c_float(e), C.c_uint(gap_len), C.byref(out)))
    return np.float32(out.value)


def parse_matrix_csv(path) -> np.ndarray:
    out = np.zeros((61, 61), np.float32)
    _check(load().coati_host_parse_matrix_csv(str(path).encode(), _p(out)))
    return out


def align_leafs(ref_seq: str, leaves, br_lens, model="mar-mg", omega=0.2, gap_open=0.001, gap_extend=None, gap_len=1):
    """Batched pairwise step of `coati msa` (coati_amd/host/align.hpp: align_leafs): every leaf aligned
    to the reference with the table of its own branch length, one GPU launch.
    Returns [(aligned_ref, aligned_leaf, score), ...]."""
    ge = float(np.float32(1.0) - np.float32(1.0) / np.float32(6.0)) if gap_extend is None else gap_extend
    n = len(leaves)
    slot = len(ref_seq) + max((len(x) for x in leaves), default=0) + 1
    buf = C.create_string_buffer(max(2 * n * slot, 1))
    arr = (C.c_char_p * n)(*[x.encode() for x in leaves])
    bl = np.ascontiguousarray(br_lens, np.float32)
    scores = np.zeros(n, np.float32)
    _check(load().coati_host_align_leafs(model.encode(), C.c_float(omega), C.c_float(gap_open), C.c_float(ge),
                                         C.c_uint(gap_len), ref_seq.encode(), arr, _p(bl), C.c_uint(n), buf,
                                         C.c_ulonglong(slot), _p(scores)))
    out = []
    for p in range(n):
        a = C.string_at(C.addressof(buf) + (2 * p) * slot).decode()
        b = C.string_at(C.addressof(buf) + (2 * p + 1) * slot).decode()
        out.append((a, b, float(scores[p])))
    return out


def newick(text: str, reroot: str = ""):
    """[(index, label, length, is_leaf, parent)] of the parsed (optionally re-rooted) guide tree."""
    buf = C.create_string_buffer(1 << 20)
    _check(load().coati_host_newick(text.encode(), reroot.encode(), buf, C.c_ulonglong(len(buf))))
    rows = []
    for line in buf.value.decode().splitlines():
        i, label, length, leaf, parent = line.split("\t")
        rows.append((int(i), label, float(length), leaf == "1", int(parent)))
    return rows


def tree_distance(text: str, ref: str, node: str, reroot: bool = False) -> float:
    out = C.c_float()
    _check(load().coati_host_tree_distance(text.encode(), ref.encode(), node.encode(), int(reroot), C.byref(out)))
    return float(out.value)


def merge_indels(sets):
    """sets: [(names, seqs, flag_length, {pos: flag})].  Returns (names, seqs, {pos: flag})."""
    spec = "\n".join(f"{','.join(n)};{','.join(s)};{cap};{','.join(f'{k}={v}' for k, v in sorted(fl.items()))}"
                     for n, s, cap, fl in sets)
    buf = C.create_string_buffer(1 << 20)
    _check(load().coati_host_merge_indels(spec.encode(), buf, C.c_ulonglong(len(buf))))
    names, seqs, flags = buf.value.decode().split(";")
    return names.split(","), seqs.split(","), {int(kv.split("=")[0]): int(kv.split("=")[1]) for kv in flags.split(",") if kv}


def batch_reader_check(path) -> int:
    """The --batch driver's indexed FASTA reader against read_input on one file: 0 = identical records,
    k + 1 = record k differs, -1 = the fast reader declines the file (not a FASTA path)."""
    out = C.c_long(0)
    _check(load().coati_host_batch_reader_check(str(path).encode(), C.byref(out)))
    return int(out.value)


def batch_shard_check(path, world: int, rank: int):
    """One rank's host side of `coati-alignpair --batch --devices` without a device -> (s0, s1, first_difference): its shard
    [s0, s1) from the file's index, parsed + encoded through the block pipeline and compared with the generic reader
    (0 = identical, k + 1 = pair k differs)."""
    s0, s1, out = C.c_ulonglong(0), C.c_ulonglong(0), C.c_long(0)
    _check(load().coati_host_batch_shard_check(str(path).encode(), world, rank, C.byref(s0), C.byref(s1), C.byref(out)))
    return int(s0.value), int(s1.value), int(out.value)
