"""oracle -- TEST INFRASTRUCTURE.  CPU restatements of the reference algorithms (gf_oracle.c, hope_oracle.py, n2v_oracle.c; snap_stream.py / .c:
the node2vec binary restated on its own random stream and pinned to its output).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under gem_amd/ imports it (tests/test_capi.py::test_product_never_imports_oracle checks).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_build', 'liboracle.so')
REF_GF = os.path.join(_HERE, '_ref', 'gf')
REF_N2V = os.path.join(_HERE, '_ref', 'node2vec')
REF_FAKETIME = os.path.join(_HERE, '_ref', 'libfaketime.so')      # shim/faketime.c: LD_PRELOAD it and REF_N2V is deterministic (snap_stream.py)
_lib = None


def build(force=False):
    """make -C oracle : compiles liboracle.so and (only where /root/reference exists)
    the real reference binaries into oracle/_ref/."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith('.c')]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(['make', '-s', '-C', _HERE, '_build/liboracle.so'])
    if os.path.exists('/root/reference/gem/c_src/gf.cpp') and not (os.path.exists(REF_GF) and os.path.exists(REF_N2V) and os.path.exists(REF_FAKETIME)):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'ref'])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        i32p, f32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.oracle_gf_train_f32.argtypes = [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_float, C.c_float,
                                          C.c_int32, f32p]
        L.oracle_gf_train_f32.restype = None
        L.oracle_gf_train_f64.argtypes = [C.c_int64, C.c_int64, i32p, i32p, f64p, C.c_int32, C.c_double, C.c_double,
                                          C.c_int32, f64p]
        L.oracle_gf_train_f64.restype = None
        L.oracle_gf_objective.argtypes = [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, f32p, f64p]
        L.oracle_gf_objective.restype = None
        L.oracle_gf_cpp_init.argtypes = [C.c_uint32, C.c_int64, C.c_int32, f32p]
        L.oracle_gf_cpp_init.restype = None
        i64p, u32p = C.POINTER(C.c_int64), C.POINTER(C.c_uint32)
        L.oracle_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p]
        L.oracle_perm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.oracle_perm.restype = C.c_uint32
        L.oracle_alias_build_f32.argtypes = [C.c_int32, f32p, f32p, i32p, i32p]
        L.oracle_n2v_alias_rows.argtypes = [C.c_int64, i64p, f32p, f32p, i32p]
        L.oracle_alias_build_hub.argtypes = [C.c_int32, f32p, f32p, i32p]
        L.oracle_alias_build_hub.restype = None
        L.oracle_n2v_walks.argtypes = [C.c_int64, i64p, i32p, f32p, i32p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_uint64,
                                       C.c_int32, C.c_int64, C.c_int64, i32p, C.c_int64, i32p]
        L.oracle_n2v_vocab.argtypes = [C.c_int64, C.c_int64, i32p, i32p]
        L.oracle_unigram_build.argtypes = [C.c_int64, i32p, f64p, i32p]
        L.oracle_sgns_train.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int32, i32p, C.c_int32, C.c_int32, C.c_float,
                                        C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, f32p, i32p, C.c_uint64,
                                        C.c_int32, f32p, f32p]
        L.oracle_sgns_train_vocab_order.argtypes = [C.c_int64, i32p, C.c_int32, C.c_int64, C.c_int32, i32p, C.c_int32, C.c_int32, C.c_float,
                                                    C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, f32p, i32p, C.c_uint64,
                                                    C.c_int32, f32p, f32p]
        L.oracle_sgns_train_vocab_order.restype = None
        L.oracle_sgns_train_wide.argtypes = [C.c_int64, i32p, C.c_int32, C.c_int64, C.c_int32, i32p, C.c_int32, C.c_int32, C.c_float,
                                                    C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, f32p, i32p, C.c_uint64,
                                                    C.c_int32, f32p, f32p]
        L.oracle_sgns_train_wide.restype = None
        L.oracle_sgns_train_wide.restype = C.c_int32
        L.oracle_sgns_init.argtypes = [C.c_int64, C.c_int32, C.c_uint64, f32p, f32p]
        L.oracle_sgns_pairs.restype = C.c_int64
        L.oracle_sgns_pairs.argtypes = [C.c_int64, C.c_int32, i32p, C.c_int32, C.c_int32, C.c_int64, C.c_uint64, i32p, i32p]
        L.oracle_sgns_train_part_slots.restype = C.c_int64
        L.oracle_sgns_train_part_slots.argtypes = [C.c_int32, C.c_int64, C.c_int32, i32p, i64p, C.c_int64, C.c_int32, C.c_float, C.c_int64, C.c_int64, C.c_int32,
                                                   C.c_int32, C.c_int32, C.c_int32, C.c_int64, f32p, i32p, C.c_int64, i32p, C.c_uint64, C.c_int32, C.c_int32, f32p, f32p]
        L.oracle_sgns_train_part.restype = C.c_int64
        L.oracle_sgns_train_part.argtypes = [C.c_int32, C.c_int64, C.c_int32, i32p, i64p, C.c_int64, C.c_int32, C.c_float, C.c_int64, C.c_int64, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int64, f32p, i32p, C.c_uint64, C.c_int32, C.c_int32, f32p, f32p]
        for f in ('oracle_philox', 'oracle_alias_build_f32', 'oracle_n2v_alias_rows', 'oracle_n2v_walks', 'oracle_n2v_vocab',
                  'oracle_unigram_build', 'oracle_sgns_train', 'oracle_sgns_init'):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def gf_train_f32(n, src, dst, w, d, eta, regu, max_iter, X0):
    """gf.cpp:152-164 in fp32; returns a new array."""
    X = np.ascontiguousarray(X0, dtype=np.float32).copy()
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    lib().oracle_gf_train_f32(n, len(src), _p(src, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_float), d, eta, regu,
                              max_iter, _p(X, C.c_float))
    return X


def gf_train_f64(n, src, dst, w, d, eta, regu, max_iter, X0):
    """gf.py:93-100 in fp64; returns a new array."""
    X = np.ascontiguousarray(X0, dtype=np.float64).copy()
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    lib().oracle_gf_train_f64(n, len(src), _p(src, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_double), d, eta, regu,
                              max_iter, _p(X, C.c_double))
    return X


def gf_cpp_init(seed, n, d):
    """gf.cpp:41-52 init_embedding() for the 32-bit seed the binary derives from its clock: float32 [n, d]."""
    X = np.empty((n, d), dtype=np.float32)
    lib().oracle_gf_cpp_init(int(seed) & 0xffffffff, n, d, _p(X, C.c_float))
    return X


def gf_cpp_seed(fake_clock):
    """The `unsigned seed` of gf.cpp:46 for GEM_FAKE_CLOCK="<sec>[.<nsec>]": nanoseconds since the epoch truncated to 32 bits."""
    sec, _, nsec = str(fake_clock).partition('.')
    return (int(sec) * 1000000000 + int(nsec or 0)) & 0xffffffff


def gf_objective(n, src, dst, w, d, X):
    """gf.cpp:94-113: returns (f1, f2)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    out = np.zeros(2)
    lib().oracle_gf_objective(n, len(src), _p(src, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_float), d, _p(X, C.c_float),
                              _p(out, C.c_double))
    return float(out[0]), float(out[1])


# ------------------------------------------------------------------ node2vec
def sorted_csr(n, src, dst, w=None):
    """CSR with columns sorted inside each row (what the device uses for walks)."""
    src = np.asarray(src); dst = np.asarray(dst)
    perm = np.lexsort((dst, src))
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=row_ptr[1:])
    col = np.ascontiguousarray(dst[perm], dtype=np.int32)
    ww = None if w is None else np.ascontiguousarray(np.asarray(w, dtype=np.float32)[perm])
    return row_ptr, col, ww


def n2v_alias_rows(row_ptr, w):
    n = len(row_ptr) - 1
    U = np.zeros(len(w), dtype=np.float32); K = np.zeros(len(w), dtype=np.int32)
    lib().oracle_n2v_alias_rows(n, _p(row_ptr, C.c_int64), _p(w, C.c_float), _p(U, C.c_float), _p(K, C.c_int32))
    return U, K


def start_nodes(row_ptr, col):
    """Nodes that occur in the edge list (as source or target), ascending: the only nodes the reference binary knows."""
    n = len(row_ptr) - 1
    present = np.diff(row_ptr) > 0
    present[np.asarray(col)] = True
    return np.flatnonzero(present).astype(np.int32)


def n2v_walks(row_ptr, col, U, K, p, q, num_walks, walk_len, seed, flags, walk_begin=0, walk_end=None):
    n = len(row_ptr) - 1
    start = start_nodes(row_ptr, col)
    if walk_end is None:
        walk_end = len(start) * num_walks
    out = np.empty((walk_end - walk_begin, walk_len), dtype=np.int32)
    lib().oracle_n2v_walks(n, _p(row_ptr, C.c_int64), _p(col, C.c_int32), _p(U, C.c_float), _p(K, C.c_int32), p, q, num_walks,
                           walk_len, seed, flags, walk_begin, walk_end, _p(start, C.c_int32), len(start), _p(out, C.c_int32))
    return out


def n2v_vocab(n, walks):
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    c = np.zeros(n, dtype=np.int32)
    lib().oracle_n2v_vocab(n, walks.size, _p(walks, C.c_int32), _p(c, C.c_int32))
    return c


def unigram_build(counts):
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    U = np.zeros(len(counts)); K = np.zeros(len(counts), dtype=np.int32)
    lib().oracle_unigram_build(len(counts), _p(counts, C.c_int32), _p(U, C.c_double), _p(K, C.c_int32))
    return U.astype(np.float32), K


def sgns_init(n, d, seed):
    P = np.empty((n, d), dtype=np.float32); N = np.empty((n, d), dtype=np.float32)
    lib().oracle_sgns_init(n, d, seed, _p(P, C.c_float), _p(N, C.c_float))
    return P, N


def sgns_train(walks, window, alpha0, epochs, epoch, tokens_total, token_offset, walk_id_offset, UT, KT, seed, flags, SynPos,
               SynNeg, neg=5):
    """In place on SynPos/SynNeg (float32 C-contiguous)."""
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    n, d = SynPos.shape
    lib().oracle_sgns_train(n, d, walks.shape[0], walks.shape[1], _p(walks, C.c_int32), window, neg, alpha0, epochs, epoch,
                            tokens_total, token_offset, walk_id_offset, _p(UT, C.c_float), _p(KT, C.c_int32), seed, flags,
                            _p(SynPos, C.c_float), _p(SynNeg, C.c_float))


def unigram_build_vocab_order(counts, walks, flags):
    """InitUnigramTable in the binary's layout (LearnVocab renames the tokens by first appearance in the walk matrix; flag 16 of the library): returns
    (slot_tab int32[N], UTn float32[n], KTn int32[n], back int32[N], U' float32[N], K' int32[N]) -- N = nodes that occur, slot_tab[slot] = the node a slot
    names (under flags & 2: the node of KTable'[slot]), UTn / KTn the table indexed by NODE (alias as a node), back = renamed id -> node."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n = len(counts)
    flat = np.asarray(walks).ravel()
    ok = flat >= 0
    first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, flat[ok], np.nonzero(ok)[0])
    back = np.argsort(first, kind='stable')[:int((first < np.iinfo(np.int64).max).sum())].astype(np.int32)
    U, K = unigram_build(np.ascontiguousarray(counts[back]))
    slot_tab = np.ascontiguousarray(back[K] if (flags & 2) else back, dtype=np.int32)
    UTn = np.zeros(n, np.float32); KTn = np.zeros(n, np.int32)
    UTn[back] = U; KTn[back] = back[K]
    return slot_tab, UTn, KTn, back, U, K


def sgns_train_vocab_order(walks, window, alpha0, epochs, epoch, tokens_total, token_offset, walk_id_offset, slot_tab, UTn, KTn, seed, flags, SynPos,
                           SynNeg, neg=5):
    """oracle_sgns_train with the unigram table in the binary's layout (unigram_build_vocab_order).  In place on SynPos / SynNeg."""
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    n, d = SynPos.shape
    lib().oracle_sgns_train_vocab_order(len(slot_tab), _p(slot_tab, C.c_int32), d, walks.shape[0], walks.shape[1], _p(walks, C.c_int32), window, neg,
                                        alpha0, epochs, epoch, tokens_total, token_offset, walk_id_offset, _p(UTn, C.c_float), _p(KTn, C.c_int32), seed,
                                        flags, _p(SynPos, C.c_float), _p(SynNeg, C.c_float))



def sgns_train_wide(walks, window, alpha0, epochs, epoch, tokens_total, token_offset, walk_id_offset, slot_tab, UT, KT, seed, flags, SynPos,
                    SynNeg, neg=5):
    """oracle_sgns_train_wide: sgns_train (slot_tab=None; under flags & 2 the slot names KT[slot], as oracle_sgns_train does) or
    sgns_train_vocab_order (slot_tab given) with the dot product in 32 interleaved partial sums -- about 4x the speed, for the hours-long R-MAT
    passes.  In place on SynPos / SynNeg."""
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    n, d = SynPos.shape
    if slot_tab is None:
        n_slots, st = n, (_p(KT, C.c_int32) if (flags & 2) else None)
    else:
        n_slots, st = len(slot_tab), _p(slot_tab, C.c_int32)
    rc = lib().oracle_sgns_train_wide(n_slots, st, d, walks.shape[0], walks.shape[1], _p(walks, C.c_int32), window, neg, alpha0, epochs, epoch,
                                      tokens_total, token_offset, walk_id_offset, _p(UT, C.c_float), _p(KT, C.c_int32), seed, flags,
                                      _p(SynPos, C.c_float), _p(SynNeg, C.c_float))
    if rc != 0:
        raise ValueError('oracle_sgns_train_wide needs d % 32 == 0')

def n2v_train(n, src, dst, w, d, walk_len, num_walks, window, epochs, p, q, seed, flags):
    """Whole pipeline on the CPU, sequential: the meaning of the reference binary for one seed.  flags & 16: the unigram table in the binary's
    own layout (first-appearance order), else in node-id order."""
    row_ptr, col, ww = sorted_csr(n, src, dst, w)
    uniform = ww is None or all(np.all(ww[row_ptr[v]:row_ptr[v + 1]] == ww[row_ptr[v]]) for v in range(n) if row_ptr[v + 1] > row_ptr[v])
    U = K = None
    if not uniform:
        U, K = n2v_alias_rows(row_ptr, ww)
    walks = n2v_walks(row_ptr, col, U, K, p, q, num_walks, walk_len, seed, flags)
    P, N = sgns_init(n, d, seed)
    tot = walks.size
    if flags & 16:
        slot_tab, UTn, KTn = unigram_build_vocab_order(n2v_vocab(n, walks), walks, flags)[:3]
        for ep in range(epochs):
            sgns_train_vocab_order(walks, window, 0.025, epochs, ep, tot, ep * tot, 0, slot_tab, UTn, KTn, seed, flags, P, N)
        return P, walks
    UT, KT = unigram_build(n2v_vocab(n, walks))
    for ep in range(epochs):
        sgns_train(walks, window, 0.025, epochs, ep, tot, ep * tot, 0, UT, KT, seed, flags, P, N)
    return P, walks


def sgns_pairs(walks, window, epoch, walk_id_offset, seed):
    """(context, word) pairs TrainModel forms, in walk order: int32 [np, 2]."""
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    a = (walks.shape[0], walks.shape[1], _p(walks, C.c_int32), window, epoch, walk_id_offset, seed)
    npairs = lib().oracle_sgns_pairs(*a, None, None)
    ctx = np.empty(npairs, np.int32); word = np.empty(npairs, np.int32)
    lib().oracle_sgns_pairs(*a, _p(ctx, C.c_int32), _p(word, C.c_int32))
    return np.stack([ctx, word], axis=1)


def unigram_build_parts(counts, parts):
    """Concatenated per-partition tables (partition p = counts[p::parts]); returns (UT, KT, offsets)."""
    UT, KT, off = [], [], [0]
    for p in range(parts):
        u, k = unigram_build(np.ascontiguousarray(counts[p::parts]))
        UT.append(u); KT.append(k); off.append(off[-1] + len(u))
    return np.concatenate(UT), np.concatenate(KT), off


def unigram_build_parts_vocab_order(counts, corpus, parts, flags):
    """The per-partition tables in the binary's layout (GEMHIP_N2V_VOCAB_ORDER on the N-GPU schedule): partition p = the nodes v % parts == p that occur, in
    order of first appearance in the WHOLE corpus (walks of all ranks in walk-id order; -1 tokens are padding); Vose over their counts in that order.
    Returns per partition (slot_tab int32[N_p] of LOCAL indices -- the local index of the entry a slot names, under flags & 2 of its alias --,
    UT float32[n_p], KT int32[n_p] indexed by local index v // parts, alias as a local index)."""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n = len(counts)
    flat = np.asarray(corpus).ravel()
    ok = flat >= 0
    first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, flat[ok], np.nonzero(ok)[0])
    back = np.argsort(first, kind='stable')[:int((first < np.iinfo(np.int64).max).sum())].astype(np.int64)
    out = []
    for p in range(parts):
        Lp = back[back % parts == p]
        n_p = (n - p + parts - 1) // parts
        UT = np.zeros(n_p, np.float32); KT = np.zeros(n_p, np.int32)
        if len(Lp) == 0:
            out.append((np.zeros(0, np.int32), UT, KT)); continue
        U, K = unigram_build(np.ascontiguousarray(counts[Lp]))
        loc = (Lp // parts).astype(np.int32)
        UT[loc] = U; KT[loc] = loc[K]
        out.append((np.ascontiguousarray(loc[K] if (flags & 2) else loc, dtype=np.int32), UT, KT))
    return out


def sgns_train_part(walks, wids, window, alpha0, alpha_tokens_total, token_offset, epoch, parts, ctx_part, word_part, UTp, KTp, seed, flags,
                    SynPos, SynNeg, walk_id_offset=0, local_rows=False, slot_tab=None):
    """One bucket (contexts of partition ctx_part, centre words of partition word_part) of the partitioned schedule in walk order, in place.
    UTp / KTp: the unigram table restricted to word_part (local indices); wids: global walk id per walk or None.  SynPos / SynNeg: the FULL
    tables, or (local_rows) the partition buffers of ctx_part / word_part.  Returns the number of pairs trained."""
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    wids = None if wids is None else np.ascontiguousarray(wids, dtype=np.int64)
    if slot_tab is not None:
        slot_tab = np.ascontiguousarray(slot_tab, dtype=np.int32)
        return lib().oracle_sgns_train_part_slots(SynPos.shape[1], walks.shape[0], walks.shape[1], _p(walks, C.c_int32), _p(wids, C.c_int64), walk_id_offset,
                                                  window, alpha0, alpha_tokens_total, token_offset, epoch, parts, ctx_part, word_part, len(UTp),
                                                  _p(UTp, C.c_float), _p(KTp, C.c_int32), len(slot_tab), _p(slot_tab, C.c_int32), seed, flags,
                                                  1 if local_rows else 0, _p(SynPos, C.c_float), _p(SynNeg, C.c_float))
    return lib().oracle_sgns_train_part(SynPos.shape[1], walks.shape[0], walks.shape[1], _p(walks, C.c_int32), _p(wids, C.c_int64), walk_id_offset,
                                        window, alpha0, alpha_tokens_total, token_offset, epoch, parts, ctx_part, word_part, len(UTp),
                                        _p(UTp, C.c_float), _p(KTp, C.c_int32), seed, flags, 1 if local_rows else 0, _p(SynPos, C.c_float),
                                        _p(SynNeg, C.c_float))
