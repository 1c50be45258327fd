"""The reference's prediction pickle, byte for byte, without building its Python lists.

`Evaluator.forward_one_batch` of the reference turns every query's predictions into nested Python lists
(code/eval_vsrl_corr.py:247-273: `.tolist()` of boxes [nsrl, ncmp, nfrm, 7], scores, indices and the annotation ids) and
rank 0 `pickle.dump`s the list of per-query dicts (:104, :156). Per 512 queries that is 0.9 M Python floats in 130 k lists:
120 ms of `tolist` + 40 ms of `pickle.dumps` on one core - a 3.2 k queries/s ceiling for ANY forward behind it (measured,
DESIGN.md). `dumps_records(cols)` writes the SAME BYTES `pickle.dumps([{k: cols[k][i].tolist() ...} ...])` produces
(protocol 4, the default of the reference's Python >= 3.8) straight from the numpy columns:

* opcodes (pickle protocol 4): list = `]` MEMOIZE [MARK items APPENDS | item APPEND] in batches of 1000, dict = `}` MEMOIZE
  MARK key value ... SETITEMS (APPEND / SETITEM only for a container of exactly one item), float = `G` + big-endian double, int = `K` / `M` / `J` + 1 / 2 / 4 little-endian bytes (LONG1
  beyond int32), a key string = SHORT_BINUNICODE + MEMOIZE the first time and BINGET / LONG_BINGET of its memo index afterwards
  (every container takes a memo slot, so the indices follow from the structure of the first record);
* framing (C pickler, `_pickle.c::_Pickler_OpcodeBoundary`): the check runs at the START of every object's `save()`; a frame is
  committed there once its payload has reached 64 KiB - so a frame ends at the first object start at or past 64 KiB, `FRAME` +
  8-byte length in front of every payload of >= 4 bytes;
* the first record is encoded by the exact (slow) encoder; every later record has the same byte layout up to the width of its
  integers: one template, the floats of all records written with one fancy assignment, variable-width integers placed by
  width class.

`tests/test_fast_pickle.py` compares against `pickle.dumps` for every record layout of the evaluator (sep / svsq / temp / spat,
batch boundaries at 1000, ids of every width, negative ids). Falls back to `pickle.dumps` of the lists for anything it does
not model (integers beyond int32, non-finite shapes, other protocols)."""
from __future__ import annotations

import pickle
import struct
from typing import Dict, List, Sequence

import numpy as np

_FRAME_TARGET = 64 * 1024
_BATCH = 1000


# ---- exact encoder of one value (nested lists of floats / ints, dicts with str keys): bytes + object starts ----------------
class _Enc:
    def __init__(self):
        self.out = bytearray()
        self.begins: List[int] = []          # offsets at which an object's save() starts
        self.memo: Dict[str, int] = {}       # key string -> memo index
        self.nmemo = 0
        self.float_slots: List[int] = []     # offset of the 8 payload bytes of every float, in encounter order
        self.int_slots: List[int] = []       # offset of the OPCODE byte of every int, in encounter order

    def memoize(self):
        self.out += b"\x94"
        self.nmemo += 1

    def save(self, v):
        self.begins.append(len(self.out))
        if isinstance(v, float):
            self.out += b"G"
            self.float_slots.append(len(self.out))
            self.out += struct.pack(">d", v)
        elif isinstance(v, bool):
            raise TypeError("bool")
        elif isinstance(v, int):
            self.int_slots.append(len(self.out))
            self.out += _int_bytes(v)
        elif isinstance(v, str):
            if v in self.memo:
                i = self.memo[v]
                self.out += (b"h" + bytes([i])) if i < 256 else (b"j" + struct.pack("<I", i))
            else:
                b = v.encode("utf-8")
                assert len(b) < 256
                self.out += b"\x8c" + bytes([len(b)]) + b
                self.memo[v] = self.nmemo
                self.memoize()
        elif isinstance(v, list):
            self.out += b"]"
            self.memoize()
            # (the C pickler's batch_list_exact: APPEND only for a list of exactly one item; otherwise every batch of up to
            # 1000 items is MARK ... APPENDS, a trailing batch of one item included)
            if len(v) == 1:
                self.save(v[0])
                self.out += b"a"
            else:
                for lo in range(0, len(v), _BATCH):
                    self.out += b"("
                    for x in v[lo:lo + _BATCH]:
                        self.save(x)
                    self.out += b"e"
        elif isinstance(v, dict):
            self.out += b"}"
            self.memoize()
            items = list(v.items())
            if len(items) == 1:
                self.save(items[0][0]); self.save(items[0][1])
                self.out += b"s"
            else:
                for lo in range(0, len(items), _BATCH):
                    self.out += b"("
                    for k, x in items[lo:lo + _BATCH]:
                        self.save(k); self.save(x)
                    self.out += b"u"
        else:
            raise TypeError(type(v))


def _int_bytes(v: int) -> bytes:
    if 0 <= v <= 0xFF:
        return b"K" + bytes([v])
    if 0 <= v <= 0xFFFF:
        return b"M" + struct.pack("<H", v)
    if -0x80000000 <= v <= 0x7FFFFFFF:
        return b"J" + struct.pack("<i", v)
    raise OverflowError(v)


def _frame(body: np.ndarray, begins: np.ndarray) -> bytes:
    """Protocol-4 framing of the opcode stream behind PROTO (see the module docstring)."""
    parts = [b"\x80\x04"]
    n = body.size
    start = 0
    mv = memoryview(body)
    while True:
        k = int(np.searchsorted(begins, start + _FRAME_TARGET, side="left"))
        end = int(begins[k]) if k < begins.size else n
        if end - start >= 4:
            parts.append(b"\x95" + struct.pack("<Q", end - start))
        parts.append(mv[start:end])
        if end >= n:
            break
        start = end
    return b"".join(parts)


def _lists(cols: Dict[str, np.ndarray], i: int) -> dict:
    return {k: v[i].tolist() for k, v in cols.items()}


def dumps_reference(cols: Dict[str, np.ndarray]) -> bytes:
    """What the reference does: Python lists, then pickle (the slow path; also the oracle of the tests)."""
    n = len(next(iter(cols.values())))
    ls = {k: v.tolist() for k, v in cols.items()}
    return pickle.dumps([{k: v[i] for k, v in ls.items()} for i in range(n)], protocol=4)


def dumps_records(cols: Dict[str, np.ndarray]) -> bytes:
    """cols: key -> array with the record axis first (floating or integer dtype), keys in record order.
    -> the bytes of pickle.dumps([{k: cols[k][i].tolist() for k in cols} for i in range(n)], protocol=4)."""
    keys = list(cols)
    cols = {k: np.asarray(v) for k, v in cols.items()}
    n = len(cols[keys[0]]) if keys else 0
    for k, v in cols.items():
        if len(v) != n or v.dtype.kind not in "fiu" or (v.dtype.kind == "u" and v.dtype.itemsize == 8):
            return dumps_reference(cols)
    if n < 3 or not keys:
        return dumps_reference(cols)
    ints = {k: v.astype(np.int64).reshape(n, -1) for k, v in cols.items() if v.dtype.kind in "iu"}
    for v in ints.values():
        if v.size and (v.min() < -0x80000000 or v.max() > 0x7FFFFFFF):
            return dumps_reference(cols)
    # ---- record 0 (memoizes the keys) and the template (record 1: keys by memo reference), exact encoder
    enc = _Enc()
    enc.out += b"]"; enc.begins.append(0); enc.memoize()            # the outer list
    head_len = len(enc.out)
    nb0 = min(n, _BATCH)
    if nb0 > 1:
        enc.out += b"("
    enc.save(_lists(cols, 0))
    rec0_end = len(enc.out)
    pre = bytes(enc.out)
    pre_begins = np.asarray(enc.begins, dtype=np.int64)
    t = _Enc()
    t.memo, t.nmemo = dict(enc.memo), enc.nmemo
    t.save(_lists(cols, 1))
    T = np.frombuffer(bytes(t.out), dtype=np.uint8)
    tb = np.asarray(t.begins, dtype=np.int64)
    fslots = np.asarray(t.float_slots, dtype=np.int64)
    islots = np.asarray(t.int_slots, dtype=np.int64)
    # slot order = encounter order = key order, row-major within a key
    nf = sum(int(np.prod(cols[k].shape[1:], dtype=np.int64)) for k in keys if cols[k].dtype.kind == "f")
    ni = sum(int(np.prod(cols[k].shape[1:], dtype=np.int64)) for k in keys if cols[k].dtype.kind != "f")
    assert nf == fslots.size and ni == islots.size
    fl = [cols[k].reshape(n, -1) for k in keys if cols[k].dtype.kind == "f"]
    F = np.concatenate(fl, axis=1).astype(">f8") if fl else np.zeros((n, 0), ">f8")
    il = [ints[k] for k in keys if cols[k].dtype.kind != "f"]
    I = np.concatenate(il, axis=1) if il else np.zeros((n, 0), np.int64)
    # the template encodes record 1's ints: normalise every int slot to the 2-byte form `K x`
    w1 = np.where((I[1] >= 0) & (I[1] <= 0xFF), 2, np.where((I[1] >= 0) & (I[1] <= 0xFFFF), 3, 5)) if ni else np.zeros(0, np.int64)
    if ni:
        keep = np.ones(T.size, bool)
        shift = np.zeros(T.size + 1, np.int64)
        for s, w in zip(islots, w1):
            keep[s + 2:s + w] = False
            shift[s + w:] += w - 2
        T = T[keep].copy()
        tb = tb - shift[tb]
        fslots = fslots - shift[fslots]
        islots = islots - shift[islots]
        T[islots] = ord("K")
    LT = T.size
    # ---- records 1 .. n-1: grouped by the byte widths of their integers (a handful of distinct layouts in practice)
    R = n - 1
    Iv = I[1:]
    W = np.where((Iv >= 0) & (Iv <= 0xFF), 2, np.where((Iv >= 0) & (Iv <= 0xFFFF), 3, 5)) if ni else np.zeros((R, 0), np.int64)
    var = np.nonzero((W != 2).any(axis=0))[0] if ni else np.zeros(0, np.int64)        # slots that are ever wider than 2 bytes
    const_i = np.setdiff1d(np.arange(ni), var) if ni else np.zeros(0, np.int64)
    if var.size:
        pats, inv = np.unique(W[:, var], axis=0, return_inverse=True)
        inv = inv.reshape(-1)
    else:
        pats, inv = np.zeros((1, 0), np.int64), np.zeros(R, np.int64)
    vslots = islots[var]
    rec_len = np.empty(R, np.int64)
    rows_of, mats, tbs = [], [], []
    fcol = (fslots[:, None] + np.arange(8)[None, :]).reshape(-1)
    for pi in range(pats.shape[0]):
        rows = np.nonzero(inv == pi)[0]
        wv = pats[pi]
        # template of this layout: the variable slots widened from 2 to wv bytes
        ins = np.zeros(LT + 1, np.int64)                              # bytes inserted in front of template offset x
        for sl, w in zip(vslots, wv):
            ins[sl + 2:] += w - 2
        Lp = LT + int(ins[-1])
        pos = np.arange(LT) + ins[:LT]                                # where every template byte goes
        Tp = np.zeros(Lp, np.uint8)
        Tp[pos] = T
        M = np.broadcast_to(Tp, (rows.size, Lp)).copy()
        if nf:
            M[:, pos[fcol]] = F[1:][rows].view(np.uint8).reshape(rows.size, -1)
        if const_i.size:
            M[:, pos[islots[const_i] + 1]] = Iv[rows][:, const_i].astype(np.uint8)
        for sl, w, v in zip(vslots, wv, var):
            op, dt = {2: ("K", "<u1"), 3: ("M", "<u2"), 5: ("J", "<i4")}[int(w)]
            p0 = int(pos[sl])
            M[:, p0] = ord(op)
            M[:, p0 + 1:p0 + int(w)] = np.ascontiguousarray(Iv[rows, v]).astype(dt).view(np.uint8).reshape(rows.size, int(w) - 1)
        rec_len[rows] = Lp
        rows_of.append(rows); mats.append(M); tbs.append(pos[tb])
    # batch punctuation between records of the outer list: APPENDS of a full batch + MARK of the next one
    gidx = np.arange(1, n)
    pre_bytes = np.where((gidx % _BATCH) == 0, 2, 0).astype(np.int64)
    rec_start = rec0_end + np.cumsum(rec_len + pre_bytes) - rec_len
    begins_r = np.empty((R, tb.size), np.int64)
    row_src = [None] * R
    for rows, M, tbp in zip(rows_of, mats, tbs):
        begins_r[rows] = rec_start[rows, None] + tbp[None, :]
        for k, r in enumerate(rows.tolist()):
            row_src[r] = M[k]
    parts = [pre]
    for r in range(R):
        if pre_bytes[r]:
            parts.append(b"e(")
        parts.append(row_src[r])
    parts.append(b"e.")
    body = np.frombuffer(b"".join(parts), dtype=np.uint8)
    begins = np.concatenate([pre_begins, begins_r.reshape(-1)])
    return _frame(body, begins)


def records_columns(cols: Dict[str, Sequence]) -> Dict[str, np.ndarray]:
    return {k: np.asarray(v) for k, v in cols.items()}
