"""Host mirror of the reference's indel calling loop (call_indels, lofreq_call.c:619-726).

`IndelColumns` is the flattened form of the indel fields of a run of plp_col_t columns (plp.h:113-130;
ins_event / del_event, utils.h:101-135): what a binding hands to `lfq_call_indels_batch`.  The statistical
test of one indel event is snpcaller() over all reads of the column (snpcaller.c:502-623), so the device
work is the SNV path's kernels run on pseudo-columns -- see include/lofreq_amd.h.
"""
import ctypes as C

import numpy as np

from . import _lib

_I32 = ("coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun")


class IndelColumns:
    """Build with `IndelColumns.from_columns(list_of_dicts)`; each dict is one pileup column:

        ref (1-char str), coverage_plp, num_tails, num_non_indels, hrun,
        ins / dels: {"non_fw", "non_rv", "ne_q": [...], "ne_mq": [...],
                     "events": [{"key": "AC", "fw": 3, "rv": 2, "q": [...], "aq": [...], "mq": [...],
                                 "sq": [...] (optional)}, ...]}      (events in first-occurrence order)

    num_ins / num_dels default to the number of reads carrying an event of that side.
    """

    def __init__(self):
        self.ncols = 0
        self.ref_base = np.zeros(0, np.uint8)
        for n in _I32:
            setattr(self, n, np.zeros(0, np.int32))
        self.sides = [None, None]
        self.keys = [[], []]
        self.cons_indel = None          # uint8 per column when known (pileup_indel_columns)

    @staticmethod
    def from_columns(cols):
        o = IndelColumns()
        o.ncols = len(cols)
        o.ref_base = np.frombuffer("".join(c["ref"] for c in cols).encode(), dtype=np.uint8).copy()
        side_names = ("ins", "dels")
        nev_reads = [[sum(len(e["q"]) for e in c.get(sn, {}).get("events", [])) for c in cols] for sn in side_names]
        o.coverage_plp = np.asarray([c["coverage_plp"] for c in cols], np.int32)
        o.num_tails = np.asarray([c.get("num_tails", 0) for c in cols], np.int32)
        o.num_non_indels = np.asarray([c["num_non_indels"] for c in cols], np.int32)
        o.num_ins = np.asarray([c.get("num_ins", nev_reads[0][i]) for i, c in enumerate(cols)], np.int32)
        o.num_dels = np.asarray([c.get("num_dels", nev_reads[1][i]) for i, c in enumerate(cols)], np.int32)
        o.hrun = np.asarray([c.get("hrun", 0) for c in cols], np.int32)
        for sd, sn in enumerate(side_names):
            non_fw, non_rv, ne_off, ne_q, ne_mq, ev_off = [], [], [0], [], [], [0]
            key_off, key_chars, ev_fw, ev_rv, rd_off = [0], [], [], [], [0]
            rd = {k: [] for k in ("q", "aq", "mq", "sq")}
            keys = []
            for c in cols:
                s = c.get(sn, {})
                non_fw.append(s.get("non_fw", 0))
                non_rv.append(s.get("non_rv", 0))
                ne_q.extend(s.get("ne_q", []))
                ne_mq.extend(s.get("ne_mq", []))
                ne_off.append(len(ne_q))
                for e in s.get("events", []):
                    keys.append(e["key"])
                    key_chars.append(e["key"])
                    key_off.append(key_off[-1] + len(e["key"]))
                    ev_fw.append(e["fw"])
                    ev_rv.append(e["rv"])
                    n = len(e["q"])
                    rd["q"].extend(e["q"])
                    rd["aq"].extend(e.get("aq", [-1] * n))
                    rd["mq"].extend(e["mq"])
                    rd["sq"].extend(e.get("sq", [-1] * n))
                    rd_off.append(len(rd["q"]))
                ev_off.append(len(keys))
            o.keys[sd] = keys
            o.sides[sd] = {
                "non_fw": np.asarray(non_fw, np.int32), "non_rv": np.asarray(non_rv, np.int32),
                "ne_off": np.asarray(ne_off, np.int64), "ne_q": np.asarray(ne_q, np.int16),
                "ne_mq": np.asarray(ne_mq, np.int16), "ev_off": np.asarray(ev_off, np.int64),
                "key_off": np.asarray(key_off, np.int64),
                "key_chars": np.frombuffer(("".join(key_chars) + "\0").encode(), dtype=np.uint8).copy(),
                "ev_fw": np.asarray(ev_fw, np.int32), "ev_rv": np.asarray(ev_rv, np.int32),
                "rd_off": np.asarray(rd_off, np.int64),
                "rd_q": np.asarray(rd["q"], np.int16), "rd_aq": np.asarray(rd["aq"], np.int16),
                "rd_mq": np.asarray(rd["mq"], np.int16), "rd_sq": np.asarray(rd["sq"], np.int16),
            }
        return o

    def flat(self):
        """dict of arrays (the layout both this library's C struct and the oracle's take)"""
        d = {"ncols": self.ncols, "ref_base": self.ref_base}
        for n in _I32:
            d[n] = getattr(self, n)
        for k in self.sides[0]:
            d[k] = [self.sides[0][k], self.sides[1][k]]
        return d

    def c_struct(self):
        s = _lib.IndelColumnsC()
        s.ncols = self.ncols
        s.ref_base = self.ref_base.ctypes.data
        for n in _I32:
            setattr(s, n, getattr(self, n).ctypes.data)
        for sd in range(2):
            for n, a in self.sides[sd].items():
                setattr(s.side[sd], n, a.ctypes.data)
        return s

    def ref_alt(self, col, side, event):
        """ins_to_str / del_to_str (lofreq_call.c:272-303): VCF REF / ALT strings of an event"""
        r = chr(self.ref_base[col])
        key = self.keys[side][event]
        return (r, r + key) if side == 0 else (r + key, r)


def call_indels(caller, cols, conf, records_capacity=None):
    """call_indels over a batch -> (records[INDEL_RECORD_DTYPE] in reference order, number of tests).
    Mutates conf.bonf_indel / conf.num_indel_tests like the reference."""
    L = _lib.load()
    # columns that came from this context's device pileup and are still current go back as the original struct: the
    # library then builds the pseudo-columns on the device from the resident quality arrays
    if (getattr(cols, "_c_ptr", None) is not None and cols._c_caller is caller
            and cols._c_gen == getattr(caller, "_indel_gen", -1)):
        cs = cols._c_ptr.contents
    else:
        cs = cols.c_struct()
    nev = len(cols.keys[0]) + len(cols.keys[1])
    cap = int(records_capacity if records_capacity is not None else max(nev, 16))
    rec = np.zeros(cap, dtype=_lib.INDEL_RECORD_DTYPE)
    n, nt = C.c_int64(0), C.c_int64(0)
    _lib.check(L.lfq_call_indels_batch(caller.h, C.byref(conf.c), C.byref(cs), rec.ctypes.data, cap,
                                       C.byref(n), C.byref(nt)), "lfq_call_indels_batch")
    return rec[: n.value], nt.value


def format_indel_record(chrom, pos0, cols, r, filter_str=None):
    """VCF line of one indel record (vcf_write_var, vcf.c:469-497, 608-629)"""
    L = _lib.load()
    ref, alt = cols.ref_alt(int(r["col"]), int(r["side"]), int(r["event"]))
    buf = C.create_string_buffer(1024 + len(ref) + len(alt))
    L.lfq_format_indel_record(buf, len(buf), chrom.encode(), int(pos0), ref.encode(), alt.encode(), int(r["qual"]),
                              int(r["dp"]), C.c_float(float(r["af"])), int(r["sb"]), int(r["ref_fw"]),
                              int(r["ref_rv"]), int(r["alt_fw"]), int(r["alt_rv"]), int(r["hrun"]),
                              None if filter_str is None else filter_str.encode())
    return buf.value.decode()


def filter_indel_records(recs, indelqual_thresh, apply_defaults=True):
    """keep mask of `lofreq filter` as run by `lofreq call` on indel records"""
    keep = np.zeros(len(recs), np.int32)
    recs = np.ascontiguousarray(recs)
    _lib.check(_lib.load().lfq_filter_indel_records(recs.ctypes.data, len(recs), int(indelqual_thresh),
                                                    1 if apply_defaults else 0, keep.ctypes.data),
               "lfq_filter_indel_records")
    return keep.astype(bool)
