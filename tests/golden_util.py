"""Load tests/golden/*.json (generated from the reference's own lofreq 2.1.4 binary by
oracle/make_golden.py) into packed host tracks."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "snv_*.json")))


def indel_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "indel_*.json")))


def load_indels(path):
    """-> (fixture dict, column dicts for lofreq_amd.indel.IndelColumns.from_columns)"""
    fx = json.load(open(path))
    cols = []
    for c in fx["columns"]:
        d = {k: c[k] for k in ("ref", "coverage_plp", "num_tails", "num_non_indels", "num_ins", "num_dels", "hrun")}
        for sn in ("ins", "dels"):
            s = c[sn]
            d[sn] = {"non_fw": s["non_fw"], "non_rv": s["non_rv"], "ne_q": dec(s["ne_q"]).tolist(),
                     "ne_mq": s["ne_mq"],
                     "events": [{"key": e["key"], "fw": e["fw"], "rv": e["rv"], "q": dec(e["q"]).tolist(),
                                 "aq": dec(e["aq"]).tolist(), "mq": e["mq"], "sq": dec(e["sq"]).tolist()}
                                for e in s["events"]]}
        cols.append(d)
    return fx, cols


def dec(s):
    return np.array([(-1 if ch == " " else ord(ch) - 33) for ch in s], dtype=np.int64)


def load(path):
    fx = json.load(open(path))
    nts, bqs, baqs, mqs, off, refs = [], [], [], [], [0], []
    has_baq = "-B" not in fx["call_args"]
    for col in fx["columns"]:
        n_col = 0
        for code, nt in enumerate("ACGTN"):
            o = col["obs"].get(nt)
            if not o:
                continue
            bq = dec(o["bq"])
            n = len(bq)
            fw = col["fwrv"][nt][0]
            assert fw + col["fwrv"][nt][1] == n
            strand = (np.arange(n) >= fw).astype(np.uint8)
            nts.append((np.full(n, code, np.uint8)) | (strand << 3))
            bqs.append(bq.astype(np.uint8))
            mqs.append((dec(o["mq"]) if isinstance(o["mq"], str) else np.asarray(o["mq"])).astype(np.uint8))
            if has_baq:
                b = dec(o["baq"])
                baqs.append(np.where(b < 0, 255, b).astype(np.uint8))
            n_col += n
        off.append(off[-1] + n_col)
        refs.append(ord(col["ref"]))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint8)
    host = dict(nt=cat(nts), bq=cat(bqs), baq=cat(baqs) if has_baq else None, mq=cat(mqs), sq=None,
                col_off=np.asarray(off, np.uint64), ref_base=np.asarray(refs, np.uint8))
    return fx, host


def conf_kwargs(call_args):
    """lofreq call options (lofreq_call.c:1068-1304) -> varcall_conf fields"""
    kw = {}
    no_default_filter = False
    flag = 3
    it = iter(call_args)
    for a in it:
        if a == "--no-default-filter":
            no_default_filter = True
        elif a == "--call-indels":
            flag |= 8                       # VARCALL_USE_IDAQ stays on only when indels are called (:1325-1328)
        elif a == "--only-indels":
            pass
        elif a == "-A":
            flag &= ~8
        elif a == "-b":
            kw["bonf_dynamic"] = 0
            kw["bonf_subst"] = int(next(it))
        elif a == "-B":
            flag &= ~1
        elif a == "-q":
            kw["min_bq"] = int(next(it))
        elif a == "-Q":
            kw["min_alt_bq"] = int(next(it))
        elif a == "-a":
            kw["sig"] = float(next(it))
        elif a == "-s":
            flag |= 4
        elif a == "-T":
            next(it)                        # def_nm_q belongs to the per-read source quality, not to varcall_conf
        elif a == "-S":
            next(it)
        else:
            raise ValueError(a)
    kw["flag"] = flag
    return kw, no_default_filter


def strip_hqa(line):
    return line.split(";HQA=")[0]


def baq_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "baq_*.json")))


def parse_cigar(s):
    import re
    return [(op, int(n)) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", s)]


SEQ_LETTERS = "ACGTN=MRSVWYHKDB"       # include/lofreq_amd.h: base codes 0..4 and, for the other letters of a BAM base, 5..15


def load_baq(path):
    """-> (fixture, list of reads as dicts: pos0, cigar [(op, len)], seq codes, qual phred, lb bytes or None)"""
    fx = json.load(open(path))
    code = {c: i for i, c in enumerate(SEQ_LETTERS)}
    reads = []
    for r in fx["reads"]:
        reads.append({"pos0": r["pos0"], "cigar": parse_cigar(r["cigar"]),
                      "seq": np.array([code.get(c, 4) for c in r["seq"].upper()], np.uint8),
                      "qual": np.array([ord(c) - 33 for c in r["qual"]], np.uint8),
                      "lb": None if r["lb"] is None else np.frombuffer(r["lb"].encode(), np.uint8)})
    return fx, reads


def chain_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "chain_*.json")))


def pileup_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "pileup_*.json")))


def srcq_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "srcq_*.json")))


def load_srcq(path):
    """-> (fixture, reads as dicts for the batch APIs, def_nm_q, ign mask over the genome or None)"""
    fx = json.load(open(path))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    reads = [{"pos0": r[0], "cigar": parse_cigar(r[3]), "seq": np.array([code.get(c, 4) for c in r[4]], np.uint8),
              "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16)}
             for r in fx["reads"]]
    nmq = int(fx["args"][fx["args"].index("-T") + 1]) if "-T" in fx["args"] else -1
    ign = None
    if fx["ign"]:
        ign = np.zeros(len(fx["genome"]), np.uint8)
        ign[fx["ign"]] = 1
    return fx, reads, nmq, ign


def py_pileup(reads, min_plp_bq=3):
    """plain restatement of which (read, qpos) land in which column / nucleotide list, in pileup order
    (compile_plp_col, plp.c:905-960): -> {pos0: {letter: [(read index, qpos), ...]}}"""
    cols = {}
    for ri, r in enumerate(reads):
        x, y = r["pos0"], 0
        for op, l in r["cigar"]:
            if op in "M=X":
                for i in range(l):
                    if r["qual"][y + i] >= min_plp_bq:
                        cols.setdefault(x + i, {}).setdefault("ACGTN"[min(int(r["seq"][y + i]), 4)], []).append((ri, y + i))
                x += l
                y += l
            elif op in "IS":
                y += l
            elif op in "DN":
                x += l
    return cols


def plpindel_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "plpindel_*.json")))


def load_plpindel(path, with_alnqual_tags=True):
    """-> (fixture, reads as dicts incl. bi / bd and, if wanted, the lb / ai / ad tags `lofreq alnqual` wrote)"""
    fx = json.load(open(path))
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    tag = lambda t: None if t is None else np.frombuffer(t.encode(), np.uint8)
    reads = []
    for r in fx["reads"]:
        d = {"pos0": r[0], "cigar": parse_cigar(r[3]), "seq": np.array([code.get(c, 4) for c in r[4]], np.uint8),
             "qual": np.array([ord(c) - 33 for c in r[5]], np.uint8), "mapq": r[2], "reverse": bool(r[1] & 16),
             "bi": tag(r[6]), "bd": tag(r[7])}
        if with_alnqual_tags:
            d.update(lb=tag(r[8]), ai=tag(r[9]), ad=tag(r[10]))
        reads.append(d)
    return fx, reads


def uniq_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "uniq_detlim*.json")))


def uniq_binom_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "uniq_binom*.json")))


def load_uniq(path):
    """-> (fixture, packed host tracks of the variants' columns (no BAQ: uniq's mpileup), af float32 per column)"""
    fx = json.load(open(path))
    nts, bqs, mqs, off, refs, af = [], [], [], [0], [], []
    for v in fx["variants"]:
        n_col = 0
        for code, nt in enumerate("ACGTN"):
            o = v["obs"].get(nt)
            if not o:
                continue
            bq = dec(o["bq"])
            n = len(bq)
            fw = v["fwrv"][nt][0]
            strand = (np.arange(n) >= fw).astype(np.uint8)
            nts.append(np.full(n, code, np.uint8) | (strand << 3))
            bqs.append(bq.astype(np.uint8))
            mqs.append(np.array([int(o["mq"][2 * i:2 * i + 2], 16) for i in range(n)], np.uint8))
            n_col += n
        off.append(off[-1] + n_col)
        refs.append(ord(v["ref"]))
        af.append(np.float32(v["af"]))          # strtof of the VCF's AF string (lofreq_uniq.c:262)
    host = dict(nt=np.concatenate(nts), bq=np.concatenate(bqs), baq=None, mq=np.concatenate(mqs), sq=None,
                col_off=np.asarray(off, np.uint64), ref_base=np.asarray(refs, np.uint8))
    return fx, host, np.asarray(af, np.float32)


def py_indel_pileup(reads, ref, min_plp_idq=0):
    """Plain restatement of the indel part of compile_plp_col (plp.c:1019-1192) over htslib's pileup entries
    (resolve_cigar2): -> {pos0: dict(cov, tails, non_indels, n_ins, n_dels, non_fw[2], non_rv[2], ne[2] = [(q, mq)],
    ev[2] = ordered {key: [(q, aq, mq, sq, rev)]})} -- the independent check of lfq_pileup_indel_columns."""
    cols = {}

    def col(p):
        return cols.setdefault(p, dict(cov=0, tails=0, non_indels=0, n_ins=0, n_dels=0, non_fw=[0, 0], non_rv=[0, 0],
                                       ne=[[], []], ev=[{}, {}]))
    for r in reads:
        cig = r["cigar"]
        end = r["pos0"] + sum(l for o, l in cig if o in "MDN=X") - 1
        x, y = r["pos0"], 0
        lq = len(r["seq"])
        rev = 1 if r["reverse"] else 0
        for k, (op, l) in enumerate(cig):
            if op in "M=XDN":
                is_del = op in "DN"
                indel_last = 0
                if k + 1 < len(cig):
                    o2, l2 = cig[k + 1]
                    if o2 == "D":
                        indel_last = -l2
                    elif o2 == "I":
                        indel_last = l2
                    elif o2 == "P" and k + 2 < len(cig):
                        l3 = 0
                        for o3, ll in cig[k + 2:]:
                            if o3 == "I":
                                l3 += ll
                            elif o3 in "DMN=X":
                                break
                        indel_last = l3
                for j in range(l):
                    c = col(x + j)
                    qpos = min(y if is_del else y + j, lq - 1)
                    iq = int(r["bi"][qpos]) - 33 if r.get("bi") is not None else 0
                    dq = int(r["bd"][qpos]) - 33 if r.get("bd") is not None else 0
                    indel = indel_last if j == l - 1 else 0
                    c["cov"] += 1
                    if not is_del and x + j == end:
                        c["tails"] += 1
                    if iq < min_plp_idq or dq < min_plp_idq:
                        continue
                    mq, sq = r["mapq"], (-1 if r.get("sq") is None else r["sq"])
                    if indel > 0:
                        key = "".join(SEQ_LETTERS[b] for b in r["seq"][qpos + 1:qpos + 1 + indel])      # seq_nt16_str letters (plp.c:1092)
                        aq = int(r["ai"][qpos]) - 33 if r.get("ai") is not None else -1
                        c["ev"][0].setdefault(key, []).append((iq, aq, mq, sq, rev))
                        c["n_ins"] += 1
                        c["ne"][1].append((dq, mq))
                        c["non_rv" if rev else "non_fw"][1] += 1
                    elif indel < 0:
                        key = "".join((ref[g] if g < len(ref) else "N") for g in range(x + j + 1, x + j + 1 - indel)).upper()
                        aq = int(r["ad"][qpos]) - 33 if r.get("ad") is not None else -1
                        c["ev"][1].setdefault(key, []).append((dq, aq, mq, sq, rev))
                        c["n_dels"] += 1
                        c["ne"][0].append((iq, mq))
                        c["non_rv" if rev else "non_fw"][0] += 1
                    else:
                        c["non_indels"] += 1
                        c["ne"][0].append((iq, mq))
                        c["ne"][1].append((dq, mq))
                        for sd in range(2):
                            c["non_rv" if rev else "non_fw"][sd] += 1
                x += l
                if not is_del:
                    y += l
            elif op in "IS":
                y += l
    return cols
