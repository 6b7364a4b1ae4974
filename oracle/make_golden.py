#!/usr/bin/env python
"""Generate tests/golden/*.json from the REFERENCE ITSELF (its prebuilt lofreq 2.1.4 binary).

Runs only in the build container (needs /root/reference/dist; `make -C oracle ref` unpacks the binary
to oracle/_ref/lofreq214).  For every fixture:
  * a seeded SAM + FASTA is written (SAM text is auto-detected by the binary's htslib),
  * `lofreq plpsummary` dumps, per pileup column, the per-nucleotide BQ / BAQ / MQ arrays and strand
    counts -- exactly the input of the hot path (lofreq_call.c:438-599),
  * `lofreq call` produces the VCF and the "Number of substitution tests performed" line.
The JSON holds inputs (columns) and expected outputs (VCF records, test count): data only.

Known 2.1.4-vs-HEAD deltas on this path: HEAD appends ;HQA= to SNV INFO and counts raw alt
bases before the min_bq filter.  tests/test_golden.py compares modulo ;HQA= and runs the oracle with
raw_counts_after_minbq=1.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# named `lofreq`: `lofreq call` runs `lofreq filter` through the shell (lofreq_call.c:1506-1551), and every run below puts
# this directory in front of PATH
LOFREQ = os.path.join(HERE, "_ref", "bin", "lofreq")
OUT = os.environ.get("LFQ_GOLDEN_OUT") or os.path.join(HERE, "..", "tests", "golden")


def write_fixture(tmp, seed, glen, nreads, planted, mapqs, min_q=3):
    rng = np.random.default_rng(seed)
    genome = "".join(rng.choice(list("ACGT"), glen))
    open(os.path.join(tmp, "t.fa"), "w").write(">chr1\n" + genome + "\n")
    reads = []
    rl = 100
    for _ in range(nreads):
        pos = int(rng.integers(0, glen - rl + 1))
        seq = list(genome[pos:pos + rl])
        qual = np.clip(np.round(rng.normal(33, 7, rl)), min_q, 41).astype(int)
        for j in range(rl):
            if rng.random() < 10 ** (-qual[j] / 10.0):
                seq[j] = rng.choice([c for c in "ACGT" if c != seq[j]])
            p = planted.get(pos + j)
            if isinstance(p, list):              # several planted alleles at one position: [(base, frac), ...]
                u = rng.random()
                for base, frac in p:
                    if u < frac:
                        seq[j] = base
                        break
                    u -= frac
            elif p and rng.random() < p[1]:
                seq[j] = p[0]
        flag = 16 if rng.random() < 0.5 else 0
        reads.append((pos, flag, int(rng.choice(mapqs)), "".join(seq), "".join(chr(33 + q) for q in qual)))
    reads.sort()
    with open(os.path.join(tmp, "t.sam"), "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen)
        for i, (pos, flag, mapq, seq, q) in enumerate(reads):
            f.write("r%d\t%d\tchr1\t%d\t%d\t%dM\t*\t0\t0\t%s\t%s\n" % (i, flag, pos + 1, mapq, rl, seq, q))
    return genome


def parse_plpsummary(text):
    cols = []
    cur = None
    for line in text.splitlines():
        if not line.strip():
            continue
        if not line.startswith(" "):
            f = line.split("\t")
            cur = {"pos0": int(f[1]) - 1, "ref": f[2], "cons": f[3], "fwrv": {}, "obs": {}}
            for tok in f[4:9]:
                nt, c = tok.split(":")
                fw, rv = c.split("/")
                cur["fwrv"][nt] = [int(fw), int(rv)]
            cols.append(cur)
        else:
            f = line.strip().split("\t")
            key = f[0]
            if key in "ACGTN" and len(key) == 1:
                track = f[1].split("=")[0].strip()
                vals = [int(x) for x in f[2].split()] if len(f) > 2 else []
                cur["obs"].setdefault(key, {})[track] = vals
    return cols


def enc(vals):
    """phred list -> compact ASCII (value + 33; -1 -> ' ', which no value maps to: 93 + 33 is '~')"""
    return "".join(" " if v < 0 else chr(33 + v) for v in vals)


def run(name, seed, glen, nreads, planted, mapqs, call_args, keep_cols=None):
    """keep_cols: store only these positions (and their VCF lines) -- for deep fixtures, where every column is
    10 000 observations; needs a fixed Bonferroni factor in call_args so that columns are independent"""
    with tempfile.TemporaryDirectory() as tmp:
        genome = write_fixture(tmp, seed, glen, nreads, planted, mapqs)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        plp_args = [a for a in call_args if a in ("-B",)]
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa"] + plp_args + ["t.sam"], cwd=tmp, check=True,
                             capture_output=True, text=True).stdout
        env = dict(os.environ)
        env["PATH"] = os.path.dirname(os.path.abspath(LOFREQ)) + ":" + env["PATH"]
        res = subprocess.run([LOFREQ, "call", "-f", "t.fa", "-o", "out.vcf"] + call_args + ["t.sam"], cwd=tmp,
                             check=True, capture_output=True, text=True, env=env)
        ntests = None
        for line in res.stderr.splitlines():
            if "Number of substitution tests performed" in line:
                ntests = int(line.split(":")[-1])
        vcf = [l for l in open(os.path.join(tmp, "out.vcf")).read().splitlines() if not l.startswith("#")]
    cols = parse_plpsummary(plp)
    if keep_cols is not None:
        assert "-b" in call_args
        cols = [c for c in cols if c["pos0"] in keep_cols]
        vcf = [l for l in vcf if int(l.split("\t")[1]) - 1 in keep_cols]
    packed = []
    for c in cols:
        o = {}
        for nt, tr in c["obs"].items():
            mq = tr.get("MQ", [])
            o[nt] = {"bq": enc(tr.get("BQ", [])), "baq": enc(tr["BAQ"]) if "BAQ" in tr else None,
                     "mq": enc(mq) if keep_cols is not None and max(mq + [0]) <= 93 else mq}
        packed.append({"pos0": c["pos0"], "ref": c["ref"], "fwrv": c["fwrv"], "obs": o})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "call_args": call_args, "genome": genome, "columns": packed, "vcf": vcf, "num_snv_tests": ntests}
    if keep_cols is not None:
        fix["column_subset"] = True          # num_snv_tests counts every column of the run, not only the stored ones
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d columns, %d vcf records, %s tests, %d bytes" % (name, len(packed), len(vcf), ntests,
                                                                    os.path.getsize(path)))


# ---- indel fixtures ---------------------------------------------------------------------------------

def write_indel_fixture(tmp, seed, glen, nreads, sites, mapqs, noise=0.0008, planted_snvs=None):
    """sites: {pos0: [(kind, key_or_len, frac), ...]}; kind '+' inserts key after pos0, '-' deletes len
    bases after pos0.  Per-base BI/BD indel qualities are random.  Returns (genome, reads)."""
    rng = np.random.default_rng(seed)
    genome = "".join(rng.choice(list("ACGT"), glen))
    # a few homopolymer runs so that HRUN varies
    g = list(genome)
    for p0 in range(40, glen - 20, 55):
        g[p0:p0 + 5] = g[p0] * 5
    genome = "".join(g)
    open(os.path.join(tmp, "t.fa"), "w").write(">chr1\n" + genome + "\n")
    rl = 90
    reads = []
    for _ in range(nreads):
        pos = int(rng.integers(0, glen - rl - 12))
        seq, cigar, run, gp, gpos = [], [], 0, pos, []
        while len(seq) < rl and gp < glen - 8:
            seq.append(genome[gp])
            gpos.append(gp)
            run += 1
            ev = None
            if run > 4 and len(seq) < rl - 8:
                for kind, k, frac in sites.get(gp, []):
                    if rng.random() < frac:
                        ev = (kind, k)
                        break
                if ev is None and rng.random() < noise:
                    ev = ("+", str(rng.choice(list("ACGT")))) if rng.random() < 0.5 else ("-", 1)
            if ev:
                cigar.append("%dM" % run)
                run = 0
                if ev[0] == "+":
                    seq.extend(ev[1])
                    gpos.extend([None] * len(ev[1]))
                    cigar.append("%dI" % len(ev[1]))
                else:
                    cigar.append("%dD" % ev[1])
                    gp += ev[1]
            gp += 1
        if run == 0:
            continue
        cigar.append("%dM" % run)
        n = len(seq)
        qual = "".join(chr(33 + int(q)) for q in np.clip(np.round(rng.normal(35, 4, n)), 8, 41))
        if planted_snvs is not None:        # base errors by quality + planted SNVs (only when asked: keeps older fixtures)
            for j in range(n):
                if rng.random() < 10 ** (-(ord(qual[j]) - 33) / 10.0):
                    seq[j] = str(rng.choice([c for c in "ACGT" if c != seq[j]]))
                p = planted_snvs.get(gpos[j])
                if p and rng.random() < p[1]:
                    seq[j] = p[0] if p[0] != genome[gpos[j]] else "ACGT"[("ACGT".index(p[0]) + 1) % 4]
        bi = "".join(chr(33 + int(q)) for q in rng.integers(25, 50, n))
        bd = "".join(chr(33 + int(q)) for q in rng.integers(25, 50, n))
        flag = 16 if rng.random() < 0.5 else 0
        reads.append((pos, flag, int(rng.choice(mapqs)), "".join(cigar), "".join(seq), qual, bi, bd))
    reads.sort()
    with open(os.path.join(tmp, "t.sam"), "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen)
        for i, (pos, flag, mapq, cg, sq, q, bi, bd) in enumerate(reads):
            f.write("r%d\t%d\tchr1\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tBI:Z:%s\tBD:Z:%s\n"
                    % (i, flag, pos + 1, mapq, cg, sq, q, bi, bd))
    return genome, reads


def indel_strands(genome, reads):
    """Independent of the reference: per column, strands of reads overlapping it and of reads carrying an
    insertion / deletion right after it.  -> {pos0: {"cov": [fw, rv], "+": {key: [fw, rv]}, "-": {...}}}"""
    import re
    out = {}
    for pos, flag, _mq, cg, seq, *_ in reads:
        st = 1 if flag & 16 else 0
        gp, qp, last = pos, 0, None
        for n, op in re.findall(r"(\d+)([MID])", cg):
            n = int(n)
            if op == "M":
                for k in range(n):
                    out.setdefault(gp + k, {"cov": [0, 0], "+": {}, "-": {}})["cov"][st] += 1
                gp += n
                qp += n
                last = gp - 1
            elif op == "I":
                out[last]["+"].setdefault(seq[qp:qp + n], [0, 0])[st] += 1
                qp += n
            else:
                out[last]["-"].setdefault(genome[gp:gp + n], [0, 0])[st] += 1
                for k in range(n):     # deleted bases still count as pileup coverage (is_del)
                    out.setdefault(gp + k, {"cov": [0, 0], "+": {}, "-": {}})["cov"][st] += 1
                gp += n
    return out


def parse_plpsummary_indels(text):
    cols, cur = [], None
    for line in text.splitlines():
        if not line.strip():
            continue
        if not line.startswith(" "):
            f = line.split("\t")
            cur = {"pos0": int(f[1]) - 1, "ref": f[2], "cons": f[3]}
            for tok in f[9:]:
                k, v = tok.split(":")
                cur[k] = int(v)
            cur["ins"] = {"ne": {}, "events": []}
            cur["dels"] = {"ne": {}, "events": []}
            cols.append(cur)
            continue
        f = line.strip().split("\t")
        key = f[0]
        if key[0] not in "+-":
            continue
        side = cur["ins"] if key[0] == "+" else cur["dels"]
        track = f[1].split("=")[0].strip()
        vals = [int(x) for x in f[2].split()] if len(f) > 2 else []
        if key[1:] == "0":
            side["ne"][track] = vals
        else:
            if not side["events"] or side["events"][-1]["key"] != key[1:]:
                side["events"].append({"key": key[1:]})
            side["events"][-1][track] = vals
    return cols


def pack_indel_cols(plp, genome, reads):
    strands = indel_strands(genome, reads)
    packed = []
    for c in parse_plpsummary_indels(plp):
        if not (c["ins"]["events"] or c["dels"]["events"]):
            continue
        s = strands[c["pos0"]]
        col = {"pos0": c["pos0"], "ref": c["ref"], "num_tails": c["tails"], "hrun": c["hrun"]}
        n_ins = sum(len(e["IQ"]) for e in c["ins"]["events"])
        n_del = sum(len(e["IDQ"]) for e in c["dels"]["events"])
        col["num_ins"], col["num_dels"] = n_ins, n_del
        col["coverage_plp"] = len(c["ins"]["ne"].get("IDQ", [])) + n_ins
        assert col["coverage_plp"] == len(c["dels"]["ne"].get("IDQ", [])) + n_del == sum(s["cov"]), (c["pos0"],)
        col["num_non_indels"] = col["coverage_plp"] - n_ins - n_del
        for sn, sign, qk in (("ins", "+", "IQ"), ("dels", "-", "IDQ")):
            ev_fw = sum(v[0] for v in s[sign].values())
            ev_rv = sum(v[1] for v in s[sign].values())
            side = {"non_fw": s["cov"][0] - ev_fw, "non_rv": s["cov"][1] - ev_rv,
                    "ne_q": enc(c[sn]["ne"].get("IDQ", [])), "ne_mq": c[sn]["ne"].get("MQ", []), "events": []}
            for e in c[sn]["events"]:
                fw, rv = s[sign][e["key"]]
                assert fw + rv == len(e[qk]), (c["pos0"], e["key"])
                side["events"].append({"key": e["key"], "fw": fw, "rv": rv, "q": enc(e[qk]), "aq": enc(e["AQ"]),
                                       "mq": e["MQ"], "sq": enc(e["SQ"])})
            col[sn] = side
        packed.append(col)
    return packed


def run_indel(name, seed, glen, nreads, sites, mapqs, call_args, alnqual=True):
    with tempfile.TemporaryDirectory() as tmp:
        genome, reads = write_indel_fixture(tmp, seed, glen, nreads, sites, mapqs)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        bam = "t.sam"
        if alnqual:
            with open(os.path.join(tmp, "t.aq.bam"), "wb") as f:
                subprocess.check_call([LOFREQ, "alnqual", "-b", "t.sam", "t.fa"], cwd=tmp, stdout=f)
            bam = "t.aq.bam"
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa", bam], cwd=tmp, check=True, capture_output=True,
                             text=True).stdout
        env = dict(os.environ)
        env["PATH"] = os.path.dirname(os.path.abspath(LOFREQ)) + ":" + env["PATH"]
        res = subprocess.run([LOFREQ, "call", "--call-indels", "--only-indels", "-f", "t.fa", "-o", "out.vcf"]
                             + call_args + [bam], cwd=tmp, check=True, capture_output=True, text=True, env=env)
        ntests = None
        for line in res.stderr.splitlines():
            if "Number of indel tests performed" in line:
                ntests = int(line.split(":")[-1])
        vcf = [l for l in open(os.path.join(tmp, "out.vcf")).read().splitlines() if not l.startswith("#")]
    packed = pack_indel_cols(plp, genome, reads)
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "call_args": ["--call-indels", "--only-indels"] + call_args, "alnqual": alnqual, "columns": packed,
           "vcf": vcf, "num_indel_tests": ntests}
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d indel columns, %d vcf records, %s tests, %d bytes" % (name, len(packed), len(vcf), ntests,
                                                                          os.path.getsize(path)))


def main_indels():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    sites = {70: [("+", "AC", 0.10)], 100: [("-", 3, 0.08)], 130: [("+", "G", 0.03), ("+", "GGT", 0.03)],
             160: [("-", 1, 0.02), ("+", "T", 0.02)], 190: [("+", "A", 0.5)], 215: [("-", 2, 0.01)],
             240: [("-", 5, 0.3), ("+", "CCCC", 0.05)]}
    run_indel("indel_default", 21, 330, 900, sites, mq_mix, ["--no-default-filter"])
    run_indel("indel_fixedbonf_noidaq", 22, 330, 900, sites, mq_mix, ["--no-default-filter", "-b", "100", "-A"])
    run_indel("indel_default_filter", 23, 330, 700, sites, [60] * 10 + [20], [])


# ---- BAQ fixtures -------------------------------------------------------------------

def run_baq(name, seed, glen, nreads, sites, mapqs, extra=()):
    """reads with and without indels -> `lofreq alnqual` (SAM out) -> the lb:Z tag of every read"""
    with tempfile.TemporaryDirectory() as tmp:
        genome, reads = write_indel_fixture(tmp, seed, glen, nreads, sites, mapqs)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        sam = subprocess.run([LOFREQ, "alnqual"] + list(extra) + ["t.sam", "t.fa"], cwd=tmp, check=True,
                             capture_output=True, text=True).stdout
    out = []
    for line in sam.splitlines():
        if line.startswith("@"):
            continue
        f = line.split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        out.append({"pos0": int(f[3]) - 1, "flag": int(f[1]), "cigar": f[5], "seq": f[9], "qual": f[10],
                    "lb": tags.get("lb"), "ai": tags.get("ai"), "ad": tags.get("ad")})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "alnqual_args": list(extra), "genome": genome, "reads": out}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d reads, %d with lb, %d bytes" % (name, len(out), sum(1 for r in out if r["lb"]), os.path.getsize(path)))


# ---- reads -> VCF chain fixtures: BAQ computed by `lofreq call` itself ---------------------------------------

def run_chain(name, seed, glen, nreads, planted, mapqs, call_args):
    """all-M reads (base qualities >= 6 so that the 2.1.4 / HEAD raw-count delta cannot show) -> `lofreq call`
    with its on-the-fly BAQ -> VCF.  The fixture holds the READS, not the columns: the test has to run BAQ,
    build the pileup and call."""
    with tempfile.TemporaryDirectory() as tmp:
        genome = write_fixture(tmp, seed, glen, nreads, planted, mapqs, min_q=6)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        env = dict(os.environ)
        env["PATH"] = os.path.dirname(os.path.abspath(LOFREQ)) + ":" + env["PATH"]
        res = subprocess.run([LOFREQ, "call", "-f", "t.fa", "-o", "out.vcf"] + call_args + ["t.sam"], cwd=tmp,
                             check=True, capture_output=True, text=True, env=env)
        ntests = None
        for line in res.stderr.splitlines():
            if "Number of substitution tests performed" in line:
                ntests = int(line.split(":")[-1])
        vcf = [l for l in open(os.path.join(tmp, "out.vcf")).read().splitlines() if not l.startswith("#")]
        reads = []
        for line in open(os.path.join(tmp, "t.sam")):
            if line.startswith("@"):
                continue
            f = line.rstrip("\n").split("\t")
            reads.append([int(f[3]) - 1, int(f[1]), int(f[4]), f[5], f[9], f[10]])
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "call_args": call_args, "genome": genome, "reads": reads, "vcf": vcf, "num_snv_tests": ntests}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d reads, %d vcf records, %s tests, %d bytes" % (name, len(reads), len(vcf), ntests, os.path.getsize(path)))


def main_chain():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    planted = {60: ("A", 0.05), 61: ("C", 0.05), 90: ("C", 0.10), 120: ("G", 0.03), 150: ("T", 0.5), 180: ("C", 1.0),
               200: ("T", 0.07), 230: ("A", 0.02)}
    run_chain("chain_default", 41, 300, 700, planted, mq_mix, [])
    run_chain("chain_nofilter", 42, 300, 700, planted, mq_mix, ["--no-default-filter"])


def run_pileup(name, seed, glen, nreads, sites, mapqs):
    """reads with indels + their lb tags (alnqual) and the binary's own column dump (plpsummary) of the same BAM"""
    with tempfile.TemporaryDirectory() as tmp:
        genome, reads = write_indel_fixture(tmp, seed, glen, nreads, sites, mapqs)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        with open(os.path.join(tmp, "t.aq.bam"), "wb") as f:
            subprocess.check_call([LOFREQ, "alnqual", "-b", "t.sam", "t.fa"], cwd=tmp, stdout=f)
        sam = subprocess.run([LOFREQ, "alnqual", "t.sam", "t.fa"], cwd=tmp, check=True, capture_output=True,
                             text=True).stdout
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa", "t.aq.bam"], cwd=tmp, check=True, capture_output=True,
                             text=True).stdout
    out = []
    for line in sam.splitlines():
        if line.startswith("@"):
            continue
        f = line.split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        out.append([int(f[3]) - 1, int(f[1]), int(f[4]), f[5], f[9], f[10], tags.get("lb")])
    cols = []
    for c in parse_plpsummary(plp):
        o = {}
        for nt, tr in c["obs"].items():
            o[nt] = {"bq": enc(tr.get("BQ", [])), "baq": enc(tr["BAQ"]) if "BAQ" in tr else None, "mq": tr.get("MQ", [])}
        cols.append({"pos0": c["pos0"], "ref": c["ref"], "fwrv": c["fwrv"], "obs": o})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "genome": genome, "reads": out, "columns": cols}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d reads, %d columns, %d bytes" % (name, len(out), len(cols), os.path.getsize(path)))


def main_pileup():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    sites = {70: [("+", "AC", 0.10)], 100: [("-", 3, 0.08)], 130: [("+", "G", 0.03), ("+", "GGT", 0.03)],
             190: [("+", "A", 0.5)], 240: [("-", 12, 0.3)]}
    run_pileup("pileup_indels", 51, 330, 300, sites, mq_mix)


# ---- source quality fixtures: per-read sq as seen through the binary's own column dump (plpsummary -s) -----

def write_srcq_fixture(tmp, seed, glen, nreads, sites, mapqs):
    """the indel reads of write_indel_fixture with read-specific mismatch rates (0 .. 25 %), base qualities down to
    2 (below source_qual's min_bq 6), occasional soft clips and N bases"""
    genome, reads = write_indel_fixture(tmp, seed, glen, nreads, sites, mapqs, noise=0.004)
    rng = np.random.default_rng(seed + 1000)
    out = []
    for (pos, flag, mapq, cg, sq, q, bi, bd) in reads:
        rate = float(rng.choice([0.0] * 14 + [0.01, 0.03, 0.08, 0.25]))
        seq = list(sq)
        qual = [ord(c) - 33 for c in q]
        if "I" not in cg and "D" not in cg:         # planted SNVs (reads without indels: offsets are direct)
            for p0, alt, frac in ((50, "A", 0.3), (85, "C", 0.4), (120, "G", 0.5), (160, "T", 0.3), (210, "C", 0.2),
                                  (211, "G", 0.2), (275, "A", 0.6)):
                if pos <= p0 < pos + len(seq) and rng.random() < frac:
                    seq[p0 - pos] = alt if genome[p0] != alt else "T"
        for j in range(len(seq)):
            if rng.random() < rate:
                seq[j] = str(rng.choice([c for c in "ACGTN" if c != seq[j]]))
            if rng.random() < 0.1:
                qual[j] = int(rng.integers(2, 20))
        if rng.random() < 0.2:
            k = int(rng.integers(1, 6))
            seq = list(rng.choice(list("ACGT"), k)) + seq
            qual = [int(x) for x in rng.integers(10, 40, k)] + qual
            cg = "%dS" % k + cg
        if rng.random() < 0.2:
            k = int(rng.integers(1, 6))
            seq = seq + list(rng.choice(list("ACGT"), k))
            qual = qual + [int(x) for x in rng.integers(10, 40, k)]
            cg = cg + "%dS" % k
        out.append((pos, flag, mapq, cg, "".join(seq), "".join(chr(33 + x) for x in qual)))
    with open(os.path.join(tmp, "t.sam"), "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen)
        for i, (pos, flag, mapq, cg, sq, q) in enumerate(out):
            f.write("r%d\t%d\tchr1\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\n" % (i, flag, pos + 1, mapq, cg, sq, q))
    return genome, out


def enc_mq(vals):
    return "".join("%02x" % v for v in vals)


def run_srcq(name, seed, glen, nreads, sites, mapqs, extra=(), ign=()):
    """`lofreq plpsummary -s -B`: BQ / MQ / SQ of every column; `lofreq call -s -B`: the VCF with the source
    quality merged in (snpcaller.c:303-341).  ign: 0-based positions written to a VCF for -S/--ign-vcf."""
    with tempfile.TemporaryDirectory() as tmp:
        genome, reads = write_srcq_fixture(tmp, seed, glen, nreads, sites, mapqs)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        sargs = ["-s"] + list(extra)
        if ign:
            with open(os.path.join(tmp, "ign.vcf"), "w") as f:
                f.write("##fileformat=VCFv4.0\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
                for p0 in ign:
                    f.write("chr1\t%d\t.\t%s\t%s\t100\tPASS\tDP=10\n" % (p0 + 1, genome[p0], "A" if genome[p0] != "A" else "C"))
            sargs += ["-S", "ign.vcf"]
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa", "-B"] + sargs + ["t.sam"], cwd=tmp, check=True,
                             capture_output=True, text=True).stdout
        env = dict(os.environ)
        env["PATH"] = os.path.dirname(os.path.abspath(LOFREQ)) + ":" + env["PATH"]
        # source quality is harsh (every read with two or more non-matches counts as a near-certain error), so
        # that nothing is significant at the defaults: fixed Bonferroni factor 1 and sig 0.9 give records to compare
        call_args = ["--no-default-filter", "-b", "1", "-a", "0.9"]
        res = subprocess.run([LOFREQ, "call", "-f", "t.fa", "-B", "-o", "out.vcf"] + call_args + sargs + ["t.sam"],
                             cwd=tmp, check=True, capture_output=True, text=True, env=env)
        ntests = None
        for line in res.stderr.splitlines():
            if "Number of substitution tests performed" in line:
                ntests = int(line.split(":")[-1])
        vcf = [l for l in open(os.path.join(tmp, "out.vcf")).read().splitlines() if not l.startswith("#")]
    cols = []
    for c in parse_plpsummary(plp):
        o = {}
        for nt, tr in c["obs"].items():
            o[nt] = {"bq": enc(tr.get("BQ", [])), "mq": enc_mq(tr.get("MQ", [])),
                     "sq": enc([-1 if v > 253 else v for v in tr.get("SQ", [])])}
        cols.append({"pos0": c["pos0"], "ref": c["ref"], "fwrv": c["fwrv"], "obs": o})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "args": sargs, "call_args": call_args, "ign": list(ign), "genome": genome,
           "encoding": "bq / sq: chr(33 + value), sq ' ' = 49314 (PROB_TO_PHREDQUAL(LDBL_MIN)); mq: 2 hex digits",
           "reads": [[r[0], r[1], r[2], r[3], r[4], r[5]] for r in reads], "columns": cols, "vcf": vcf,
           "num_snv_tests": ntests}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    nsq = sorted({v for c in cols for o in c["obs"].values() for v in o["sq"]})
    print("%s: %d reads, %d columns, %d vcf records, %d bytes; distinct SQ: %s" % (name, len(reads), len(cols), len(vcf),
                                                                                   os.path.getsize(path), nsq[:40]))


def main_srcq():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    sites = {70: [("+", "AC", 0.10)], 100: [("-", 3, 0.08)], 130: [("+", "G", 0.03), ("+", "GGT", 0.03)],
             190: [("+", "A", 0.5)], 240: [("-", 12, 0.3)]}
    run_srcq("srcq_default", 61, 330, 220, sites, mq_mix)
    run_srcq("srcq_ign_nmq", 62, 330, 220, sites, mq_mix, extra=("-T", "20"), ign=(70, 99, 100, 150, 189, 190, 240))


# ---- reads -> indel columns -> indel (and SNV) calls: the pileup's indel fields ------------------------------

def run_plpindel(name, seed, glen, nreads, sites, mapqs, call_args, planted=None):
    """reads with BI / BD (+ lb / ai / ad from `lofreq alnqual`), the binary's indel column dump (plpsummary) and
    the VCFs of `lofreq call --call-indels` with and without --only-indels"""
    with tempfile.TemporaryDirectory() as tmp:
        planted = planted or {50: ("A", 0.3), 120: ("G", 0.08), 121: ("T", 0.05), 205: ("C", 0.02), 260: ("A", 0.5)}
        genome, reads = write_indel_fixture(tmp, seed, glen, nreads, sites, mapqs, planted_snvs=planted)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        with open(os.path.join(tmp, "t.aq.bam"), "wb") as f:
            subprocess.check_call([LOFREQ, "alnqual", "-b", "t.sam", "t.fa"], cwd=tmp, stdout=f)
        sam = subprocess.run([LOFREQ, "alnqual", "t.sam", "t.fa"], cwd=tmp, check=True, capture_output=True,
                             text=True).stdout
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa", "t.aq.bam"], cwd=tmp, check=True,
                             capture_output=True, text=True).stdout
        env = dict(os.environ)
        env["PATH"] = os.path.dirname(os.path.abspath(LOFREQ)) + ":" + env["PATH"]
        out = {}
        for tag, extra in (("indels", ["--only-indels"]), ("all", [])):
            res = subprocess.run([LOFREQ, "call", "--call-indels", "-f", "t.fa", "-o", "out_%s.vcf" % tag] + extra
                                 + call_args + ["t.aq.bam"], cwd=tmp, check=True, capture_output=True, text=True, env=env)
            nt = {}
            for line in res.stderr.splitlines():
                if "tests performed" in line:
                    nt["indel" if "indel" in line else "snv"] = int(line.split(":")[-1])
            vcf = [l for l in open(os.path.join(tmp, "out_%s.vcf" % tag)).read().splitlines() if not l.startswith("#")]
            out[tag] = {"vcf": vcf, "num_tests": nt}
    rd = []
    for line in sam.splitlines():
        if line.startswith("@"):
            continue
        f = line.split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        rd.append([int(f[3]) - 1, int(f[1]), int(f[4]), f[5], f[9], f[10], tags.get("BI"), tags.get("BD"),
                   tags.get("lb"), tags.get("ai"), tags.get("ad")])
    packed = pack_indel_cols(plp, genome, reads)
    cons = {c["pos0"]: c["cons"] for c in parse_plpsummary_indels(plp)}
    for col in packed:
        col["cons"] = cons[col["pos0"]]
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "call_args": ["--call-indels"] + call_args, "genome": genome,
           "read_fields": "pos0, flag, mapq, cigar, seq, qual, BI, BD, lb, ai, ad", "reads": rd, "columns": packed,
           "only_indels": out["indels"], "all": out["all"]}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d reads, %d indel columns, %d / %d vcf records, tests %s / %s, %d bytes"
          % (name, len(rd), len(packed), len(out["indels"]["vcf"]), len(out["all"]["vcf"]), out["indels"]["num_tests"],
             out["all"]["num_tests"], os.path.getsize(path)))


def main_plpindel():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    sites = {70: [("+", "AC", 0.10)], 100: [("-", 3, 0.08)], 130: [("+", "G", 0.03), ("+", "GGT", 0.03)],
             160: [("-", 1, 0.02), ("+", "T", 0.02)], 190: [("+", "A", 0.5)], 215: [("-", 2, 0.01)],
             240: [("-", 5, 0.3), ("+", "CCCC", 0.05)]}
    run_plpindel("plpindel_default", 71, 330, 350, sites, mq_mix, ["--no-default-filter"])
    # consensus indels: SNVs planted at (and next to) columns where most reads carry an insertion / deletion
    sites2 = {70: [("+", "AC", 0.85)], 100: [("-", 3, 0.9)], 160: [("-", 1, 0.55), ("+", "T", 0.1)], 230: [("+", "G", 0.6)]}
    planted2 = {70: ("A", 0.3), 71: ("C", 0.3), 100: ("G", 0.3), 160: ("T", 0.4), 230: ("C", 0.35), 280: ("A", 0.2)}
    run_plpindel("plpindel_consindel", 72, 330, 300, sites2, mq_mix, ["--no-default-filter"], planted=planted2)


# ---- lofreq uniq --use-det-lim -------------------------------------------------------------

def run_uniq(name, seed, glen, nreads, mapqs):
    """variants with assorted AFs against a BAM: `lofreq uniq --use-det-lim --output-all` says which ones would
    have been detectable (UNIQ flag); the columns are what uniq's own mpileup sees (no BAQ, MAPQ >= 1,
    lofreq_uniq.c:461-465) = `plpsummary -B -m 1`"""
    with tempfile.TemporaryDirectory() as tmp:
        genome = write_fixture(tmp, seed, glen, nreads, {}, mapqs)
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        with open(os.path.join(tmp, "t.bam"), "wb") as f:      # alnqual -b: the handiest SAM -> BAM converter here
            subprocess.check_call([LOFREQ, "alnqual", "-b", "t.sam", "t.fa"], cwd=tmp, stdout=f)
        subprocess.check_call([LOFREQ, "index", "t.bam"], cwd=tmp)
        rng = np.random.default_rng(seed + 5)
        afs = [0.001, 0.004, 0.008, 0.012, 0.02, 0.03, 0.05, 0.08, 0.15, 0.3, 0.6, 1.0]
        var = []
        for p0 in range(5, glen - 5, 3):
            ref = genome[p0]
            alt = str(rng.choice([c for c in "ACGT" if c != ref]))
            var.append((p0, ref, alt, float(rng.choice(afs))))
        with open(os.path.join(tmp, "v.vcf"), "w") as f:
            f.write("##fileformat=VCFv4.0\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
            for p0, ref, alt, af in var:
                f.write("chr1\t%d\t.\t%s\t%s\t100\tPASS\tDP=100;AF=%f\n" % (p0 + 1, ref, alt, af))
        res = subprocess.run([LOFREQ, "uniq", "--use-det-lim", "--output-all", "-v", "v.vcf", "-o", "-", "t.bam"],
                             cwd=tmp, check=True, capture_output=True, text=True).stdout
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa", "-B", "-m", "1", "t.bam"], cwd=tmp, check=True,
                             capture_output=True, text=True).stdout
    uniq = {}
    for line in res.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        uniq[int(f[1]) - 1] = "UNIQ" in f[7].split(";")
    cols = {c["pos0"]: c for c in parse_plpsummary(plp)}
    out = []
    for p0, ref, alt, af in var:
        c = cols.get(p0)
        if c is None or p0 not in uniq:
            continue
        o = {nt: {"bq": enc(tr.get("BQ", [])), "mq": enc_mq(tr.get("MQ", []))} for nt, tr in c["obs"].items()}
        out.append({"pos0": p0, "ref": ref, "alt": alt, "af": "%f" % af, "fwrv": c["fwrv"], "obs": o, "uniq": uniq[p0]})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "encoding": "bq: chr(33 + value); mq: 2 hex digits; af: the string written to the VCF (strtof)",
           "variants": out}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d variants, %d UNIQ, %d bytes" % (name, len(out), sum(v["uniq"] for v in out), os.path.getsize(path)))


def run_uniq_binom(name, seed, glen, nreads, mapqs):
    """The default mode of `lofreq uniq` (uniq_snv's binomial branch, lofreq_uniq.c:335-393, + apply_uniq_filter_mtc,
    :140-206): variants called in one sample tested against this BAM, in which some of them are present at another
    (or the same) frequency.  Per variant the binary's UQ= value (binom() through cdflib's cdfbin) and whether it ends
    up PASS under the default FDR correction (alpha 0.001, ntests = number of variants)."""
    rng = np.random.default_rng(seed + 5)
    genome_rng = np.random.default_rng(seed)
    genome = "".join(genome_rng.choice(list("ACGT"), glen))
    afs = [0.004, 0.01, 0.02, 0.05, 0.08, 0.15, 0.3, 0.6, 0.95]
    present = [0.0, 0.0, 0.0, 0.003, 0.01, 0.03, 0.1, 0.3, 0.6]
    var, planted = [], {}
    for p0 in range(5, glen - 5, 3):
        ref = genome[p0]
        alt = str(rng.choice([c for c in "ACGT" if c != ref]))
        af = float(rng.choice(afs))
        here = float(rng.choice(present))
        var.append((p0, ref, alt, af))
        if here > 0:
            planted[p0] = (alt, here)
    with tempfile.TemporaryDirectory() as tmp:
        g2 = write_fixture(tmp, seed, glen, nreads, planted, mapqs)
        assert g2 == genome
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        with open(os.path.join(tmp, "t.bam"), "wb") as f:
            subprocess.check_call([LOFREQ, "alnqual", "-b", "t.sam", "t.fa"], cwd=tmp, stdout=f)
        subprocess.check_call([LOFREQ, "index", "t.bam"], cwd=tmp)
        with open(os.path.join(tmp, "v.vcf"), "w") as f:
            f.write("##fileformat=VCFv4.0\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
            for p0, ref, alt, af in var:
                f.write("chr1\t%d\t.\t%s\t%s\t100\tPASS\tDP=100;AF=%f\n" % (p0 + 1, ref, alt, af))
        res = subprocess.run([LOFREQ, "uniq", "--output-all", "-v", "v.vcf", "-o", "-", "t.bam"],
                             cwd=tmp, check=True, capture_output=True, text=True).stdout
        plp = subprocess.run([LOFREQ, "plpsummary", "-f", "t.fa", "-B", "-m", "1", "t.bam"], cwd=tmp, check=True,
                             capture_output=True, text=True).stdout
    got = {}
    for line in res.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        info = dict(x.split("=") for x in f[7].split(";") if "=" in x)
        got[int(f[1]) - 1] = (int(info["UQ"]) if "UQ" in info else None, f[6])
    cols = {c["pos0"]: c for c in parse_plpsummary(plp)}
    out = []
    for p0, ref, alt, af in var:
        c = cols.get(p0)
        if c is None or p0 not in got:
            continue
        o = {nt: {"bq": enc(tr.get("BQ", [])), "mq": enc_mq(tr.get("MQ", []))} for nt, tr in c["obs"].items()}
        out.append({"pos0": p0, "ref": ref, "alt": alt, "af": "%f" % af, "fwrv": c["fwrv"], "obs": o,
                    "uq": got[p0][0], "filter": got[p0][1]})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "encoding": "bq: chr(33 + value); mq: 2 hex digits; af: the string written to the VCF (strtof)",
           "mtc": "fdr", "alpha": 0.001, "variants": out}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d variants, %d PASS, UQ range %s..%s, %d bytes" % (
        name, len(out), sum(v["filter"] == "PASS" for v in out), min(v["uq"] for v in out if v["uq"] is not None),
        max(v["uq"] for v in out if v["uq"] is not None), os.path.getsize(path)))


def main_uniq():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    run_uniq("uniq_detlim", 81, 400, 700, mq_mix)
    run_uniq_binom("uniq_binom", 82, 400, 1500, mq_mix)


# ---- lofreq filter, every mode (lofreq_filter.c) ---------------------------------------------------------------

FILTER_CASES = [
    [],                                                       # the defaults: SB by FDR at 0.001 + compound rule, DP >= 10
    ["--no-defaults"],
    ["--no-defaults", "-v", "40", "-V", "900"],
    ["--no-defaults", "-a", "0.02", "-A", "0.6"],
    ["--no-defaults", "-Q", "60", "-K", "45"],
    ["--no-defaults", "-q", "bonf", "-r", "0.0001"],
    ["--no-defaults", "-q", "holm", "-r", "0.00001", "-s", "5000"],
    ["--no-defaults", "-q", "fdr", "-r", "0.000001", "-s", "100000"],
    ["--no-defaults", "-k", "bonf", "-l", "0.001", "-m", "300"],
    ["--no-defaults", "-k", "fdr", "-l", "0.00001"],
    ["--no-defaults", "-B", "30"],
    ["--no-defaults", "-B", "30", "--sb-no-compound"],
    ["--no-defaults", "-B", "30", "--sb-incl-indels"],
    ["--no-defaults", "-b", "bonf", "-c", "0.001"],
    ["--no-defaults", "-b", "holm", "-c", "0.01", "--sb-no-compound", "--sb-incl-indels"],
    ["-b", "fdr", "-c", "0.05", "-v", "25", "-q", "fdr", "-k", "holm", "-a", "0.004"],
    ["--no-defaults", "--only-snvs", "-Q", "80"],
    ["--no-defaults", "--only-indels", "-k", "bonf"],
]


def run_filter(name, seed, n, quals=(20, 44, 45, 46, 59, 60, 61, 75, 90, 140, 400, 3000, 49314), sbs=(0, 0, 1, 3, 12, 29, 30, 31, 45, 80, 200, 2147483647), case_list=None):
    """a VCF of `n` seeded variants (SNVs and indels; QUAL, DP, AF, SB, DP4 over the ranges the filters cut at, one missing
    QUAL) through the binary's `lofreq filter --print-all` for every option set of FILTER_CASES: the FILTER column of every
    variant and the ##FILTER lines it added; and without --print-all: which variants are written"""
    rng = np.random.default_rng(seed)
    lines, pos = [], 100
    for i in range(n):
        pos += int(rng.integers(1, 40))
        is_indel = rng.random() < 0.3
        dp = int(rng.choice([5, 9, 10, 11, 24, 25, 60, 300, 899, 900, 901, 4000]))
        alt = int(max(1, round(dp * float(rng.choice([0.003, 0.01, 0.02, 0.05, 0.3, 0.6, 0.95])))))
        alt = min(alt, dp)
        skew = float(rng.choice([0.5, 0.5, 0.8, 0.86, 0.97, 1.0]))
        afw = int(round(alt * skew)) if rng.random() < 0.5 else alt - int(round(alt * skew))
        arv = alt - afw
        rfw = int((dp - alt) * float(rng.choice([0.5, 0.5, 0.3])))
        rrv = dp - alt - rfw
        sb = int(rng.choice(list(sbs)))
        qual = int(rng.choice(list(quals)))
        qs = "." if i == 7 else str(qual)
        if is_indel:
            ref, al = ("AC", "A") if rng.random() < 0.5 else ("A", "AGG")
            info = "DP=%d;AF=%f;SB=%d;DP4=%d,%d,%d,%d;INDEL;HRUN=%d" % (dp, alt / float(dp), sb, rfw, rrv, afw, arv, int(rng.integers(1, 6)))
        else:
            ref, al = "A", "G"
            info = "DP=%d;AF=%f;SB=%d;DP4=%d,%d,%d,%d" % (dp, alt / float(dp), sb, rfw, rrv, afw, arv)
        lines.append("chr1\t%d\t.\t%s\t%s\t%s\t.\t%s" % (pos, ref, al, qs, info))
    head = "##fileformat=VCFv4.0\n##source=lofreq call\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "in.vcf"), "w").write(head + "\n".join(lines) + "\n")
        for args in (case_list or FILTER_CASES):
            res = {"args": args}
            for tag, extra in (("all", ["--print-all"]), ("passed", [])):
                p = subprocess.run([LOFREQ, "filter", "-i", "in.vcf", "-o", "out.vcf"] + args + extra, cwd=tmp,
                                   capture_output=True, text=True)
                assert p.returncode == 0, (args, p.stderr)
                out = open(os.path.join(tmp, "out.vcf")).read().splitlines()
                os.remove(os.path.join(tmp, "out.vcf"))
                body = [l.split("\t") for l in out if not l.startswith("#")]
                if tag == "all":
                    res["filter_lines"] = [l for l in out if l.startswith("##FILTER")]
                    res["pos"] = [int(f[1]) for f in body]
                    res["filter"] = [f[6] for f in body]
                else:
                    res["passed_pos"] = [int(f[1]) for f in body]
            cases.append(res)
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "vcf": lines, "cases": cases}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d variants, %d option sets, %d bytes" % (name, n, len(cases), os.path.getsize(path)))


def main_filter():
    run_filter("filter_modes", 91, 90)
    # no QUAL / SB large enough for pow(10, -q / 10) to underflow: with a correction requested the AF filter still works
    # here (in filter_modes it is switched off by the errno the first pass leaves behind, lofreq_filter.c:252-261)
    run_filter("filter_modes_small", 92, 40, quals=(20, 45, 60, 75, 140, 400), sbs=(0, 1, 12, 30, 31, 80, 200),
               case_list=[["-q", "fdr", "-a", "0.02", "-A", "0.5"], ["--no-defaults", "-b", "bonf", "-a", "0.02"],
                      ["--no-defaults", "-k", "holm", "-A", "0.5", "-V", "500"]])



def run_baq_iupac(name, seed):
    """IUPAC ambiguity codes in reads and reference (bam_md_ext.c:176-200: the repeat scan of idaq compares LETTERS): a
    genome with R / Y / M / N letters, insertions whose inserted bases repeat the reference letters behind them -- ambiguity
    letters included, identical and different ones -- and ambiguity codes inside match blocks -> `lofreq alnqual` -> lb / ai /
    ad of every read."""
    rng = np.random.default_rng(seed)
    glen, rl = 400, 80
    g = list(rng.choice(list("ACGT"), glen))
    sites = {}
    # (site, letters of the reference right behind it, what the read inserts)
    plan = [(60, "R", "R"), (100, "RR", "R"), (140, "YN", "YN"), (180, "M", "R"), (220, "N", "R"), (260, "R", "N"),
            (300, "AR", "AR"), (330, "K", "K")]
    for p0, after, ins in plan:
        g[p0 + 1:p0 + 1 + len(after)] = list(after)
        sites[p0] = ins
    for p0 in (75, 155, 245):                      # ambiguity letters inside match blocks
        g[p0] = "S"
    genome = "".join(g)
    reads = []
    for i in range(160):
        pos = int(rng.integers(0, glen - rl - 6))
        seq, cigar, run, gp = [], [], 0, pos
        while len(seq) < rl and gp < glen - 2:
            c = genome[gp]
            if c not in "ACGT" and rng.random() < 0.5:
                c = str(rng.choice(list("ACGT")))          # half of the reads resolve an ambiguous reference letter
            seq.append(c)
            run += 1
            if gp in sites and run > 3 and len(seq) < rl - 8 and rng.random() < 0.6:
                cigar.append("%dM" % run)
                run = 0
                seq.extend(sites[gp])
                cigar.append("%dI" % len(sites[gp]))
            gp += 1
        if run == 0:
            continue
        cigar.append("%dM" % run)
        if rng.random() < 0.1:
            seq[int(rng.integers(0, len(seq)))] = str(rng.choice(list("RYWN")))
        qual = "".join(chr(33 + int(q)) for q in np.clip(np.round(rng.normal(35, 4, len(seq))), 8, 41))
        reads.append((pos, 16 if rng.random() < 0.5 else 0, 60, "".join(cigar), "".join(seq), qual))
    reads.sort()
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "t.fa"), "w").write(">chr1\n" + genome + "\n")
        with open(os.path.join(tmp, "t.sam"), "w") as f:
            f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:%d\n" % glen)
            for i, (pos, flag, mapq, cg, sq, q) in enumerate(reads):
                f.write("r%d\t%d\tchr1\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\n" % (i, flag, pos + 1, mapq, cg, sq, q))
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        sam = subprocess.run([LOFREQ, "alnqual", "t.sam", "t.fa"], cwd=tmp, check=True, capture_output=True, text=True).stdout
    out = []
    for line in sam.splitlines():
        if line.startswith("@"):
            continue
        f = line.split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        out.append({"pos0": int(f[3]) - 1, "flag": int(f[1]), "cigar": f[5], "seq": f[9], "qual": f[10],
                    "lb": tags.get("lb"), "ai": tags.get("ai"), "ad": tags.get("ad")})
    fix = {"name": name, "generator": "oracle/make_golden.py", "reference_binary": "lofreq 2.1.4 (dist tgz)",
           "alnqual_args": [], "genome": genome, "reads": out}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d reads, %d with ai, %d bytes" % (name, len(out), sum(1 for r in out if r["ai"]), os.path.getsize(path)))


def main_baq():
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    sites = {70: [("+", "AC", 0.10)], 100: [("-", 3, 0.08)], 130: [("+", "G", 0.03), ("+", "GGT", 0.03)],
             160: [("-", 1, 0.02), ("+", "T", 0.02)], 190: [("+", "A", 0.5)], 215: [("-", 2, 0.01)],
             240: [("-", 12, 0.3), ("+", "CCCCCCCCCC", 0.05)]}
    run_baq("baq_extended", 31, 330, 250, sites, mq_mix)
    run_baq("baq_plain", 32, 330, 250, sites, mq_mix, extra=("-e",))
    run_baq_iupac("baq_iupac", 33)


# ---- real-size fixtures: the reads are NOT stored, only how to make them (tests/golden_reads.py) ------------------------------

def run_big(name, params, call_args, note):
    """A C1- or C4-shaped run of the 2.1.4 binary (BASELINE.json configs[0] / [3]): seeded reads from tests/golden_reads.py
    -> SAM -> `lofreq call` (+ its `lofreq filter` epilogue unless --no-default-filter, lofreq_call.c:1506-1551).  The fixture
    holds the generator's parameters and version, the SHA-256 of the SAM text, the binary's VCF lines and its test counts."""
    import time
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    import golden_reads as gr
    with tempfile.TemporaryDirectory() as tmp:
        R = gr.make(**params)
        open(os.path.join(tmp, "t.fa"), "w").write(">chr1\n" + R["ref"].decode() + "\n")
        sha = gr.write_sam(R, os.path.join(tmp, "t.sam"))
        subprocess.check_call([LOFREQ, "faidx", "t.fa"], cwd=tmp)
        env = dict(os.environ)
        env["PATH"] = os.path.dirname(os.path.abspath(LOFREQ)) + ":" + env["PATH"]
        t0 = time.time()
        res = subprocess.run([LOFREQ, "call", "-f", "t.fa", "-o", "out.vcf"] + call_args + ["t.sam"], cwd=tmp,
                             check=True, capture_output=True, text=True, env=env)
        secs = time.time() - t0
        nt = {}
        for line in res.stderr.splitlines():
            if "tests performed" in line:
                nt["indel" if "indel" in line else "snv"] = int(line.split(":")[-1])
        vcf = [l for l in open(os.path.join(tmp, "out.vcf")).read().splitlines() if not l.startswith("#")]
    n_ind = sum(1 for l in vcf if "INDEL" in l.split("\t")[7])
    fix = {"name": name, "generator": {"module": "tests/golden_reads.py", "version": gr.GENERATOR_VERSION, "params": params},
           "reference_binary": "lofreq 2.1.4 (dist tgz)", "call_args": call_args, "note": note,
           "n_reads": int(R["n"]), "n_bases": int(R["seq_off"][-1]), "sam_sha256": sha, "num_tests": nt, "vcf": vcf,
           "n_snv_lines": len(vcf) - n_ind, "n_indel_lines": n_ind, "binary_seconds_in_the_build_container": round(secs, 1)}
    path = os.path.join(OUT, name + ".json")
    json.dump(fix, open(path, "w"), separators=(",", ":"))
    print("%s: %d reads, %d SNV + %d indel vcf records, tests %s, %.1f s in the binary, %d bytes"
          % (name, R["n"], len(vcf) - n_ind, n_ind, nt, secs, os.path.getsize(path)))


def main_big():
    # C1 shape (tests/bonf_auto_vs_dyn.sh:10-30: denv2, 10.7 kb, depth 10^3..10^4): `lofreq call` defaults = extended BAQ on the
    # fly, dynamic Bonferroni, the `lofreq filter` epilogue.  Qualities >= 6: the 2.1.4 / HEAD raw-count delta cannot show.
    run_big("big_c1_default", dict(seed=601, glen=10700, depth_lo=1000, depth_hi=5000, min_q=6), [],
            "C1 shape, lofreq call defaults")
    # the same shape with base qualities from 2 (alt bases below min_bq 6 exist: AF differs between 2.1.4 and HEAD where
    # such bases carry the alt allele; everything else must still agree) and without the default filter
    run_big("big_c1_lowbq_nofilter", dict(seed=602, glen=10700, depth_lo=1000, depth_hi=3000, min_q=2, low_q_frac=0.04),
            ["--no-default-filter"], "C1 shape, low base qualities (documented 2.1.4-vs-HEAD raw-count delta), no default filter")
    # C4 shape at a size the binary does in a minute: 500x, planted insertions / deletions, BI / BD tags, --call-indels
    run_big("big_c4_indels", dict(seed=603, glen=24000, depth_lo=500, depth_hi=500, min_q=6, snv_every=60, indel_every=240),
            ["--call-indels"], "C4 shape (500x, --call-indels, defaults otherwise)")


def main():
    if "--big-only" in sys.argv:
        return main_big()
    if "--indels-only" in sys.argv:
        return main_indels()
    if "--baq-only" in sys.argv:
        return main_baq()
    if "--chain-only" in sys.argv:
        return main_chain()
    if "--pileup-only" in sys.argv:
        return main_pileup()
    if "--srcq-only" in sys.argv:
        return main_srcq()
    if "--plpindel-only" in sys.argv:
        return main_plpindel()
    if "--uniq-only" in sys.argv:
        return main_uniq()
    if "--deep10k-only" in sys.argv:
        return main_deep10k()
    if "--filter-only" in sys.argv:
        return main_filter()
    if not os.path.exists(LOFREQ):
        sys.exit("reference binary missing: run `make -C oracle ref` in the build container")
    mq_mix = [60] * 24 + [40, 30, 20, 10, 0, 255]
    planted_a = {60: ("A", 0.05), 61: ("C", 0.05), 62: ("G", 0.05), 90: ("C", 0.10), 91: ("A", 0.10),
                 92: ("T", 0.10), 120: ("G", 0.03), 121: ("A", 0.03), 122: ("C", 0.03), 150: ("T", 0.5),
                 151: ("A", 0.5), 152: ("G", 0.5), 180: ("C", 1.0), 181: ("A", 1.0), 200: ("T", 0.07),
                 201: ("C", 0.07), 202: ("A", 0.07)}
    run("snv_default", 11, 260, 500, planted_a, mq_mix, [])
    run("snv_nofilter_fixedbonf", 12, 260, 500, planted_a, mq_mix, ["--no-default-filter", "-b", "780"])
    run("snv_nobaq_dynamic_nofilter", 13, 260, 500, planted_a, mq_mix, ["-B", "--no-default-filter"])
    planted_b = {50: ("A", 0.01), 51: ("C", 0.004), 70: ("G", 0.2), 71: ("T", 0.2), 72: ("A", 0.6),
                 100: ("C", 0.03), 101: ("G", 0.03)}
    run("snv_deep", 14, 160, 1800, planted_b, [60] * 12 + [50, 3], ["--no-default-filter", "-b", "480"])
    run("snv_minbq_sig", 15, 220, 600, planted_a, mq_mix, ["-q", "20", "-Q", "25", "-a", "0.001", "-b", "660",
                                                         "--no-default-filter"])
    if "--snv-only" in sys.argv:
        return
    main_deep10k()
    main_indels()
    main_baq()
    main_chain()
    main_pileup()


def main_deep10k():
    """BASELINE.json configs[2] regime on the reference binary itself: 10 000 reads covering positions 30..99 of a
    130 bp contig (depth 10 000 there), planted 0.5 / 1 / 5 / 50 % alleles, two columns with a second, minor allele
    (QUAL 49314, the FE-clamp quirk that loses or keeps minor alleles).  14 of the columns stored."""
    planted = {40: ("A", 0.005), 45: ("C", 0.01), 50: ("G", 0.05), 55: ("T", 0.5),
               60: [("A", 0.5), ("C", 0.006)], 65: [("G", 0.05), ("T", 0.01)], 72: ("C", 0.002), 80: [("A", 0.3), ("G", 0.3)]}
    keep = {38, 39, 40, 41, 45, 50, 55, 60, 65, 70, 72, 75, 80, 90}
    run("snv_deep10k", 16, 130, 10000, planted, [60] * 18 + [40, 20], ["--no-default-filter", "-b", "390"], keep_cols=keep)


if __name__ == "__main__":
    main()
