#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own test data.  Run in the dev container only
(needs /root/reference); the outputs are committed so nothing reads /root/reference at test time.

What it produces (all DATA, no reference source text):
  * copies of the bundled input BAMs the freq golden tests use (reference test/data/*.bam),
  * copies of the reference's expected outputs for those tests (reference test/expected/test*.tsv|bedmethyl),
  * pseudo-reference patch lists (position, base) replacing the genome FASTAs that the reference's
    test/test.sh:31-41 downloads and that are absent offline:
      - chr22: every row of test/expected/*.bed with a `ref_kmer` column carries `ref_position` and a
        forward-strand 5-mer; painting them onto an all-N chr22 gives every base the '?'-flag BAMs
        (example-ont, example-hifi, dna_5mCG_5hmCG) ever look up,
      - chr1: hap.bam and eb.bam carry MD:Z; CIGAR+MD+SEQ reconstruct the reference bases under
        every aligned read base.
    Conflicts between sources are counted and must be zero.
"""
import glob
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pybam  # noqa: E402

REF = "/root/reference/test"

BAMS = ["example-ont.bam", "example-hifi.bam", "hap.bam", "eb.bam", "dna_5mCG_5hmCG_mm_chr22.bam",
        "dna_5mC_5hmC_mm_chr22.bam", "dna_4mC_5mC_mm_chr22.bam", "dna_6mA_mm_chr22.bam", "dRNA.bam",
        "dna_5mCG_5hmCG_mm_with_secondary_chr22.bam"]
EXPECTED = ["test3.tsv", "test4.bedmethyl", "test5.tsv", "test5a.tsv", "test5b.tsv", "test5c.tsv",
            "test6.bedmethyl", "test7.tsv", "test8.tsv", "test9.tsv", "test12.tsv", "test16.tsv",
            "test2.tsv", "test2c.tsv", "test10.tsv", "test11.tsv", "test15.tsv",
            "test1.tsv", "test2a.tsv", "test2b.tsv", "test2c_wild.tsv", "test17a.tsv",
            "test18.tsv", "dna_5mCG_5hmCG_mm_with_secondary_chr22_summary.tsv",
            "dna_5mCG_5hmCG_mm_with_secondary_chr22_summary_sec.tsv", "dna_5mCG_5hmCG_mm_with_secondary_chr22_summary_nosup.tsv",
            "dna_5mCG_5hmCG_mm_with_secondary_chr22_summary_sec_nosup.tsv"]


def chr22_patches():
    length = None
    bam = pybam.BamFile(os.path.join(REF, "data", "example-ont.bam"))
    length = bam.target_len[bam.target_name.index("chr22")]
    bam.close()
    known = {}
    conflicts = 0
    nfiles = 0
    for path in sorted(glob.glob(os.path.join(REF, "expected", "*.bed"))):
        with open(path) as f:
            hdr = f.readline().rstrip("\n").split("\t")
            if "ref_kmer" not in hdr:
                continue
            ci, pi, ki = hdr.index("chrom"), hdr.index("ref_position"), hdr.index("ref_kmer")
            used = False
            for line in f:
                t = line.rstrip("\n").split("\t")
                if t[ci] != "chr22":
                    continue
                p = int(t[pi])
                if p < 0:
                    continue
                kmer = t[ki].upper()
                for i, ch in enumerate(kmer):
                    q = p - 2 + i
                    if q < 0 or q >= length or ch in "-.":
                        continue
                    if q in known and known[q] != ch:
                        conflicts += 1
                    known[q] = ch
                used = True
            nfiles += used
    print("chr22: %d files, %d bases, %d conflicts" % (nfiles, len(known), conflicts))
    assert conflicts == 0
    return length, known


def chr1_patches():
    known = {}
    conflicts = 0
    length = None
    for name in ("hap.bam", "eb.bam"):
        bam = pybam.BamFile(os.path.join(REF, "data", name))
        length = bam.target_len[bam.target_name.index("chr1")]
        for r in bam:
            md = r.md()
            if md is None or (r.flag & 4) or bam.target_name[r.tid] != "chr1":
                continue
            seq = r.seq_str()
            md = md.decode()
            # tokenise MD: numbers, ^DEL, single mismatch letters
            toks = []
            i = 0
            while i < len(md):
                if md[i].isdigit():
                    j = i
                    while j < len(md) and md[j].isdigit():
                        j += 1
                    toks.append(("=", int(md[i:j])))
                    i = j
                elif md[i] == "^":
                    j = i + 1
                    while j < len(md) and md[j].isalpha():
                        j += 1
                    toks.append(("^", md[i + 1:j]))
                    i = j
                else:
                    toks.append(("X", md[i]))
                    i += 1
            ti = 0
            tleft = toks[0][1] if toks and toks[0][0] == "=" else 0
            q, p = 0, r.pos

            def put(pos, ch):
                nonlocal conflicts
                ch = ch.upper()
                if pos in known and known[pos] != ch:
                    conflicts += 1
                known[pos] = ch
            for c in r.cigar:
                op, ln = int(c) & 15, int(c) >> 4
                if op in (0, 7, 8):
                    for _ in range(ln):
                        while ti < len(toks) and toks[ti][0] == "=" and tleft == 0:
                            ti += 1
                            if ti < len(toks) and toks[ti][0] == "=":
                                tleft = toks[ti][1]
                        if ti < len(toks) and toks[ti][0] == "=":
                            put(p, seq[q])
                            tleft -= 1
                        elif ti < len(toks) and toks[ti][0] == "X":
                            put(p, toks[ti][1])
                            ti += 1
                            if ti < len(toks) and toks[ti][0] == "=":
                                tleft = toks[ti][1]
                        q += 1
                        p += 1
                elif op == 2:
                    while ti < len(toks) and toks[ti][0] == "=" and tleft == 0:
                        ti += 1
                        if ti < len(toks) and toks[ti][0] == "=":
                            tleft = toks[ti][1]
                    if ti < len(toks) and toks[ti][0] == "^":
                        for k, ch in enumerate(toks[ti][1]):
                            put(p + k, ch)
                        ti += 1
                        if ti < len(toks) and toks[ti][0] == "=":
                            tleft = toks[ti][1]
                    p += ln
                elif op == 3:
                    p += ln
                elif op in (1, 4):
                    q += ln
        bam.close()
    print("chr1: %d bases, %d conflicts" % (len(known), conflicts))
    assert conflicts == 0
    return length, known


def save_patches(name, contig, length, known):
    pos = np.array(sorted(known), dtype=np.int32)
    base = np.array([ord(known[int(p)]) for p in pos], dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name), contig=np.array(contig), length=np.int64(length),
                        pos=pos, base=base)


def main():
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    os.makedirs(os.path.join(HERE, "expected"), exist_ok=True)
    for b in BAMS:
        shutil.copyfile(os.path.join(REF, "data", b), os.path.join(HERE, "data", b))
    for e in EXPECTED:
        shutil.copyfile(os.path.join(REF, "expected", e), os.path.join(HERE, "expected", e))
    length, known = chr22_patches()
    save_patches("pseudo_chr22.npz", "chr22", length, known)
    length, known = chr1_patches()
    save_patches("pseudo_chr1.npz", "chr1", length, known)


if __name__ == "__main__":
    main()
