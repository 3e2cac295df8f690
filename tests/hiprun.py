"""Helpers that drive the HIP path (through the C ABI) the same way oracle.freq drives the oracle."""
import numpy as np

from oracle import oracle as O
from oracle import pybam


def to_oracle_rows(rows):
    out = np.zeros(len(rows), dtype=O.ROW_DTYPE)
    out["tid"], out["pos"], out["strand"], out["code"] = rows["tid"], rows["pos"], rows["strand"], rows["code"]
    out["ins_off"], out["hp"], out["n_called"], out["n_mod"] = rows["ins_offset"], rows["hp"], rows["n_called"], rows["n_mod"]
    return out


def make_engine(mods, th, target_names, target_lens, contigs, **kw):
    import minimod_amd
    ctg = [(n, l, contigs.get(n)) for n, l in zip(target_names, target_lens)]
    # the parity tests' batches are far smaller than what the default lets k_stream_reads have: send it every read it can take,
    # so that the goldens, the KATs and the error cases go through it as well (the tile pipeline still gets what it hands
    # on, everything with --insertions / --haplotypes / view, and the stream_mode=1 runs)
    kw.setdefault("stream_mode", 2)
    return minimod_amd.FreqEngine([(c, x, t) for (c, x), t in zip(mods, th)], ctg, **kw)


def hip_freq(bam_path, contigs, c="m", m=None, insertions=False, haplotypes=False, allow_secondary=False,
             skip_supplementary=False, K=512, B=20 * 1000 * 1000, **ekw):
    """`minimod freq` with the hot path on the GPU; returns (rows as oracle dtype, names, code names)."""
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(m, len(mods))
    eng = None
    tickets = []
    for bam, batch, _st in pybam.load_batches(bam_path, K=K, B=B, allow_secondary=allow_secondary,
                                             skip_supplementary=skip_supplementary):
        if eng is None:
            eng = make_engine(mods, th, bam.target_name, bam.target_len, contigs, insertions=insertions,
                              haplotypes=haplotypes, **ekw)
        if len(batch["reads"]):
            eng.process(batch)
    rows = eng.finalize()
    names, codes = eng.names, eng.code_names()
    eng.close()
    return to_oracle_rows(rows), names, codes


def hip_rows_from_records(recs, ref, c, **kw):
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(None, len(mods))
    eng = make_engine(mods, th, ["chrT"], [len(ref)], {"chrT": ref.encode()}, **kw)
    eng.process(pybam.flatten(recs))
    rows = to_oracle_rows(eng.finalize())
    codes = eng.code_names()
    eng.close()
    return [(int(r["pos"]), "+-"[r["strand"]], int(r["n_called"]), int(r["n_mod"]), int(r["ins_off"]), int(r["hp"]),
             codes[r["code"]]) for r in rows]


def to_oracle_view_rows(rows, reads, base, haplotypes):
    """engine VIEW_ROW_DTYPE rows of one batch -> oracle VIEW_DTYPE rows (read index running over batches from `base`)."""
    out = np.zeros(len(rows), dtype=O.VIEW_DTYPE)
    rd = reads[rows["read"]]
    out["read"] = rows["read"].astype(np.int64) + base
    out["tid"], out["pos"], out["strand"] = rd["tid"], rows["pos"], (rd["flag"] & 0x10) != 0
    out["code"], out["ins_off"], out["read_pos"], out["prob"] = rows["code"], rows["ins_offset"], rows["read_pos"], rows["prob"]
    out["hp"] = rd["hp"] if haplotypes else -1
    return out


def hip_view(bam_path, contigs, c="m", insertions=False, haplotypes=False, allow_secondary=False,
             skip_supplementary=False, K=512, B=20 * 1000 * 1000, **ekw):
    """`minimod view` with the hot path on the GPU; returns (rows as oracle view dtype, qnames, names, code names)."""
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(None, len(mods))
    eng = None
    parts, qnames = [], []
    for bam, batch, _st in pybam.load_batches(bam_path, K=K, B=B, allow_secondary=allow_secondary,
                                             skip_supplementary=skip_supplementary):
        if eng is None:
            eng = make_engine(mods, th, bam.target_name, bam.target_len, contigs, insertions=insertions,
                              haplotypes=haplotypes, view=True, **ekw)
        if len(batch["reads"]):
            rows = eng.view(batch)
            parts.append(to_oracle_view_rows(rows, batch["reads"], len(qnames), haplotypes))
            qnames += batch["qnames"]
    rows = np.concatenate(parts) if parts else np.zeros(0, dtype=O.VIEW_DTYPE)
    names, codes = eng.names, eng.code_names()
    eng.close()
    return rows, qnames, names, codes


def hip_view_from_records(recs, ref, c, **kw):
    mods = O.parse_mod_codes(c)
    th = O.parse_mod_threshes(None, len(mods))
    eng = make_engine(mods, th, ["chrT"], [len(ref)], {"chrT": ref.encode()}, view=True, **kw)
    batch = pybam.flatten(recs)
    rows = eng.view(batch)
    codes = eng.code_names()
    eng.close()
    return [(int(r["read"]), int(r["pos"]), int(r["read_pos"]), codes[r["code"]], int(r["prob"]), int(r["ins_offset"])) for r in rows]
