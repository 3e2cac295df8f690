"""Random differential run of the device BGZF inflate on VALID streams no compressor would write: DEFLATE streams (RFC 1951) made
here token by token -- stored / fixed / dynamic blocks in any order, random complete Huffman codes up to 15 bits (lopsided ones
on purpose), code-length runs (16 / 17 / 18) that cross the literal/distance boundary, HLIT / HDIST / HCLEN larger than needed,
a single distance code, no distance code, an end-of-block-only block, matches of every length at every distance incl. overlapping
ones.  zlib on the host is the judge of the generator (a stream it does not decode to the intended bytes is the generator's bug and
is counted, not sent); the device must return exactly those bytes with status 0, or refuse the block (status != 0: the host
decoder's then -- reported as `refused`, not as a problem).
usage: tools/fuzz_inflate.py <first seed> <count> [streams per seed]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_XB = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_XB = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self):
        self.out = bytearray(); self.acc = 0; self.n = 0
    def put(self, v, nb):          # nb bits of v, least significant first
        self.acc |= (v & ((1 << nb) - 1)) << self.n; self.n += nb
        while self.n >= 8:
            self.out.append(self.acc & 255); self.acc >>= 8; self.n -= 8
    def code(self, c, nb):         # a Huffman code: most significant bit first
        r = 0
        for _ in range(nb): r = (r << 1) | (c & 1); c >>= 1
        self.put(r, nb)
    def align(self):
        if self.n: self.put(0, 8 - self.n)
    def done(self):
        self.align(); return bytes(self.out)


def len_sym(l):
    s = max(i for i in range(29) if LEN_BASE[i] <= l)
    if l == 258: s = 28
    return s, l - LEN_BASE[s]

def dist_sym(d):
    s = max(i for i in range(30) if DIST_BASE[i] <= d)
    return s, d - DIST_BASE[s]

def random_lengths(rng, used, maxlen, shape):
    """code lengths of a COMPLETE prefix code over the symbols `used` (>= 2 of them), none longer than maxlen"""
    k = len(used)
    assert 2 <= k <= (1 << maxlen)
    depth = [1, 1]
    while len(depth) < k:
        cand = [i for i, d in enumerate(depth) if d < maxlen]
        if shape == 0: i = cand[int(rng.integers(len(cand)))]                  # any leaf
        elif shape == 1: i = max(cand, key=lambda j: (depth[j], rng.random()))   # the deepest: a vine, codes of maxlen bits
        else: i = min(cand, key=lambda j: (depth[j], rng.random()))              # the shallowest: balanced
        d = depth.pop(i); depth += [d + 1, d + 1]
    order = list(used); rng.shuffle(order)
    return {s: d for s, d in zip(order, depth)}

def canonical(lengths, n):
    """{symbol: length} -> {symbol: (code, length)} as RFC 1951 3.2.2 numbers them"""
    cnt = [0] * 16
    for l in lengths.values(): cnt[l] += 1
    nxt = [0] * 16; c = 0
    for b in range(1, 16):
        c = (c + cnt[b - 1]) << 1; nxt[b] = c
    out = {}
    for s in range(n):
        l = lengths.get(s, 0)
        if l: out[s] = (nxt[l], l); nxt[l] += 1
    return out

def random_tokens(rng, out, budget):
    """tokens of one block appended to the decoded bytes `out`: ('L', byte) / ('M', length, distance)"""
    toks = []
    n_tok = int(rng.choice([0, 1, 3, 20, 200, 2000, 12000]))
    p_match = float(rng.choice([0.0, 0.1, 0.5, 0.9, 1.0]))
    alpha = rng.integers(0, 256, int(rng.choice([1, 2, 4, 16, 64, 256])))
    dmode = int(rng.integers(0, 5))
    lmode = int(rng.integers(0, 4))
    for _ in range(n_tok):
        room = budget - len(out)
        if room <= 0: break
        if len(out) > 0 and room >= 3 and rng.random() < p_match:
            top = min(len(out), 32768)
            if dmode == 0: d = 1 + int(rng.integers(top))
            elif dmode == 1: d = 1 + int(rng.integers(min(top, 4)))              # overlapping copies
            elif dmode == 2: d = max(1, top - int(rng.integers(min(top, 8))))    # the window's far end
            elif dmode == 3: d = min(top, int(DIST_BASE[int(rng.integers(30))]) + int(rng.integers(2)))   # bucket edges
            else: d = min(top, 1 << int(rng.integers(16)))
            if lmode == 0: l = 3 + int(rng.integers(256))
            elif lmode == 1: l = int(rng.choice([3, 4, 10, 11, 257, 258, 227, 226, 130, 131]))
            elif lmode == 2: l = 3 + int(rng.integers(8))
            else: l = 258
            l = min(l, room)
            if l < 3: continue
            toks.append(("M", l, d))
            for _ in range(l): out.append(out[-d])
        else:
            b = int(alpha[int(rng.integers(len(alpha)))])
            toks.append(("L", b)); out.append(b)
    return toks

def write_tokens(bw, toks, lit, dist):
    for t in toks:
        if t[0] == "L": bw.code(*lit[t[1]])
        else:
            s, x = len_sym(t[1]); bw.code(*lit[257 + s])
            if LEN_XB[s]: bw.put(x, LEN_XB[s])
            s, x = dist_sym(t[2]); bw.code(*dist[s])
            if DIST_XB[s]: bw.put(x, DIST_XB[s])
    bw.code(*lit[256])

def fixed_codes():
    ll = {s: (8 if s < 144 else 9 if s < 256 else 7 if s < 280 else 8) for s in range(288)}
    return canonical(ll, 288), canonical({s: 5 for s in range(30)}, 30)

def dynamic_block(rng, bw, toks):
    used_l = {256} | {t[1] for t in toks if t[0] == "L"} | {257 + len_sym(t[1])[0] for t in toks if t[0] == "M"}
    used_d = {dist_sym(t[2])[0] for t in toks if t[0] == "M"}
    for _ in range(int(rng.choice([0, 0, 1, 5, 40]))): used_l.add(int(rng.integers(286)))       # coded but never sent
    for _ in range(int(rng.choice([0, 0, 1, 5, 20]))): used_d.add(int(rng.integers(30)))
    shape = int(rng.integers(0, 3))
    if len(used_l) == 1:
        ll = {256: 1} if rng.random() < 0.5 else None                                          # one code of one bit (incomplete, allowed)
        if ll is None: used_l.add(int(rng.integers(256)))
    else: ll = None
    if ll is None: ll = random_lengths(rng, sorted(used_l), 15, shape)
    if len(used_d) == 0: dl = {} if rng.random() < 0.5 else {int(rng.integers(30)): 1}
    elif len(used_d) == 1: dl = {next(iter(used_d)): 1} if rng.random() < 0.5 else None
    else: dl = None
    if dl is None:
        if len(used_d) == 1: used_d.add((next(iter(used_d)) + 1 + int(rng.integers(29))) % 30)
        dl = random_lengths(rng, sorted(used_d), 15, int(rng.integers(0, 3)))
    hlit = max(257, max(ll) + 1); hlit = min(286, hlit + int(rng.choice([0, 0, 1, 8, 29])))
    hdist = max(1, (max(dl) + 1) if dl else 1); hdist = min(30, hdist + int(rng.choice([0, 0, 1, 8, 29])))
    seq = [ll.get(s, 0) for s in range(hlit)] + [dl.get(s, 0) for s in range(hdist)]           # ONE sequence: runs may cross the boundary
    # code-length symbols
    rle = rng.random() < 0.8
    cl = []; i = 0
    while i < len(seq):
        v = seq[i]; run = 1
        while i + run < len(seq) and seq[i + run] == v: run += 1
        if rle and v == 0 and run >= 3 and rng.random() < 0.9:
            r = min(run, 138 if rng.random() < 0.8 else 10); r = int(rng.integers(3, r + 1)) if rng.random() < 0.3 else r
            cl.append((18, r - 11) if r >= 11 else (17, r - 3)); i += r
        elif rle and i > 0 and seq[i - 1] == v and run >= 3 and rng.random() < 0.9:
            r = min(run, 6); cl.append((16, r - 3)); i += r
        else:
            cl.append((v, 0)); i += 1
    used_c = {c for c, _ in cl}
    while len(used_c) < 2: used_c.add(int(rng.integers(19)))
    for _ in range(int(rng.choice([0, 0, 2, 6]))): used_c.add(int(rng.integers(19)))
    cll = random_lengths(rng, sorted(used_c), 7, int(rng.integers(0, 3)))
    clc = canonical(cll, 19)
    hclen = max(4, 1 + max(k for k in range(19) if CL_ORDER[k] in cll)); hclen = min(19, hclen + int(rng.choice([0, 0, 1, 15])))
    bw.put(hlit - 257, 5); bw.put(hdist - 1, 5); bw.put(hclen - 4, 4)
    for k in range(hclen): bw.put(cll.get(CL_ORDER[k], 0), 3)
    for c, x in cl:
        bw.code(*clc[c])
        if c == 16: bw.put(x, 2)
        elif c == 17: bw.put(x, 3)
        elif c == 18: bw.put(x, 7)
    write_tokens(bw, toks, canonical(ll, 286), canonical(dl, 30))

def random_stream(rng):
    """-> (raw DEFLATE bytes, decoded bytes)"""
    budget = int(rng.choice([0, 1, 100, 5000, 30000, 65280]))
    out = bytearray(); bw = Bits()
    n_blocks = int(rng.choice([1, 1, 2, 3, 8]))
    for b in range(n_blocks):
        last = b == n_blocks - 1
        kind = int(rng.integers(0, 4))   # dynamic twice as likely
        bw.put(1 if last else 0, 1)
        if kind == 0:
            n = min(int(rng.choice([0, 1, 7, 300, 9000, 65280])), budget - len(out))
            data = bytes(rng.integers(0, int(rng.choice([2, 256])), n, dtype=np.uint8))
            bw.put(0, 2); bw.align(); bw.put(n, 16); bw.put(n ^ 0xFFFF, 16)
            bw.out += data; out += data
        elif kind == 1:
            toks = random_tokens(rng, out, budget)
            bw.put(1, 2); write_tokens(bw, toks, *fixed_codes())
        else:
            toks = random_tokens(rng, out, budget)
            bw.put(2, 2); dynamic_block(rng, bw, toks)
    return bw.done(), bytes(out)

def make(seed, per_seed):
    rng = np.random.default_rng(seed)
    streams, bad = [], 0
    while len(streams) < per_seed:
        raw, want = random_stream(rng)
        if len(raw) > 65000: continue          # no BGZF block holds it
        try:
            ok = zlib.decompress(raw, -15) == want
        except zlib.error:
            ok = False
        if not ok: bad += 1; continue
        streams.append((raw, want))
    return streams, bad

def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    per_seed = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    from minimod_amd import bgzf
    inf = bgzf.Inflater(slots=1, max_blocks=4096, max_cbytes=64 << 20, max_obytes=64 << 20)
    problems = refused = total = gen_bad = 0
    refused_codes = {}
    for seed in range(first, first + count):
        streams, bad = make(seed, per_seed); gen_bad += bad
        blocks = [(raw, len(want), zlib.crc32(want) & 0xFFFFFFFF) for raw, want in streams]
        k = 0
        while k < len(blocks):           # launches of at most 48 MB of decoded bytes
            j, o = k, 0
            while j < len(blocks) and o + blocks[j][1] <= (48 << 20): o += blocks[j][1]; j += 1
            got, status = inf.inflate(blocks[k:j])
            for i in range(k, j):
                total += 1
                s = int(status[i - k])
                if s != 0:
                    refused += 1; refused_codes[s] = refused_codes.get(s, 0) + 1
                    print("REFUSED seed %d stream %d status %d (%d -> %d bytes)" % (seed, i, s, len(blocks[i][0]), blocks[i][1]))
                elif got[i - k] != streams[i][1]:
                    problems += 1
                    print("PROBLEM seed %d stream %d: status 0 and other bytes" % (seed, i))
            k = j
    inf.close()
    print("fuzz_inflate: seeds %d..%d, %d streams, %d problems, %d refused %s, %d generator rejects" %
          (first, first + count - 1, total, problems, refused, refused_codes or "", gen_bad))
    return 1 if problems else 0

if __name__ == "__main__":
    sys.exit(main())
