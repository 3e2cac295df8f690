"""The library's block-keeping allocator (csrc/devmem.h): a handle's memory is kept and handed out again, never returned to the driver while
the process lives -- a virtual address the driver took back and handed out again was read through its old translation by one XCD's
workgroups when a dozen processes shared the GPU (profiles/r6_site_index_root_cause.txt).  No reference counterpart: CPU code."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stats(L):
    out = (ctypes.c_int64 * 7)()
    L.mm_devmem_stats(out)
    return dict(zip(("dev_held", "dev_kept", "pin_held", "pin_kept", "hits", "misses", "returned"), [int(x) for x in out]))


def test_a_second_handle_reuses_the_first_ones_blocks():
    import minimod_amd
    from minimod_amd import engine, synth
    from oracle import oracle as O
    L = engine.load_library()
    L.mm_devmem_stats.argtypes = [ctypes.POINTER(ctypes.c_int64)]
    L.mm_devmem_trim.restype = ctypes.c_int64
    ref = synth.reference(3, 1 << 20)
    b = synth.batch(ref, 0, 300, seed=5, n_reads_total=300)
    rows = []
    st = []
    for i in range(3):
        eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
        eng.process(b)
        rows.append(eng.finalize().copy())
        st.append(_stats(L))
        eng.close()
    orc = O.Oracle([("m", "CG")], [0.8], ["chrS"]); orc.add_contig("chrS", ref); orc.process(b)
    want = orc.rows()
    for r in rows:   # a kept block is not a zeroed one: the rows do not care
        assert len(r) == len(want) and np.array_equal(r["pos"], want["pos"]) and np.array_equal(r["n_called"], want["n_called"]) and np.array_equal(r["n_mod"], want["n_mod"])
    after = _stats(L)
    # the second and third handle asked the driver for nothing the first had not asked for, and nothing went back
    assert st[2]["misses"] == st[1]["misses"], st
    assert st[2]["hits"] > st[1]["hits"] > st[0]["hits"], st
    assert after["returned"] == st[0]["returned"] and after["dev_kept"] > 0 and after["dev_held"] < st[2]["dev_held"], (st, after)
    # a long-lived process may give the kept blocks back at a quiet moment
    freed = int(L.mm_devmem_trim())
    end = _stats(L)
    assert freed >= after["dev_kept"] and end["dev_kept"] == 0 and end["pin_kept"] == 0 and end["returned"] > after["returned"], (freed, after, end)
    eng = minimod_amd.FreqEngine([("m", "CG", 0.8)], [("chrS", len(ref), ref)])
    eng.process(b)
    r = eng.finalize(); eng.close()
    assert np.array_equal(r["n_called"], want["n_called"])
