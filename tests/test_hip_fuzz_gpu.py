"""The random differential campaigns of tools/fuzz_*.py, a bounded number of seeds each, inside the driver-run suite: random
batches (mixed group shapes, the options, long reads and CIGARs, several contigs, interval shards with halo slabs, corrupted
records) through every HIP mode against the oracle.  The tools are run as they are (`python tools/fuzz_X.py <first> <count>`),
all six at once; a campaign passes when its last line reports 0 problems."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAMPAIGNS = ["fuzz_mixed", "fuzz_options", "fuzz_long", "fuzz_contigs", "fuzz_shards", "fuzz_errors", "fuzz_tie"]   # (fuzz_tie, round 5: the device-side
# replay of the reference's row order against the host's serial restatement, the same calls to both)
FIRST = int(os.environ.get("MM_FUZZ_FIRST", "31000"))
COUNT = int(os.environ.get("MM_FUZZ_SEEDS", "50"))


@pytest.fixture(scope="module")
def campaigns():
    # (the error campaign gets four times the seeds: a corrupted record of the kind that shows a deviation -- a reverse read whose
    # leading clip overshoots the sequence -- comes once in sixty batches; it is the quickest of the six)
    procs = {c: subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", c + ".py"), str(FIRST if c != "fuzz_errors" else 400),
                                  str(COUNT if c not in ("fuzz_errors", "fuzz_tie") else (4 * COUNT if c == "fuzz_errors" else max(4, COUNT // 4)))], stdout=subprocess.PIPE,
                                 stderr=subprocess.STDOUT, cwd=ROOT) for c in CAMPAIGNS}
    out = {}
    for c, p in procs.items():
        try:
            txt, _ = p.communicate(timeout=1500)
        except subprocess.TimeoutExpired:
            p.kill()
            txt, _ = p.communicate()
            txt += b"\nTIMEOUT"
        out[c] = (p.returncode, txt.decode(errors="replace"))
    return out


@pytest.mark.parametrize("name", CAMPAIGNS)
def test_random_campaign_agrees_with_the_oracle(name, campaigns):
    rc, txt = campaigns[name]
    assert rc == 0, txt[-3000:]
    last = txt.strip().splitlines()[-1]
    m = re.search(r"seeds (\d+)\.\.(\d+) done in \d+ s, (\d+) problems", last)
    assert m, txt[-3000:]
    want = 4 * COUNT if name == "fuzz_errors" else (max(4, COUNT // 4) if name == "fuzz_tie" else COUNT)
    assert int(m.group(2)) - int(m.group(1)) + 1 == want and int(m.group(3)) == 0, txt[-3000:]
    if name == "fuzz_errors":   # the malformed-input deviations DESIGN.md section 7 used to list are closed
        k = re.search(r"(\d+) known deviations", last)
        assert k and int(k.group(1)) == 0, last
