"""Drop-in proof: the REFERENCE's own command line tool, built from its unmodified caller
sources (ocl_aho_grep.c, ocl_worker.c, file_traverse.c, utils.c -- compiled where they lie by
`make -C oracle dropin`, binary under oracle/_ref/) and linked against OUR libacmatch.so, run on
the reference's own fixture.  Its -v lines and STATS block must agree with the oracle.
"""
import os
import re
import subprocess

import numpy as np
import pytest

import fixtures
import orc

pytestmark = pytest.mark.gpu

BIN = os.path.join(orc.ORACLE_DIR, "_ref", "ocl_aho_grep_acm")
LINE = re.compile(r"^Pattern (-?\d+) \('(.*)'\) found in file '(.*)' at offset (\d+) \[relative: (-?\d+)\]$")


def run_cli(args):
    p = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=120, errors="replace")
    assert p.returncode == 0, p.stderr[-2000:]
    hits = [LINE.match(l) for l in p.stdout.splitlines()]
    stats = dict(re.findall(r"^([A-Za-z ()]+):\s+([\d.]+)$", p.stdout, flags=re.M))
    return [h.groups() for h in hits if h], stats, p.stdout


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ocl_aho_grep_acm not built")
@pytest.mark.parametrize("workers", [1, 2])
def test_reference_cli_on_reference_fixture(gpu, workers):
    text_path = os.path.join(orc.DATA, "ref_tests", "input.txt")
    pat_path = os.path.join(orc.DATA, "ref_tests", "patterns.txt")
    hits, stats, out = run_cli(["-f", text_path, "-p", pat_path, "-B", "2048", "-D", "0", "-G", "8",
                                "-L", "1024", "-w", str(workers), "-v"])
    o = fixtures.oracle_for("tests")
    text = np.fromfile(text_path, dtype=np.uint8)
    pos, pat, _ = o.scan(text)
    assert int(stats["Matches"]) == pos.size == 24
    assert int(stats["Matches reported"]) == 24
    assert int(stats["Automaton states"]) == 198
    assert int(stats["Processed bytes"]) == text.size
    assert int(stats["Kernel launches"]) == 1
    assert len(hits) == 24
    for (iid, name, fname, off, rel), p, k in zip(hits, pos.tolist(), pat.tolist()):
        b, want_iid = o.pattern(k)
        assert int(iid) == want_iid and name.encode() == b and fname == text_path
        assert int(off) == p + 1                      # offset of the last byte + 1 (databuf.c:771)
        assert int(rel) == p + 1 - (p // 2048) * 2048  # relative to the chunk (ocl_aho_grep.c:285)


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/ocl_aho_grep_acm not built")
def test_reference_cli_hex_signatures_multiple_rounds(gpu, tmp_path):
    """-x -m 12 on a 1 MiB planted corpus with a 256 KiB buffer: four rounds, last_state carried."""
    name = "clamav2000_m12"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 20, "seed": 31, "n_plant": 300}, pats)
    f = tmp_path / "corpus.bin"
    f.write_bytes(text.tobytes())
    sigs = orc.clamav_file(2000, str(tmp_path))
    hits, stats, out = run_cli(["-f", str(f), "-p", sigs, "-x", "-m", "12", "-B", "4096", "-D", "0",
                                "-G", "64", "-L", "1024", "-w", "1", "-v"])
    pos, pat, _ = o.scan(text)
    assert int(stats["Matches"]) == pos.size
    assert int(stats["Kernel launches"]) == 4
    assert int(stats["Automaton states"]) == o.num_states
    assert int(stats["Matches reported"]) == pos.size
    # the tool prints the raw signature bytes with %s, so lines whose signature contains a NUL or
    # a newline do not parse; the ones that do must be the expected records, in order
    got = [(int(h[0]), int(h[3])) for h in hits]
    want = [(o.pattern(k)[1], (p % (64 * 4096)) + 1) for p, k in zip(pos.tolist(), pat.tolist())]
    it = iter(want)
    assert all(g in it for g in got), "parsed -v lines are not a subsequence of the expected records"
    assert len(got) > len(want) // 2
