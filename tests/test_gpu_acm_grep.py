"""acm_grep, the native CLI: flags, -v line format, STATS block, rounds with a
carried state, directory input, several workers, text mode -- against the oracle, and against the
reference's own CLI binary (built from its sources against libacmatch.so) where that exists."""
import os
import re
import subprocess

import numpy as np
import pytest

import fixtures
import orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gpu_pattern_matching_amd", "acm_grep")
REF_CLI = os.path.join(orc.ORACLE_DIR, "_ref", "ocl_aho_grep_acm")
LINE = re.compile(r"^Pattern (-?\d+) \('(.*)'\) found in file '(.*)' at offset (\d+) \[relative: (-?\d+)\]$")


def run(binary, args):
    p = subprocess.run([binary] + args, capture_output=True, text=True, timeout=180, errors="replace")
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    hits = [m.groups() for m in (LINE.match(l) for l in p.stdout.splitlines()) if m]
    stats = dict(re.findall(r"^([A-Za-z ()]+):\s+([\d.]+)$", p.stdout, flags=re.M))
    return hits, stats, p.stdout


def test_fixture_matches_oracle_and_reference_cli(gpu):
    text_path = os.path.join(orc.DATA, "ref_tests", "input.txt")
    pat_path = os.path.join(orc.DATA, "ref_tests", "patterns.txt")
    args = ["-f", text_path, "-p", pat_path, "-B", "2048", "-D", "0", "-G", "8", "-L", "1024", "-w", "1", "-v"]
    hits, stats, out = run(CLI, args)
    o = fixtures.oracle_for("tests")
    pos, pat, _ = o.scan(np.fromfile(text_path, dtype=np.uint8))
    assert int(stats["Matches"]) == int(stats["Matches reported"]) == pos.size == 24
    assert int(stats["Automaton states"]) == 198
    assert int(stats["Processed bytes"]) == 9479 and int(stats["Kernel launches"]) == 1
    want = [(str(o.pattern(k)[1]), o.pattern(k)[0].decode(), text_path, str(p + 1), str(p + 1 - p // 2048 * 2048))
            for p, k in zip(pos.tolist(), pat.tolist())]
    assert hits == want
    if os.path.exists(REF_CLI):      # the reference's own tool, same flags: identical match lines
        ref_hits, ref_stats, _ = run(REF_CLI, args)
        assert ref_hits == hits
        for key in ("Matches", "Matches reported", "Automaton states", "Processed bytes", "Kernel launches"):
            assert ref_stats[key] == stats[key]


def test_directory_workers_and_rounds(gpu, tmp_path):
    """A directory of files, two workers, buffers smaller than the files: every file's matches are
    found, whichever worker and round they fall into (state carried per worker)."""
    name = "clamav2000_m12"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    d = tmp_path / "inputs"
    d.mkdir()
    total, per_file = 0, {}
    for i in range(5):
        t = fixtures.text_for({"kind": "clamav", "n": 300000 + 1111 * i, "seed": 40 + i, "n_plant": 120}, pats)
        (d / ("f%d.bin" % i)).write_bytes(t.tobytes())
        per_file[str(d / ("f%d.bin" % i))] = t
        total += t.size
    sigs = orc.clamav_file(2000, str(tmp_path))
    hits, stats, out = run(CLI, ["-f", str(d), "-p", sigs, "-x", "-m", "12", "-B", "4096", "-D", "0", "-G", "32",
                                 "-L", "1024", "-w", "2", "-v"])
    assert int(stats["Processed files"]) == 5 and int(stats["Processed bytes"]) == total
    # a worker scans its files as ONE stream (the reference carries last_state across files too,
    # databuf.c:622): count per worker stream = serial scan of the concatenation in readdir order
    order = [os.path.join(str(d), e) for e in os.listdir(str(d))]
    # acm_grep walks the directory with readdir(), the same order os.listdir reports on this fs
    expect = 0
    for w in range(2):
        stream = np.concatenate([per_file[f] for f in order[w::2]])
        expect += o.scan(stream)[0].size
    assert int(stats["Matches"]) == expect
    assert int(stats["Matches reported"]) == expect
    assert int(stats["Kernel launches"]) >= total // (32 * 4096)


def test_device_list(gpu, tmp_path):
    """-D takes a list of devices: worker i runs on entry i mod its length, every entry holds its
    own copy of the automaton, the files are dealt to the workers as with one device
    (ocl_aho_grep.c:87).  On a one-GPU box the list names device 0 twice: two copies, four workers,
    same counts as the oracle; a device that does not exist is the reference's error."""
    name = "clamav2000_m12"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    d = tmp_path / "inputs"
    d.mkdir()
    per_file = {}
    for i in range(7):
        t = fixtures.text_for({"kind": "clamav", "n": 200000 + 777 * i, "seed": 70 + i, "n_plant": 90}, pats)
        (d / ("f%d.bin" % i)).write_bytes(t.tobytes())
        per_file[str(d / ("f%d.bin" % i))] = t
    sigs = orc.clamav_file(2000, str(tmp_path))
    base = ["-f", str(d), "-p", sigs, "-x", "-m", "12", "-B", "4096", "-G", "32", "-L", "1024", "-w", "4"]
    hits, stats, out = run(CLI, base + ["-D", "0,0"])
    order = [os.path.join(str(d), e) for e in os.listdir(str(d))]
    expect = sum(o.scan(np.concatenate([per_file[f] for f in order[w::4]]))[0].size for w in range(4))
    assert int(stats["Matches"]) == expect and int(stats["Processed files"]) == 7
    p = subprocess.run([CLI] + base + ["-D", "0,99"], capture_output=True, text=True, timeout=180, errors="replace")
    assert p.returncode != 0 and "invalid dev pos" in p.stderr


def test_text_mode(gpu, tmp_path):
    o = fixtures.oracle_for("sentiment")
    text = fixtures.text_for({"kind": "words", "n": 200000, "seed": 9}, None).tobytes()
    lines, pos, rng = [], 0, np.random.default_rng(9)
    while pos < len(text):
        ln = int(rng.integers(10, 100))
        lines.append(text[pos:pos + ln].replace(b"\n", b" ") + b"\n")
        pos += ln
    f = tmp_path / "lines.txt"
    f.write_bytes(b"".join(lines))
    pat_path = os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt")
    hits, stats, out = run(CLI, ["-f", str(f), "-p", pat_path, "-t", "-B", "256", "-D", "0", "-G", "1024",
                                 "-L", "1024", "-w", "1", "-R", "64"])
    stream = np.frombuffer(b"".join(lines), dtype=np.uint8)
    assert int(stats["Matches"]) == o.scan(stream)[0].size
    assert int(stats["Processed lines"]) == len(lines)
    assert int(stats["Processed bytes"]) == stream.size


def test_usage_errors(gpu):
    p = subprocess.run([CLI, "-f", "x"], capture_output=True, text=True)
    assert p.returncode != 0 and "No pattern file" in p.stdout


def test_all_patterns_flag(gpu, tmp_path):
    """-A (extension): one line per pattern ending at an offset, in match-list
    order; without it the output is the reference's (head of the list only)."""
    words = [b"abc", b"xabc", b"bc", b"c", b"abc"]          # suffixes of each other + a duplicate
    pat_path = tmp_path / "pats.txt"
    pat_path.write_bytes(b"\n".join(words) + b"\n")
    rng = np.random.default_rng(17)
    text = np.frombuffer(b"abcx ", dtype=np.uint8)[rng.integers(0, 5, size=50000)]
    text_path = tmp_path / "in.txt"
    text.tofile(str(text_path))
    o = orc.Oracle()
    o.load(str(pat_path))
    o.compile()
    base = ["-f", str(text_path), "-p", str(pat_path), "-B", "4096", "-D", "0", "-G", "16", "-L", "1024", "-R", "4096", "-w", "1", "-v"]
    head_pos, head_pat, _ = o.scan(text)
    all_pos, all_pat, _ = o.scan_all(text)
    assert all_pos.size > head_pos.size
    for flags, pos, pat in (([], head_pos, head_pat), (["-A"], all_pos, all_pat)):
        hits, stats, _ = run(CLI, base + flags)
        assert int(stats["Matches"]) == int(stats["Matches reported"]) == pos.size
        got = [(int(h[3]) - 1, h[1]) for h in hits]
        want = [(int(p), o.pattern(int(k))[0].decode()) for p, k in zip(pos, pat)]
        assert got == want
    o.close()


def test_all_patterns_on_nested_patterns(gpu, tmp_path):
    """-A where patterns nest (aaa, aaaa, aaaaa over a run of a's): more records than text bytes.
    Up to four per byte the expanded planes hold them all; beyond that acm_grep stops with an
    error instead of reading past its planes."""
    text = np.full(20000, ord("a"), dtype=np.uint8)
    text[::997] = ord("b")
    text_path = tmp_path / "in.txt"
    text.tofile(str(text_path))
    for npat, fits in ((3, True), (16, False)):      # (-G is rounded up to 16 chunks: the planes hold 4 x 64 Ki records)
        words = [b"a" * (3 + k) for k in range(npat)]
        pat_path = tmp_path / ("pats%d.txt" % npat)
        pat_path.write_bytes(b"\n".join(words) + b"\n")
        args = ["-f", str(text_path), "-p", str(pat_path), "-B", "4096", "-D", "0", "-G", "8", "-L", "1024",
                "-R", "16384", "-w", "1", "-v", "-A"]
        if fits:
            o = orc.Oracle()
            o.load(str(pat_path))
            o.compile()
            all_pos, all_pat, _ = o.scan_all(text)
            assert all_pos.size > 2 * text.size
            hits, stats, _ = run(CLI, args)
            assert int(stats["Matches"]) == all_pos.size
            got = [(int(h[3]) - 1, h[1]) for h in hits]
            assert got == [(int(p), o.pattern(int(k))[0].decode()) for p, k in zip(all_pos, all_pat)]
            o.close()
        else:
            p = subprocess.run([CLI] + args, capture_output=True, text=True, timeout=180, errors="replace")
            assert p.returncode == 1 and "-A produced" in p.stderr


def test_follow_mode_sees_appended_data(gpu, tmp_path):
    """-F (ocl_aho_grep.c:96-99): the worker keeps polling its files; data appended later is scanned
    from the state the earlier data left (a signature cut by the append boundary is found), and
    SIGINT ends the run with the STATS block."""
    import signal
    import time
    name = "clamav2000_m12"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    path, hx, ml = fixtures.set_source(name)
    whole = fixtures.text_for({"kind": "clamav", "n": 180000, "seed": 91, "n_plant": 90}, pats).copy()
    cut = 100000
    sig = np.frombuffer(pats[7], dtype=np.uint8)
    whole[cut - 5:cut - 5 + sig.size] = sig              # straddles the append boundary
    f = tmp_path / "growing.bin"
    whole[:cut].tofile(str(f))
    args = [CLI, "-f", str(f), "-p", path, "-x", "-m", "12", "-B", "4096", "-D", "0", "-G", "64", "-L", "1024",
            "-w", "1", "-v", "-F"]
    p = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        time.sleep(6)
        with open(str(f), "ab") as fh:
            fh.write(whole[cut:].tobytes())
        time.sleep(6)
        p.send_signal(signal.SIGINT)
        out, err = p.communicate(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    out = out.decode(errors="replace")
    assert p.returncode == 0, out[-1500:] + err.decode(errors="replace")[-1500:]
    pos, pat, _ = o.scan(whole)
    stats = dict(re.findall(r"^([A-Za-z ()]+):\s+([\d.]+)$", out, flags=re.M))
    assert int(stats["Matches"]) == pos.size
    assert int(stats["Processed bytes"]) == whole.size
    assert int(stats["Kernel launches"]) >= 2            # the appended part came in a later round
    assert int(stats["Matches reported"]) == pos.size
    # the -v lines whose (binary) pattern text survives line splitting come in scan order
    iids = [int(m.group(1)) for m in (LINE.match(l) for l in out.splitlines()) if m]
    want = iter(o.pattern(int(k))[1] for k in pat)
    assert len(iids) > pos.size // 2 and all(any(i == w for w in want) for i in iids)
