"""Shared helpers for the test-suite: fixture pattern sets, texts, oracle cache."""
import functools
import os
import tempfile

import numpy as np

import orc
import synth

_TMP = tempfile.mkdtemp(prefix="acm_fixtures_")


def set_source(name):
    """(path, hex, max_len) of a golden.json set name."""
    base, _, m = name.partition("_m")
    max_len = int(m) if m else -1
    if base.startswith("clamav"):
        return orc.clamav_file(int(base[len("clamav"):]), _TMP), True, max_len
    path, hx = orc.pattern_set(base)
    return path, hx, max_len


@functools.lru_cache(maxsize=2)
def oracle_for(name):
    path, hx, max_len = set_source(name)
    o = orc.Oracle()
    o.load(path, hx, max_len)
    o.compile()
    return o


def patterns_of(name):
    o = oracle_for(name)
    return [o.pattern(i)[0] for i in range(o.num_patterns)]


def text_for(spec, pats):
    kind = spec["kind"]
    if kind == "file":
        return np.fromfile(os.path.join(orc.DATA, spec["path"]), dtype=np.uint8)
    if kind == "clamav":
        return synth.clamav_corpus(spec["n"], spec["seed"], pats, spec["n_plant"])
    if kind == "words":
        words = open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()
        return synth.word_corpus(spec["n"], spec["seed"], words)
    if kind == "repeat":
        p = pats[spec["pattern"]]
        return np.frombuffer((p * (spec["n"] // len(p) + 1))[: spec["n"]], dtype=np.uint8).copy()
    if kind == "zeros":
        return np.zeros(spec["n"], dtype=np.uint8)
    raise KeyError(kind)


def spec_id(spec):
    return "-".join(str(spec[k]) for k in ("kind", "n", "seed", "path") if k in spec).replace("/", "_")
