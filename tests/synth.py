"""Seeded synthetic corpora for the BASELINE.json configs (test/bench helper).

The reference's 32MB.N.bin files are unseeded /dev/urandom dumps that are not
in the repository (test.sh:1-9, .MISSING_LARGE_BLOBS), and uniform random
bytes never match a ClamAV signature -- so the corpus is uniform
bytes from a seeded generator plus planted signatures, a share of them forced
across tile / chain / shard boundaries.
"""
import numpy as np


def clamav_corpus(n_bytes, seed, patterns, n_plant=4096, boundaries=(64, 4096, 1 << 22)):
    """uint8[n_bytes]: uniform random + n_plant planted patterns.

    patterns: list of bytes.  About a quarter of the plants are centred on a
    multiple of one of `boundaries` so that matches straddle chain, wave-tile
    and shard borders.
    """
    rng = np.random.default_rng(seed)
    text = rng.integers(0, 256, size=n_bytes, dtype=np.uint8)
    if not patterns or n_plant <= 0:
        return text
    pick = rng.integers(0, len(patterns), size=n_plant)
    where = rng.integers(0, max(n_bytes - 1, 1), size=n_plant)
    for k in range(n_plant):
        p = patterns[int(pick[k])]
        if len(p) == 0 or len(p) > n_bytes:
            continue
        pos = int(where[k])
        if k % 4 == 0:
            b = boundaries[(k // 4) % len(boundaries)]
            if n_bytes > b:
                edge = (pos // b) * b
                if edge == 0:
                    edge = b
                pos = edge - int(rng.integers(1, len(p) + 1))
        pos = max(0, min(pos, n_bytes - len(p)))
        text[pos:pos + len(p)] = np.frombuffer(p, dtype=np.uint8)
    return text


def word_corpus(n_bytes, seed, hot_words, neutral_words=None, hot_share=0.5):
    """Space-separated words; hot_words are drawn with probability hot_share.

    Stand-in for the missing imdb/twitter sentiment datasets.
    """
    rng = np.random.default_rng(seed)
    if neutral_words is None:
        neutral_words = [b"the", b"of", b"and", b"to", b"in", b"is", b"that", b"for", b"it", b"as",
                         b"with", b"was", b"on", b"be", b"by", b"at", b"this", b"have", b"from",
                         b"or", b"one", b"had", b"not", b"but", b"what", b"all", b"were", b"when",
                         b"we", b"there", b"can", b"an", b"your", b"which", b"their", b"said"]
    hot = [w if isinstance(w, bytes) else w.encode() for w in hot_words]
    neu = [w if isinstance(w, bytes) else w.encode() for w in neutral_words]
    out = bytearray()
    # draw in blocks to keep this fast
    while len(out) < n_bytes:
        k = 65536
        use_hot = rng.random(k) < hot_share
        hi = rng.integers(0, len(hot), size=k)
        ni = rng.integers(0, len(neu), size=k)
        words = [hot[hi[i]] if use_hot[i] else neu[ni[i]] for i in range(k)]
        out += b" ".join(words) + b" "
    return np.frombuffer(bytes(out[:n_bytes]), dtype=np.uint8).copy()


def load_hex_patterns(path, limit=None, max_len=-1):
    pats = []
    with open(path, "rb") as f:
        for i, line in enumerate(f):
            if limit is not None and i >= limit:
                break
            h = line.strip()
            if max_len != -1:
                h = h[: 2 * max_len]
            pats.append(bytes.fromhex(h.decode()))
    return pats
