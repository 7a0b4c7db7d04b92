"""Seeded synthetic inputs restating benches/kzg_benches.rs:7-44: blobs of pseudo-random bytes with byte 0 of every
32-byte field element forced to 0 (so every element is < 2^248 < r).  The reference uses an unseeded thread_rng;
here the stream is splitmix64 seeded with 0x48440000 + blob_index, little-endian, so fixtures are reproducible."""
import numpy as np

BYTES_PER_BLOB = 131072


def _splitmix64_stream(seed, n_words):
    out = np.empty(n_words, dtype=np.uint64)
    x = np.uint64(seed)
    with np.errstate(over="ignore"):
        for i in range(n_words):
            x = x + np.uint64(0x9E3779B97F4A7C15)
            z = x
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            out[i] = z ^ (z >> np.uint64(31))
    return out


def _splitmix64_vec(seed, n_words):
    """Vectorised splitmix64: state_i = seed + (i+1)*gamma."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n_words + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_blob(index):
    words = _splitmix64_vec(0x48440000 + index, BYTES_PER_BLOB // 8)
    b = bytearray(words.astype("<u8").tobytes())
    b[0::32] = bytes(len(b) // 32)   # arr[i*32] = 0   (benches/kzg_benches.rs:19-21)
    return bytes(b)


def random_field_element(index):
    words = _splitmix64_vec(0x48450000 + index, 4)
    b = bytearray(words.astype("<u8").tobytes())
    b[0] = 0
    return bytes(b)


def splitmix64_bytes(seed, n_bytes):
    """n_bytes of the same little-endian splitmix64 stream (minimal-preset fixtures, tests/golden/make_minimal_fixtures.py)."""
    return _splitmix64_vec(seed, (n_bytes + 7) // 8).astype("<u8").tobytes()[:n_bytes]
