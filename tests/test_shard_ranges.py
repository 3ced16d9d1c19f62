"""Host-only entry points a multi-rank run plans with: szh_shard_ranges (how a file list is cut over
the ranks) and szh_wav_info (a file's length from its header).  No GPU needed."""
import wave

import numpy as np
import pytest

import streamz_amd as sz


def brute_force(counts, world):
    """the specification restated: boundary r = the index >= the previous boundary whose prefix sum is
    nearest to r / world of the total, the smallest such index"""
    counts = [int(c) for c in counts]
    n, total = len(counts), sum(counts)
    prefix = [0]
    for c in counts:
        prefix.append(prefix[-1] + c)
    out = [0]
    for r in range(1, world):
        if total == 0:
            out.append(n * r // world)
            continue
        best = min(range(out[-1], n + 1), key=lambda i: (abs(prefix[i] * world - total * r), i))
        out.append(best)
    return out + [n]


@pytest.mark.parametrize("counts,world,want", [
    ([10, 10, 10, 10], 2, [0, 2, 4]),
    ([10, 10, 10, 10], 4, [0, 1, 2, 3, 4]),
    ([6614] * 1000, 8, [0, 125, 250, 375, 500, 625, 750, 875, 1000]),
    ([100, 1, 1, 1, 1], 2, [0, 1, 5]),             # one long file: the cut falls behind it
    ([1, 1, 1, 1, 100], 2, [0, 4, 5]),
    ([5, 5], 4, [0, 0, 1, 1, 2]),                   # more ranks than files: two ranks stay empty
    ([], 3, [0, 0, 0, 0]),
    ([0, 0, 0, 0], 2, [0, 2, 4]),                   # nothing to balance on: by file count
    ([0, 0, 7, 0, 0, 7, 0], 2, [0, 3, 7]),          # empty files (too short for a window) travel with a neighbour
    ([3], 1, [0, 1]),
])
def test_known_cuts(counts, world, want):
    assert sz.shard_ranges(counts, world) == want == brute_force(counts, world)


def test_random_lists_against_the_restated_rule():
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = int(rng.integers(0, 40))
        counts = rng.integers(0, 2000, n) * (rng.random(n) > 0.2)
        world = int(rng.integers(1, 12))
        got = sz.shard_ranges(counts, world)
        assert got == brute_force(counts, world)
        assert got[0] == 0 and got[-1] == n and all(a <= b for a, b in zip(got, got[1:]))
        if n and counts.sum():   # no rank holds more than its share plus one file
            loads = [int(counts[a:b].sum()) for a, b in zip(got, got[1:])]
            assert max(loads) <= counts.sum() / world + counts.max()


def test_counts_beyond_32_bits():
    big = [3_000_000_000_000, 1, 3_000_000_000_000, 2]
    assert sz.shard_ranges(big, 2) == brute_force(big, 2) == [0, 2, 4]


def test_bad_arguments():
    with pytest.raises(sz.StreamzError):
        sz.shard_ranges([1, 2], 0)


def test_wav_info_reads_the_header_only(tmp_path):
    rng = np.random.default_rng(0)
    for ch, frames in ((1, 12345), (2, 1001), (3, 700)):
        p = str(tmp_path / f"c{ch}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(ch)
            w.setsampwidth(2)
            w.setframerate(44100)
            w.writeframes(rng.integers(-3000, 3000, frames * ch).astype(np.int16).tobytes())
        assert sz.wav_info(p) == (frames, 44100, ch)
        assert len(sz.wav_read(p)) == frames          # what the decoder then delivers
    # a data chunk that claims more than the file holds is cut at the end of the file, as the reader does
    raw = bytearray(open(str(tmp_path / "c1.wav"), "rb").read())
    raw[40:44] = (10 ** 9).to_bytes(4, "little")
    open(str(tmp_path / "long.wav"), "wb").write(bytes(raw))
    assert sz.wav_info(str(tmp_path / "long.wav"))[0] == 12345 == len(sz.wav_read(str(tmp_path / "long.wav")))
    with pytest.raises(sz.StreamzError):
        sz.wav_info(str(tmp_path / "absent.wav"))
    open(str(tmp_path / "junk.wav"), "wb").write(b"not a wav file at all")
    with pytest.raises(sz.StreamzError):
        sz.wav_info(str(tmp_path / "junk.wav"))
    with pytest.raises(sz.StreamzError) as ei:
        sz.wav_info(str(tmp_path / "clip.mp3"))
    assert ei.value.code == -5
