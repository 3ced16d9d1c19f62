"""Host-side file formats, CPU only: NPY/NPZ writer and reader against numpy, WAV ingest
(lib.rs:509-529, 172-183), and the lane-level extraction math against a direct DFT."""
import os
import subprocess
import wave

import numpy as np
import pytest

import streamz_amd as sz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "streamz_amd", "csrc")


@pytest.fixture(scope="module")
def npy_tool(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bin") / "npy_tool")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + CSRC, "-o", out,
                           os.path.join(ROOT, "tests", "cpu", "npy_tool.cpp"), os.path.join(CSRC, "npy.cpp"), "-lz"])
    return out


def test_npz_written_by_the_library_is_read_by_numpy(npy_tool, tmp_path):
    p = str(tmp_path / "m.npz")
    subprocess.check_call([npy_tool, "write", p])
    z = np.load(p)
    assert sorted(z.files) == ["empty", "sample_rate", "speaker_0_files", "w1"]
    np.testing.assert_array_equal(z["w1"], np.array([[1.5, -2.0, 3.25], [4.0, 5.0, -6.5]], np.float32))
    assert z["w1"].dtype == np.float32 and z["sample_rate"].dtype == np.int64 and z["sample_rate"][0] == 44100
    assert bytes(z["speaker_0_files"]).decode() == "a/b.wav\nc.wav"
    assert z["empty"].shape == (0,)
    import zipfile
    with zipfile.ZipFile(p) as zf:
        assert all(i.compress_type == zipfile.ZIP_STORED for i in zf.infolist())  # NpzWriter::new is uncompressed
        assert zf.testzip() is None  # CRCs


@pytest.mark.parametrize("compressed", [False, True])
def test_npz_written_by_numpy_is_read_by_the_library(npy_tool, tmp_path, compressed):
    p = str(tmp_path / "n.npz")
    arrs = {"w1": np.arange(12, dtype=np.float32).reshape(3, 4), "bits": np.array([16], np.int64),
            "speaker_1_files": np.frombuffer(b"x.wav", np.uint8)}
    (np.savez_compressed if compressed else np.savez)(p, **arrs)
    out = subprocess.check_output([npy_tool, "dump", p]).decode().splitlines()
    got = {l.split()[0]: l.split()[1:] for l in out}
    assert got["w1"][:4] == ["<f4", "2", "3", "4"]
    assert got["bits"][:3] == ["<i8", "1", "1"]
    assert got["speaker_1_files"][:3] == ["|u1", "1", "5"]
    for k, a in arrs.items():
        s = 0
        for b in a.tobytes():
            s = (s * 131 + b) % (1 << 64)
        assert int(got[k][-1]) == s


def _write_wav(path, data, channels=1, rate=44100, width=2):
    with wave.open(path, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(width)
        w.setframerate(rate)
        w.writeframes(data.tobytes())


def test_wav_mono_and_stereo_downmix(tmp_path):
    rng = np.random.default_rng(0)
    mono = rng.integers(-32768, 32767, 5000).astype(np.int16)
    p = str(tmp_path / "m.wav")
    _write_wav(p, mono)
    np.testing.assert_array_equal(sz.wav_read(p), mono)
    st = rng.integers(-32768, 32767, (3001, 2)).astype(np.int16)
    p2 = str(tmp_path / "S.WAV")  # the extension is matched case-insensitively (lib.rs:510-514)
    _write_wav(p2, st, channels=2)
    want = np.trunc(st.astype(np.int32).sum(axis=1) / 2).astype(np.int16)  # i32 sum / channels, truncating
    np.testing.assert_array_equal(sz.wav_read(p2), want)


def test_wav_rejections(tmp_path):
    p = str(tmp_path / "r.wav")
    _write_wav(p, np.zeros(100, np.int16), rate=16000)
    with pytest.raises(sz.StreamzError) as e:
        sz.wav_read(p)
    assert e.value.code == -5 and "44100" in str(e.value)
    with pytest.raises(sz.StreamzError) as e:
        sz.wav_read(str(tmp_path / "missing.wav"))
    assert e.value.code == -4
    with pytest.raises(sz.StreamzError) as e:
        sz.wav_read(str(tmp_path / "clip.mp3"))
    assert e.value.code == -5
    p8 = str(tmp_path / "b8.wav")
    _write_wav(p8, np.zeros(100, np.uint8), width=1)
    with pytest.raises(sz.StreamzError):
        sz.wav_read(p8)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_lane_emulation_of_the_extraction_phases(tmp_path, kind):
    """The per-lane phases of the extraction kernel, run lane by lane on the CPU, reproduce a
    direct double-precision DFT / mel / log / DCT of one window."""
    exe = str(tmp_path / "emu")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + CSRC, "-o", exe,
                           os.path.join(ROOT, "tests", "cpu", "emulate_extract.cpp"),
                           os.path.join(CSRC, "extract_tables.cpp")])
    for seed in (1, 2):
        e_pow, e_log, e_mfcc = [float(v) for v in subprocess.check_output([exe, str(seed), str(kind)]).split()]
        assert e_pow < 1e-6      # power spectrum error relative to its largest bin
        assert e_log < 2e-5      # log mel energies, absolute
        assert e_mfcc < 1e-4     # raw MFCCs (range about -450 .. +60), absolute
