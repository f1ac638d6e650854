"""CPU-side checks of the C ABI: the library loads without a GPU, exports every symbol include/fmx.h
declares, the .fmi reader/writer round-trips, and the match entry points fail loudly without a device."""
import os
import re

import numpy as np
import pytest

import fmi_reader
import golden_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol(fmx):
    hdr = open(os.path.join(ROOT, "include", "fmx.h")).read()
    declared = set(re.findall(r"\b(fmx_[a-z_]+)\s*\(", hdr))
    L = fmx.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(fmx.EXPORTS), declared ^ set(fmx.EXPORTS)


@pytest.mark.parametrize("fmi", ["tm1.fmi", "tm1.old.fmi"])
def test_fmi_reader_matches_independent_decode(fmx, fmi):
    path = os.path.join(golden_util.GOLDEN, fmi)
    d = fmi_reader.read_fmi(path)
    assert d["consumed"] == d["size"]
    ix = fmx.Index.load(path)
    info = ix.info()
    assert (info.pt, info.n_sentences, info.n_suffixes, info.vocab_size, info.max_tokens_in_pattern, info.sorted) == \
           (49, 8, 33, 9, 300, 1)
    assert list(ix.array("suf_sid")) == [s for s, _ in d["suffixes"]]
    assert list(ix.array("suf_pos")) == [p for _, p in d["suffixes"]]
    assert list(ix.array("buffer")) == d["buffer"]
    assert list(ix.array("sent_pos")) == d["sent_pos"]
    assert list(ix.array("qva")) == d["qva"]
    assert list(ix.array("sfreq")) == d["sfreq"]
    assert [ix.sentence_id(i) for i in range(8)] == d["ids"]
    assert list(ix.vocab_lookup(["aa", "bb", "zz", "｟num｠", "."])) == [2, 3, 1, 6, 8]
    ix.finalize()
    assert ix.info().real_differs == 1
    # derived record of suffix 0: sentence 1, position 1
    rec = ix.array("suf_rec")[0]
    off, rem, slen = int(rec & 0xffffffff), int((rec >> 32) & 0xffff), int(rec >> 48)
    assert (off, rem, slen) == (d["sent_pos"][1] + 1, 6, 6)


def test_missing_fmi_fails_like_the_reference(fmx):
    """test/test.cc:22-26 (nofmi): importing a non-existing file is an error."""
    with pytest.raises(fmx.FmxError):
        fmx.Index.load(os.path.join(golden_util.GOLDEN, "non_existing.fmi"))


def test_corrupt_fmi_is_rejected(fmx, tmp_path):
    data = open(os.path.join(golden_util.GOLDEN, "tm1.fmi"), "rb").read()
    p = tmp_path / "trunc.fmi"
    p.write_bytes(data[:700])
    with pytest.raises(fmx.FmxError):
        fmx.Index.load(str(p))
    p.write_bytes(b"XXXX" + data[4:])
    with pytest.raises(fmx.FmxError):
        fmx.Index.load(str(p))


def test_fmi_writer_roundtrip(fmx, tmp_path):
    """export -> import keeps every array, the ids, real tokens and max_tokens_in_pattern
    (test/test.cc:102-120, :273-303)."""
    ix = fmx.Index.create(pt=49, max_tokens=7)
    ix.add_tm("1=aa bb", ["aa", "bb"], real=["L", "L"])
    ix.add_tm("2=aa Cc", ["aa", "cc"], real=["L", "C"], itoks={1: "T", 2: " "})
    ix.add_tm("3", ["bb", "aa", "bb"])
    assert ix.add_tm("4", ["a"] * 8) == -1  # longer than max_tokens: skipped like the reference
    ix.sort(-1, 1)
    out = tmp_path / "rt.fmi"
    ix.save(str(out))
    d = fmi_reader.read_fmi(str(out))
    assert d["consumed"] == d["size"] and d["max_tokens"] == 7 and d["pt"] == 49
    assert d["ids"] == ["1=aa bb", "2=aa Cc", "3"]
    assert d["real"][1] == ("L\tC", {1: "T", 2: " "})
    assert d["real"][2] == ("bb\taa\tbb", {})
    ix2 = fmx.Index.load(str(out))
    for k in ["suf_sid", "suf_pos", "buffer", "sent_pos", "qva", "sfreq"]:
        assert np.array_equal(ix.array(k), ix2.array(k)), k
    assert ix2.info().max_tokens_in_pattern == 7


def test_match_without_device_fails_loudly(fmx):
    if fmx.device_count() > 0:
        pytest.skip("a GPU is present")
    ix = fmx.Index.create()
    ix.add_tm("a", ["x", "y", "z"])
    ix.sort(-1, 1)
    ix.finalize()
    with pytest.raises(fmx.FmxError):
        ix.to_device(0)
    with pytest.raises(fmx.FmxError):
        ix.match_batch(fmx.Params(), np.asarray([0, 3], np.uint32), np.asarray([2, 3, 4], np.uint32))
    with pytest.raises(fmx.FmxError):
        ix.sort(0)  # already sorted -> ok; new index needed for the GPU sort failure
    ix3 = fmx.Index.create()
    ix3.add_tm("a", ["x"])
    with pytest.raises(fmx.FmxError):
        ix3.sort(0)
