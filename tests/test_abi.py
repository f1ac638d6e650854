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


def test_fmi_writer_reproduces_the_reference_fixture_bytes(fmx, tmp_path):
    """export_binarized_fuzzy_matcher (src/fuzzy_matcher_binarization.cc:16-30): a reference-written file, loaded and
    saved again, comes back byte for byte — headers, class-info records, versions (this fixture is SuffixArrayIndex v0:
    no _max_tokens_in_pattern trailer, include/fuzzy/suffix_array_index.hxx:33-43), every array, ids, real tokens, and the
    unordered_maps in the order and with the bucket counts the file had."""
    src = os.path.join(golden_util.GOLDEN, "tm1.fmi")
    out = tmp_path / "again.fmi"
    fmx.Index.load(src).save(str(out))
    assert out.read_bytes() == open(src, "rb").read()


def _without_bucket_counts(data, decoded):
    data = bytearray(data)
    for off in decoded["itok_bucket_fields"]:
        data[off:off + 8] = bytes(8)
    return bytes(data)


def test_fmi_old_layout_is_rewritten_in_the_current_layout(fmx, tmp_path):
    """tm1.old.fmi (SuffixArray v0: vector<pair<unsigned, unsigned short>> with struct padding) saved again is the
    SuffixArray v1 layout of tm1.fmi: identical bytes everywhere except what belongs to the C++ library that wrote each
    file — the iteration order of the vocabulary's unordered_map and the bucket counts of the maps; decoded, the two
    are equal."""
    ref = os.path.join(golden_util.GOLDEN, "tm1.fmi")
    out = tmp_path / "upgraded.fmi"
    fmx.Index.load(os.path.join(golden_util.GOLDEN, "tm1.old.fmi")).save(str(out))
    a, b = out.read_bytes(), open(ref, "rb").read()
    da, db = fmi_reader.read_fmi(str(out)), fmi_reader.read_fmi(ref)
    assert da["map_span"] == db["map_span"] and len(a) == len(b)
    lo, hi = db["map_span"]
    assert a[:lo] == b[:lo] and _without_bucket_counts(a, da)[hi:] == _without_bucket_counts(b, db)[hi:]
    for k in da:
        assert da[k] == db[k], k


def test_fmi_writer_on_a_rebuilt_tm1(fmx, tmp_path):
    """tm1 rebuilt from its raw sentences (text front-end, same penalty-token flags as the fixture) and exported: the
    current class versions (SuffixArrayIndex v1 = the fixture's bytes + the u64 _max_tokens_in_pattern trailer), and
    outside the vocabulary map the bytes of the reference-built fixture."""
    ref = os.path.join(golden_util.GOLDEN, "tm1.fmi")
    lines = [ln for ln in open(os.path.join(golden_util.GOLDEN, "tm1"), encoding="utf-8").read().split("\n") if ln != ""]
    ix = fmx.Index.create(pt=1 | 16 | 32)  # pt_tag | pt_nbr | pt_cas (test/test.cc:102-120)
    for i, ln in enumerate(lines):
        assert ix.add_text(f"{i + 1}={ln}", ln) == i
    ix.sort(-1, 1)
    out = tmp_path / "rebuilt.fmi"
    ix.save(str(out))
    a, b = out.read_bytes(), open(ref, "rb").read()
    da, db = fmi_reader.read_fmi(str(out)), fmi_reader.read_fmi(ref)
    assert da["consumed"] == da["size"] and da["max_tokens"] == 300
    for k in da:
        if k not in ("map_span", "size", "consumed"):
            assert da[k] == db[k], k
    lo, hi = db["map_span"]
    assert da["map_span"] == (lo, hi)
    # the SuffixArrayIndex class version (one byte of the header) and the trailer are the only differences
    assert len(a) == len(b) + 8 and a[-8:] == (300).to_bytes(8, "little")
    ver = [i for i in range(lo) if a[i] != b[i]]
    assert len(ver) == 1 and a[ver[0]] == 1 and b[ver[0]] == 0, ver
    assert _without_bucket_counts(a, da)[hi:len(b)] == _without_bucket_counts(b, db)[hi:]


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


def test_many_penalty_strings(fmx):
    """The penalty-string distance table grows by one row and one column per new string (no limit of 4096 any more;
    the reference interns nothing and recomputes the byte distance per cell, include/fuzzy/edit_distance.hxx:7-35)."""
    def lev(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]
    ix = fmx.Index.create()
    strings = [""] + [f"{k:x}" + "~" * (k % 5) for k in range(1, 5000)]
    ids = [ix.intern_itok(s) for s in strings[1:]]
    assert ids == list(range(1, 5000))
    assert ix.intern_itok(strings[4321]) == 4321  # already known
    n = ix.info().n_itok
    assert n == 5000
    d = ix.array("itok_dist").reshape(n, n)
    rng = np.random.default_rng(3)
    for a, b in rng.integers(0, n, size=(300, 2)):
        assert d[a, b] == lev(strings[a].encode(), strings[b].encode()), (a, b)
    assert (d.diagonal() == 0).all() and d[0, 7] == len(strings[7])


def test_index_from_arrays_adopts_the_reference_arrays(fmx, tmp_path):
    """fmx_index_from_arrays: the SuffixArray members of a reference-built index (tm1.fmi) adopted as they
    are -> the same arrays, the same derived tables; malformed input is refused with a message."""
    src = fmx.Index.load(os.path.join(golden_util.GOLDEN, "tm1.fmi"))
    arr = {k: np.array(src.array(k)) for k in ("suf_sid", "suf_pos", "buffer", "sent_pos", "qva", "sfreq")}
    ix = fmx.Index.from_arrays(arr["suf_sid"], arr["suf_pos"], arr["buffer"], arr["sent_pos"], qva=arr["qva"], sfreq=arr["sfreq"], pt=49)
    info = ix.info()
    assert (info.n_sentences, info.n_suffixes, info.sorted, info.pt) == (8, 33, 1, 49)
    for k in arr:
        assert np.array_equal(np.array(ix.array(k)), arr[k]), k
    ix.finalize()
    src.finalize()
    assert np.array_equal(np.array(ix.array("idf")), np.array(src.array("idf")))
    # qva and sfreq recomputed when absent
    ix2 = fmx.Index.from_arrays(arr["suf_sid"], arr["suf_pos"], arr["buffer"], arr["sent_pos"])
    assert np.array_equal(np.array(ix2.array("qva")), arr["qva"]) and np.array_equal(np.array(ix2.array("sfreq"))[2:], arr["sfreq"][2:])
    # refused: two suffixes swapped (order), a suffix outside its sentence, a wrong _quickVocabAccess, a broken buffer
    bad = arr["suf_sid"].copy(), arr["suf_pos"].copy()
    bad[0][[3, 4]] = bad[0][[4, 3]]
    bad[1][[3, 4]] = bad[1][[4, 3]]
    with pytest.raises(fmx.FmxError, match="order"):
        fmx.Index.from_arrays(bad[0], bad[1], arr["buffer"], arr["sent_pos"])
    pos = arr["suf_pos"].copy()
    pos[0] = 60
    with pytest.raises(fmx.FmxError, match="outside"):
        fmx.Index.from_arrays(arr["suf_sid"], pos, arr["buffer"], arr["sent_pos"])
    q = arr["qva"].copy()
    q[3] += 1
    with pytest.raises(fmx.FmxError, match="_quickVocabAccess"):
        fmx.Index.from_arrays(arr["suf_sid"], arr["suf_pos"], arr["buffer"], arr["sent_pos"], qva=q)
    with pytest.raises(fmx.FmxError):
        fmx.Index.from_arrays(arr["suf_sid"], arr["suf_pos"], arr["buffer"][:-1], arr["sent_pos"])


def test_vocab_lookup_mt_and_attached_spellings(fmx):
    """fmx_vocab_lookup_mt == fmx_vocab_lookup token by token (several threads, unknown words -> 1, the empty word -> 0), on a
    string-built index and on an id-built one with spellings attached afterwards (fmx_index_set_forms)."""
    import ctypes as C
    ix = fmx.Index.load(os.path.join(golden_util.GOLDEN, "tm1.fmi"))
    words = ["aa", "bb", "zz", "｟num｠", ".", "", "cc", "nope"] * 3000
    ref = np.asarray(ix.vocab_lookup(words), np.uint32)
    enc = [w.encode("utf-8") for w in words]
    arr = (C.c_char_p * len(enc))(*enc)
    out = np.zeros(len(enc), np.uint32)
    for nt in (1, 5):
        out[:] = 7
        assert fmx.lib().fmx_vocab_lookup_mt(ix.h, arr, len(enc), out.ctypes.data_as(C.c_void_p), nt) == 0
        assert np.array_equal(out, ref)
    # an index built from ids gets its spellings afterwards
    b = fmx.Index.create()
    b.add_ids(np.asarray([0, 3, 5], np.uint32), np.asarray([2, 3, 4, 3, 5], np.uint32), 6)
    sp = [None, None, b"two", b"three", b"four", b"five"]
    sarr = (C.c_char_p * len(sp))(*sp)
    assert fmx.lib().fmx_index_set_forms(b.h, sarr, len(sp)) == 0
    assert list(b.vocab_lookup(["three", "five", "six", "two"])) == [3, 5, 1, 2]
    q = [b"four", b"two", b"seven"]
    qa = (C.c_char_p * 3)(*q)
    o3 = np.zeros(3, np.uint32)
    assert fmx.lib().fmx_vocab_lookup_mt(b.h, qa, 3, o3.ctypes.data_as(C.c_void_p), 2) == 0
    assert list(o3) == [4, 2, 1]


def test_respelled_ids_reach_every_lookup(fmx):
    """fmx_index_set_forms a second time with the same count but other spellings (round-3 advice: the spelling table was keyed
    on the vocabulary's size): fmx_vocab_lookup and fmx_vocab_lookup_mt both see the new words and forget the old ones."""
    import ctypes as C
    b = fmx.Index.create()
    b.add_ids(np.asarray([0, 3, 5], np.uint32), np.asarray([2, 3, 4, 3, 5], np.uint32), 6)

    def both(words):
        enc = [w.encode() for w in words]
        arr = (C.c_char_p * len(enc))(*enc)
        out = np.zeros(len(enc), np.uint32)
        assert fmx.lib().fmx_vocab_lookup_mt(b.h, arr, len(enc), out.ctypes.data_as(C.c_void_p), 2) == 0
        one = list(b.vocab_lookup(words))
        assert one == list(out)
        return one

    for sp, probe, want in (([None, None, b"two", b"three", b"four", b"five"], ["two", "four", "vier", "five"], [2, 4, 1, 5]),
                            ([None, None, b"two", b"three", b"vier", b"fuenf"], ["two", "four", "vier", "five", "fuenf"], [2, 1, 4, 1, 5]),
                            ([None, None, b"two", b"three", b"vier", b"five"], ["five", "fuenf", "vier"], [5, 1, 4])):
        sarr = (C.c_char_p * len(sp))(*sp)
        assert fmx.lib().fmx_index_set_forms(b.h, sarr, len(sp)) == 0
        assert both(probe) == want


def test_text_batch_arguments_are_checked_up_front(fmx):
    """fmx_match_batch_text: offsets that count spellings the text cannot hold (n_bytes == 0 with q_off[nq] > 0, fewer bytes than
    spellings) or that do not start at 0 are FMX_E_INVALID before anything runs (include/fmx.h)."""
    import ctypes as C
    b = fmx.Index.create()
    b.add_ids(np.asarray([0, 3], np.uint32), np.asarray([2, 3, 4], np.uint32), 5)
    b.sort()
    b.finalize()
    out = np.zeros((2, 1), fmx.MATCH_DTYPE)
    cnt = np.zeros(2, np.uint32)
    prm = fmx.Params()
    text = np.frombuffer(b"a\0b\0c\0", np.uint8).copy()
    for q_off, n_bytes in ((np.asarray([0, 2, 3], np.uint32), 0), (np.asarray([0, 2, 9], np.uint32), 6), (np.asarray([1, 2, 3], np.uint32), 6)):
        tb = fmx.TextBatch(2, q_off.ctypes.data, text.ctypes.data, n_bytes)
        rc = fmx.lib().fmx_match_batch_text(b.h, C.byref(prm), C.byref(tb), out.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), None, 1, None)
        assert rc == -1, (rc, fmx.lib().fmx_last_error())   # FMX_E_INVALID


def test_host_side_under_sanitizers_with_mutated_fmi_files(tmp_path):
    """The .fmi reader / writer, the index builder, the text front-end and the C ABI's host side compiled with
    -fsanitize=address,undefined (tests/cpp/Makefile; the device side replaced by a stub that reports "no device") and driven by
    tests/cpp/fmi_fuzz.cc: the reference's fixtures round-trip (tm1.fmi byte for byte), the text fixtures are tokenised in every
    penalty-token mode (also cut at every byte) and indexed, and 1 000 mutated copies of tm1.fmi / tm1.old.fmi — bit flips,
    overwritten bytes, blown-up 64-bit counts, truncations, extensions — go through fmx_index_load and, when they load, through
    finalize / lookups / save / load.  Any out-of-bounds access, overflow or leak ends the run with a sanitizer report."""
    import gzip
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-C", cpp, "fmi_fuzz"], stdout=subprocess.DEVNULL)
    tm2 = tmp_path / "tm2_head.txt"
    with gzip.open(os.path.join(golden_util.GOLDEN, "tm2.en.gz"), "rt", encoding="utf-8") as f:
        tm2.write_text("".join(line for _, line in zip(range(300), f)), encoding="utf-8")
    r = subprocess.run([os.path.join(cpp, "fmi_fuzz"), os.path.join(golden_util.GOLDEN, "tm1.fmi"), os.path.join(golden_util.GOLDEN, "tm1.old.fmi"),
                        "--", "1000", str(tmp_path), os.path.join(golden_util.GOLDEN, "tm1"), os.path.join(golden_util.GOLDEN, "test-tm1"), str(tm2)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "fmi_fuzz ok" in r.stdout and "1000 mutations" in r.stdout
