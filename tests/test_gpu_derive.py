"""The accelerator arrays of DESIGN.md §3 are built ON THE DEVICE at fmx_index_to_device (fmx_derive.hip.hh).  Here they
are copied back out of HBM and compared with HostIndex::derive_heavy, the host loops the CPU emulation of the kernels runs
on (tests/test_emul_vs_oracle.py pins those against the oracle): per-suffix arrays byte for byte, hash tables as sets
(slot placement depends on the order concurrent inserts land in)."""
import os

import numpy as np
import pytest

import golden_util
import workloads

pytestmark = pytest.mark.gpu


def _entries(tab):
    live = tab[tab["lo"] != 0xffffffff]
    return np.sort(live, order=["lo", "cnt", "w1", "w2"])


def _bucket_fields(tab):
    used = (tab["dir_b"] >> 16) & 3
    over = (tab["dir_b"] >> 18) & 1
    tags = np.stack([tab["dir_a"] & 0xffff, tab["dir_a"] >> 16, tab["dir_b"] & 0xffff], axis=1)
    lcp = np.stack([(tab["dir_b"] >> (20 + 3 * k)) & 7 for k in range(3)], axis=1)
    return used, over, tags, lcp


def _bucket_entries(tab):
    """Live entries of the 5-gram buckets as sorted rows (w1..w4, lo, cnt, t5..t8, lcp, tag)."""
    if len(tab) == 0:
        return np.zeros((0, 12), np.uint32)
    used, _, tags, lcp = _bucket_fields(tab)
    rows = []
    for k in range(3):
        sel = used > k
        e = tab["e"][sel, k]
        rows.append(np.column_stack([e["w"], e["lo"], e["cnt"], e["t"], lcp[sel, k], tags[sel, k]]).astype(np.uint32))
    rows = np.concatenate(rows) if rows else np.zeros((0, 12), np.uint32)
    return rows[np.lexsort(rows.T[::-1])]


def _check_buckets_searchable(ix, tab, label):
    """Every entry sits in its home bucket or in a later one with every bucket from home up to it flagged `over` (what a lookup
    follows); lcp = tokens 5..8 shared by the group's first and last suffix; unused tag slots are zero."""
    nb = len(tab)
    key, rec, buf = ix.array("suf_key")["t"], ix.array("suf_rec"), ix.array("buffer")
    off = (rec & np.uint64(0xffffffff)).astype(np.int64)
    used, over, tags, lcps = _bucket_fields(tab)
    M32 = np.uint64(0xffffffff)

    def mul(a, c):
        return (a * np.uint64(c)) & M32

    def h5(w0, w):
        h = mul(w0.astype(np.uint64), 0x9E3779B1)
        for (sh, c, col) in ((15, 0x85EBCA6B, 0), (13, 0xC2B2AE35, 1), (16, 0x27D4EB2F, 2), (15, 0x165667B1, 3)):
            h = ((h ^ (h >> np.uint64(sh))) + mul(w[:, col].astype(np.uint64), c)) & M32
        h ^= h >> np.uint64(16); h = mul(h, 0x7FEB352D); h ^= h >> np.uint64(15)
        return h

    def tag5(w0, w):
        h = mul(w[:, 3].astype(np.uint64), 0x9E3779B1)
        for (sh, c, col) in ((13, 0xC2B2AE35, 2), (15, 0x27D4EB2F, 1), (16, 0x165667B1, 0)):
            h = ((h ^ (h >> np.uint64(sh))) + mul(w[:, col].astype(np.uint64), c)) & M32
        h = ((h ^ (h >> np.uint64(13))) + mul(w0.astype(np.uint64), 0x85EBCA6B)) & M32
        h ^= h >> np.uint64(15); h = mul(h, 0x846CA68B)
        return h >> np.uint64(16)

    for k in range(3):
        assert np.all(tags[used <= k, k] == 0) and np.all(lcps[used <= k, k] == 0), label
        at = np.nonzero(used > k)[0]
        if not len(at):
            continue
        e = tab["e"][at, k]
        lo, cnt = e["lo"].astype(np.int64), e["cnt"].astype(np.int64)
        assert np.array_equal(key[lo][:, :4], e["w"]) and np.array_equal(key[lo][:, 4:], e["t"]), label
        first, last = key[lo][:, 4:], key[lo + cnt - 1][:, 4:]
        eq = first == last
        lcp = np.where(eq.all(axis=1), 4, np.argmin(eq, axis=1))
        assert np.array_equal(lcp, lcps[at, k]), label
        w0 = buf[off[lo]]
        assert np.array_equal(tag5(w0, e["w"]), tags[at, k].astype(np.uint64)), label
        home = ((h5(w0, e["w"]) * np.uint64(nb)) >> np.uint64(32)).astype(np.int64)
        dist = (at - home) % nb
        assert dist.max() < 64, (label, int(dist.max()))
        for d in range(1, int(dist.max()) + 1):       # displaced entries: every bucket they passed is flagged
            far = dist >= d
            assert np.all(over[(home[far] + d - 1) % nb] != 0), label


def _to_device_layout(ix, off):
    """Token offsets of the host's packed _sentence_buffer -> the device layout (DeviceIndex::upload keeps a sentence that fits
    one 128-byte line inside one line): same sentence, same position in it."""
    host_pos, dev_pos = ix.array("sent_pos").astype(np.int64), ix.device_array("sent_pos").astype(np.int64)
    sid = np.searchsorted(host_pos, off.astype(np.int64), side="right") - 1
    return (dev_pos[sid] + (off.astype(np.int64) - host_pos[sid])).astype(np.uint32)


def _check_device_layout(ix, label):
    """The device copy of the sentence buffer: every sentence's words unchanged, positions ascending, and no sentence of at
    most 32 words crossing a 128-byte line."""
    hb, hp = ix.array("buffer"), ix.array("sent_pos").astype(np.int64)
    db, dp = ix.device_array("buffer"), ix.device_array("sent_pos").astype(np.int64)
    assert len(hp) == len(dp), label
    size = hb[hp].astype(np.int64) + 2
    assert np.all(dp[1:] >= dp[:-1] + size[:-1]) and dp[-1] + size[-1] <= len(db), label
    fits = size <= 32
    assert np.all((dp[fits] // 32) == ((dp[fits] + size[fits] - 1) // 32)), label
    idx_h = np.repeat(hp, size) + (np.arange(size.sum()) - np.repeat(np.cumsum(size) - size, size))
    idx_d = np.repeat(dp, size) + (np.arange(size.sum()) - np.repeat(np.cumsum(size) - size, size))
    assert np.array_equal(hb[idx_h], db[idx_d]), label


def _compare(ix, label):
    _check_device_layout(ix, label)
    for name in ["suf_rec", "suf_sig", "suf_key", "suf_b"]:
        host, dev = ix.array(name).copy(), ix.device_array(name)
        # the host twins hold offsets into the packed buffer, the device arrays into the device layout
        if name == "suf_rec":
            host = (host & np.uint64(0xffffffff00000000)) | _to_device_layout(ix, (host & np.uint64(0xffffffff)).astype(np.uint32)).astype(np.uint64)
        if name == "suf_b":
            host["off"] = _to_device_layout(ix, host["off"])
        assert host.shape == dev.shape and np.array_equal(host.view(np.uint8), dev.view(np.uint8)), (label, name)
    for name in ["gram", "gram7"]:
        host, dev = ix.array(name), ix.device_array(name)
        assert len(host) == len(dev), (label, name, len(host), len(dev))   # same capacity rule
        assert np.array_equal(_entries(host), _entries(dev)), (label, name)
        if len(dev):
            # open addressing stays searchable: every live entry is reachable from its home slot without crossing an empty one
            assert (dev["lo"] != 0xffffffff).sum() * 2 <= len(dev), (label, name)
    # the 5-gram buckets: same number of buckets, the same entries (with their lcp) wherever concurrent inserts put them
    host, dev = ix.array("gram5"), ix.device_array("gram5")
    assert len(host) == len(dev), (label, "gram5", len(host), len(dev))
    he, de = _bucket_entries(host), _bucket_entries(dev)
    assert he.shape == de.shape and np.array_equal(he, de), (label, "gram5")
    if len(dev):
        _check_buckets_searchable(ix, dev, label)
    return len(ix.device_array("gram")), len(de)


def test_device_built_arrays_equal_the_host_build_synthetic(fmx, gpu_device):
    c_off, c_tok, _, _, vs = workloads.synthetic(30_000, 10, 25)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    n3, n5 = _compare(ix, "synthetic 30k")
    assert n3 > 0


@pytest.mark.parametrize("seed,vocab", [(1, 3), (2, 6), (3, 40)])
def test_device_built_arrays_equal_the_host_build_dense(fmx, gpu_device, seed, vocab):
    """Tiny vocabularies: 5-gram groups of hundreds of suffixes, so the 7-gram level is populated and buckets overflow; one- and
    two-token sentences (suffixes that end a run); duplicates."""
    rng = np.random.default_rng(seed)
    c_off, c_tok = workloads.dense_corpus(rng, 6000, vocab, min_len=1, max_len=14)
    ix = fmx.Index.create(max_tokens=20)
    ix.add_ids(c_off, c_tok, vocab + 2)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    n3, n5 = _compare(ix, f"dense vocab {vocab}")
    assert n3 > 0 and (vocab > 6 or n5 > 0), (n3, n5)


@pytest.mark.parametrize("fmi", ["tm1.fmi", "tm1.old.fmi"])
def test_device_built_arrays_of_the_reference_fixture(fmx, gpu_device, fmi):
    ix = fmx.Index.load(os.path.join(golden_util.GOLDEN, fmi))
    ix.finalize()
    ix.to_device(gpu_device)
    _compare(ix, fmi)


def test_single_sentence_and_single_token(fmx, gpu_device):
    for toks in ([5], [5, 5], [2, 3, 4], [7] * 9):
        ix = fmx.Index.create()
        ix.add_ids(np.asarray([0, len(toks)], np.uint32), np.asarray(toks, np.uint32), 10)
        ix.sort(gpu_device)
        ix.finalize()
        ix.to_device(gpu_device)
        _compare(ix, str(toks))
