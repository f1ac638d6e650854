"""The index the reference CLI writes BY DEFAULT (-p nbr,tag,cas, cli/src/FuzzyMatch-cli.cc:328): built from cased TEXT, every
sentence's real tokens (case features, digits — src/fuzzy_match.cc:123-192) differ from its normalised tokens, so every candidate
pays the real-token term of _edit_distance (src/edit_distance.cc:41-71).  The GPU path on such an index — text batches with a real-token
stream, id batches with interned real ids, raw lines tokenised on the device — against the oracle, BIT FOR BIT.  The oracle's index is
built independently of the product's tokenizer: vocabulary ids from the generator's ids (first-appearance order), real tokens from
the generator's rules, its own sort."""
import os
import tempfile

import numpy as np
import pytest

import oracle_lib
import workloads
import synth  # noqa: E402  (tools/ is put on the path by workloads)

pytestmark = pytest.mark.gpu

PT_DEFAULT = 1 | 16 | 32  # tag, nbr, cas


def build_default_index(fmx, gpu_device, c_off, c_tok, via_fmi=True):
    lines = bytes(synth.pack_lines_cased(c_off, c_tok)).decode("ascii").split("\n")[:-1]
    assert len(lines) == len(c_off) - 1
    ix = fmx.Index.create(pt=PT_DEFAULT)
    ix.add_text_bulk([str(i) for i in range(len(lines))], lines)
    ix.sort(gpu_device)
    ix.finalize()
    if via_fmi:
        path = os.path.join(tempfile.gettempdir(), f"fmx_default_{os.getpid()}.fmi")
        try:
            ix.save(path)
            ix.close()
            ix = fmx.Index.load(path)
            ix.finalize()
        finally:
            if os.path.exists(path):
                os.unlink(path)
    ix.to_device(gpu_device)
    return ix


def build_default_oracle(c_off, c_tok):
    m, vs = synth.cased_vocab_map(c_tok)
    o = oracle_lib.OracleIndex()
    o.add_ids_bulk(c_off, m[c_tok], vs)
    packed, boff = oracle_lib.tab_joined(synth.pack_cased_stream(c_off, c_tok, 1), c_off)
    o.sort(oracle_lib.effective_cpus())
    o.set_real_bulk(packed, boff)
    return o, m, vs


@pytest.fixture(scope="module")
def default_index(fmx, gpu_device):
    c_off, c_tok, q_off, q_tok, _ = workloads.synthetic(60000, 3000, 25)
    ix = build_default_index(fmx, gpu_device, c_off, c_tok)
    o, m, vs = build_default_oracle(c_off, c_tok)
    return ix, o, m, vs, c_off, c_tok, q_off, q_tok


def test_text_built_index_equals_the_generators_ids(default_index):
    """fmx-cli -c's ingestion (tokenizer -> VocabIndexer -> buffer) against ids derived from the generator alone."""
    ix, o, m, vs, c_off, c_tok, q_off, q_tok = default_index
    info = ix.info()
    assert info.real_differs and info.vocab_size == vs and info.n_sentences == len(c_off) - 1
    buf = ix.array("buffer")
    sp = ix.array("sent_pos").astype(np.int64)
    lens = np.diff(c_off).astype(np.int64)
    assert np.array_equal(buf[sp], lens)
    want = m[c_tok]
    # sentence i's tokens sit at sp[i] + 1 ..: gather them all
    idx = np.repeat(sp + 1 - c_off[:-1].astype(np.int64), lens) + np.arange(len(c_tok), dtype=np.int64)
    assert np.array_equal(buf[idx], want)
    a = o.arrays()
    assert np.array_equal(a["suf_sid"], ix.array("suf_sid")) and np.array_equal(a["suf_pos"], ix.array("suf_pos"))


SETS = {"C3": dict(fuzzy=0.7, nmatch=5), "C4": dict(fuzzy=0.5, nmatch=5, min_subseq_ratio=0.5, idf_penalty=1.0),
        "C5": dict(fuzzy=0.7, nmatch=5, contrast=0.3, contrast_buffer=32),
        "costs": dict(fuzzy=0.5, nmatch=4, costs=(0.5, 1, 2)), "wide": dict(fuzzy=0.4, nmatch=0, min_subseq_length=2, min_subseq_ratio=0),
        "no_perfect": dict(fuzzy=0.6, nmatch=3, no_perfect=True)}


@pytest.mark.parametrize("name", list(SETS))
def test_default_index_text_ids_and_raw_lines_bit_exact(fmx, default_index, name):
    ix, o, m, vs, c_off, c_tok, q_off, q_tok = default_index
    kw = SETS[name]
    cap = 16
    q_real = oracle_lib.tab_joined(synth.pack_cased_stream(q_off, q_tok, 1), q_off)
    oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, m[q_tok], cap, nthreads=oracle_lib.effective_cpus(), real=q_real)
    # the real-token term is active: some scores are neither 1 nor what ids alone give
    o_ids, c_ids = o.match_batch(oracle_lib.Params(**kw), q_off, m[q_tok], cap, nthreads=oracle_lib.effective_cpus(),
                                 real=oracle_lib.tab_joined(synth.pack_cased_stream(q_off, q_tok, 0), q_off))
    assert oo.tobytes() != o_ids.tobytes()
    prm = fmx.Params(**kw)
    # (i) text batch: normalised spellings + real-token spellings, both looked up on the device
    norm = synth.pack_cased_stream(q_off, q_tok, 0)
    real = synth.pack_cased_stream(q_off, q_tok, 1)
    go, gc, gt = ix.match_batch_text(prm, q_off, norm, cap=cap, real_text=real)
    workloads.compare(oo, oc, go, gc, name + "/text")
    # (ii) id batch with interned real ids
    toks = bytes(real).split(b"\0")[:-1]
    table = {t: ix.lookup_real(t.decode()) for t in set(toks)}
    r_ids = np.asarray([table[t] for t in toks], np.uint32)
    ids = ix.vocab_lookup([t.decode() for t in bytes(norm).split(b"\0")[:-1]])
    assert np.array_equal(ids, m[q_tok])
    go, gc, gt = ix.match_batch(prm, q_off, ids, q_real=r_ids, cap=cap)
    workloads.compare(oo, oc, go, gc, name + "/ids")
    # (iii) raw lines, tokenised / case-folded / looked up on the device: match(const std::string&, ...)
    lines = synth.pack_lines_cased(q_off, q_tok)
    go, gc, gt = ix.match_batch_raw_lines(prm, lines, len(q_off) - 1, cap=cap)
    workloads.compare(oo, oc, go, gc, name + "/raw")


def test_hybrid_real_token_path_bit_exact(fmx, default_index):
    """Option "real_hybrid" (off by default: slower at C3, profiles/r06_real_hybrid_ab.md): under unit costs the bit-parallel kernel answers
    the pairs in which no equal word meets a different real token and leaves the others to the float kernel — same records."""
    ix, o, m, vs, c_off, c_tok, q_off, q_tok = default_index
    ix.set_option("real_hybrid", 1)
    try:
        for name in ("C3", "C5", "no_perfect", "wide"):
            kw = SETS[name]
            cap = 16
            q_real = oracle_lib.tab_joined(synth.pack_cased_stream(q_off, q_tok, 1), q_off)
            oo, oc = o.match_batch(oracle_lib.Params(**kw), q_off, m[q_tok], cap, nthreads=oracle_lib.effective_cpus(), real=q_real)
            go, gc, gt = ix.match_batch_text(fmx.Params(**kw), q_off, synth.pack_cased_stream(q_off, q_tok, 0), cap=cap,
                                             real_text=synth.pack_cased_stream(q_off, q_tok, 1))
            workloads.compare(oo, oc, go, gc, name + "/hybrid")
    finally:
        ix.set_option("real_hybrid", 0)
