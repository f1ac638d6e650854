"""fmx_match_batch_raw_lines (round 5): FuzzyMatch::match(const std::string&, ...) — src/fuzzy_match.cc:393-412, the reference
CLI's only input (cli/src/FuzzyMatch-cli.cc:219-233) — for a block of raw lines.  Lines of ASCII letters, digits, punctuation and
blanks are tokenised on the DEVICE by a byte-class machine (fmx_vocab.hip.hh: k_raw_emit); every other line goes through the
host front-end (fmx_text.cc).

CPU part: the machine's rules, restated here line for line, against the host tokenizer (which the reference's text fixtures pin,
tests/test_text_frontend.py) on random ASCII lines under every penalty-token set the device path serves.  GPU part: raw lines ==
the id batch the host front-end makes of the same lines (pinned to the oracle and to the reference's printed scores there):
test-tm1 / test-tm1-nonbr-nocas against tm1, test-tm2 and 2 000 Europarl sentences (accents, digits, punctuation, placeholders:
both paths in one call) against tm2, random ASCII."""
import gzip
import os

import numpy as np
import pytest

import golden_util
from test_text_frontend import PT, read_cases, tm1_lines, tm2_lines, _gpu_index, _gpu_batch

G = golden_util.GOLDEN


def device_rules(pt, line):
    """What k_raw_emit does with an ASCII line: (norm, real) token lists, or None when the line is flagged for the host."""
    norm, real = [], []
    cls = []
    for ch in line.encode("utf-8"):
        c = chr(ch)
        if ("a" <= c <= "z") or ("A" <= c <= "Z"):
            cls.append(1)
        elif "0" <= c <= "9":
            cls.append(2)
        elif c == " " or 9 <= ch <= 13:
            cls.append(0)
        elif 32 < ch < 128:
            cls.append(3)
        else:
            return None
    b = line
    i, n = 0, len(b)
    while i < n:
        if cls[i] == 0:
            i += 1
            continue
        j = i + 1
        if cls[i] != 3:
            while j < n and cls[j] == cls[i]:
                j += 1
        tok = b[i:j]
        word = cls[i] == 1 and (pt & PT["cas"])
        if cls[i] == 2 and (pt & PT["nbr"]):
            norm.append("｟num｠")
            real.append(tok)
        elif word:
            norm.append(tok.lower())
            n_up = sum(1 for c in tok if c.isupper())
            first = tok[0].isupper()
            real.append("L" if n_up == 0 else ("C" if first and n_up == 1 else ("U" if n_up == len(tok) and len(tok) >= 2 else "M")))
        else:
            norm.append(tok)
            real.append(tok)
        i = j
    return norm, real


def random_ascii_lines(rng, n):
    words = ["aa", "bb", "cc", "dd", "ee", "Dd", "AA", "aB", "ABc", "x", "Y", "the", "The", "THE", "commission", "Commission"]
    out = []
    for _ in range(n):
        parts = []
        for _ in range(int(rng.integers(0, 14))):
            r = rng.random()
            if r < 0.55:
                parts.append(words[int(rng.integers(0, len(words)))])
            elif r < 0.7:
                parts.append(str(int(rng.integers(0, 3000))))
            elif r < 0.85:
                punct = ".,;:!?()[]'\"-/%&$#@*+=<>~^_`{|}\\"
                parts.append(punct[int(rng.integers(0, len(punct)))])
            else:
                parts.append("a1b" if r < 0.9 else "12.5%")
            if rng.random() < 0.8:
                parts.append(" " * int(rng.integers(1, 3)) if rng.random() < 0.9 else "\t")
        out.append("".join(parts))
    return out


@pytest.mark.parametrize("pt", [0, PT["nbr"], PT["cas"], PT["nbr"] | PT["cas"], PT["tag"] | PT["nbr"] | PT["cas"]])
def test_device_rules_equal_the_host_tokenizer_on_ascii(fmx, pt):
    rng = np.random.default_rng(11)
    lines = random_ascii_lines(rng, 1500) + ["aa bb cc dd.", "aa bb 14 dd", "Mr. O'Neil paid 4.40 EUR", "", "   ", "a", "A", "AB", "Ab", "aB", "9", "x9y", "...", "\x7f"]
    n = 0
    for ln in lines:
        got = device_rules(pt, ln)
        assert got is not None
        t = fmx.tokenize(pt, ln)
        assert (t["norm"], t["real"]) == (got[0], got[1]), (pt, ln, t, got)
        assert t["itoks"] == {}
        n += len(got[0])
    assert n > 5000
    for ln in ["café", "a\x01b", "｟it＃1｠ aa", "naïve 12"]:
        assert device_rules(pt, ln) is None


def _host_batch(fmx, ix, pt, lines, prm, cap):
    toks, q_off, q_tok, q_real, q_itok = _gpu_batch(fmx, ix, pt, [{"text": ln} for ln in lines])
    if ix.info().real_differs:
        return ix.match_batch(prm, q_off, q_tok, q_real=q_real, q_itok=q_itok, cap=cap)
    return ix.match_batch(prm, q_off, q_tok, cap=cap)


def _raw(lines):
    return ("\n".join(lines) + "\n").encode("utf-8")


@pytest.mark.gpu
@pytest.mark.parametrize("pt,fixture", [(PT["tag"] | PT["nbr"] | PT["cas"], "test-tm1"), (0, "test-tm1-nonbr-nocas"),
                                        (PT["pct"] | PT["nbr"] | PT["cas"], "test-tm1-pct"), (PT["tag"] | PT["jnr"] | PT["cas"] | PT["nbr"], "test-tm1-jnr")])
def test_tm1_raw_lines(fmx, gpu_device, pt, fixture):
    """(pct / jnr: penalty tokens the device machine does not implement — every line takes the host path inside the call)"""
    lines = tm1_lines()
    ix, ids = _gpu_index(fmx, gpu_device, pt, lines, [f"{i + 1}={ln}" for i, ln in enumerate(lines)])
    cases = read_cases(fixture)
    texts = [c["text"] for c in cases]
    for kw in (dict(fuzzy=0.4, nmatch=8, min_subseq_length=2, min_subseq_ratio=0), dict(fuzzy=0.7, nmatch=3, no_perfect=True, min_subseq_length=2, min_subseq_ratio=0)):
        want = _host_batch(fmx, ix, pt, texts, fmx.Params(**kw), 8)
        got = ix.match_batch_raw_lines(fmx.Params(**kw), _raw(texts), len(texts), cap=8)
        assert np.array_equal(want[1], got[1]) and int(want[1].sum()) > 0, (fixture, kw)
        for q in range(len(texts)):
            assert want[0][q, :want[1][q]].tobytes() == got[0][q, :got[1][q]].tobytes(), (fixture, kw, texts[q])
    # a line count that disagrees with the text's line feeds is refused on the all-host path too (pct / jnr), in both directions
    for wrong in (len(texts) + 1, len(texts) - 1):
        if wrong == 0:   # (no lines: nothing is looked at)
            continue
        with pytest.raises(fmx.FmxError):
            ix.match_batch_raw_lines(fmx.Params(fuzzy=0.5, nmatch=2), _raw(texts), wrong, cap=2)


@pytest.mark.gpu
def test_tm2_raw_lines(fmx, gpu_device):
    en, fr = tm2_lines()
    pt = PT["tag"] | PT["nbr"] | PT["cas"]
    ix, ids = _gpu_index(fmx, gpu_device, pt, en, [f"{i + 1}={fr[i]}" for i in range(len(en))])
    texts = [c["text"] for c in read_cases("test-tm2")] + [en[i] for i in range(0, 20000, 10)] + \
            ["Résumé 2024 ｟it＃1｠ of the Commission 12 May", "", "   ", "The Commission", "naïve café 12", "a\x01b cc",
             "the Commi\x00ssion proposes"]   # (a NUL inside a word does not end it on the host: such a line must not be cut on the device)
    n_ascii = sum(1 for t in texts if device_rules(pt, t) is not None)
    assert 100 < n_ascii < len(texts)   # both paths inside one call
    n_rec = 0
    for kw in [dict(fuzzy=0.5, nmatch=2, no_perfect=True), dict(fuzzy=0.6, nmatch=5), dict(fuzzy=0.5, nmatch=3, idf_penalty=0.5)]:
        want = _host_batch(fmx, ix, pt, texts, fmx.Params(**kw), 5)
        got = ix.match_batch_raw_lines(fmx.Params(**kw), _raw(texts), len(texts), cap=5)
        assert np.array_equal(want[1], got[1]), kw
        for q in range(len(texts)):
            assert want[0][q, :want[1][q]].tobytes() == got[0][q, :got[1][q]].tobytes(), (kw, q, texts[q])
        n_rec += int(want[1].sum())
    assert n_rec > 3000
    # the reference's printed scores, through the raw-line path alone
    from test_text_frontend import check_case
    cases = read_cases("test-tm2")
    got = ix.match_batch_raw_lines(fmx.Params(fuzzy=0.5, nmatch=2, no_perfect=True, min_subseq_length=3, min_subseq_ratio=0.3), _raw([c["text"] for c in cases]), len(cases), cap=2)
    for qi, case in enumerate(cases):
        rec = got[0][qi, :got[1][qi]]
        check_case(case, [float(s) for s in rec["score"]], [ids[s] for s in rec["s_id"]])


@pytest.mark.gpu
def test_raw_lines_on_an_id_index_equal_line_batches(fmx, gpu_device):
    """An index built from ids with spellings attached (bench.py's): no penalty tokens, real tokens = normalised tokens.  Lower-case
    ASCII words: raw lines == pre-tokenised lines (fmx_match_batch_lines); several passes; a wrong line count is refused."""
    import workloads
    c_off, c_tok, q_off, q_tok, vs = workloads.synthetic(30_000, 9_000, 20)
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    assert vs < 26 ** 4
    words = [b"", b""] + [b"w" + bytes([97 + (k % 26), 97 + ((k // 26) % 26), 97 + ((k // 676) % 26), 97 + ((k // 17576) % 26)]) for k in range(2, vs)]
    offs = np.full(len(words), 0xffffffffffffffff, np.uint64)
    pos = 0
    for k in range(2, len(words)):
        offs[k] = pos
        pos += len(words[k]) + 1
    ix.set_forms_packed(np.frombuffer(b"".join(w + b"\0" for w in words[2:]), dtype=np.uint8).copy(), offs)
    ix.to_device(gpu_device)
    nq = len(q_off) - 1
    text = b"".join(b" ".join(words[t] for t in q_tok[q_off[q]:q_off[q + 1]]) + b"\n" for q in range(nq))
    prm = fmx.Params(fuzzy=0.7, nmatch=5)
    want = ix.match_batch(prm, q_off, q_tok, cap=5)
    for sub in (1 << 20, 2500):
        ix.set_option("sub_queries", sub)
        got = ix.match_batch_raw_lines(prm, text, nq, cap=5)
        lines = ix.match_batch_lines(prm, text, nq, cap=5)
        assert np.array_equal(want[1], got[1]) and np.array_equal(want[1], lines[1])
        assert want[0].tobytes() == got[0].tobytes()
    ix.set_option("sub_queries", 1 << 20)
    with pytest.raises(fmx.FmxError):
        ix.match_batch_raw_lines(prm, text, nq + 1, cap=5)
