"""tools/fmx-cli: the reference CLI's flags, output format (`score<TAB>id` pairs, %.9g like
boost::lexical_cast<float>) and stderr lines (STEP / NMATCH): raw text like the reference (the library's restricted
text front-end), or pre-tokenised input with --tokenised; actions index / match / subseq."""
import os
import subprocess

import numpy as np
import pytest

import golden_util
import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "fmx-cli")


def _build(fmx):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)


def test_cli_help_and_loud_failure_without_gpu(fmx):
    _build(fmx)
    r = subprocess.run([CLI, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--fuzzy" in r.stderr and "--contrast-buffer" in r.stderr
    r = subprocess.run([CLI, "-a", "match"], capture_output=True, text=True)
    assert r.returncode == 3 and "index file or corpus needs to be provided" in r.stderr  # cli:445-447
    r = subprocess.run([CLI, "-i", "/nonexistent.fmi", "-a", "match"], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR" in r.stderr
    if fmx.device_count() == 0:
        r = subprocess.run([CLI, "-i", os.path.join(golden_util.GOLDEN, "tm1.fmi"), "-a", "match"], input="aa bb cc\n",
                           capture_output=True, text=True)
        assert r.returncode == 1 and "ERROR" in r.stderr and r.stdout == ""  # no CPU fallback


@pytest.mark.gpu
def test_cli_index_then_match(fmx, gpu_device, tmp_path):
    """-c builds a .fmi (GPU suffix sort + Boost-archive writer), -i/-a match prints the reference's format;
    checked against the oracle on the same tokenised corpus."""
    _build(fmx)
    rng = np.random.default_rng(11)
    words = ["w%d" % i for i in range(40)]
    sents = [[words[int(x)] for x in rng.integers(0, 40, size=int(rng.integers(3, 12)))] for _ in range(300)]
    corpus = tmp_path / "corpus.tok"
    corpus.write_text("".join("id%d\t%s\n" % (i, " ".join(s)) for i, s in enumerate(sents)))
    r = subprocess.run([CLI, "--tokenised", "-c", str(corpus), "-a", "index"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(str(corpus) + ".fmi") and "STEP\tImporting TM" in r.stderr and "STEP\tSorting Index" in r.stderr
    queries = []
    for i in range(60):
        s = list(sents[int(rng.integers(0, 300))])
        if i % 3 == 0 and len(s) > 3:
            s[int(rng.integers(0, len(s)))] = "zzz"
        if i % 7 == 0:
            s = []
        queries.append(s)
    inp = "".join(" ".join(q) + "\n" for q in queries)
    r = subprocess.run([CLI, "--tokenised", "-i", str(corpus) + ".fmi", "-a", "match", "-f", "0.5", "-n", "3", "--ml", "2", "--mr", "0",
                        "--batch", "17"], input=inp, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")[:-1]
    assert len(lines) == len(queries)
    o = oracle_lib.OracleIndex()
    for s in sents:
        o.add_tm(s)
    o.sort(2)
    prm = oracle_lib.Params(fuzzy=0.5, nmatch=3, min_subseq_length=2, min_subseq_ratio=0)
    n_nonempty = 0
    for q, line in zip(queries, lines):
        res, _ = o.match_tokens(prm, q) if q else (np.zeros(0, oracle_lib.MATCH_DTYPE), 0)
        exp = "\t".join("%.9g\tid%d" % (m["score"], m["s_id"]) for m in res)
        assert line == exp, (q, line, exp)
        n_nonempty += 1 if len(res) else 0
    assert "NMATCH\t%d\t/\t%d" % (n_nonempty, len(queries)) in r.stderr and "STEP\tMatching" in r.stderr


@pytest.mark.gpu
def test_cli_raw_text_tm2_like_the_reference_ci(fmx, gpu_device, tmp_path):
    """.github/workflows/cmake.yml:41 + test/test.cc:217-221: `-c tm2.en.gz,tm2.fr.gz --add-target` builds
    tm2.en.gz.fmi from the gzipped bitext; `-i ... -a match` on the raw test-tm2 sentences prints what the reference's
    test expects (ids '<line>=<target>', scores within its 0.009 tolerance)."""
    import shutil
    import test_text_frontend as tf
    _build(fmx)
    for f in ("tm2.en.gz", "tm2.fr.gz"):
        shutil.copy(os.path.join(golden_util.GOLDEN, f), tmp_path / f)
    r = subprocess.run([CLI, "-c", f"{tmp_path}/tm2.en.gz,{tmp_path}/tm2.fr.gz", "--add-target"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(f"{tmp_path}/tm2.en.gz.fmi"), r.stderr
    cases = tf.read_cases("test-tm2")
    inp = "".join(c["text"] + "\n" for c in cases)
    r = subprocess.run([CLI, "-i", f"{tmp_path}/tm2.en.gz.fmi", "-a", "match", "-f", "0.5", "-n", "2", "-P"], input=inp,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")[:-1]
    assert len(lines) == len(cases)
    for c, line in zip(cases, lines):
        f = line.split("\t") if line else []
        tf.check_case(c, [float(f[i]) for i in range(0, len(f), 2)], [f[i + 1] for i in range(0, len(f), 2)])
    assert "NMATCH\t12\t/\t100" in r.stderr
    # -a subseq (cli/src/FuzzyMatch-cli.cc:234-248): score, length, id, and the text of the sub-n-gram
    r = subprocess.run([CLI, "-i", f"{tmp_path}/tm2.en.gz.fmi", "-a", "subseq", "-n", "3"], input="I declare adjourned the session of the European Parliament .\nzzzz qqqq\n",
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split("\n")
    f = out[0].split("\t")
    assert len(f) == 4 and int(f[1]) >= 6 and f[3] in "I declare adjourned the session of the European Parliament ." and out[1] == ""
    assert "NMATCH\t1\t/\t2" in r.stderr


@pytest.mark.gpu
def test_cli_blocks_and_formatting_tasks_keep_order_and_bytes(fmx, gpu_device, tmp_path):
    """The streaming pipeline on an input of many blocks: 40 000 lines (some empty, one 20 KB long, the last without a line feed,
    sentence ids of 3 to 40 characters so that both forms of the id table are used) through block sizes that cut them differently
    — one line per block upwards; blocks beyond 8192 lines are formatted as several tasks by whichever workers are free — give
    byte-identical output, in input order, equal to the records of the library's own id batch."""
    _build(fmx)
    rng = np.random.default_rng(23)
    words = ["v%d" % i for i in range(300)]
    sents = [[words[int(x)] for x in rng.integers(0, 300, size=int(rng.integers(2, 14)))] for _ in range(3000)]
    ids = ["s%d" % i if i % 5 else "sentence-with-a-long-identifier-%010d" % i for i in range(len(sents))]
    corpus = tmp_path / "corpus.tok"
    corpus.write_text("".join("%s\t%s\n" % (ids[i], " ".join(s)) for i, s in enumerate(sents)))
    r = subprocess.run([CLI, "--tokenised", "-c", str(corpus), "-a", "index"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    queries = []
    for i in range(40_000):
        s = list(sents[int(rng.integers(0, len(sents)))])
        if i % 4 == 0:
            s[int(rng.integers(0, len(s)))] = "unseen"
        if i % 97 == 0:
            s = []
        queries.append(s)
    queries[1234] = ["unseen"] * 5000   # a 30 KB line
    inp = "\n".join(" ".join(q) for q in queries)   # no line feed after the last line
    outs = []
    for batch in ("1000", "8192", "65536", "3"):
        lines_in = inp if batch != "3" else "\n".join(inp.split("\n")[:500])
        r = subprocess.run([CLI, "--tokenised", "-i", str(corpus) + ".fmi", "-a", "match", "-f", "0.6", "-n", "4", "--ml", "2", "--mr", "0",
                            "--batch", batch], input=lines_in, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2]
    assert outs[0].count("\n") == len(queries)
    assert outs[3] == "".join(l + "\n" for l in outs[0].split("\n")[:500])
    # the same records through the library (ids in, records out), printed like the CLI prints them
    ix = fmx.Index.load(str(corpus) + ".fmi")
    ix.finalize()
    ix.to_device(gpu_device)
    voc = {w: int(v) for w, v in zip(words + ["unseen"], ix.vocab_lookup(words + ["unseen"]))}
    off = np.zeros(len(queries) + 1, np.uint32)
    off[1:] = np.cumsum([len(q) for q in queries])
    tok = np.array([voc[t] for q in queries for t in q], np.uint32)
    go, gc, _ = ix.match_batch(fmx.Params(fuzzy=0.6, nmatch=4, min_subseq_length=2, min_subseq_ratio=0), off, tok, cap=4)
    exp = "".join("\t".join("%.9g\t%s" % (go[q, k]["score"], ids[int(go[q, k]["s_id"])]) for k in range(int(gc[q]))) + "\n" for q in range(len(queries)))
    assert outs[0] == exp
