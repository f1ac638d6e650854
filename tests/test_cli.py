"""tools/fmx-cli: the reference CLI's flags, output format (`score<TAB>id` pairs, %.9g like
boost::lexical_cast<float>) and stderr lines (STEP / NMATCH), on pre-tokenised input."""
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
    r = subprocess.run([CLI, "-c", str(corpus), "-a", "index"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(str(corpus) + ".fmi") and "STEP\tIndexing" in r.stderr
    queries = []
    for i in range(60):
        s = list(sents[int(rng.integers(0, 300))])
        if i % 3 == 0 and len(s) > 3:
            s[int(rng.integers(0, len(s)))] = "zzz"
        if i % 7 == 0:
            s = []
        queries.append(s)
    inp = "".join(" ".join(q) + "\n" for q in queries)
    r = subprocess.run([CLI, "-i", str(corpus) + ".fmi", "-a", "match", "-f", "0.5", "-n", "3", "--ml", "2", "--mr", "0",
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
