"""Loaders for tests/golden (fixtures copied from / transcribed from the reference's own tests)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def known_answers():
    return json.load(open(os.path.join(GOLDEN, "known_answers.json"), encoding="utf-8"))["cases"]


def tm1_cases():
    """test/data/test-tm1 rows -> dicts with the hand-derived tokenisation attached."""
    tok = json.load(open(os.path.join(GOLDEN, "test-tm1.tokenised.json"), encoding="utf-8"))["queries"]
    cases = []
    for line in open(os.path.join(GOLDEN, "test-tm1"), encoding="utf-8"):
        line = line.rstrip("\n")
        if not line or line.startswith("#"):
            continue
        f = line.split("\t")
        exp = [(float(f[5 + 2 * i]), f[6 + 2 * i]) for i in range((len(f) - 5) // 2)]
        q = tok[f[1]]
        cases.append({"id": f[0], "text": f[1], "fuzzy": float(f[2]), "no_perfect": f[3] == "true",
                      "nmatch": int(f[4]), "expect": exp, "norm": q["norm"], "real": q["real"],
                      "itoks": {int(k): v for k, v in q["itoks"].items()}})
    return cases


def check_tm1_case(case, scores, ids):
    """The reference's own acceptance rule, test/test.cc:57-99."""
    exp = case["expect"]
    assert len(ids) == len(exp), (case["id"], ids, exp)
    for (es, eid), s, i in zip(exp, scores, ids):
        assert i == eid, (case["id"], ids, exp)
        assert (es == 1) == (s == 1), (case["id"], s, es)
        assert abs(es - s) <= 0.009, (case["id"], s, es)
