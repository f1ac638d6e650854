"""The oracle's bulk real-token entries (fmo_index_set_real_bulk, fmo_match_batch_real) against its per-sentence / per-pattern ones
(the ones the reference's fixtures pin), on cased synthetic text: same records, and the real-token term is active."""
import numpy as np

import oracle_lib
import workloads
import synth  # noqa: E402  (tools/ is put on the path by workloads)


def test_bulk_real_entries_equal_per_call_entries():
    c_off, c_tok, q_off, q_tok, _ = workloads.synthetic(3000, 200, 20)
    m, vs = synth.cased_vocab_map(c_tok)
    c_real = bytes(synth.pack_cased_stream(c_off, c_tok, 1)).split(b"\0")[:-1]
    q_real = bytes(synth.pack_cased_stream(q_off, q_tok, 1)).split(b"\0")[:-1]
    a = oracle_lib.OracleIndex()
    a.add_ids_bulk(c_off, m[c_tok], vs)
    a.sort(2)
    a.set_real_bulk(*oracle_lib.tab_joined(synth.pack_cased_stream(c_off, c_tok, 1), c_off))
    b = oracle_lib.OracleIndex()
    b.add_ids_bulk(c_off, m[c_tok], vs)
    b.sort(2)
    for s in range(len(c_off) - 1):
        b.set_real(s, [t.decode() for t in c_real[c_off[s]:c_off[s + 1]]])
    n_real_cost = 0
    for kw in (dict(fuzzy=0.6, nmatch=5), dict(fuzzy=0.5, nmatch=4, idf_penalty=0.7, costs=(0.5, 1, 2))):
        prm = oracle_lib.Params(**kw)
        oo, oc = a.match_batch(prm, q_off, m[q_tok], 8, nthreads=2, real=oracle_lib.tab_joined(synth.pack_cased_stream(q_off, q_tok, 1), q_off))
        for q in range(len(q_off) - 1):
            ref, _ = b.match_ids(prm, m[q_tok[q_off[q]:q_off[q + 1]]], real=[t.decode() for t in q_real[q_off[q]:q_off[q + 1]]], cap=8)
            assert len(ref) == oc[q] and ref.tobytes() == oo[q, :oc[q]].tobytes(), (kw, q)
            ids_only, _ = b.match_ids(prm, m[q_tok[q_off[q]:q_off[q + 1]]],
                                      real=[("L" if t[:1] in b"CL" else t.decode()) for t in q_real[q_off[q]:q_off[q + 1]]], cap=8)
            n_real_cost += int(ids_only.tobytes() != ref.tobytes())
    assert n_real_cost > 5, n_real_cost


def test_cased_text_rules():
    """pack_lines_cased / pack_cased_stream agree with each other: first word capitalised, numbers as digits behind the placeholder."""
    c_off, c_tok, _, _, _ = workloads.synthetic(500, 1, 25)
    lines = bytes(synth.pack_lines_cased(c_off, c_tok)).decode().split("\n")[:-1]
    norm = bytes(synth.pack_cased_stream(c_off, c_tok, 0)).decode().split("\0")[:-1]
    real = bytes(synth.pack_cased_stream(c_off, c_tok, 1)).decode().split("\0")[:-1]
    n_num = 0
    for s, ln in enumerate(lines):
        toks = ln.split(" ")
        assert len(toks) == c_off[s + 1] - c_off[s]
        for i, t in enumerate(toks):
            nrm, rl = norm[c_off[s] + i], real[c_off[s] + i]
            if t.isdigit():
                assert nrm == "｟num｠" and rl == t and synth.is_number(int(t))
                n_num += 1
            else:
                assert nrm == t.lower() and rl == ("C" if i == 0 else "L") and (t[0].isupper() == (i == 0))
    assert 0.01 < n_num / len(norm) < 0.03
