"""Contrastive rerank against matches the caller's vector already holds (src/fuzzy_match.cc:622-650: every round's
penalties run over ALL of `matches`, the entries present before the call first).  The reference has no test for it
(test/test.cc never calls match() with a contrastive factor on a non-empty vector), so the oracle's restatement is pinned
by a hand-derived case here; the GPU path (fmx_match_one_prior -> k_contrast<2> with priors) is compared with the oracle
bit for bit on randomised corpora."""
import numpy as np
import pytest

import oracle_lib
import workloads


def test_oracle_prior_hand_derived():
    """A = a b c d, B = a b c e, C = a b c f; pattern a b c d, factor 1, mean.  Candidates (score): A 1, B .75, C .75.
    Empty vector: A, then B and C tie at .75 - .75 = 0 -> first in list order (s_id) = B, then C.
    Vector already holds B: round 1 penalties vs B are A .75, B 1, C .75 -> keys .25, -.25, 0 -> A;
    round 2 (memory B, A): B (1 + .75) / 2 = .875 -> -.125, C (.75 + .75) / 2 = .75 -> 0 -> C; round 3: B, mean of
    (1, .75, .75)."""
    o = oracle_lib.OracleIndex()
    for s in (["a", "b", "c", "d"], ["a", "b", "c", "e"], ["a", "b", "c", "f"]):
        o.add_tm(s)
    o.sort(1)
    prm = oracle_lib.Params(fuzzy=0.5, nmatch=3, min_subseq_length=2, min_subseq_ratio=0, contrast=1.0, contrast_buffer=-1)
    wids = o.vocab.get_index(["a", "b", "c", "d"])
    plain, _ = o.match_ids(prm, wids, cap=8)
    assert list(plain["s_id"]) == [0, 1, 2]
    none, _ = o.match_ids_prior(prm, wids, [], cap=8)
    assert none.tobytes() == plain.tobytes()
    got, total = o.match_ids_prior(prm, wids, [o.vocab.get_index(["a", "b", "c", "e"])], cap=8)
    assert total == 3 and list(got["s_id"]) == [0, 2, 1]
    assert got["penalty"][0] == np.float32(0.75) and got["penalty"][1] == np.float32(0.75)
    assert got["penalty"][2] == np.float32((np.float32(1.0) + np.float32(0.75) + np.float32(0.75)) / np.float32(3))
    # maximum instead of mean: B's penalty stays 1 (against itself)
    prm_max = oracle_lib.Params(fuzzy=0.5, nmatch=3, min_subseq_length=2, min_subseq_ratio=0, contrast=1.0, reduce=1, contrast_buffer=-1)
    got, _ = o.match_ids_prior(prm_max, wids, [o.vocab.get_index(["a", "b", "c", "e"])], cap=8)
    assert list(got["s_id"]) == [0, 2, 1] and list(got["penalty"]) == [np.float32(0.75), np.float32(0.75), np.float32(1.0)]
    o.close()


def test_oracle_prior_memo_is_keyed_on_ids():
    """src/fuzzy_match.cc:627, 637-650: the cost of a (candidate, entry of `matches`) pair is memoised under the two S_IDS, so an
    entry whose id an earlier entry carries is scored through the EARLIER entry's cost, whatever its own tokens are.  Index:
    A = a b c d (0), B = a b c e (1), C = a b c f (2); pattern a b c d, factor 1, mean, three new matches.
    (1) The vector holds P1 = "a b c e" and P2 = four unknown words, both with s_id 7.  Penalties against P1: A .75, B 1, C .75;
        against P2 by its tokens 0, 0, 0 — but through the memo .75, 1, .75.  With ids: A (.75 + .75) / 2 = .75 is picked, then C
        (.75 x 3 / 3 = .75, key 0 against B's .75 - 2.75 / 3), then B with (1 + 1 + .75 + .75) / 4 = .875.  Without ids (every
        entry an id of its own) A's penalty is (.75 + 0) / 2 = .375.
    (2) The vector holds P = "a b c f" (C's tokens) with s_id 0 — A's id.  A is picked first (.75 against P); from then on the
        memo answers "(candidate, 0)" with the cost against P: C's penalties are 1 (P), 1 (A, through the memo; .75 by A's own
        tokens), .75 (B) -> 2.75 / 3 at its turn; without ids (1 + .75 + .75) / 3."""
    o = oracle_lib.OracleIndex()
    for s in (["a", "b", "c", "d"], ["a", "b", "c", "e"], ["a", "b", "c", "f"]):
        o.add_tm(s)
    o.sort(1)
    prm = oracle_lib.Params(fuzzy=0.5, nmatch=3, min_subseq_length=2, min_subseq_ratio=0, contrast=1.0, contrast_buffer=-1)
    wids = o.vocab.get_index(["a", "b", "c", "d"])
    f32 = np.float32
    p1, p2 = o.vocab.get_index(["a", "b", "c", "e"]), o.vocab.get_index(["x", "y", "z", "w"])
    got, _ = o.match_ids_prior(prm, wids, [p1, p2], cap=8, prior_sid=[7, 7])
    assert list(got["s_id"]) == [0, 2, 1] and list(got["penalty"]) == [f32(0.75), f32(0.75), f32(0.875)]
    got, _ = o.match_ids_prior(prm, wids, [p1, p2], cap=8, prior_sid=[7, 8])
    assert list(got["s_id"])[0] == 0 and got["penalty"][0] == f32(0.375)
    assert got.tobytes() == o.match_ids_prior(prm, wids, [p1, p2], cap=8)[0].tobytes()   # no ids: every entry an id of its own
    pc = o.vocab.get_index(["a", "b", "c", "f"])
    got, _ = o.match_ids_prior(prm, wids, [pc], cap=8, prior_sid=[0])
    assert list(got["s_id"]) == [0, 1, 2]
    assert list(got["penalty"]) == [f32(0.75), f32(0.75), f32((f32(1.0) + f32(1.0) + f32(0.75)) / f32(3))]
    got, _ = o.match_ids_prior(prm, wids, [pc], cap=8, prior_sid=[5])
    assert list(got["s_id"]) == [0, 1, 2] and got["penalty"][2] == f32((f32(1.0) + f32(0.75) + f32(0.75)) / f32(3))
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_gpu_prior_matches_bit_exact(fmx, gpu_device, seed):
    rng = np.random.default_rng(seed)
    c_off, c_tok = workloads.dense_corpus(rng, 3000, 40, 3, 14)
    long_sent = rng.integers(2, 42, size=90).astype(np.uint32)          # one indexed sentence beyond 64 tokens
    c_tok = np.concatenate([c_tok, long_sent]).astype(np.uint32)
    c_off = np.concatenate([c_off, [c_off[-1] + 90]]).astype(np.uint32)
    vs = 64
    ix = fmx.Index.create()
    ix.add_ids(c_off, c_tok, vs)
    ix.sort(gpu_device)
    ix.finalize()
    ix.to_device(gpu_device)
    o = workloads.build_oracle(c_off, c_tok, vs)
    n_checked = n_changed = n_memo = 0
    for it in range(120):
        s = int(rng.integers(0, len(c_off) - 1))
        pat = c_tok[c_off[s]:c_off[s + 1]].copy()
        if len(pat) > 2 and rng.random() < 0.7:
            pat[int(rng.integers(0, len(pat)))] = int(rng.integers(2, 42))
        kw = dict(fuzzy=0.35, nmatch=int(rng.integers(1, 5)), min_subseq_length=2, min_subseq_ratio=0,
                  contrast=float(rng.choice([0.3, 1.0])), reduce=int(rng.integers(0, 2)), contrast_buffer=int(rng.choice([-1, 6, 40])))
        # the prior entries: what a first call returned, or arbitrary token sequences (also empty / longer than 64 tokens)
        first, _ = ix.match_one(fmx.Params(**dict(kw, nmatch=2)), pat, cap=2)
        prior = [c_tok[c_off[r["s_id"]]:c_off[r["s_id"] + 1]] for r in first]
        k = it % 4
        if k == 1:
            prior.append(rng.integers(2, 42, size=int(rng.integers(1, 12))).astype(np.uint32))
        elif k == 2:
            prior = [long_sent, np.zeros(0, np.uint32)] + prior
        elif k == 3:
            prior = []
        ref, total = o.match_ids_prior(oracle_lib.Params(**kw), pat, prior, cap=8)
        got, gtot = ix.match_one_prior(fmx.Params(**kw), pat, prior, cap=8)
        assert gtot == total and got.tobytes() == ref.tobytes(), (it, kw, prior, ref, got)
        plain, _ = o.match_ids(oracle_lib.Params(**kw), pat, cap=8)
        n_changed += int(plain.tobytes() != ref.tobytes())
        n_checked += len(ref)
        # ... and with the entries' s_ids (round 5): the memo of pair costs is keyed on ids (src/fuzzy_match.cc:637) — ids shared
        # among the prior entries, and ids of sentences this call will pick, with tokens that are NOT those sentences'
        if prior:
            wanted = [int(r["s_id"]) for r in plain] or [0]
            sids = [int(rng.choice(wanted + [wanted[0], 100000 + (it % 3)])) for _ in prior]
            alien = [rng.integers(2, 42, size=max(1, len(x) + int(rng.integers(-1, 2)))).astype(np.uint32) if rng.random() < 0.5 else x for x in prior]
            ref_i, tot_i = o.match_ids_prior(oracle_lib.Params(**kw), pat, alien, cap=8, prior_sid=sids)
            got_i, gtot_i = ix.match_one_prior(fmx.Params(**kw), pat, alien, cap=8, prior_sid=sids)
            assert gtot_i == tot_i and got_i.tobytes() == ref_i.tobytes(), (it, kw, alien, sids, ref_i, got_i)
            ref_n, _ = o.match_ids_prior(oracle_lib.Params(**kw), pat, alien, cap=8)
            n_memo += int(ref_n.tobytes() != ref_i.tobytes())
    assert n_checked > 150 and n_changed > 20 and n_memo > 5, (n_checked, n_changed, n_memo)
    o.close()
