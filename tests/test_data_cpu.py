"""CPU: SURVEY 8f row 1 -- the data oracle against the fixture captured from the reference's own dataloader methods
(oracle/make_golden.py::data_case), the shard file format, and the caption table against the reference's sample
enumeration (_make_infoset)."""
import json
import os

import numpy as np
import pytest

import nacf_amd  # noqa: F401
from nacf_amd.data import CaptionTable, FeatureShard, write_feature_shard
from oracle import nacf_data_oracle as D
from util import load_gold


def _caps(g):
    caps = [row[:n].tolist() for row, n in zip(g["caps"], g["cap_len"])]
    poss = [row[:n].tolist() for row, n in zip(g["poss"], g["cap_len"])]
    return caps, poss


@pytest.mark.parametrize("dt,vw,max_len,beta", [("NARFormer", True, 10, [0.35, 0.9]), ("NARFormer", False, 30, [0.0, 1.0]),
                                                ("ARFormer", True, 10, [0, 1]), ("ARFormer", False, 30, [0, 1])])
@pytest.mark.parametrize("mode", ["train", "validate"])
def test_source_target_tables_match_reference(dt, vw, max_len, beta, mode):
    g = load_gold("tiny_data")
    caps, poss = _caps(g)
    opt = dict(decoding_type=dt, visual_word_generation=vw, max_len=max_len, beta=beta, seed=7)
    rng = np.random.RandomState(opt["seed"])            # same stream as the reference's self.random
    key = "%s.%s.%d.%s" % (dt, "vw" if vw else "plain", max_len, mode)
    got = {k: [] for k in ("tokens", "labels", "tokens_1", "labels_1")}
    for c, p in zip(caps, poss):
        r = D.make_source_target(c, p, opt, g["demanded"], g["is_be"], mode == "train", rng=rng)
        for k, v in r.items():
            got[k].append(v)
    checked = 0
    for k in got:
        if key + "." + k in g.files:
            assert np.array_equal(np.array(got[k]), g[key + "." + k]), (key, k)
            checked += 1
    assert checked >= 2


def test_frame_ids_resampling_length_targets_match_reference():
    g = load_gold("tiny_data")
    for total, n, ref in json.loads(str(g["frames_json"])):
        assert D.get_frame_ids(total, n, "equally_sampling") == ref
    rng = np.random.RandomState(11)
    assert [D.get_frame_ids(60, 8, "segment_random", rng) for _ in range(4)] == json.loads(str(g["seg_random_json"]))
    for s, t_, ref in json.loads(str(g["resampling_json"])):
        assert D.resampling(s, t_) == ref
    assert D.select_frames(60, 8, 8, 1, "equally_sampling") == D.get_frame_ids(60, 8, "equally_sampling")
    assert D.select_frames(5, 8, 8, 1, "equally_sampling") == D.resampling(5, 8)
    assert D.select_frames(60, 8, 60, 2, "equally_sampling") == list(range(60))


def test_feature_shard_round_trip(tmp_path):
    rs = np.random.RandomState(0)
    clips = [rs.rand(t, 16).astype(np.float32) for t in (60, 12, 60, 33)]
    path = os.path.join(str(tmp_path), "feats_i.nacf")
    write_feature_shard(path, clips, video_ids=np.array([7, 3, 11, 5]))
    sh = FeatureShard(path)
    assert (sh.N, sh.T, sh.D) == (4, 60, 16) and sh.lengths.tolist() == [60, 12, 60, 33]
    for i, c in enumerate(clips):
        assert np.array_equal(sh.array[i, :c.shape[0]], c) and not sh.array[i, c.shape[0]:].any()
    assert sh.row_of(np.array([11, 7, 5])).tolist() == [2, 0, 3]
    with pytest.raises(KeyError):
        sh.row_of(np.array([4]))
    with open(path, "r+b") as f:
        f.write(b"XXXXXXXX")
    with pytest.raises(ValueError):
        FeatureShard(path)


@pytest.mark.parametrize("mode,ncap", [("train", 0), ("train", 2), ("validate", 0)])
def test_caption_table_matches_reference_infoset(mode, ncap, tmp_path):
    g = load_gold("tiny_data")
    corpus = json.loads(str(g["corpus_json"]))
    info = dict(itow={int(k): v for k, v in corpus["itow"].items()}, itop={int(k): v for k, v in corpus["itop"].items()},
                itoc={int(k): v for k, v in corpus["itoc"].items()}, length_info=corpus["length_info"])
    opt = dict(max_len=10, n_caps_per_video=ncap, seed=3, demand=["VERB", "NOUN"])
    table, vids = CaptionTable.from_corpus(corpus["captions"], corpus["pos_tags"], info, corpus["splits"][mode], opt, mode,
                                           rng=np.random.RandomState(opt["seed"]))
    ref = json.loads(str(g["infoset_json"]))["%s.%d" % (mode, ncap)]
    assert len(table) == len(ref)
    for i, row in enumerate(ref):
        v = table.video[i]
        assert [int(vids[v]), int(table.cap_id[i]), int(table.category[v])] == row[:3]
        np.testing.assert_allclose(table.length_target[v], np.array(row[3:], dtype=np.float32), rtol=0, atol=1e-7)
        c = corpus["captions"]["video%d" % vids[v]][table.cap_id[i]]
        assert table.caps[i, :table.cap_len[i]].tolist() == c and not table.caps[i, table.cap_len[i]:].any()
    assert table.tag_demanded.tolist() == [int(info["itop"][i] in ("VERB", "NOUN")) for i in range(len(info["itop"]))]
    assert [i for i, b in enumerate(table.word_is_be) if b] == [6, 7, 8, 9, 10]
    p = os.path.join(str(tmp_path), "t.npz")
    table.save(p)
    t2 = CaptionTable.load(p)
    assert all(np.array_equal(getattr(table, k), getattr(t2, k)) for k in CaptionTable.FIELDS)
