"""CPU: SURVEY 8f rows 3-4 host logic against what the REFERENCE returned on the same inputs
(tests/golden/tiny_host.json and tiny_nacf_trajectory.npz, written by oracle/make_golden.py from
misc/logger.py, misc/utils.py, coco-caption/pycocoevalcap and the reference's own training loop)."""
import csv
import json
import os

import numpy as np
import pytest
import torch

from nacf_amd.misc import cocoeval
from nacf_amd.misc.logger import AverageMeter, CsvLogger, k_PriorityQueue
from nacf_amd.misc.utils import analyze_length_novel_unique, duplicate, to_sentence
from oracle import nacf_oracle as O
from util import gold_opt, load_gold

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def host():
    with open(os.path.join(GOLD, "tiny_host.json")) as f:
        return json.load(f)


def test_bleu_rouge_cider_match_reference_scorers(host):
    m = host["metrics"]
    gts, res = m["gts"], m["res"]
    b, b_each = cocoeval.bleu(gts, res)
    assert np.allclose(b, m["bleu"], rtol=0, atol=1e-12)
    assert np.allclose(b_each, m["bleu_each"], rtol=0, atol=1e-12)
    r, r_each = cocoeval.rouge_l(gts, res)
    assert abs(r - m["rouge"]) < 1e-12 and np.allclose(r_each, m["rouge_each"], rtol=0, atol=1e-12)
    c, c_each = cocoeval.cider(gts, res)
    assert abs(c - m["cider"]) < 1e-10 and np.allclose(c_each, m["cider_each"], rtol=0, atol=1e-10)
    flat = cocoeval.score(gts, res)
    assert abs(flat["Bleu_4"] - m["bleu"][3]) < 1e-12 and abs(flat["CIDEr"] - m["cider"]) < 1e-10
    # exact match / nothing in common (fixture plants both)
    ids = sorted(gts)
    assert r_each[ids.index("video3")] == pytest.approx(1.0) and r_each[ids.index("video5")] == 0.0
    assert c_each[ids.index("video5")] == 0.0


def test_cocoscorer_shape_and_tokeniser(host):
    m = host["metrics"]
    GT = {k: [{"image_id": k, "cap_id": i, "caption": c} for i, c in enumerate(v)] for k, v in m["gts"].items()}
    RES = {k: [{"image_id": k, "caption": v[0]}] for k, v in m["res"].items()}
    sc = cocoeval.COCOScorer()
    total, each = sc.score(GT, RES, RES.keys())
    assert sorted(total) == ["Bleu_1", "Bleu_2", "Bleu_3", "Bleu_4", "CIDEr", "METEOR", "ROUGE_L"]
    assert abs(total["CIDEr"] - m["cider"]) < 1e-10 and abs(total["Bleu_4"] - m["bleu"][3]) < 1e-12
    assert "METEOR" not in sc.available and total["METEOR"] == 0.0
    assert set(each) == set(RES) and all("CIDEr" in v for v in each.values())
    # a METEOR backend plugs in
    sc2 = cocoeval.COCOScorer(meteor=lambda g, r: (0.25, [0.25] * len(g)))
    assert sc2.score(GT, RES, RES.keys())[0]["METEOR"] == 0.25 and "METEOR" in sc2.available
    tok = cocoeval.tokenize({"v": [{"caption": "A man's dog (small), runs -- fast... Really?!"}]})
    assert tok["v"] == ["a man's dog small runs fast really"]


def test_text_helpers_match_reference(host):
    for s_in, s_out, report in host["duplicate"]:
        assert list(duplicate(s_in)) == [s_out, report], s_in
    vocab = {int(k): v for k, v in host["analyze"]["vocab"].items()}
    for hyp, sent in host["to_sentence"]:
        assert to_sentence(hyp, vocab) == sent
    a = host["analyze"]
    out = analyze_length_novel_unique(a["gt_data"], a["preds"], vocab, a["splits"], n=1)
    assert out[0] == pytest.approx(a["ave_length"], abs=1e-12) and out[1] == a["novel"] and out[2] == a["unique"]
    assert out[3] == a["usage"] and out[4] == a["grams"] and out[5] == a["gram4"]


def test_k_best_queue_matches_reference(host, tmp_path):
    for run in host["kbest"]:
        root = tmp_path / ("k%d_%d" % (run["k_best"], run["tolerence"]))
        root.mkdir()
        opt = {"checkpoint_path": str(root), "tolerence": run["tolerence"]}
        folder = str(root / "tmp_models")
        q = k_PriorityQueue(k_best_model=run["k_best"], folder_path=folder, standard=run["standard"])
        for step in run["trace"]:
            res = dict(step["res"])
            ep = res["epoch"]
            (root / "checkpoint.pth.tar").write_text("epoch %d" % ep)
            name = "model_%04d.pth.tar" % ep
            ok, info = q.check(res, opt, os.path.join(folder, name), name)
            assert bool(ok) == step["ok"] and info == step["info"], (run["k_best"], ep, info, step["info"])
            assert q.continuous_failed_count == step["failed"] and q.qsize() == step["qsize"]
            assert res["Sum"] == pytest.approx(step["sum"], abs=1e-15)
            assert q.best_res.get("epoch", -1) == step["best_epoch"]
            if run["k_best"] > 1:
                assert sorted(os.listdir(folder)) == step["kept"]
            else:
                assert (root / "best.pth.tar").read_text() == step["best_file"]
        assert not run["trace"][-1]["ok"]          # every fixture run ends on the tolerance stop


def test_csv_logger_and_meter(tmp_path):
    lg = CsvLogger(filepath=str(tmp_path), filename="rec.csv", fieldsnames=["epoch", "train_loss", "CIDEr"])
    lg.write({"epoch": 0, "train_loss": 3.5, "CIDEr": 0.4, "not_a_column": 1})
    lg.write({"epoch": 1, "train_loss": 3.0, "CIDEr": 0.5})
    with pytest.raises(KeyError):
        lg.write({"epoch": 2, "train_loss": 2.0})
    lg.write_text("hello", print_t=False)
    rows = list(csv.DictReader(open(tmp_path / "rec.csv")))
    assert [r["epoch"] for r in rows] == ["0", "1"] and rows[1]["CIDEr"] == "0.5" and set(rows[0]) == {"epoch", "train_loss", "CIDEr"}
    assert (tmp_path / "log.txt").read_text() == "hello\n"
    again = CsvLogger(filepath=str(tmp_path), filename="rec.csv", fieldsnames=["epoch", "train_loss", "CIDEr"])
    assert len(list(csv.DictReader(open(again.csv_path)))) == 2        # an existing record is appended to, not reset
    m = AverageMeter()
    m.update(2.0, 3)
    m.update(4.0, 1)
    assert m.avg == pytest.approx(2.5) and m.count == 4
    m.update(6.0, 2, multiply=False)
    assert m.sum == pytest.approx(16.0)


def test_oracle_replays_reference_training_trajectory():
    """2 epochs x 3 steps of the reference loop (lr decay between epochs): loss per step, lr per step, final weights"""
    g = load_gold("tiny_nacf_trajectory")
    opt = gold_opt(g)
    sd = O.init_state_dict(opt, seed=0)
    st, lr = {}, opt["learning_rate"]
    E, S = int(g["epochs"]), int(g["steps"])
    for ep in range(E):
        for it in range(S):
            k = "b%d." % (ep * S + it)
            t = lambda n: torch.from_numpy(g[k + n])          # noqa: E731
            assert abs(float(g["lrs"][ep * S + it]) - lr) < 1e-12
            loss, _, _ = O.train_step(sd, opt, [t("feats0"), t("feats1")], [t("tokens_1"), t("tokens")], t("category"),
                                      [t("labels_1"), t("labels")], t("tgt_length"), st, lr=lr)
            assert abs(float(loss) - float(g["losses"][ep * S + it])) < 5e-5 * max(1.0, float(loss))
        lr = max(opt["minimum_learning_rate"], opt["decay"] * lr)
    assert abs(lr - float(g["final_lr"])) < 1e-12
    for key in g.files:
        if key.startswith("solid."):
            name = key[len("solid."):]
            mask = torch.from_numpy(g[key])
            if mask.any():
                assert float((sd[name] - torch.from_numpy(g["after." + name])).abs()[mask].max()) < 3e-4, name
