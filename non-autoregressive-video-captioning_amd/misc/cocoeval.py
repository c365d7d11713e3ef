"""Caption metrics without a JVM (reference: misc/cocoeval.py:46-104 over
coco-caption/pycocoevalcap/{bleu,rouge,cider}).

`COCOScorer().score(GT, RES, IDs)` keeps the reference's call and its return value
({'Bleu_1'..'Bleu_4', 'METEOR', 'ROUGE_L', 'CIDEr'}, per-video dict).  The three metrics the reference computes in
Python are restated here and are pinned to the reference's own scorers by tests/golden/tiny_metrics.json:

  * bleu(gts, res)     corpus BLEU-1..4, brevity penalty against the CLOSEST reference length, the 1e-15 / 1e-9
                       guards of bleu_scorer.py:211-256
  * rouge_l(gts, res)  F-measure (beta 1.2) of the best LCS precision and the best LCS recall over the references
                       (rouge.py:47-74)
  * cider(gts, res)    tf-idf weighted n-gram cosine (n=1..4, idf from the reference sets of the scored videos,
                       clipped numerator, Gaussian length penalty sigma 6 on the BIGRAM count, x10; cider_scorer.py:103-178)

Two pieces of the reference shell out to Java and cannot run here (no JVM, jars not vendored): the Stanford PTB
tokeniser and METEOR 1.5.  `tokenize` is a plain-Python stand-in for "-lowerCase" PTB tokenisation followed by the
reference's punctuation filter (ptbtokenizer.py:21-22,78-79) -- identical on the lower-cased, punctuation-free
captions the corpora hold (prepare_corpora.py), not pinned beyond that.  METEOR is reported only when a callable
is supplied (`COCOScorer(meteor=fn)`); otherwise 'METEOR' is 0.0 and is left out of `available`.
"""
import math
import os
import re

import numpy as np

PUNCTUATIONS = ("''", "'", "``", "`", "-LRB-", "-RRB-", "-LCB-", "-RCB-", ".", "?", "!", ",", ":", "-", "--", "...", ";")
_SPLIT = re.compile(r"\.\.\.|--|``|''|[A-Za-z0-9]+(?:'[A-Za-z]+)?|[^\sA-Za-z0-9]")
_BRACKETS = {"(": "-LRB-", ")": "-RRB-", "{": "-LCB-", "}": "-RCB-"}


class suppress_stdout_stderr(object):
    """silence file descriptors 1 and 2 for the duration of the block (cocoeval.py:15-43)"""

    def __enter__(self):
        self._null = [os.open(os.devnull, os.O_RDWR) for _ in range(2)]
        self._saved = (os.dup(1), os.dup(2))
        os.dup2(self._null[0], 1)
        os.dup2(self._null[1], 2)

    def __exit__(self, *_):
        os.dup2(self._saved[0], 1)
        os.dup2(self._saved[1], 2)
        for fd in self._null + list(self._saved):
            os.close(fd)


def tokenize(captions_for_image):
    """{id: [{'caption': str}, ...]} -> {id: [str, ...]}: lower-cased tokens, punctuation tokens dropped"""
    out = {}
    for k, items in captions_for_image.items():
        out[k] = []
        for c in items:
            toks = [_BRACKETS.get(t, t) for t in _SPLIT.findall(c['caption'].replace('\n', ' ').lower())]
            out[k].append(' '.join(t for t in toks if t not in PUNCTUATIONS))
    return out


def _ngrams(sentence, n=4):
    """(length, {ngram tuple: count}) with keys in the order length 1..n, left to right"""
    words = sentence.split()
    counts = {}
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            g = tuple(words[i:i + k])
            counts[g] = counts.get(g, 0) + 1
    return len(words), counts


def _check(gts, res):
    ids = sorted(gts.keys())
    assert ids == sorted(res.keys())
    for i in ids:
        assert isinstance(res[i], list) and len(res[i]) == 1 and isinstance(gts[i], list) and len(gts[i]) >= 1
    return ids


# ---- BLEU -------------------------------------------------------------------------------------------------------------
def bleu(gts, res, n=4):
    """-> ([BLEU-1..n], [[per-video BLEU-k] for k])"""
    small, tiny = 1e-9, 1e-15
    per_video = [[] for _ in range(n)]
    tot_guess, tot_correct = [0] * n, [0] * n
    tot_test = tot_ref = 0

    def finish(correct, guess, testlen, reflen):
        vals, prod = [], 1.0
        for k in range(n):
            prod *= (float(correct[k]) + tiny) / (float(guess[k]) + small)
            vals.append(prod ** (1.0 / (k + 1)))
        ratio = (testlen + tiny) / (reflen + small)
        if ratio < 1:
            vals = [v * math.exp(1 - 1 / ratio) for v in vals]
        return vals

    for vid in _check(gts, res):
        ref_lens, ref_max = [], {}
        for ref in gts[vid]:
            rl, counts = _ngrams(ref, n)
            ref_lens.append(rl)
            for g, c in counts.items():
                if c > ref_max.get(g, 0):
                    ref_max[g] = c
        testlen, counts = _ngrams(res[vid][0], n)
        reflen = min((abs(l - testlen), l) for l in ref_lens)[1]
        guess = [max(0, testlen - k) for k in range(n)]
        correct = [0] * n
        for g, c in counts.items():
            correct[len(g) - 1] += min(ref_max.get(g, 0), c)
        for k, v in enumerate(finish(correct, guess, testlen, reflen)):
            per_video[k].append(v)
        tot_test += testlen
        tot_ref += reflen
        for k in range(n):
            tot_guess[k] += guess[k]
            tot_correct[k] += correct[k]
    return finish(tot_correct, tot_guess, tot_test, tot_ref), per_video


# ---- ROUGE-L ----------------------------------------------------------------------------------------------------------
def _lcs(a, b):
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b):
            cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
        prev = cur
    return prev[len(b)]


def rouge_l(gts, res, beta=1.2):
    """-> (mean, per-video array)"""
    scores = []
    for vid in _check(gts, res):
        cand = res[vid][0].split(" ")
        prec = rec = 0.0
        for ref in gts[vid]:
            r = ref.split(" ")
            l = _lcs(r, cand)
            prec, rec = max(prec, l / float(len(cand))), max(rec, l / float(len(r)))
        scores.append(((1 + beta ** 2) * prec * rec) / float(rec + beta ** 2 * prec) if prec != 0 and rec != 0 else 0.0)
    scores = np.array(scores)
    return np.mean(scores), scores


# ---- CIDEr ------------------------------------------------------------------------------------------------------------
def cider(gts, res, n=4, sigma=6.0):
    """-> (mean, per-video array)"""
    ids = _check(gts, res)
    tests = [_ngrams(res[v][0], n)[1] for v in ids]
    refs = [[_ngrams(r, n)[1] for r in gts[v]] for v in ids]
    df = {}
    for rs in refs:
        for g in set(g for r in rs for g in r):
            df[g] = df.get(g, 0.0) + 1
    assert len(tests) >= max(df.values())
    log_docs = np.log(float(len(refs)))

    def vectorise(counts):
        vec = [{} for _ in range(n)]
        norm = [0.0] * n
        bigrams = 0
        for g, tf in counts.items():
            k = len(g) - 1
            w = float(tf) * (log_docs - np.log(max(1.0, df.get(g, 0.0))))
            vec[k][g] = w
            norm[k] += pow(w, 2)
            if k == 1:
                bigrams += tf
        return vec, [np.sqrt(x) for x in norm], bigrams

    scores = []
    for test, rs in zip(tests, refs):
        vh, nh, lh = vectorise(test)
        total = np.zeros(n)
        for r in rs:
            vr, nr, lr = vectorise(r)
            delta = float(lh - lr)
            val = np.zeros(n)
            for k in range(n):
                for g, w in vh[k].items():
                    wr = vr[k].get(g, 0.0)
                    val[k] += min(w, wr) * wr
                if nh[k] != 0 and nr[k] != 0:
                    val[k] /= (nh[k] * nr[k])
                val[k] *= np.e ** (-(delta ** 2) / (2 * sigma ** 2))
            total += val
        scores.append(np.mean(total) / len(rs) * 10.0)
    return np.mean(np.array(scores)), np.array(scores)


# ---- the reference's entry points -------------------------------------------------------------------------------------
def score(ref, sample):
    """{id: [tokenised str, ...]} x {id: [tokenised str]} -> {'Bleu_1'.., 'ROUGE_L', 'CIDEr'}  (cocoeval.py:162-179)"""
    out = {'Bleu_%d' % (k + 1): v for k, v in enumerate(bleu(ref, sample)[0])}
    out['ROUGE_L'] = rouge_l(ref, sample)[0]
    out['CIDEr'] = cider(ref, sample)[0]
    return out


class COCOScorer(object):
    def __init__(self, meteor=None):
        """meteor: optional callable (gts, res) -> (score, per-video scores) over tokenised captions"""
        self.meteor = meteor
        self.available = ['Bleu_1', 'Bleu_2', 'Bleu_3', 'Bleu_4', 'ROUGE_L', 'CIDEr'] + (['METEOR'] if meteor else [])

    def score(self, GT, RES, IDs):
        IDs = list(IDs)
        gts = tokenize({i: GT[i] for i in IDs})
        res = tokenize({i: RES[i] for i in IDs})
        self.eval, self.imgToEval = {}, {i: {'image_id': i} for i in IDs}
        order = sorted(IDs)                      # the scorers walk the videos in sorted-key order

        def put(name, total, each):
            self.eval[name] = total
            for i, s in zip(order, each):
                self.imgToEval[i][name] = s
        b, b_each = bleu(gts, res)
        for k in range(4):
            put('Bleu_%d' % (k + 1), b[k], b_each[k])
        if self.meteor is not None:
            put('METEOR', *self.meteor(gts, res))
        else:
            self.eval['METEOR'] = 0.0
        put('ROUGE_L', *rouge_l(gts, res))
        put('CIDEr', *cider(gts, res))
        return self.eval, self.imgToEval
