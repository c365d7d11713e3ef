"""Training / evaluation orchestration (reference: misc/run.py:34-359).

Same entry points and argument meaning as the reference -- `get_forword_results` (sic), `get_loader`, `run_eval`,
`run_train`, `train_network_all` -- over the MI355X data path and step engine:

  * batches come from `data.ShardLoader` (features resident in HBM or streamed, frame sampling and masked-LM target
    construction on the device) instead of DataLoader(VideoDataset) -- `get_loader` attaches a `.dataset` facade with
    the accessors run.py uses (get_vocab, get_references, shuffle, captions, splits);
  * `run_train` drives `runtime.engine.TrainStep`: after two launch-by-launch steps the whole step (zero_grad, forward,
    criterion, backward, clip + Adam) replays from a hipGraph; the reference's `clip_grad_value_` (run.py:260) lives
    inside the fused Adam launch; with torch.distributed initialised every rank trains its shard of the global batch
    and gradients are all-reduced over RCCL (runtime/ddp.py);
  * `run_eval` decodes with `Translator.translate_batch` and scores with the Java-free `misc.cocoeval.COCOScorer`;
  * `train_network_all` keeps the epoch loop: lr decay per epoch, evaluation every `save_checkpoint_every` epochs
    after `start_eval_epoch`, csv record, `checkpoint.pth.tar` in the reference's checkpoint format, k-best selection
    with `tolerence` early stop, teacher weights / teacher model.  Its tail `os.system('python translate.py ...')`
    (run.py:350-356) becomes an in-process evaluation of the best checkpoint on the test split.

TensorBoard writers are accepted and used only if the caller passes one (tensorboardX is not a dependency).
"""
import json
import os
import sys
import pickle
import shutil
import time
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist

from ..config import Constants
from ..models.Translator import Translator
from ..opts import persistable
from ..runtime.ddp import DataParallel, host_broadcast_int
from ..runtime.engine import TrainStep
from .cocoeval import COCOScorer
from .crit import get_criterion
from .logger import CsvLogger, k_PriorityQueue
from .optim import get_optimizer
from .utils import (analyze_length_novel_unique, duplicate, load_model_and_opt, load_satisfied_weights,
                    save_checkpoint, to_sentence)


def prepare_data(data, key, device):
    v = data.get(key, None)
    return None if v is None else v.to(device)


def get_forword_results(opt, model, data, device, only_data=False, vocab=None, **kwargs):
    """batch dictionary -> model results with the criterion's targets attached (run.py:40-86).  Accepts the
    reference's per-modality keys (`feats_m`, `feats_i`) as well as the loader's `feats` list."""
    category, labels = data['category'].to(device), data['labels'].to(device)
    if 'feats' in data:
        feats = [f.to(device) for f in data['feats']]
    else:
        feats = [data['feats_%s' % c].to(device) for c in opt['modality'].lower()]
    vw = opt.get('visual_word_generation', False)
    if only_data:                           # evaluation: the decoder inputs are built by the decoding loop
        results = model.encode(feats=feats)
    else:
        tokens = [data['tokens_1'].to(device), data['tokens'].to(device)] if vw else data['tokens'].to(device)
        results = model(feats=feats, tgt_tokens=tokens, category=category, opt=opt, vocab=vocab, **kwargs)
    start = 1
    if opt['decoding_type'] == 'NARFormer':
        results[Constants.mapping['length'][1]] = prepare_data(data, 'length_target', device)
        start = 0
    lab = labels[:, start:]
    if vw and 'labels_1' in data:           # evaluation batches carry no visual-word pass
        lab = [data['labels_1'].to(device)[:, start:], lab]
    results[Constants.mapping['lang'][1]] = lab
    if only_data:
        return results, category, labels
    return results


def get_criterion_during_evaluation(opt, **kwargs):
    """only the auxiliary (non-language) terms are tracked while decoding (crit.py:242-251)"""
    sub = defaultdict(list)
    for key in ('attribute', 'length'):
        if key in opt['crit']:
            i = opt['crit'].index(key)
            for k in ('crit', 'crit_key', 'crit_name', 'crit_scale'):
                if k in opt and len(opt[k]) > i:
                    sub[k].append(opt[k][i])
    return get_criterion(dict(sub), **kwargs) if len(sub) else None


# ---- data ---------------------------------------------------------------------------------------------------------
class CorpusView(object):
    """The parts of VideoDataset (dataloader.py:40-113) the run loop touches."""

    def __init__(self, opt, mode, corpus, loader_factory):
        self.opt, self.mode = opt, mode
        self.captions, self.pos_tags = corpus['captions'], corpus.get('pos_tags')
        info = corpus['info']
        self.itow, self.itoc, self.itop = info['itow'], info.get('itoc'), info.get('itop')
        self.splits = info['split']
        self.n_caps_per_video = opt.get('n_caps_per_video', 0) if mode == 'train' else 1
        self.references = None
        self._factory = loader_factory

    def get_references(self):
        if self.references is None:
            with open(self.opt['reference'], 'rb') as f:
                self.references = pickle.load(f)
        return self.references

    def get_preprocessed_references(self):
        return self.captions

    def get_vocab(self):
        return self.itow

    def get_vocab_size(self):
        return len(self.itow)

    def shuffle(self):
        """re-draws the captions kept per video when n_caps_per_video > 0 (dataloader.py:103-108); the batch order
        itself is re-permuted by the loader at the start of every epoch"""
        if self.n_caps_per_video:
            self._factory(redraw=True)


class _Loader(object):
    """iterable with `.dataset`, re-bindable so that `dataset.shuffle()` can swap the caption table underneath"""

    def __init__(self):
        self.inner = self.dataset = None

    def __iter__(self):
        ids = self.video_names
        for b in self.inner:
            b['video_ids'] = [ids[int(v)] for v in self.inner.table.video[b['sample_index_host']]]
            yield b

    def __len__(self):
        return len(self.inner)


def get_loader(opt, mode, print_info=False, specific=-1, device=None, **kwargs):
    """DataLoader(VideoDataset(opt, mode), batch_size, shuffle=train) of the reference (run.py:89-96).
    opt['info_corpus'] is the reference's corpus pickle; opt['feats_<m>'] names ONE feature shard per modality
    (data/shards.py; the reference's per-video HDF5 files are re-packed once with `write_feature_shard`).
    specific != -1: only the videos of that category, info['split_category'][mode][specific] (dataloader.py:151-156)."""
    from ..data import CaptionTable, FeatureShard, ShardLoader
    assert mode in ('train', 'validate', 'test')
    device = torch.device('cuda') if device is None else torch.device(device)
    with open(opt['info_corpus'], 'rb') as f:
        corpus = pickle.load(f)
    shards = []
    for c in opt['modality'].lower():
        path = opt['feats_%s' % c]
        if isinstance(path, (list, tuple)):
            if len(path) != 1:
                raise NotImplementedError('nacf_amd: one feature shard per modality (concatenate when re-packing)')
            path = path[0]
        if str(path).endswith('.hdf5'):
            raise ValueError('nacf_amd: %s is an HDF5 file; re-pack it with nacf_amd.data.write_feature_shard' % path)
        shards.append(FeatureShard(path))
    batch_size = kwargs.get('batch_size', opt['batch_size'])
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    rs = np.random.RandomState(opt.get('seed', 0))          # the reference's per-dataset RandomState(opt['seed'])
    out = _Loader()
    state = {'epoch': 0}

    if specific != -1:
        by_category = corpus['info'].get('split_category')
        if by_category is None:
            raise KeyError("nacf_amd: specific=%d needs info['split_category'] in %s" % (specific, opt['info_corpus']))
        members = by_category[mode][specific]
    else:
        members = corpus['info']['split'][mode]

    def build(redraw=False):
        table, vids = CaptionTable.from_corpus(corpus['captions'], corpus.get('pos_tags'), corpus['info'],
                                               members, opt, mode, rng=rs)
        if redraw:
            state['epoch'] += 1
        out.inner = ShardLoader(shards, table, vids, opt, batch_size=batch_size, device=device, mode=mode,
                                seed=opt.get('seed', 0) + 7919 * state['epoch'], rank=rank if mode == 'train' else 0,
                                world=world if mode == 'train' else 1,
                                **{k: kwargs[k] for k in ('placement', 'drop_last', 'hbm_budget_bytes') if k in kwargs})
        out.video_names = ['video%d' % int(v) for v in vids]
    out.dataset = CorpusView(opt, mode, corpus, build)
    build()
    if print_info:
        sp = corpus['info']['split']
        print('Dataset: train %d / validate %d / test %d videos, vocab %d, max_len %d' % (
            len(sp['train']), len(sp['validate']), len(sp['test']), len(corpus['info']['itow']), opt['max_len']))
    return out


# ---- evaluation ---------------------------------------------------------------------------------------------------
def run_eval(opt, model, crit, loader, vocab, device, json_path='', json_name='', scorer=None, teacher_model=None,
             dict_mapping={}, no_score=False, print_sent=False, analyze=False,
             collect_best_candidate_iterative_results=False, collect_path=None, extra_opt={}, summarywriter=None,
             global_step=0):
    """decode every batch of `loader`, turn ids into sentences, score against the references (run.py:99-246)"""
    opt.update(extra_opt)
    model.eval()
    if teacher_model is not None:
        teacher_model.eval()
    scorer = COCOScorer() if scorer is None else scorer
    gt_captions = loader.dataset.get_references()
    pred_captions = defaultdict(list)
    opt['collect_best_candidate_iterative_results'] = collect_best_candidate_iterative_results
    translator = Translator(model=model, opt=opt, device=device, teacher_model=teacher_model, dict_mapping=dict_mapping)
    collected_sents, collected_score = defaultdict(list), defaultdict(list)
    collect_ar = opt['decoding_type'] == 'ARFormer' and collect_best_candidate_iterative_results
    na_dedup = opt.get('duplicate', False) and opt['decoding_type'] == 'NARFormer'
    if crit is not None:
        crit.reset_loss_recorder()
    decode_time = 0.0

    for data in loader:
        with torch.no_grad():
            enc, category, labels = get_forword_results(opt, model, data, device=device, only_data=True, vocab=vocab)
            if crit is not None:
                crit.get_loss(enc)
            t_enc = None
            if teacher_model is not None:
                t_enc = get_forword_results(opt, teacher_model, data, device=device, only_data=True, vocab=vocab)[0]
            if opt['batch_size'] == 1:
                torch.cuda.synchronize()
                t0 = time.time()
            all_hyp, all_scores = translator.translate_batch(enc, category, labels, vocab, teacher_encoder_outputs=t_enc)
            if opt['batch_size'] == 1:
                torch.cuda.synchronize()
                decode_time += time.time() - t0
            if isinstance(all_hyp, torch.Tensor):
                all_hyp = (all_hyp.unsqueeze(1) if all_hyp.dim() == 2 else all_hyp).tolist()
            if isinstance(all_scores, torch.Tensor):
                all_scores = (all_scores.unsqueeze(1) if all_scores.dim() == 2 else all_scores).tolist()
        video_ids = np.array(data['video_ids']).reshape(-1)

        for k, hyps in enumerate(all_hyp):
            vid = video_ids[k]
            if not no_score:
                assert len(hyps) == 1
            for j, hyp in enumerate(hyps):
                sent = to_sentence(hyp, vocab)
                if na_dedup:
                    sent, _ = duplicate(sent)
                if print_sent:
                    print('%s: %s' % (vid, sent))
                if collect_ar:
                    pred_captions[vid].append({'caption': sent, 'score': all_scores[k][j]})
                else:
                    pred_captions[vid].append({'image_id': vid, 'caption': sent})

        if collect_best_candidate_iterative_results and not collect_ar:
            # NA decoding: the best candidate's tokens and scores after every refinement iteration
            assert isinstance(all_scores, tuple)
            it_sents, it_scores = all_scores[0].tolist(), all_scores[1].tolist()
            ids = video_ids
            if len(ids) != len(it_sents):
                ids = np.array(data['video_ids'])[:, np.newaxis].repeat(opt['length_beam_size'], axis=1).reshape(-1)
                assert len(ids) == len(it_sents)
            for vid, hyps, scores in zip(ids, it_sents, it_scores):
                assert len(hyps) == len(scores)
                n_words = None
                for j, (hyp, sc) in enumerate(zip(hyps, scores)):
                    sent = to_sentence(hyp, vocab)
                    n = len(sent.split(' '))
                    assert n_words in (None, n)
                    n_words = n
                    print('%10s(iteration %d Length %d): %s' % (vid, j, n, sent))
                    collected_sents[vid].append(sent)
                    collected_score[vid].append(sc)

    if collect_best_candidate_iterative_results:
        assert collect_path is not None
        with open(collect_path, 'wb') as f:
            pickle.dump(pred_captions if collect_ar else [collected_sents, collected_score], f)
    if opt['batch_size'] == 1:
        print(decode_time / max(1, len(loader)), len(loader))

    res, valid_score = {}, {}
    if analyze:
        ave_length, novel, unique, usage, _, gram4 = analyze_length_novel_unique(
            loader.dataset.captions, pred_captions, vocab, splits=loader.dataset.splits, n=1)
        res.update({'ave_length': ave_length, 'novel': novel, 'unique': unique, 'usage': usage, 'gram4': gram4})
    if not no_score:
        valid_score, _ = scorer.score(gt_captions, pred_captions, pred_captions.keys())
        res.update(valid_score)
        metric_sum = opt.get('metric_sum', [1, 1, 1, 1])
        candidate = [res['Bleu_4'], res['METEOR'], res['ROUGE_L'], res['CIDEr']]
        res['Sum'] = sum(v for v, keep in zip(candidate, metric_sum) if keep)
        if crit is not None:
            for n, m in zip(*crit.get_loss_info()):
                res[n] = m
    if summarywriter is not None:
        for k, v in res.items():
            summarywriter.add_scalar(k, v, global_step=global_step)
    if json_path:
        os.makedirs(json_path, exist_ok=True)
        with open(os.path.join(json_path, json_name), 'w') as f:
            json.dump({'predictions': pred_captions, 'scores': valid_score}, f)
    return res


# ---- training -----------------------------------------------------------------------------------------------------
def run_train(opt, model, crit, optimizer, loader, device, logger=None, epoch=-1, return_all_info=False, engine=None,
              **kwargs):
    """one epoch (run.py:249-269).  `engine`: a TrainStep kept across epochs by the caller so that the captured
    hipGraphs are reused; without one a fresh engine is built (two launch-by-launch steps, then capture)."""
    model.train()
    crit.reset_loss_recorder()
    vocab = loader.dataset.get_vocab()
    if engine is None:
        engine = make_engine(opt, model, crit, optimizer, device, vocab=vocab, **kwargs)
    inner = getattr(loader, 'inner', loader)
    for data in loader:
        engine(data)
        if engine.captured and hasattr(inner, 'bind_outputs'):
            inner.bind_outputs(engine.static)      # from now on the loader builds batches inside the graph's inputs
    if hasattr(inner, 'bind_outputs'):
        inner.bind_outputs(None)
    name, loss_info = crit.get_loss_info()
    if logger is not None:
        logger.write_text('\t'.join('%10s: %05.3f' % item for item in zip(name, loss_info)))
    return loss_info if return_all_info else loss_info[0]


def make_engine(opt, model, crit, optimizer, device, vocab=None, graph=None, **kwargs):
    ddp = DataParallel(model) if dist.is_initialized() and dist.get_world_size() > 1 else None
    if ddp is not None:
        ddp.broadcast_parameters()
    fwd = lambda b: get_forword_results(opt, model, b, device=device, only_data=False, vocab=vocab, **kwargs)  # noqa: E731
    return TrainStep(model, crit, optimizer, fwd, ddp=ddp, graph=graph or opt.get('hipgraph', 'auto'))


def train_network_all(opt, model, device, summarywriter=None, **kwargs):
    """the reference's training driver (run.py:272-359); returns the k-best queue's best validation result"""
    if opt.get('load_teacher_weights', False):
        assert opt.get('teacher_path', None) is not None
        model = load_satisfied_weights(model=model, checkpoint_path=opt['teacher_path'],
                                       str_mapping={'decoder.bert.': 'decoder.'})
    model.to(device)
    # this loop owns the criterion, so the vocabulary projection + log-softmax + NLL run as ONE fused function
    # (runtime/functional.py:FusedVocabXentMultiFn) unless the caller asked for materialised log-probs
    # (a runtime switch of the MODEL OBJECT: neither the caller's opt nor the checkpoint's `settings` carry it, so a
    # reloaded checkpoint returns the reference's `tgt_word_logprobs` again)
    model.opt['fused_loss'] = bool(opt.get('fused_loss', True))
    rank0 = not dist.is_initialized() or dist.get_rank() == 0
    optimizer = get_optimizer(opt, model, summarywriter=summarywriter)
    crit = get_criterion(opt, summarywriter=summarywriter)
    crit_eval = get_criterion_during_evaluation(opt)
    teacher_model = None
    if opt.get('with_teacher', False) and opt['method'] in ('NAB', 'NACF'):
        assert opt.get('teacher_path', None) is not None
        teacher_model, _ = load_model_and_opt(opt['teacher_path'], device)

    train_loader = get_loader(opt, 'train', device=device, **kwargs)
    vali_loader = get_loader(opt, 'validate', device=device)
    vocab = vali_loader.dataset.get_vocab()
    scorer = kwargs.get('scorer') or COCOScorer()
    standard = [k for k in opt.get('standard', ['METEOR', 'CIDEr']) if k in getattr(scorer, 'available', [k])]
    dropped = [k for k in opt.get('standard', ['METEOR', 'CIDEr']) if k not in standard]
    if dropped and rank0:
        print('[nacf_amd] WARNING: %s not available from this scorer (no METEOR jar / JVM here): best-model selection uses '
              '%s only -- the reference selects on METEOR + CIDEr' % (dropped, standard), file=sys.stderr)
    folder_path = os.path.join(opt['checkpoint_path'], 'tmp_models')
    best_model = k_PriorityQueue(k_best_model=opt.get('k_best_model', 1), folder_path=folder_path, standard=standard)
    logger = CsvLogger(filepath=opt['checkpoint_path'], filename='trainning_record.csv',
                       fieldsnames=['epoch', 'train_loss', 'Bleu_1', 'Bleu_2', 'Bleu_3', 'Bleu_4', 'METEOR', 'ROUGE_L',
                                    'CIDEr', 'Sum'] + crit.get_fieldsnames()) if rank0 else None
    say = (lambda text: logger.write_text(text)) if rank0 else (lambda text: None)
    engine = make_engine(opt, model, crit, optimizer, device, vocab=vocab)

    for epoch in range(opt['epochs']):
        train_loader.dataset.shuffle()
        say('epoch %d lr=%g (ss_prob=%g)' % (epoch, optimizer.get_lr(), opt.get('teacher_prob', 1)))
        train_loss = run_train(opt, model, crit, optimizer, train_loader, device, logger=logger, epoch=epoch, engine=engine)
        optimizer.epoch_update_learning_rate()

        if (epoch + 1) > opt['start_eval_epoch'] and (epoch + 1) % opt['save_checkpoint_every'] == 0:
            stop = torch.zeros(1, device=device)
            if dist.is_initialized() and dist.get_world_size() > 1 and not model.opt.get('sync_bn', False):
                # per-rank BatchNorm: every rank saw 1/world of the data.  Average the running statistics so that the
                # evaluated / saved model reflects all of it (with sync_bn they are identical already)
                for m in model.modules():
                    if isinstance(m, torch.nn.BatchNorm1d):
                        for buf in (m.running_mean, m.running_var):
                            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                            buf.div_(dist.get_world_size())
            if rank0:                        # replicas are identical: rank 0 evaluates, the others wait for its verdict
                res = run_eval(opt, model, crit_eval, vali_loader, vocab, device, teacher_model=teacher_model,
                               analyze=True, scorer=scorer, summarywriter=summarywriter, global_step=epoch)
                res['train_loss'], res['epoch'] = train_loss, epoch
                for k in logger.fieldsnames:       # a criterion column the evaluation criterion does not produce
                    res.setdefault(k, float('nan'))
                logger.write(res)
                save_checkpoint({'epoch': epoch + 1, 'state_dict': model.state_dict(), 'validate_result': res,
                                 'settings': persistable(opt)}, False, filepath=opt['checkpoint_path'],
                                filename='checkpoint.pth.tar')
                model_name = 'model_%04d.pth.tar' % res['epoch']
                go_on, info = best_model.check(res, opt, os.path.join(folder_path, model_name), model_name)
                if go_on:
                    say(info)
                else:
                    stop.fill_(1)
            # the verdict travels through the rendezvous store: the idle ranks wait on the host, not inside an RCCL call
            if host_broadcast_int(int(float(stop) > 0), 'stop_after_epoch_%d' % epoch) > 0:
                break                        # `tolerence` evaluations in a row without entering the k-best queue

    final = None
    if rank0 and not opt.get('no_test', False):
        best_path = os.path.join(opt['checkpoint_path'], 'best.pth.tar')
        if opt.get('k_best_model', 1) == 1 and os.path.exists(best_path):
            best, best_opt = load_model_and_opt(best_path, device)
            test_loader = get_loader(best_opt, 'test', device=device)
            final = run_eval(best_opt, best, None, test_loader, vocab, device, teacher_model=teacher_model,
                             analyze=True, scorer=scorer)
            say('test: ' + '\t'.join('%s %.4f' % (k, final[k]) for k in ('Bleu_4', 'METEOR', 'ROUGE_L', 'CIDEr')))
    if rank0 and opt.get('k_best_model', 1) > 1:
        shutil.rmtree(folder_path, ignore_errors=True)
    return best_model.best_res, final
