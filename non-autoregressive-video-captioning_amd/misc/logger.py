"""Training records and k-best model selection (reference: misc/logger.py:10-211).

Same class names, constructor arguments, files on disk and return values as the reference so that `train_network_all`
reads like misc/run.py:272-359:

  * CsvLogger          one csv row per evaluated epoch (only the declared columns are written) + `log.txt`
  * AverageMeter       running average
  * k_PriorityQueue    keeps the k best checkpoints by the mean of each `standard` metric relative to the best value
                       that metric has reached so far ("Sum"); every call re-scores the retained results first, because
                       the per-metric bests move (logger.py:94-116); `check` returns (True, info line) or
                       (False, failure count) once `opt['tolerence']` evaluations in a row did not enter the queue.

The queue is a binary heap ordered by `res['Sum']` with `<` only, pushed and popped in the reference's order, so ties
resolve identically (queue.PriorityQueue is heapq underneath).
"""
import csv
import heapq
import os
import shutil

import torch


class CsvLogger(object):
    def __init__(self, filepath='./', filename='validate_record.csv', data=None,
                 fieldsnames=('epoch', 'train_loss', 'val_loss', 'Bleu_4', 'METEOR', 'ROUGE_L', 'CIDEr')):
        self.log_path = filepath
        os.makedirs(filepath, exist_ok=True)
        if not filename:
            return
        self.log_name = filename
        self.csv_path = os.path.join(filepath, filename)
        self.fieldsnames = list(fieldsnames)
        if not os.path.exists(self.csv_path):
            with open(self.csv_path, 'w') as f:
                csv.DictWriter(f, fieldnames=self.fieldsnames).writeheader()
        self.data = {k: [] for k in self.fieldsnames}
        for row in (data or ()):
            self.write({k: (int(v) if k == 'epoch' else float(v)) for k, v in row.items()})

    def write(self, data):
        for k, column in self.data.items():
            column.append(data[k])              # a missing declared column is an error, as in the reference
        with open(self.csv_path, 'a') as f:
            csv.DictWriter(f, fieldnames=self.fieldsnames).writerow({k: data[k] for k in self.fieldsnames})

    def write_text(self, text, print_t=True):
        with open(os.path.join(self.log_path, 'log.txt'), 'a') as f:
            f.write('%s\n' % text)
        if print_t:
            print(text)


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1, multiply=True):
        self.val = val
        self.sum += val * n if multiply else val
        self.count += n
        self.avg = self.sum / self.count


class ModelNode(object):
    def __init__(self, res, model_path, key='Sum'):
        self.res, self.model_path, self.key = res, model_path, key

    def __lt__(self, other):
        return self.res[self.key] < other.res[self.key]


class k_PriorityQueue(object):
    def __init__(self, k_best_model, folder_path, standard=('METEOR', 'CIDEr')):
        self.k_best_model = k_best_model
        self.folder_path = folder_path
        self.key = 'Sum'
        self.continuous_failed_count = 0
        self.best_res = {self.key: 0, 'Bleu_4': 0, 'METEOR': 0, 'ROUGE_L': 0, 'CIDEr': 0}
        self.best_ = {k: 0 for k in standard}
        self._heap = []
        if k_best_model > 1:
            os.makedirs(folder_path, exist_ok=True)

    # ---- scoring -------------------------------------------------------------------------------------------------
    def score(self, res):
        """res['Sum'] = mean_k res[k] / best_k, after raising best_k to res[k] where it is exceeded"""
        rel = 0
        for k in self.best_:
            if res[k] > self.best_[k]:
                self.best_[k] = res[k]
            rel += res[k] / self.best_[k]
        res[self.key] = rel / len(self.best_)

    def update(self, res):
        self.score(res)
        self.score(self.best_res)
        old, self._heap = self._heap, []
        while old:                                  # pop in heap order, re-score, push: the reference's rebuild
            node = heapq.heappop(old)
            self.score(node.res)
            heapq.heappush(self._heap, node)

    def get_print_info(self, res):
        b = self.best_res
        cols = ''.join('\t%s %5.2f(%5.2f)' % (tag, 100 * res[k], 100 * (res[k] - b[k]))
                       for tag, k in (('B', 'Bleu_4'), ('M', 'METEOR'), ('R', 'ROUGE_L'), ('C', 'CIDEr')))
        return '%2d, %6.2f %s %6.2f%s' % (self.continuous_failed_count, 100 * res['Sum'], res['Sum'] > b['Sum'],
                                         100 * b['Sum'], cols)

    # ---- selection -----------------------------------------------------------------------------------------------
    def check(self, res, opt, model_path=None, model_name=None):
        """k == 1: the kept model is <checkpoint_path>/best.pth.tar; k > 1: <folder_path>/<model_name>, and the
        evicted model's file (model_%04d.pth.tar of its epoch) is removed."""
        self.update(res)
        src = os.path.join(opt['checkpoint_path'], 'checkpoint.pth.tar')
        single = self.k_best_model == 1
        dst = os.path.join(opt['checkpoint_path'], 'best.pth.tar') if single else os.path.join(self.folder_path, model_name)
        kept_path = dst if single else model_path
        if len(self._heap) == self.k_best_model:
            worst = heapq.heappop(self._heap)
            if res['Sum'] > worst.res['Sum']:
                self.continuous_failed_count = 0
                heapq.heappush(self._heap, ModelNode(res, kept_path))
                shutil.copy(src, dst)
                if not single:
                    os.remove(os.path.join(self.folder_path, 'model_%04d.pth.tar' % worst.res['epoch']))
            else:
                heapq.heappush(self._heap, worst)
                self.continuous_failed_count += 1
                if self.continuous_failed_count >= opt['tolerence']:
                    return False, self.continuous_failed_count
        else:
            heapq.heappush(self._heap, ModelNode(res, kept_path))
            shutil.copy(src, dst)
        info = self.get_print_info(res)
        if res['Sum'] > self.best_res['Sum']:
            self.best_res = res
        return True, info

    def check_only_one(self, res, opt, *args):
        assert self.k_best_model == 1
        return self.check(res, opt)

    def check_multiple(self, res, opt, model_path, model_name):
        assert self.k_best_model > 1
        return self.check(res, opt, model_path, model_name)

    def load(self):
        for name in os.listdir(self.folder_path):
            path = os.path.join(self.folder_path, name)
            res = torch.load(path, map_location='cpu', weights_only=False)['validate_result']
            heapq.heappush(self._heap, ModelNode(res, path))

    def qsize(self):
        return len(self._heap)

    def get(self):
        return heapq.heappop(self._heap)
