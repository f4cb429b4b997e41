"""analysis.py of the reference (the metric functions, :314-787) on the MI355X.

`results` is the dict full_model_eval.py:128-135 builds — 'y_out' (thresholded, [B,T,H,W]),
'y_gt', 's_out', 's_gt' — with float32 CUDA tensors.  All functions share one device pass
(ops.eval_metrics: pairwise intersections on the MFMA streaming kernel + one small metrics
kernel), cached on the dict.  The analyzers that render images or write CSV files (:52-311,
:790-900) are out of scope."""
import torch

import ra_ops as ops


def _m(results):
  if '_metrics' not in results:
    results['_metrics'] = ops.eval_metrics(results['y_out'], results['y_gt'], results['s_gt'])
    if 'iou_pairwise' not in results:
      results['iou_pairwise'] = results['_metrics']['iou_pairwise']
  return results['_metrics']


def _stat(results, name):
  return _m(results)['stats'][:, ops.EVAL_NAMES.index(name)]


def _inst(results, name):
  return _m(results)['inst'][:, ops.EVALI_NAMES.index(name)]


def f_iou_pairwise(a, b):
  """:329-334 for batches: a [B,N,H,W], b [B,M,H,W] binary -> [B,N,M]."""
  st = ops.pair_stats(a, b, want=('inter', 'sum_a', 'sum_b'))
  union = st['sum_a'][:, :, None] + st['sum_b'][:, None, :] - st['inter']
  return st['inter'] / (union + (union == 0).to(torch.float32))


def f_iou(a, b):
  """:314-326 for aligned [B,N,H,W] masks -> [B,N]."""
  return torch.diagonal(f_iou_pairwise(a, b), dim1=1, dim2=2)


def f_symmetric_best_dice(results):
  """:434-460 -> [B]."""
  return _stat(results, 'sbd')


def f_coverage(results, weighted=False):
  """:481-504."""
  return _stat(results, 'wt_cov' if weighted else 'unwt_cov')


def f_wt_coverage(results):
  return f_coverage(results, weighted=True)


def f_unwt_coverage(results):
  return f_coverage(results, weighted=False)


def f_fg_iou(results):
  """:533-553."""
  return _stat(results, 'fg_iou')


def f_fg_dice(results):
  """:556-576."""
  return _stat(results, 'fg_dice')


def f_fp(results):
  """:579-592."""
  return _stat(results, 'avg_fp')


def f_fn(results):
  """:595-605."""
  return _stat(results, 'avg_fn')


def f_pixel_pr(results):
  """:608-627 -> 1-D tensor over the output instances that exist."""
  return _inst(results, 'pix_pr')[_inst(results, 'has_out') > 0]


def f_pixel_re(results):
  """:630-650 -> 1-D tensor over the ground-truth instances."""
  return _inst(results, 'pix_re')[_inst(results, 'is_gt') > 0]


def f_obj_pr(results):
  """:653-671."""
  return _inst(results, 'obj_pr')[_inst(results, 'has_out') > 0]


def f_obj_re(results):
  """:674-690."""
  return _inst(results, 'obj_re')[_inst(results, 'is_gt') > 0]


def f_count_mse(results):
  """:693-708."""
  return _stat(results, 'count_mse')


def f_count_acc(results):
  """:711-726."""
  return _stat(results, 'count_acc')


def f_dic(results):
  """:729-744."""
  return _stat(results, 'dic')


def f_dic_abs(results):
  """:747-763."""
  return _stat(results, 'dic_abs')


def f_count_out(y_out):
  """:766-770."""
  sizes = ops.pair_stats(y_out, y_out[:, :1], want=('sum_a',))['sum_a']
  return (sizes > 0).to(torch.float32)


ANALYZERS = {'sbd': f_symmetric_best_dice, 'wt_cov': f_wt_coverage, 'unwt_cov': f_unwt_coverage,
             'fg_dice': f_fg_dice, 'fg_iou': f_fg_iou, 'avg_fp': f_fp, 'avg_fn': f_fn,
             'avg_pr': f_pixel_pr, 'avg_re': f_pixel_re, 'obj_pr': f_obj_pr, 'obj_re': f_obj_re,
             'count_acc': f_count_acc, 'count_mse': f_count_mse, 'dic': f_dic, 'dic_abs': f_dic_abs}


def create_analyzer(name, display_name=None, fname=None):
  """:9-49 reduced to the metric function: returns f(results) -> tensor; the StatsAnalyzer
  wrapper that accumulates and writes CSV (:790-831) is out of scope."""
  name = name.lower()
  if name not in ANALYZERS:
    raise Exception('Analyzer not found: {}'.format(name))
  return ANALYZERS[name]


def f_ins_iou(results):
  raise NotImplementedError('f_ins_iou (:404-431) indexes the whole-batch list instead of one example '
                            '(iou_pairwise vs iou_pairwise_) and cannot run in the reference either')
