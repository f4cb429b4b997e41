"""image_ops.py of the reference: the in-graph augmentation (random_transformation, :9-113).

At evaluation (phase_train false) it is pad-then-centre-crop = the identity (:70-82,106), which is
why the decode loop never calls it.  With phase_train true it draws ONE crop offset per batch (:52)
and one flip / transpose decision per batch (:85-97) and applies them to x, y, d, c alike; the
gather runs in one HIP kernel per tensor (ra_random_transform_f32).  The draws come from a
torch.Generator so that ranks can offset their streams deterministically.
The colour jitter (:99-103: random hue 0.1, saturation 0.9..1.1, brightness 0.1, contrast 0.9..1.1, one draw of each per
batch, in that order) follows the crop / flips on x (ra_colour_jitter_f32)."""
import torch

import nnlib as nn
import ra_ops as ops


def random_transformation(x, padding, phase_train, rnd_vflip=True, rnd_hflip=True, rnd_transpose=True,
                          rnd_colour=False, y=None, d=None, c=None, generator=None, draws=None, out=None):
  """x [B,H,W,3], y [B,T,H,W], d [B,H,W,8], c [B,H,W,1] -> dict with the same keys (x, y, d, c).
  draws: the step's decisions given instead of drawn ({off_y, off_x, flip_v, flip_h, transpose[, hue, saturation,
  brightness, contrast]}; tests).  out: {x, y, d, c} tensors of the inputs' shapes to write the transformed tensors into
  (the captured training step's static input buffers: no copy afterwards)."""
  out = out or {}
  results = {'x': x}
  for k, v in (('y', y), ('d', d), ('c', c)):
    if v is not None:
      results[k] = v
  if not nn._is_train(phase_train):
    return results  # centre slices of the padded tensors: the inputs themselves
  if d is not None:  # image_ops.py:42-45
    assert not rnd_vflip, 'Orientation mode is on, no random flips'
    assert not rnd_hflip, 'Orientation mode is on, no random flips'
    assert not rnd_transpose, 'Orientation mode is on, no random transpose'
  if draws is not None:
    kw = dict(padding=padding, off_y=int(draws.get('off_y', padding)), off_x=int(draws.get('off_x', padding)),
              flip_v=bool(draws.get('flip_v', False)), flip_h=bool(draws.get('flip_h', False)),
              transpose=bool(draws.get('transpose', False)))
  else:
    g = generator
    off = torch.randint(0, max(2 * padding, 1), (2,), generator=g) if padding > 0 else torch.zeros(2, dtype=torch.long)
    u = torch.rand(3, generator=g)
    # tf.random_uniform([1], 1.0 - float(flag), 1.0) < 0.5: never true when the flag is off (:85-94)
    flip_h = bool(rnd_hflip) and float(u[0]) < 0.5 and d is None
    flip_v = bool(rnd_vflip) and float(u[1]) < 0.5 and d is None
    do_tr = bool(rnd_transpose) and float(u[2]) < 0.5 and d is None
    kw = dict(padding=padding, off_y=int(off[0]), off_x=int(off[1]), flip_v=flip_v, flip_h=flip_h, transpose=do_tr)
  results['x'] = ops.random_transform(x, out=None if rnd_colour else out.get('x'), **kw)
  if rnd_colour:  # image_ops.py:99-103
    if draws is not None:
      col = dict(hue=float(draws.get('hue', 0.0)), saturation=float(draws.get('saturation', 1.0)),
                 brightness=float(draws.get('brightness', 0.0)), contrast=float(draws.get('contrast', 1.0)))
    else:
      u4 = torch.rand(4, generator=generator)
      col = dict(hue=float(-0.1 + 0.2 * u4[0]), saturation=float(0.9 + 0.2 * u4[1]), brightness=float(-0.1 + 0.2 * u4[2]),
                 contrast=float(0.9 + 0.2 * u4[3]))
    results['x'] = ops.colour_jitter(results['x'], col['hue'], col['saturation'], col['brightness'], col['contrast'])
  if y is not None:
    B, T, H, W = y.shape
    oy = out.get('y')
    results['y'] = ops.random_transform(y.reshape(B * T, H, W), out=None if oy is None else oy.reshape(B * T, H, W), **kw).reshape(B, T, H, W)
    if oy is not None:
      results['y'] = oy
  if d is not None:
    results['d'] = ops.random_transform(d, out=out.get('d'), **kw)
  if c is not None:
    results['c'] = ops.random_transform(c, out=out.get('c'), **kw)
  results['_draws'] = dict(kw, **col) if rnd_colour else kw
  return results
