"""`box_model.get_model` of the reference (box_model.py:11-669) on MI355X kernels: the
controller-only variant used to pre-train the controller weights (run_cvppp.sh:16-28).

Per step: controller CNN over concat(x, canvas[, d_in, y_in]) -> glimpse LSTM -> attention
box; the canvas is ALWAYS teacher-forced from the greedily matched ground truth times
(1 - U[0, 0.3]) noise (box_model.py:484-504), so `run` needs `y_gt` and draws (or is fed,
key `noise` [T,B,H,W]) that noise.  Outputs: `s_out` [B,T] (sigmoid) or [B,T,nc] (softmax,
box_model.py:508-519), `attn_box` [B,T,H,W], `attn_ctr`, `attn_size`, `attn_top_left`,
`attn_bot_right`, `attn_ctr_norm`, `attn_lg_size`.  With `phase_train` True / `train_step` in the
fetch list (feed `y_gt`, `s_gt`) `run` executes the training graph (box_model.py:520-652:
matched box loss + conf loss, clip + Adam; ra_train.BoxTrainStep).
"""
import full_model as fm
import nnlib as nn
import ra_engine


class BoxModel(fm.Model):
  OUTPUTS = ('s_out', 'attn_box', 'attn_ctr', 'attn_size', 'attn_top_left', 'attn_bot_right',
             'attn_ctr_norm', 'attn_lg_size', 'ctrl_out', 'canvas', 'ctrl_rnn_glimpse_map')


def get_model(opt):
  """The box model (box_model.py:11)."""
  d = fm.derive_dims(opt, box_model=True)
  model = BoxModel(opt, d, box_model=True)
  model['phase_train'] = {'value': False}
  w = fm._load_pretrained(fm._get(opt, 'pretrain_net', None))
  pt = fm._pretrained_groups(w, d, ('ctrl', 'score'))
  ccnn, cell, gmlp, cmlp = fm._register_controller(model, opt, d, pt)
  smlp = nn.mlp([d['hid'], d['nsc']], [None], wd=opt['weight_decay'], scope='score_mlp',
                model=model, init_weights=pt and pt.get('smlp'))
  model.closures = dict(ccnn=ccnn, crnn_cell=cell, gmlp=gmlp, cmlp=cmlp, smlp=smlp)
  model['global_step'] = 0.0
  model.engine = ra_engine.DecodeEngine(d, model, box_model=True)
  return model
