"""bench.py's output contract (one JSON line with the driver's fields, the roofline object and the
CPU baseline), checked on the GPU with a short run; the argument parser on the CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags_parse_without_a_gpu():
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True)
  assert out.returncode == 0
  for flag in ('--gpus', '--steps', '--warmup', '--no-cpu-baseline', '--host-input', '--pmc-group'):
    assert flag in out.stdout


@pytest.mark.gpu
def test_bench_json_line(cuda):
  env = dict(os.environ)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline'], capture_output=True, text=True, env=env, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  line = [l for l in out.stdout.strip().splitlines() if l.startswith('{')][-1]
  d = json.loads(line)
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
            'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
    assert k in d, k
  assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
  assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['vs_baseline'] is None
  assert 'workload' in d['config'] and 'model' not in d['config']
  assert abs(d['value'] - 8 * 16 * 1000.0 / d['ms_per_step']) < 1e-6 * d['value']
  r = d['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
    assert k in r, k
  assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
  assert 0.1 < r['frac'] < 1.0
  ra = d['roofline_attn']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'achieved_traffic', 'at_B32', 'launch_floor_us'):
    assert k in ra, k
  assert ra['bound'] == 'hbm' and 'in_flight' not in ra and 0.2 < ra['frac'] < 1.0 and 'frac_algorithmic' in ra['at_B32'] and set(ra['by_box_size']) == {'0.15', '0.35', '1.00'}
  tr = d['train']  # the training step timed by the same run (cfg4 shapes, 3 steps)
  assert tr['steps'] == 3 and tr['dtype'] == 'f32' and tr['ms_per_step'] > 0 and tr['fused_controller'] and tr['hip_graph']
  assert abs(tr['value'] - 8 * 16 * 1000.0 / tr['ms_per_step']) < 1e-6 * tr['value'] and tr['ranks_in_communicator'] == 1
  assert d['config']['ranks_in_communicator'] == 1
  tb = d['train_bf16']  # the same step with model_opt['compute_dtype'] = 'bf16', its own dtype field
  assert tb['dtype'] == 'bf16' and tb['steps'] == 3 and tb['ms_per_step'] > 0 and tb['hip_graph']
  for t in (tr, tb):
    rf = t['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    assert abs(rf['achieved'] - rf['bytes_per_step'] / (t['ms_per_step'] * 1e-3) / 1e9) < 1e-6 * rf['achieved']


def test_train_layer_bytes_formula():
  """bench.train_layer_bytes: 3 X + 7 U + 3 Y per layer call, by hand for a two-layer toy network."""
  sys.path.insert(0, ROOT)
  import bench
  opt = dict(ctrl_cnn_depth=[8], ctrl_cnn_pool=[2], filter_height=8, filter_width=8, attn_cnn_depth=[4], attn_cnn_pool=[1],
             attn_dcnn_depth=[4], attn_dcnn_pool=[2])
  B, S, T = 2, 16, 3
  ctrl = 2 * (B * S * S * 4 * 4) + 6 * (B * S * S * 8 * 4) + 3 * (B * S * S * 8 * 4 // 4)        # no data gradient
  acnn = 3 * (B * 8 * 8 * 4 * 4) + 7 * (B * 8 * 8 * 4 * 4) + 3 * (B * 8 * 8 * 4 * 4)
  dcnn = 3 * (B * 8 * 8 * 4 * 4) + 7 * (B * 16 * 16 * 4 * 4) + 3 * (B * 16 * 16 * 4 * 4)
  assert bench.train_layer_bytes(opt, B, S, T) == T * (ctrl + acnn + dcnn)
  # bf16 storage: U of every layer 2 bytes; every net here has ONE layer, whose output stays float32 (float32 readers)
  ctrl_s = 2 * (B * S * S * 4 * 4) + 6 * (B * S * S * 8 * 2) + 3 * (B * S * S * 8 * 4 // 4)
  acnn_s = 3 * (B * 8 * 8 * 4 * 4) + 7 * (B * 8 * 8 * 4 * 2) + 3 * (B * 8 * 8 * 4 * 4)
  dcnn_s = 3 * (B * 8 * 8 * 4 * 4) + 7 * (B * 16 * 16 * 4 * 2) + 3 * (B * 16 * 16 * 4 * 4)
  assert bench.train_layer_bytes(opt, B, S, T, True) == T * (ctrl_s + acnn_s + dcnn_s)
  full = bench.make_opt('cvppp', 512, 512, 16)
  assert 0.5 < bench.train_layer_bytes(full, 8, 512, 16, True) / bench.train_layer_bytes(full, 8, 512, 16) < 0.56


@pytest.mark.gpu
def test_cpu_baseline_object(cuda):
  sys.path.insert(0, ROOT)
  import bench
  opt = bench.make_opt('cvppp', 64, 64, 2)
  cb = bench.cpu_baseline(opt, 3, budget_s=0.5)
  assert cb['kind'] == 'port' and cb['unit'] == 'instance-timesteps/s' and cb['value'] > 0 and cb['cores'] >= 1
  assert 'sample' in cb
