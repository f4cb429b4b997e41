import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rec-attend-public_amd')
for p in (PKG, os.path.join(ROOT, 'oracle'), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
  # The product library and the C oracle are build artefacts (git-ignored); build on demand.
  if not os.path.exists(os.path.join(PKG, 'librecattend.so')):
    subprocess.check_call(['make', '-s', '-j8', '-C', os.path.join(PKG, 'csrc')])
  if not os.path.exists(os.path.join(ROOT, 'oracle', 'libhungarian_oracle.so')):
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])


@pytest.fixture(scope='session')
def cuda():
  import torch
  if not torch.cuda.is_available():
    pytest.skip('no HIP device')
  return torch.device('cuda')


@pytest.fixture(autouse=True)
def _poison_lds(request):
  """Before every GPU test, leave NaN in all LDS: a kernel that consumes shared memory it never
  wrote (e.g. halo padding multiplied by zero weights) then fails here instead of flaking."""
  if request.node.get_closest_marker('gpu') is not None:
    import torch
    if torch.cuda.is_available():
      import ra_ops
      ra_ops.poison_lds()
      torch.cuda.synchronize()
  yield
