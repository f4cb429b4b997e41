"""The C-ABI library loads (no GPU needed) and exports every symbol include/recattend.h
declares; the ctypes table (ra_native.SIGNATURES) covers exactly that set."""
import ctypes
import os
import re

import ra_native as rn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
  text = open(os.path.join(ROOT, 'include', 'recattend.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(ra_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported():
  lib = ctypes.CDLL(rn.LIB_PATH)
  syms = _header_symbols()
  assert len(syms) >= 25
  for s in syms:
    assert hasattr(lib, s), s


def _header_abi_version():
  import re
  return int(re.search(r'#define RA_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'recattend.h')).read()).group(1))


def test_binding_table_matches_header():
  assert sorted(rn.SIGNATURES) == _header_symbols()
  rn.lib()  # resolves them all
  assert rn.lib().ra_version() == rn.RA_ABI_VERSION == _header_abi_version()


def test_argument_validation_without_gpu():
  lib = rn.lib()
  assert lib.ra_conv_cout_padded(1) == 16 and lib.ra_conv_cout_padded(96) == 128
  assert lib.ra_conv_cout_padded(129) == 0
  assert lib.ra_conv_packed_floats(6, 8) == 0          # Cin % 4
  assert lib.ra_conv_packed_floats(8, 8) == 9 * 8 * 16
  # null pointers are rejected before any launch
  rc = lib.ra_conv3x3_f32(None, 4, None, 0, 1, 8, 8, 0, None, None, None, 8, 1, 1, None, -1, None, None)
  assert rc == -1 and b'bad argument' in lib.ra_last_error_string()
  rc = lib.ra_hungarian_f32(None, 1, 2, 2, None, None, None)
  assert rc == -1
  assert lib.ra_resample_bwd_workspace_floats(8, 48, 4) == 8 * 48 * 8 and lib.ra_ctrl_train_supported(256, 64, 256, 5, 9) == 1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
  monkeypatch.setattr(rn, 'LIB_PATH', str(tmp_path / 'nope.so'))
  monkeypatch.setattr(rn, '_lib', None)
  try:
    rn.lib()
    raise AssertionError('expected RecAttendError')
  except rn.RecAttendError as e:
    assert 'no CPU fallback' in str(e)
  finally:
    monkeypatch.undo()  # restores LIB_PATH and the loaded handle; reloading the module would fork
                        # RecAttendError into two classes for everything imported before
