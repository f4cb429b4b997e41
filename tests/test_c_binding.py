"""The drop-in boundary is a C ABI: build a plain-C client against include/recattend.h and
librecattend.so with gcc and run the reference's Hungarian known-answer test through it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'rec-attend-public_amd')


def test_plain_c_client(tmp_path):
  exe = str(tmp_path / 'hungarian_kat')
  subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', os.path.join(ROOT, 'tests', 'c', 'hungarian_kat.c'),
                         '-I', os.path.join(ROOT, 'include'), '-L', PKG, '-lrecattend',
                         '-Wl,-rpath,' + PKG, '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
  out = subprocess.run([exe], capture_output=True, text=True)
  assert out.returncode == 0, out.stderr
  assert out.stdout.startswith('ok version')
