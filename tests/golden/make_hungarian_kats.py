#!/usr/bin/env python
"""Extracts the known-answer / termination vectors of the reference's only test module
(hungarian_tf_tests.py) into a data fixture.  Run in the build container, where the reference
is mounted at /root/reference; the JSON it writes is what travels (data only: the weight
matrices and the expected matching / covers asserted by the reference's tests)."""
import ast
import json
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/hungarian_tf_tests.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hungarian_kats.json')


def np_array_literal(node):
  """np.array(<literal>) -> python list."""
  assert isinstance(node, ast.Call) and node.func.attr == 'array'
  return ast.literal_eval(node.args[0])


def main():
  tree = ast.parse(open(SRC).read())
  cases = []
  for cls in tree.body:
    if not isinstance(cls, ast.ClassDef):
      continue
    for fn in cls.body:
      if not (isinstance(fn, ast.FunctionDef) and fn.name.startswith('test_')):
        continue
      case = {'name': fn.name, 'line': fn.lineno, 'round_1e6': False}
      for st in ast.walk(fn):
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
          name = st.targets[0].id
          if name in ('W', 'c_0_t', 'c_1_t', 'M_t') and isinstance(st.value, ast.Call) and \
              getattr(st.value.func, 'attr', '') == 'array':
            case[{'W': 'W', 'c_0_t': 'cover_x', 'c_1_t': 'cover_y', 'M_t': 'matching'}[name]] = \
                np_array_literal(st.value)
          if name == 'W' and isinstance(st.value, ast.BinOp):
            case['round_1e6'] = True  # W = np.round(W * p) / p, p = 1e6
      cases.append(case)
  cases.sort(key=lambda c: c['line'])
  json.dump({'source': 'hungarian_tf_tests.py', 'cases': cases}, open(OUT, 'w'), indent=1)
  print('wrote %d cases to %s' % (len(cases), OUT))


if __name__ == '__main__':
  main()
