#!/usr/bin/env python3
"""What the REFERENCE constructor (pyMPC/mpc.py:76-252) does with every input of tests/error_cases.py: exception type and
message, or the shapes it stored.  Runs only in the build container (imports /root/reference with the osqp capture stub of
make_golden.py); writes tests/golden/ctor_outcomes.json (data only).

    python tests/golden/make_errors.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from make_golden import load_reference     # noqa: E402
import error_cases                         # noqa: E402


def main():
    Ref = load_reference()
    out = {name: error_cases.outcome(Ref, make) for name, make in error_cases.CASES.items()}
    with open(os.path.join(HERE, 'ctor_outcomes.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        print('%-32s %s' % (k, v[:3] if v[0] == 'error' else ('ok', v[2], v[3])))


if __name__ == '__main__':
    main()
