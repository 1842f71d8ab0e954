"""The reference's ansatz YAMLs, flattened to {dotted.key.path: scalar} -> tests/golden/ansatz_yaml.json.

    python tests/golden/make_yaml_golden.py          (build container only: reads /root/reference)

tests/test_spec_pins.py compares deepqmc_amd/spec.py (a hand transcription of these files that BOTH the HIP path and
the oracle consume) with this machine extraction, key by key, so a mistyped width / activation / flag cannot hide
behind "HIP == oracle".  List items are addressed by index; `_partial_` keys are dropped."""
import json
import os

import yaml

SRC = '/root/reference/src/deepqmc/conf/ansatz'
HERE = os.path.dirname(os.path.abspath(__file__))


def flatten(node, prefix, out):
    if isinstance(node, dict):
        for k, v in node.items():
            if k != '_partial_':
                flatten(v, f'{prefix}.{k}' if prefix else str(k), out)
    elif isinstance(node, list):
        out[prefix + '.#'] = len(node)
        for i, v in enumerate(node):
            flatten(v, f'{prefix}.{i}', out)
    else:
        out[prefix] = node


if __name__ == '__main__':
    res = {}
    for name in ('default', 'ferminet', 'psiformer', 'transpsiformer'):
        with open(os.path.join(SRC, name + '.yaml')) as f:
            tree = yaml.safe_load(f)
        flat = {}
        flatten(tree, '', flat)
        res[name] = flat
    with open(os.path.join(HERE, 'ansatz_yaml.json'), 'w') as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in res.items()})
