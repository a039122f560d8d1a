"""Resolved ``model`` dictionaries of the reference's shipped SST configs, as Python literals.

/root/reference does not exist on the GPU box, and `bench.py`'s `config_as_is` leg / `--workload sst_center` must build their
models from the config AS SHIPPED.  This script reads the config text the way mmcv.Config.fromfile
does (tests/test_configs_build.load_config: exec + `_base_` merge), takes the `model` entry and writes it with pprint - tuples,
integer keys and all - to tests/golden/configs/<name>.model.py; tests/test_config_fixtures.py checks (in the build container)
that the committed literals still equal what the config files say, and (everywhere) that they construct.

    python tests/golden/make_config_fixtures.py
"""
import os
import pprint
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))

CONFIGS = ['configs/sst_refactor/sst_waymoD5_1x_3class_8heads_v2.py',
           'configs/sst_refactor/sst_waymoD5_1x_3class_centerhead.py',
           'configs/sst_refactor/sst_waymoD1_2x_3class_centerhead.py']


def main():
    from test_configs_build import REF, load_config
    out_dir = os.path.join(HERE, 'configs')
    os.makedirs(out_dir, exist_ok=True)
    for rel in CONFIGS:
        cfg = load_config(os.path.join(REF, rel))
        name = os.path.splitext(os.path.basename(rel))[0]
        with open(os.path.join(out_dir, name + '.model.py'), 'w') as f:
            f.write(f'# resolved `model` of {rel} (tusen-ai/SST), written by tests/golden/make_config_fixtures.py - do not edit\n')
            f.write(pprint.pformat(cfg['model'], width=120, sort_dicts=False))
            f.write('\n')
        print(name, 'fp16' in cfg and cfg['fp16'])


if __name__ == '__main__':
    main()
