#!/usr/bin/env python3
"""Golden vectors for checkpoint key alignment, produced by RUNNING the reference's own
`align_and_update_state_dicts` (utils/check_point.py:14-63).  Run only in the build container:

    cd /tmp && python -B /root/repo/tests/golden/make_golden_checkpoint.py

Each case is (model keys, loaded keys); every loaded tensor carries a unique integer, so the output records which
loaded key the reference picked for every model key (-1: none).  No reference source is copied.
"""
import json
import logging
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')
sys.path.insert(1, REPO)

import torch  # noqa: E402
from utils import check_point as ref_cp  # noqa: E402  (reference)
from rtm3d_amd import weights  # noqa: E402

logging.disable(logging.CRITICAL)


def cases():
    dla = [k for k, _, _, _ in weights.state_dict_spec('DLA-34')]
    r18 = [k for k, _, _, _ in weights.state_dict_spec('RESNET-18')]
    out = {}
    out['identical_dla34'] = (dla, dla)
    out['backbone_only_stripped'] = (dla, [k[len('backbone.'):] for k in dla if k.startswith('backbone.')])     # ImageNet-style file
    out['wrapped_module_prefix'] = (dla, ['module.' + k for k in dla])                                             # no suffix relation: nothing matches
    out['model_nested_deeper'] = (['net.' + k for k in r18], r18)
    out['ambiguous_suffixes'] = (['a.b.conv1.weight', 'a.conv1.weight', 'x.b.conv1.weight', 'conv1.weight', 'b.bias'],
                                 ['conv1.weight', 'b.conv1.weight', 'a.b.conv1.weight', 'bias', 'b.bias', 'zz.b.bias'])
    out['partial_and_extra'] = (r18[:40], r18[20:60] + ['fc.weight', 'fc.bias'])
    return out


def main():
    res = {}
    for name, (mkeys, lkeys) in cases().items():
        msd = {k: torch.tensor(-1) for k in mkeys}
        lsd = {k: torch.tensor(i) for i, k in enumerate(lkeys)}
        if lkeys:
            ref_cp.align_and_update_state_dicts(msd, lsd)
        chosen = [int(msd[k]) for k in mkeys]
        res[name] = {'model_keys': mkeys, 'loaded_keys': lkeys, 'chosen': chosen}
        print(name, len(mkeys), len(lkeys), 'matched', sum(c >= 0 for c in chosen))
    with open(os.path.join(HERE, 'checkpoint_align_cases.json'), 'w') as f:
        json.dump(res, f)


if __name__ == '__main__':
    main()
